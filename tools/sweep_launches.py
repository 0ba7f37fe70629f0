"""Per-launch view of the TTT-MLP backward sweep in a rocprofv3 kernel trace of `python bench.py`.

    python tools/sweep_launches.py run_kernel_trace.csv                   # rocprofv3 --kernel-trace --output-format csv
    python tools/sweep_launches.py profiles/r4y_ttt_bwd_launches.csv.gz   # the committed reduction of round 4's trace (sweep, tail,
                                                                          # recompute and fill dispatches only, times from 0)

The default one-GPU bench runs two phases in one process: the replica path (the driver line) and the `fsdp1` point (FlatFSDP with
its RCCL collectives over one rank).  Averages hide what differs between them (profiles/r4q_*: the sweep is the only kernel that
changes, 0.82 -> 0.95 ms), so this prints, per phase: the distribution of the sweep launches' durations; which kernels of other
queues ran beside them; and the RACE between a sweep and the tail kernel of the previous chunk, which become ready together (both
wait for the same recompute): who was dispatched first, and how long the sweep took then.  A phase boundary is the first RCCL kernel
after the first sweep, or - one rank's collectives are copies, not kernels - a pause of more than 5 s between sweeps."""
import bisect
import collections
import csv
import gzip
import sys


def load(path):
    fh = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(fh))
    return ks


def is_sweep(n):
    return "mlp_bwd_cluster4" in n or n == "sweep"


def is_tail(n):
    return "mlp_bwd_tail4" in n or n == "tail"


def is_recompute(n):
    return "mlp_recompute8" in n or n == "recompute"


def phases(ks):
    sweeps = [k for k in ks if is_sweep(k[2])]
    rc = [k for k in ks if ("nccl" in k[2].lower() or "rccl" in k[2].lower()) and k[0] > sweeps[0][0]]
    boundary = rc[0][0] if rc else None
    if boundary is None:
        gaps = [(sweeps[i + 1][0] - sweeps[i][1], sweeps[i][1]) for i in range(len(sweeps) - 1)]
        g, at = max(gaps)
        if g > 5e9:
            boundary = at + 1
    first = [k for k in sweeps if boundary is None or k[0] < boundary]
    second = [k for k in sweeps if boundary is not None and k[0] >= boundary]
    return [("replica", first)] + ([("sharded (fsdp1)", second)] if second else []), boundary


def race(sel, tails, recomputes):
    """{(outcome, order): [sweep us, tail offset us, tail us, gap behind the recompute us]} for full-size sweep / tail pairs"""
    tstarts = [k[0] for k in tails]
    rends = sorted(k[1] for k in recomputes)
    d = sorted((e - s) / 1e3 for s, e, *_ in sel)
    med = d[len(d) // 2]
    out = collections.defaultdict(list)
    for s, e, *_ in sel:
        dur = (e - s) / 1e3
        if dur <= 0.6 * med:
            continue                                            # the short last chunk of a scan
        i = bisect.bisect_left(tstarts, s - 50_000)
        if not (i < len(tails) and tails[i][0] < s + 50_000 and tails[i][1] - tails[i][0] > 100_000):
            continue                                            # no full-size tail beside it (the first full chunk of a scan)
        off = (tails[i][0] - s) / 1e3
        j = bisect.bisect_right(rends, s) - 1
        gap = (s - rends[j]) / 1e3 if j >= 0 else float("nan")
        out[("slow" if dur > 1.12 * med_fast(d) else "fast", "tail first" if off < 0 else "sweep first")].append((dur, off, (tails[i][1] - tails[i][0]) / 1e3, gap))
    return out


def med_fast(sorted_durations):
    """the fast mode's typical duration: the 25th percentile of the full-size launches"""
    full = [x for x in sorted_durations if x > 0.6 * sorted_durations[len(sorted_durations) // 2]]
    return full[len(full) // 4]


def summarize(path):
    ks = load(path)
    t0 = ks[0][0]
    ph, boundary = phases(ks)
    tails = [k for k in ks if is_tail(k[2])]
    recomputes = [k for k in ks if is_recompute(k[2])]
    others = [k for k in ks if not is_sweep(k[2])]
    starts = [k[0] for k in others]
    res = {"dispatches": len(ks), "boundary_s": None if boundary is None else (boundary - t0) / 1e9, "phases": {}}
    for name, sel in ph:
        d = sorted((e - s) / 1e3 for s, e, *_ in sel)
        full = [x for x in d if x > 0.6 * d[len(d) // 2]]
        q = lambda p: full[min(len(full) - 1, int(p * len(full)))]
        foreign = collections.defaultdict(lambda: [0, 0.0])
        for s, e, *_ in sel:
            i = bisect.bisect_left(starts, s - 50_000_000)     # kernels are shorter than 50 ms
            while i < len(others) and others[i][0] < e:
                a, b, n, _q = others[i]
                if b > s and not (is_tail(n) or is_recompute(n)):
                    f = foreign[n[:80]]
                    f[0] += 1; f[1] += (min(e, b) - max(s, a)) / 1e3
                i += 1
        r = race(sel, tails, recomputes)
        res["phases"][name] = {"launches": len(sel), "mean": sum(full) / len(full), "p10": q(0.1), "p50": q(0.5), "p90": q(0.9), "max": full[-1],
                               "foreign": dict(foreign),
                               "race": {k: {"n": len(v), "sweep_us": sum(x[0] for x in v) / len(v), "tail_offset_us": sum(x[1] for x in v) / len(v),
                                            "tail_us": sum(x[2] for x in v) / len(v), "gap_behind_recompute_us": sum(x[3] for x in v) / len(v)} for k, v in r.items()}}
    return res


def main():
    res = summarize(sys.argv[1])
    print(f"{res['dispatches']} dispatches; sharded phase from t = {res['boundary_s']} s")
    for name, p in res["phases"].items():
        print(f"\n== {name}: {p['launches']} sweep launches; full-size ones: mean {p['mean']:.1f} us, p10 {p['p10']:.1f}, p50 {p['p50']:.1f}, p90 {p['p90']:.1f}, max {p['max']:.1f}")
        for n, (c, t) in sorted(p["foreign"].items(), key=lambda kv: -kv[1][1])[:6]:
            print(f"   beside a sweep (not its own tail / recompute): {c:5d} x, {t / 1e3:8.2f} ms overlapped  {n}")
        tot = sum(v["n"] for v in p["race"].values())
        for (speed, order), v in sorted(p["race"].items()):
            print(f"   {order:11s} / {speed}: {v['n']:5d} ({100 * v['n'] / max(1, tot):4.1f} %)  sweep {v['sweep_us']:7.1f} us  tail starts {v['tail_offset_us']:+5.2f} us, runs {v['tail_us']:5.1f} us;"
                  f" sweep starts {v['gap_behind_recompute_us']:4.1f} us behind its recompute")


if __name__ == "__main__":
    main()
