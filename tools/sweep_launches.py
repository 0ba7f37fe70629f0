"""Per-launch view of the TTT-MLP backward sweep in a rocprofv3 kernel trace of `python bench.py` (argv[1] = *_kernel_trace.csv).

The default one-GPU bench runs two phases in one process: the replica path (the driver line) and the `fsdp1` point (FlatFSDP with
its RCCL collectives over one rank).  Averages hide what differs between them (profiles/r4q_*: the sweep is the only kernel that
changes, 0.82 -> 0.95 ms), so this prints, per phase, the distribution of the sweep launches' durations and - launch by launch -
which kernels of other queues were running beside them.  A phase boundary is the first RCCL kernel after the warm-up collective."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
    t0 = ks[0][0]
    is_sweep = lambda n: "mlp_bwd_cluster4" in n
    is_ttt = lambda n: "mlp_bwd_tail4" in n or "mlp_recompute8" in n
    is_rccl = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
    sweeps = [k for k in ks if is_sweep(k[2])]
    rc = [k for k in ks if is_rccl(k[2])]
    # the sharded phase: from the first RCCL kernel that comes AFTER the first sweep (the warm-up all-reduce comes before any)
    first_sweep = sweeps[0][0]
    later = [k for k in rc if k[0] > first_sweep]
    boundary = later[0][0] if later else None
    print(f"{len(ks)} dispatches over {(ks[-1][1] - t0) / 1e9:.1f} s, {len(sweeps)} sweep launches, {len(rc)} RCCL kernels; "
          f"sharded phase from t = {((boundary - t0) / 1e9 if boundary else float('nan')):.1f} s")
    others = [k for k in ks if not is_sweep(k[2])]
    starts = [k[0] for k in others]
    import bisect

    def beside(s, e):
        out = []
        i = bisect.bisect_left(starts, s - 50_000_000)            # kernels are shorter than 50 ms
        while i < len(others) and others[i][0] < e:
            a, b, n, q = others[i]
            if b > s:
                out.append((min(e, b) - max(s, a), n, q))
            i += 1
        return out

    for phase, sel in (("replica", [k for k in sweeps if boundary is None or k[0] < boundary]), ("sharded (fsdp1)", [k for k in sweeps if boundary is not None and k[0] >= boundary])):
        if not sel:
            continue
        d = sorted((e - s) / 1e3 for s, e, *_ in sel)
        full = [x for x in d if x > 0.6 * d[len(d) // 2]]          # (the short last chunk of a scan is a launch of its own)
        q = lambda p: full[min(len(full) - 1, int(p * len(full)))]
        print(f"\n== {phase}: {len(sel)} launches, full-size ones: mean {sum(full) / len(full):.1f} us, p10 {q(0.1):.1f}, p50 {q(0.5):.1f}, p90 {q(0.9):.1f}, max {full[-1]:.1f}")
        med = d[len(d) // 2]
        alone, foreign = [], []
        agg = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n, qq in sel:
            dur = (e - s) / 1e3
            if dur <= 0.6 * med:
                continue
            ov = [(t, nn, q2) for t, nn, q2 in beside(s, e) if not is_ttt(nn)]
            t_for = sum(t for t, *_ in ov) / 1e3
            (foreign if t_for > 0.05 * dur else alone).append(dur)
            for t, nn, q2 in ov:
                a = agg[nn[:80]]
                a[0] += 1; a[1] += t / 1e3
        m = lambda v: (sum(v) / len(v)) if v else float("nan")
        print(f"   with only its own tail / recompute beside it: {len(alone)} launches, mean {m(alone):.1f} us;  with a foreign kernel beside it (> 5 % of its time): {len(foreign)} launches, mean {m(foreign):.1f} us")
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            print(f"   beside a sweep: {c:5d} x, {t / 1e3:8.2f} ms overlapped  {n}")
        # slow launches in time order: do they cluster?
        slow = [(s - t0) / 1e9 for s, e, *_ in sel if (e - s) / 1e3 > 1.1 * q(0.5)]
        print(f"   launches slower than 1.1 x the median: {len(slow)}" + (f", first at {slow[0]:.2f} s, last at {slow[-1]:.2f} s" if slow else ""))


if __name__ == "__main__":
    main()
