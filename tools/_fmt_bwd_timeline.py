"""Timeline of the LAST TTT-MLP backward in a rocprofv3 --kernel-trace CSV (argv[1] = *_kernel_trace.csv): begin / end of every recompute (A),
sweep (B) and tail (C) dispatch relative to the first one, the gap in front of every sweep and what ended last before it."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
tag = lambda n: "A" if "mlp_recompute8" in n else "B" if "mlp_bwd_cluster4" in n else "C" if "mlp_bwd_tail" in n else None
sel = [(s, e, tag(n), q) for s, e, n, q in ks if tag(n)]
fw = [i for i, k in enumerate(ks) if "mlp_scan" in k[2]]
t_f = ks[fw[-1]][1] if fw else 0
sel = [k for k in sel if k[0] >= t_f]
t0 = sel[0][0]
prev_b_end = None
for s, e, t, q in sel:
    extra = ""
    if t == "B":
        if prev_b_end is not None:
            last_a = max((ee for ss, ee, tt, qq in sel if tt == "A" and ee <= s + 20000), default=None)
            extra = f"   gap behind the previous sweep {(s - prev_b_end) / 1e3:7.1f} us; last recompute ended {((s - last_a) / 1e3) if last_a else float('nan'):7.1f} us before"
        prev_b_end = e
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  q{q}  {t}{extra}")
print(f"total {(sel[-1][1] - t0) / 1e3:.1f} us")
