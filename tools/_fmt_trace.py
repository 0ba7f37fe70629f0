"""Timeline of the last TTT-MLP backward in a rocprofv3 --kernel-trace CSV: begin / end of every recompute (A), sweep (B) and
tail (C) dispatch, relative to the first one (argv[1] = *_kernel_trace.csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
sel = [k for k in ks if "mlp_scan_kernel" in k[2] or "mlp_bwd_cluster" in k[2] or "mlp_bwd_tail" in k[2]]
# the last backward starts at the last recompute dispatch that is preceded by a forward scan
starts = [i for i, k in enumerate(ks) if "mlp_scan8" in k[2] or "mlp_scan_pair" in k[2]]
t_fwd_end = ks[starts[-1]][1] if starts else sel[0][0]
sel = [k for k in sel if k[0] >= t_fwd_end]
t0 = sel[0][0]
for s, e, n, q in sel:
    short = "A recompute" if "mlp_scan_kernel" in n else ("B sweep" if "cluster" in n else "C tail")
    print(f"{(s - t0) / 1e3:10.1f} us -> {(e - t0) / 1e3:10.1f} us   {(e - s) / 1e3:8.1f} us   queue {q}  {short}")
print(f"total {(sel[-1][1] - t0) / 1e3:.1f} us")
