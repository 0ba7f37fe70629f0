"""Timeline of the LAST pipelined TTT layer forward in a rocprofv3 --kernel-trace CSV (argv[1] = *_kernel_trace.csv): every dispatch
from the first GEMM in front of the last run of scan launches to the last kernel behind it, relative times, queue ids."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
scans = [i for i, k in enumerate(ks) if "mlp_scan" in k[2]]
n_parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
last = scans[-n_parts:]
lo = max(0, last[0] - 14)
hi = min(len(ks), last[-1] + 8)
t0 = ks[lo][0]
for s, e, n, q in ks[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  q{q}  {n[:70]}")
