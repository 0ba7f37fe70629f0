"""Busy / idle accounting of a rocprofv3 --kernel-trace CSV (argv[1]) over the LAST `argv[2]` seconds of the trace (note: a
bench.py trace ends ~0.1 s after the timed step - host-side barrier and result line - so align the window with tools/_fmt_gaps.py): union of all
dispatch intervals, idle gaps, and the kernels by total time (name truncated).  For comparing two runs of the same step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in ks)
win = float(sys.argv[2]) * 1e9
ks = [k for k in ks if k[0] >= t_end - win]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t_end - ks[0][0]
print(f"window {span / 1e6:.1f} ms: GPU busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms, {len(ks)} dispatches")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in ks:
    agg[n[:80]][0] += 1
    agg[n[:80]][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{t / 1e6:9.3f} ms  {c:5d}  {n}")
