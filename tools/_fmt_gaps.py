"""Largest idle gaps of the GPU inside the last `argv[2]` seconds of a rocprofv3 --kernel-trace CSV (argv[1]): start (ms before the
end of the trace), length, the dispatch before and the one after."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in ks)
ks = [k for k in ks if k[0] >= t_end - float(sys.argv[2]) * 1e9]
gaps, cur_e, prev = [], None, None
for s, e, n in ks:
    if cur_e is not None and s > cur_e:
        gaps.append((s - cur_e, cur_e, prev, n))
    if cur_e is None or e > cur_e:
        cur_e, prev = e, n
tot = sum(g[0] for g in gaps)
print(f"{len(gaps)} gaps, {tot / 1e6:.1f} ms idle in the last {(t_end - ks[0][0]) / 1e6:.1f} ms")
hist = [0, 0, 0, 0]
for g in gaps:
    hist[0 if g[0] < 2e4 else 1 if g[0] < 2e5 else 2 if g[0] < 2e6 else 3] += g[0]
print("idle by gap length: <20us %.1f ms, 20-200us %.1f ms, 0.2-2ms %.1f ms, >2ms %.1f ms" % tuple(h / 1e6 for h in hist))
for d, at, a, b in sorted(gaps, reverse=True)[:14]:
    print(f"{(t_end - at) / 1e6:8.1f} ms before the end: idle {d / 1e6:7.3f} ms  after [{a[:60]}]  before [{b[:60]}]")
