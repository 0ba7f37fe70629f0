#!/usr/bin/env python
"""GPU idle time inside a training step from a rocprofv3 --kernel-trace CSV of `bench.py --role worker`: the union of all kernel intervals
(any queue) over the last N seconds of the trace, the idle gaps by size, and the kernels around the largest ones.

    python tools/step_gaps.py run_kernel_trace.csv [--window-s 11.4]"""
import argparse, csv
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--window-s", type=float, default=11.0); a = ap.parse_args()
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.csv))))
t_end = max(e for _, e, _ in rows); t_lo = t_end - int(a.window_s * 1e9)
rows = [r for r in rows if r[0] >= t_lo]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []; last_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last_name, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last_name = n
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f"window {span / 1e9:.3f} s, {len(rows)} kernels, busy (union) {busy / 1e9:.3f} s = {100 * busy / span:.2f} %, idle {1e3 * (span - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
    g = [x[0] for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo / 1e3:7.0f} .. {hi / 1e3:9.0f} us: {len(g):6d}  total {sum(g) / 1e6:8.1f} ms")
import collections
by = collections.defaultdict(lambda: [0, 0])
for d, a_, b_ in gaps:
    k = (a_[:48], b_[:48]); by[k][0] += d; by[k][1] += 1
for (a_, b_), (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"  {d / 1e6:8.2f} ms in {c:5d} gaps  after {a_:48s} before {b_}")
