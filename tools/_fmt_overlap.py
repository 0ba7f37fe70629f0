"""Which kernels of OTHER queues ran while the TTT-MLP backward sweep was running?  (argv[1] = rocprofv3 *_kernel_trace.csv of an
FSDP run.)  Prints, per queue, the kernel names with their total time and the part of it that overlapped a sweep dispatch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
sweeps = sorted((s, e) for s, e, n, q in ks if "mlp_bwd_cluster" in n)
main_q = collections.Counter(q for s, e, n, q in ks if "mlp_bwd_cluster" in n).most_common(1)[0][0]
print(f"{len(ks)} dispatches, {len(sweeps)} sweep dispatches on queue {main_q}; queues: {dict(collections.Counter(q for *_, q in ks))}")


def overlap(s, e):
    t = 0
    for a, b in sweeps:
        if b <= s:
            continue
        if a >= e:
            break
        t += min(e, b) - max(s, a)
    return t


agg = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n, q in ks:
    if q == main_q:
        continue
    key = (q, n[:90])
    agg[key][0] += 1
    agg[key][1] += e - s
    agg[key][2] += overlap(s, e)
for (q, n), (c, tot, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"queue {q}  calls {c:5d}  total {tot / 1e6:9.3f} ms  beside a sweep {ov / 1e6:9.3f} ms  {n}")
