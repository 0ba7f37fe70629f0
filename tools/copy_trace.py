#!/usr/bin/env python
"""Memory copies of a traced run (rocprofv3 --memory-copy-trace CSV): per direction the number of copies, bytes, summed duration, the rate
while copying, the union of copy intervals, and - with the kernel trace - GPU busy time (union of kernel intervals) inside / outside them.

    python tools/copy_trace.py <memory_copy_trace.csv> [<kernel_trace.csv>] [--min-bytes 50000000]"""
import argparse, csv, collections
ap = argparse.ArgumentParser(); ap.add_argument("copies"); ap.add_argument("kernels", nargs="?"); ap.add_argument("--min-bytes", type=float, default=5e7)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.copies)))
print("columns:", list(rows[0].keys()) if rows else None, "rows:", len(rows))
key_b = next((k for k in ("Bytes", "Size", "bytes") if rows and k in rows[0]), None)
by = collections.defaultdict(list)
for r in rows:
    nb = float(r.get(key_b, 0) or 0) if key_b else 0.0
    if nb >= a.min_bytes or not key_b:             # (rocprofv3 7.2 writes no size column: every copy counts, the rates below are then 0)
        by[r.get("Direction", r.get("Operation", "?"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nb))
for d, v in by.items():
    v.sort()
    tot = sum(e - s for s, e, _ in v); nb = sum(b for _, _, b in v)
    un, cs, ce = 0, v[0][0], v[0][1]
    for s, e, _ in v[1:]:
        if s > ce:
            un += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    un += ce - cs
    rates = sorted(b / max(e - s, 1) for s, e, b in v)
    print(f"{d}: {len(v)} copies >= {a.min_bytes / 1e6:.0f} MB, {nb / 2**30:.1f} GiB, summed {tot / 1e6:.1f} ms = {nb / max(tot, 1):.1f} GB/s while copying "
          f"(median copy {rates[len(rates) // 2]:.1f}, slowest decile {rates[len(rates) // 10]:.1f} GB/s), union {un / 1e6:.1f} ms")
