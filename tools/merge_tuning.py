#!/usr/bin/env python
"""Merge TunableOp result files: lines of NEW replace the lines of BASE with the same (operator, shape) key, the rest of NEW is appended;
validator lines are BASE's.    python tools/merge_tuning.py BASE.csv NEW.csv OUT.csv"""
import sys
base, new, out = sys.argv[1:4]
key = lambda l: tuple(l.split(",")[:2])
nl = [l for l in open(new).read().splitlines() if l and not l.startswith("Validator")]
nk = {key(l): l for l in nl}
res, seen = [], set()
for l in open(base).read().splitlines():
    if l and not l.startswith("Validator") and key(l) in nk:
        res.append(nk[key(l)]); seen.add(key(l))
    else:
        res.append(l)
res += [l for l in nl if key(l) not in seen]
open(out, "w").write("\n".join(res) + "\n")
print(f"{len(seen)} replaced, {len(nl) - len(seen)} appended")
