"""Formats the JSON line of tools/op_bench.py for the logs under profiles/ (argv[1] = a label)."""
import json, sys
d = json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith("{")][-1])
out = [sys.argv[1] if len(sys.argv) > 1 else ""]
for k in ("fwd", "bwd"):
    if k in d:
        out += [k, f'{d[k]["avg_ms"]:.3f} ms (min {d[k]["min_ms"]:.3f})', f'{d[k]["us_per_step"]:.2f} us/step']
ph = d.get("phase_cycles_per_step")
if ph:
    out += ["| sweep stages", str(ph[16:24]), "| owner hand-over", str(ph[24:28]), "| deriver (stage, z1b, reverse, barriers)", str(ph[28:32])]
    if len(ph) >= 36:
        out += ["| owners before Bb (Bd..Ba arrival, wait Ba, consume_step, wait Bb)", str(ph[32:36])]
print(" ".join(out))
