import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ph = d.get("phase_cycles_per_step", [0] * 32)
print(sys.argv[1], "bwd", round(d["bwd"]["avg_ms"], 3), "ms", round(d["bwd"]["us_per_step"], 2), "us/step | sweep stages", ph[16:24], "| owner hand-over", ph[24:28])
