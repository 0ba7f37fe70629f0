#!/bin/bash
# round 6, call G: deriver variant 2 (one sigmoid evaluation per step) - oracle / determinism test, interleaved A/B at NC = 804 and 282, stamps
cd /root/repo; mkdir -p gpurun_out/r6g; O=$GRAFT_REPO_ROOT/gpurun_out/r6g
timeout 600 python -m pytest tests/test_parity_r6_gpu.py -x -q -m gpu -k "deriver_variant or barrier_inside" 2>&1 | tail -3
for nc in 804 282; do
timeout 300 python tools/op_bench.py --nc $nc --iters 16 --ab deriver_variant --ab-values 1,2 --ab-restore 2 --phases > $O/op_nc${nc}_ab_variant.json 2>/dev/null
python - <<PY
import json
d=[json.loads(l) for l in open("$O/op_nc${nc}_ab_variant.json") if l.startswith("{")][0]
print("nc $nc", {k: round(v["bwd_avg_ms"],3) for k,v in d["ab"].items() if isinstance(v, dict)})
ph=d["phase_cycles_per_step"]; print("stamps variant 2:", {k: round(ph[k]) for k in range(16,42)})
PY
done
