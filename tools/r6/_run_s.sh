#!/bin/bash
# round 6, call S (experiment build, not committed): knock-out timings of the sweep's roles - which phase really bounds a step
cd /root/repo
for k in 0 1 2 4 8 16 64 128 3 7 24 9 72 192 255; do
timeout 200 python tools/op_bench.py --nc 804 --iters 6 --ab-fixed knock=$k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knock $k bwd', round(d['bwd']['avg_ms'],3))"
python - <<PY 2>/dev/null
import sys; sys.path.insert(0,'ttt-video-dit_amd')
import test_time_training as e; e.load_library(); 
try: e.sweep_error_clear()
except Exception: pass
PY
done
