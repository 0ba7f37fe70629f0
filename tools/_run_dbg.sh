#!/bin/bash
# gpurun helper: attention backward timing per kernel (rocprof stats) + tests
export TMPDIR=/tmp
R=$PWD
timeout 200 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -1
timeout 200 python tools/attn_bench.py --no-sdpa --iters 9 2>/dev/null | tail -1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o a -- python $R/tools/attn_bench.py --iters 5 --no-sdpa > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pa/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
