#!/bin/bash
# gpurun helper: attention kernel check + timing + LDS conflict counters
mkdir -p gpurun_out/dbg
export TMPDIR=/tmp
R=$PWD
timeout 200 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -1
timeout 200 python tools/attn_bench.py --no-sdpa --iters 7 2>/dev/null | tail -1
cd /tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "attn_" --output-format csv -d /tmp/pb -o a -- python $R/tools/attn_bench.py --iters 2 --no-sdpa > $R/gpurun_out/dbg/attn_lds.log 2>&1
python - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/pb/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)): agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(set(x[0] for x in agg)):
    c, a = sum(agg[(k,'SQ_LDS_BANK_CONFLICT')])/len(agg[(k,'SQ_LDS_BANK_CONFLICT')]), sum(agg[(k,'SQ_LDS_IDX_ACTIVE')])/len(agg[(k,'SQ_LDS_IDX_ACTIVE')])
    print(k, 'conflict', f'{c:.3g}', 'active', f'{a:.3g}', 'frac', round(c/max(a,1),3))
PY
