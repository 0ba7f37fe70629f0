#!/bin/bash
# gpurun helper: A/B of the backward's prefetch-helper count / lead, attention bench
mkdir -p gpurun_out/dbg
for cfg in "2 1" "1 1" "3 1" "2 2" "3 2"; do
  set -- $cfg
  timeout 120 python tools/op_bench.py --iters 7 --helpers $1 --lead $2 > gpurun_out/dbg/op.json 2>&1
  tail -1 gpurun_out/dbg/op.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('helpers $1 lead $2 bwd ms', round(d['bwd']['avg_ms'],3), round(d['bwd']['min_ms'],3))"
done
timeout 200 python tools/attn_bench.py --no-sdpa --iters 7 2>/dev/null | tail -1
