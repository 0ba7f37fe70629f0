#!/bin/bash
# gpurun helper: A/B of the backward sweep variants (same box, interleaved)
mkdir -p gpurun_out/dbg
for v in 1 2 1 2; do
  timeout 120 python tools/op_bench.py --phases --iters 7 --sweep-variant $v > gpurun_out/dbg/op.json 2>&1
  tail -1 gpurun_out/dbg/op.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v bwd ms', round(d['bwd']['avg_ms'],3), round(d['bwd']['min_ms'],3), [int(x) for x in d['phase_cycles_per_step'][16:26]])"
done
timeout 100 python tools/debug_bwd_v2.py 2>/dev/null | grep -E "dW1 |dXK" | tail -4
