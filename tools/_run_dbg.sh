#!/bin/bash
# DEBUG helper for gpurun: A/B timing of backward-sweep variants
mkdir -p gpurun_out/dbg
for v in 0 1 0 1; do
  timeout 120 python tools/op_bench.py --phases --iters 5 --sweep-variant $v > gpurun_out/dbg/op_v$v.json 2>&1
  tail -1 gpurun_out/dbg/op_v$v.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', 'bwd ms', round(d['bwd']['avg_ms'],3), [int(x) for x in d['phase_cycles_per_step'][16:26]])"
done
