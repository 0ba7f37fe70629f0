#!/bin/bash
# DEBUG helper for gpurun: backward correctness (multi-chunk) + A/B of the recompute/sweep overlap
mkdir -p gpurun_out/dbg
timeout 100 python tools/debug_bwd_v2.py > gpurun_out/dbg/debug_bwd2.log 2>&1
grep -E "===|dXK|dW1 |NaN" gpurun_out/dbg/debug_bwd2.log | tail -9
for f in "" "--no-overlap" ""; do
  timeout 120 python tools/op_bench.py --iters 7 $f > gpurun_out/dbg/op.json 2>&1
  tail -1 gpurun_out/dbg/op.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f bwd ms', round(d['bwd']['avg_ms'],3), round(d['bwd']['min_ms'],3), 'fwd ms', round(d['fwd']['avg_ms'],3))"
done
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "mfma or fused" 2>&1 | tail -1
