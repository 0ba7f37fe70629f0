#!/bin/bash
# round 6, call D: where the sweep's barrier Bc falls inside the derivers' reverse step (debug option deriver_split 0 .. 4), interleaved
# pairs in one process each, schedule 2
cd /root/repo; mkdir -p gpurun_out/r6d; O=$GRAFT_REPO_ROOT/gpurun_out/r6d
for pair in 1,3 1,4 3,4 2,3; do
timeout 300 python tools/op_bench.py --nc 804 --iters 16 --overlap 2 --ab deriver_split --ab-values $pair 2>/dev/null | python -c "import sys,json; print('split $pair', {k: round(v['bwd_avg_ms'],3) for k,v in json.loads(sys.stdin.read().strip().splitlines()[-1])['ab'].items() if isinstance(v, dict)})"
done
timeout 300 python -m pytest tests/test_parity_r6_gpu.py -x -q -m gpu -k "barrier_inside" 2>&1 | tail -2
