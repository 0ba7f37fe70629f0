#!/bin/bash
# Round 4, last call: gate kernel in front of the tail (debug option tail_delay_us), op level only: parity, then time per delay.
cd /root/repo; mkdir -p gpurun_out/r4za; O=$GRAFT_REPO_ROOT/gpurun_out/r4za
export TMPDIR=/tmp
for v in 0 10 20 40 0 20; do
  timeout 60 python tools/op_bench.py --nc 804 --iters 8 --ab-fixed tail_delay_us=$v > $O/op_delay${v}.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/op_delay${v}.json').read().strip().splitlines()[-1]); print('delay $v us: bwd', round(d['bwd']['avg_ms'],3), 'min', round(d['bwd']['min_ms'],3))"
done
