#!/bin/bash
# Round 4, call W: does a streaming cast kernel in front of the TTT-MLP backward slow its sweeps (op level)?
cd /root/repo; mkdir -p gpurun_out/r4w; O=$GRAFT_REPO_ROOT/gpurun_out/r4w
export TMPDIR=/tmp
for d in 0 154 0 154 600; do
  timeout 120 python tools/op_bench.py --nc 804 --iters 8 --disturb $d > $O/op_disturb$d.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/op_disturb$d.json').read().strip().splitlines()[-1]); print('disturb $d MB: fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3), 'min', round(d['bwd']['min_ms'],3))"
done
