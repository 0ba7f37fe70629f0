#!/bin/bash
# Round 4, call T (one box): is the TTT-MLP backward sensitive to WHERE its buffers lie?  (in the training step the same kernels
# take 10.9 - 12.8 ms depending on the allocation history).  Workspace base / input bases skewed inside their allocations.
cd /root/repo; mkdir -p gpurun_out/r4t; O=$GRAFT_REPO_ROOT/gpurun_out/r4t
export TMPDIR=/tmp
one() { tag=$1; shift; timeout 120 python tools/op_bench.py --nc 804 --iters 8 "$@" > $O/op_$tag.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/op_$tag.json').read().strip().splitlines()[-1]); print('$tag fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3), d['ptr_mod_2MiB'])"; }
one base
for s in 256 4096 65536 1048576; do one ws$s --ws-skew $s; done
for s in 256 4352 69632 266240 1052672; do one io$s --io-skew $s; done
one base2
