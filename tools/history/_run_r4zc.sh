#!/bin/bash
# Round 4, last GPU seconds: flags_memset_early - bit identity, op-level A/B.
cd /root/repo; mkdir -p gpurun_out/r4zc; O=$GRAFT_REPO_ROOT/gpurun_out/r4zc
export TMPDIR=/tmp
timeout 40 python -m pytest tests/test_parity_r4_gpu.py -m gpu -x -q -k "schedule_options" > $O/test.log 2>&1; echo "test rc=$?"; tail -1 $O/test.log
timeout 40 python tools/op_bench.py --nc 804 --iters 12 --ab flags_memset_early > $O/op_ab.json 2>&1
python -c "import json,sys; d=json.loads(open('$O/op_ab.json').read().strip().splitlines()[-1]); print('op A/B', d['ab'])"
