#!/bin/bash
# Round 4, call Z: the tail kernel's side stream at the lowest priority: op-level A/B, parity of the backward, then the default line
# with its fsdp1 point (does the sharded path's sweep return to the replica path's 0.82 ms?).
cd /root/repo; mkdir -p gpurun_out/r4z; O=$GRAFT_REPO_ROOT/gpurun_out/r4z
export TMPDIR=/tmp
timeout 120 python tools/op_bench.py --nc 804 --iters 12 --ab tail_low_priority > $O/op_ab.json 2>&1
python -c "import json,sys; d=json.loads(open('$O/op_ab.json').read().strip().splitlines()[-1]); print('op A/B', d['ab'])"
timeout 200 python -m pytest tests/test_parity_r4_gpu.py -m gpu -x -q -k "sweep_schedule" > $O/test.log 2>&1; echo "test rc=$?"; tail -1 $O/test.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('default', round(d['value'],1), round(d['ms_per_step'],1), d['config']['remat_free_layers'], 'ttt bwd', round(r['avg_launch_ms'],3), 'attn bwd', round(r['other']['attn_bwd']['avg_ms'],3), 'fsdp1', d.get('fsdp1'))"
