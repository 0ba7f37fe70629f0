#!/bin/bash
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
timeout 300 python -m pytest tests/test_parity_r2_gpu.py -q -rf -s -k "cluster" 2>&1 | grep -v "^$" | grep -v "^   " | tail -30 | cut -c1-400 | tee $O/cluster_tests.txt
python -c "
import sys; sys.path.insert(0,'ttt-video-dit_amd'); import test_time_training as e; e.load_library(); print('sweep_error', e.sweep_error())" | tee -a $O/cluster_bench.txt
for c in 0 -1; do
  timeout 200 python tools/op_bench.py --nc 282 --iters 5 --cluster $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cluster=$c bwd', d['bwd']['avg_ms'], 'ms', d['bwd']['us_per_step'], 'us/step')" | tee -a $O/cluster_bench.txt
done
timeout 200 python tools/op_bench.py --nc 282 --iters 3 --cluster -1 --phases 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('phases', d['phase_cycles_per_step'][16:24], 'recompute', d['phase_cycles_per_step'][:14])" | tee -a $O/cluster_bench.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_op -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 282 --iters 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_op -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200 | tee $O/op_kernel_stats.txt
