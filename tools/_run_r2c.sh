#!/bin/bash
# round 2, call C: op-level diagnosis on inputs captured inside the DiT; role-specialised cluster backward: parity + timing
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
timeout 300 python tools/_diag_dit2.py 0 2>&1 | grep -v Warning | tail -30 | tee $O/diag_dit2.txt
timeout 300 python -m pytest tests/test_parity_r2_gpu.py -q -rf -s -k "cluster" 2>&1 | grep -v "^$" | tail -70 | cut -c1-400 | tee $O/cluster_tests.txt
python -c "
import sys; sys.path.insert(0,'ttt-video-dit_amd'); import test_time_training as e; e.load_library(); print('sweep_error', e.sweep_error())" | tee -a $O/cluster_bench.txt
for c in 0 -1; do
  timeout 200 python tools/op_bench.py --nc 282 --iters 5 --cluster $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cluster=$c bwd', d['bwd']['avg_ms'], 'ms', d['bwd']['us_per_step'], 'us/step')" | tee -a $O/cluster_bench.txt
done
timeout 200 python tools/op_bench.py --nc 282 --iters 3 --cluster -1 --phases 2>/dev/null | tail -1 | tee -a $O/cluster_bench.txt
timeout 200 python tools/op_bench.py --nc 804 --iters 3 --cluster -1 2>/dev/null | tail -1 | cut -c1-700 | tee -a $O/cluster_bench.txt
