#!/bin/bash
# round 2, call B: DiT gradient diagnosis, cluster-form backward (quick version) parity + timing, 9 s bench line
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 200 python tools/_diag_dit.py 2>&1 | grep -v Warning | tail -12 | tee $O/diag_dit.txt
timeout 300 python -m pytest tests/test_parity_r2_gpu.py -q -rf -s -k "cluster" 2>&1 | grep -v "^$" | tail -60 | tee $O/cluster_tests.txt
for c in 0 -1; do
  timeout 200 python tools/op_bench.py --nc 282 --iters 5 --cluster $c 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cluster=$c bwd', d['bwd']['avg_ms'], 'ms', d['bwd']['us_per_step'], 'us/step')" | tee -a $O/cluster_bench.txt
done
timeout 200 python tools/op_bench.py --nc 282 --iters 3 --cluster -1 --phases 2>/dev/null | tail -1 | tee -a $O/cluster_bench.txt
python -c "
import sys; sys.path.insert(0,'ttt-video-dit_amd'); import test_time_training as e; e.load_library(); print('sweep_error', e.sweep_error())" | tee -a $O/cluster_bench.txt
timeout 1200 python bench.py --steps 2 --warmup 1 2>$O/bench_9s.err | tail -1 > $O/bench_9s.json
tail -c 1200 $O/bench_9s.json; grep "bench " $O/bench_9s.err | tail -12
