#!/bin/bash
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
timeout 300 python -m pytest tests/test_parity_r2_gpu.py -q -rf -s -k "cluster" 2>&1 | grep -v "^$" | grep -v "^   " | tail -4 | cut -c1-330 | tee $O/cluster_tests.txt
python -c "
import sys; sys.path.insert(0,'ttt-video-dit_amd'); import test_time_training as e; e.load_library(); print('sweep_error', e.sweep_error())" | tee -a $O/cluster_bench.txt
for lead in 1 99; do
  timeout 200 python tools/op_bench.py --nc 282 --iters 5 --cluster -1 --lead $lead --phases 2>/dev/null | tail -1 | python tools/_fmt_phases.py "lead=$lead" | tee -a $O/cluster_bench.txt
done
timeout 200 python tools/op_bench.py --nc 804 --iters 3 --cluster -1 2>/dev/null | tail -1 | python tools/_fmt_phases.py "NC=804" | tee -a $O/cluster_bench.txt
