#!/bin/bash
# round 3, call B: first run of the revision-4 TTT-MLP backward (slim step record + deriver waves): parity, then A/B timing
mkdir -p gpurun_out/r3b
( timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py -x -q -m gpu -k "mfma_mlp_vs_oracle or bwd_cluster or bwd_tail or at_benchmarked_length_vs_oracle or handover or model_regime" -s 2>&1 | tail -120 ) > gpurun_out/r3b/pytest.log
tail -40 gpurun_out/r3b/pytest.log
for rev in 3 4; do
  for nc in 282 804; do
    timeout 200 python tools/op_bench.py --nc $nc --bwd-rev $rev --iters 10 2>&1 | tail -3 > gpurun_out/r3b/op_rev${rev}_nc${nc}.txt
    echo "rev $rev nc $nc"; cat gpurun_out/r3b/op_rev${rev}_nc${nc}.txt
  done
done
