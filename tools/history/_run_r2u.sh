#!/bin/bash
# round 2, call U: final tree (token-team LayerNorm backward kernels): model-level GPU tests that run them + the 9 s bench line
mkdir -p gpurun_out/r2u
O=gpurun_out/r2u
timeout 900 python -m pytest tests/test_parity_r2_gpu.py tests/test_kernels_gpu.py tests/test_zz_replica_gpu.py -m gpu -q -rf -k "dit_on_hip or transformer_layer or cogvideox or fused_module or prepost or replica or head_sharded" 2>&1 | tail -5 | cut -c1-300 | tee $O/pytest_model.txt
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -3; cut -c1-330 $O/bench_9s.json
