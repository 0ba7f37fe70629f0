#!/bin/bash
# gpurun helper (next round): parity of the opt-in kernel variants, then their A/B timing in one call
mkdir -p gpurun_out/ab
TTT_TEST_VARIANTS=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "variant" 2>&1 | tail -5 | tee gpurun_out/ab/variant_tests.txt
for body in "" "--body"; do
  timeout 100 python tools/cs16_bench.py --no-generic --batch 2 $body 2>&1 | grep "^mfma" | sed "s/^/mlp16 body='$body' /" | tee -a gpurun_out/ab/ab.txt
done
for slots in 0 4; do
  timeout 100 python tools/cs16_bench.py --linear --no-generic --lds-slots $slots 2>&1 | grep "bwd" | sed "s/^/lds_slots=$slots /" | tee -a gpurun_out/ab/ab.txt
done
