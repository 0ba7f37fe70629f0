#!/bin/bash
# gpurun helper: CS = 16 MFMA forward scans (TTT-MLP, TTT-Linear) - parity tests, then timing against the generic kernels
mkdir -p gpurun_out/dbg
timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "cs16" -s 2>&1 | tail -25 > gpurun_out/dbg/cs16_tests.txt
cat gpurun_out/dbg/cs16_tests.txt | cut -c1-400
timeout 100 python tools/cs16_bench.py --linear 2>&1 | tail -4 | tee gpurun_out/dbg/cs16_bench_linear.txt
