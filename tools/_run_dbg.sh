#!/bin/bash
# gpurun helper: CS = 16 MFMA scans (TTT-MLP forward, TTT-Linear forward + backward) - parity tests, then timing
mkdir -p gpurun_out/dbg
timeout 250 python -m pytest tests/test_kernels_gpu.py -x -q -k "cs16 or lin" -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/dbg/cs16_tests.txt
cut -c1-330 gpurun_out/dbg/cs16_tests.txt
timeout 100 python tools/cs16_bench.py --linear 2>&1 | tail -6 | tee gpurun_out/dbg/cs16_bench_linear.txt
