#!/bin/bash
# gpurun helper: CS = 16 MFMA forward scan - parity tests, then timing against the generic kernel
mkdir -p gpurun_out/dbg
timeout 150 python -m pytest tests/test_kernels_gpu.py -x -q -k "cs16" -s 2>&1 | tail -25 > gpurun_out/dbg/cs16_tests.txt
cat gpurun_out/dbg/cs16_tests.txt
timeout 100 python tools/cs16_bench.py --phases 2>&1 | tail -8 | tee gpurun_out/dbg/cs16_bench.txt
