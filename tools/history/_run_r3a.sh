#!/bin/bash
# round 3, call A: the new parity tests (model regime at 48 heads, 30 s / 63 s lengths, multi-scene kernel contract, hand-over failure path)
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_parity_r3_gpu.py -x -q -m gpu -s 2>&1 | tail -80 > gpurun_out/r3a/pytest_r3.log
tail -30 gpurun_out/r3a/pytest_r3.log
