#!/bin/bash
# gpurun helper: glue-kernel tests
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "adaln or fused_glue" 2>&1 | tail -25
