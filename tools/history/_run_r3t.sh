#!/bin/bash
# call T: cluster sweep with CUs stolen by another stream (what RCCL kernels do at N > 1)
cd /root/repo; mkdir -p gpurun_out/r3t
timeout 75 python -m pytest tests/test_parity_r3_gpu.py -x -q -m gpu -s -k stolen > gpurun_out/r3t/stolen_cus.log 2>&1; echo "rc=$?"
grep -E "stolen=|passed|failed|Error|assert" gpurun_out/r3t/stolen_cus.log | cut -c1-220 | tail -12
