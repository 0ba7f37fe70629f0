#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r6t
timeout 900 python -X faulthandler -m pytest tests/test_scan_pair_gpu.py -x -v -m gpu > gpurun_out/r6t/tests_full.log 2>&1
head -60 gpurun_out/r6t/tests_full.log
dmesg 2>/dev/null | tail -5
