#!/bin/bash
# call V: the two-tiles-per-stage attention backward kernels against the shipped ones on the device (bit-identity)
cd /root/repo; mkdir -p gpurun_out/r3v
timeout 45 python -m pytest tests/test_attention_gpu.py -x -q -m gpu -k two_tiles > gpurun_out/r3v/attn_stage2.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r3v/attn_stage2.log | cut -c1-200
