#!/bin/bash
# call U: the "fc2" keep kind on the device (bit-identity inside the DiT)
cd /root/repo; mkdir -p gpurun_out/r3u
timeout 60 python -m pytest tests/test_parity_r3_gpu.py -x -q -m gpu -k remat_keep > gpurun_out/r3u/remat_keep.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3u/remat_keep.log | cut -c1-200
