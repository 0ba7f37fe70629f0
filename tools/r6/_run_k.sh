#!/bin/bash
# round 6, call K: CU-mask mapping probe + the sweep beside a masked / unmasked GEMM stream; the new masked-stream test
cd /root/repo; mkdir -p gpurun_out/r6k; O=$GRAFT_REPO_ROOT/gpurun_out/r6k
timeout 600 python tools/cu_mask_probe.py > $O/cu_mask_probe.json 2> $O/probe.err; echo "probe rc=$?"; tail -3 $O/probe.err; cat $O/cu_mask_probe.json | head -120
timeout 600 python -m pytest tests/test_parity_r6_gpu.py -x -q -m gpu -k masked 2>&1 | tail -3
