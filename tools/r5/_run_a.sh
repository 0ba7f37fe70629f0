#!/bin/bash
# round 5, call A: the single-pass attention backward on the device - parity tests, then the interleaved A/B against the kernel pair
cd /root/repo; mkdir -p gpurun_out/r5a; O=$GRAFT_REPO_ROOT/gpurun_out/r5a
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_attention_gpu.py -x -q -m gpu > $O/attn_tests.log 2>&1; echo "attn tests rc=$?"; tail -5 $O/attn_tests.log
timeout 300 python tools/attn_bench.py --fused --no-sdpa --rounds 5 --iters 5 > $O/attn_fused_ab.json 2> $O/attn_fused_ab.err; echo "ab rc=$?"; cat $O/attn_fused_ab.json; tail -3 $O/attn_fused_ab.err
