#!/bin/bash
# Round 4, the last GPU seconds: forward with 64 query rows per wave (attn_fwd_wide): opt-in device test, then the interleaved A/B.
cd /root/repo; mkdir -p gpurun_out/r4zb; O=$GRAFT_REPO_ROOT/gpurun_out/r4zb
export TMPDIR=/tmp
TTT_TEST_VARIANTS=1 timeout 40 python -m pytest tests/test_attention_gpu.py -m gpu -x -q -s -k forward_wide > $O/test.log 2>&1; echo "test rc=$?"; grep -h "attn_fwd_wide=\|passed\|failed\|Error" $O/test.log | tail -10
timeout 45 python tools/attn_bench.py --no-sdpa --fwd-wide --rounds 5 --iters 5 > $O/attn_fwd_wide_ab.json 2>&1; tail -1 $O/attn_fwd_wide_ab.json | cut -c1-900
