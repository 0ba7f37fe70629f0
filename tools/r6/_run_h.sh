#!/bin/bash
# round 6, call H: attention backward variants (attn_body.h variant bits: 1 packed fp32 chains, 2 s_setprio around the MFMA clusters,
# 4 static priorities by wave age), interleaved in one process; attention GPU tests on the shipped variant
cd /root/repo; mkdir -p gpurun_out/r6h; O=$GRAFT_REPO_ROOT/gpurun_out/r6h
timeout 600 python tools/attn_bench.py --no-sdpa --iters 6 --variants 0,1,0 > $O/attn_variants.json 2>$O/attn_variants.err; tail -c 1500 $O/attn_variants.json; tail -3 $O/attn_variants.err

