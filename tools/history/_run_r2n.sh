#!/bin/bash
# round 2, call N: LayerNorm backward glue kernels (four tokens per block iteration, loads one group ahead): tests + timing
mkdir -p gpurun_out/r2n
O=$GRAFT_REPO_ROOT/gpurun_out/r2n
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -s -k "layernorm_backward_kernels or adaln or prepost or fused_glue or gate" 2>&1 | grep -v "^$" | tail -14 | cut -c1-600 | tee $O/pytest_glue.txt
timeout 300 python tools/glue_bench.py 2>/dev/null | tee $O/glue_bench_9s.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gs -- python $GRAFT_REPO_ROOT/tools/glue_bench.py > /tmp/gs.log 2>&1
f=$(find /tmp/gs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep "prepost" "$f" | cut -d, -f1-4 | tee $O/glue_kernel_stats.txt
