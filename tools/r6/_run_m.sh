#!/bin/bash
# round 6, call M: kernel statistics of the TTT-MLP backward under rocprofv3 --kernel-trace --stats, per-step tail vs group-sequential tail
cd /root/repo; mkdir -p gpurun_out/r6m; O=$GRAFT_REPO_ROOT/gpurun_out/r6m
export TMPDIR=/tmp; cd /tmp
for t5 in 0 1; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$t5 -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 6 --ab-fixed tail5=$t5 > /dev/null 2>&1
f=$(find /tmp/kt_$t5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_tail5_${t5}_kernel_stats.csv && echo "tail5=$t5" && head -8 "$f" | cut -c1-200
done
