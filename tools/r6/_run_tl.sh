#!/bin/bash
# round 6, call TL: kernel timeline of one TTT-MLP backward at NC = 804 (sweeps on the main queue; recompute / tail beside them)
cd /root/repo; mkdir -p gpurun_out/r6tl; O=$GRAFT_REPO_ROOT/gpurun_out/r6tl
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 4 > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); echo $f
python $GRAFT_REPO_ROOT/tools/_fmt_bwd_timeline.py $f > $O/bwd_timeline_nc804.txt; cat $O/bwd_timeline_nc804.txt
