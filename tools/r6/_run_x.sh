#!/bin/bash
# round 6, call X: kernel timeline of one pipelined TTT layer forward (4 parts, pair scan)
cd /root/repo; mkdir -p gpurun_out/r6x; O=$GRAFT_REPO_ROOT/gpurun_out/r6x
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/tools/ttt_layer_bench.py --parts 4 --rounds 1 > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); echo $f
python $GRAFT_REPO_ROOT/tools/_fmt_layer_fwd.py $f 4 > $O/layer_fwd_timeline_parts4.txt; cat $O/layer_fwd_timeline_parts4.txt
