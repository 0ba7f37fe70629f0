#!/bin/bash
# round 2, call V: FSDP2 over a one-rank mesh (the code path of N > 1): do the all-gather / reduce-scatter copy kernels run on their
# own queues beside the TTT backward sweep?  rocprofv3 kernel trace of 4 layers, one timed step
mkdir -p gpurun_out/r2v
O=$GRAFT_REPO_ROOT/gpurun_out/r2v
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --fsdp on --layers 4 --remat-free-layers 2 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
grep '^{"metric"' /tmp/kt.log | cut -c1-200
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/_fmt_overlap.py "$f" | tee $O/fsdp1_stream_overlap.txt
