#!/bin/bash
# round 2, call X: GPU idle gaps inside the timed step of the replica path (4 layers)
mkdir -p gpurun_out/r2x
O=$GRAFT_REPO_ROOT/gpurun_out/r2x
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --fsdp off --layers 4 --remat-free-layers 2 --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > /tmp/kt.log 2>&1
ms=$(grep '^{"metric"' /tmp/kt.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "replica path, 4 layers: $ms ms/step" | tee $O/gaps_replica.txt
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/_fmt_gaps.py "$f" $(python -c "print($ms/1e3)") | tee -a $O/gaps_replica.txt
