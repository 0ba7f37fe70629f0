#!/bin/bash
# round 6, call GP: GPU idle time inside the training step (kernel trace of a short worker run)
cd /root/repo; mkdir -p gpurun_out/r6gp; O=$GRAFT_REPO_ROOT/gpurun_out/r6gp
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --role worker --gpus 1 --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench.json 2> $O/bench.err
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); echo $f
grep -h "^{" $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
python $GRAFT_REPO_ROOT/tools/step_gaps.py $f --window-s 11.0 | tee $O/step_gaps.txt
