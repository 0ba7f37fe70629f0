#!/bin/bash
# round 2, call M: tail kernel of a chunk on a side stream under the next sweep - tests, A/B, kernel trace
mkdir -p gpurun_out/r2m
O=$GRAFT_REPO_ROOT/gpurun_out/r2m
timeout 600 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q -rf -k "tail_under_next_sweep" 2>&1 | tail -5 | cut -c1-400 | tee $O/pytest_new.txt
for cfg in "282 0 0" "282 1 0" "804 0 0" "804 1 0" "804 1 4" "804 1 8" "282 1 4" "282 1 3"; do
  set -- $cfg
  timeout 300 python tools/op_bench.py --nc $1 --overlap $2 --gpc $3 --iters 5 2>/dev/null | python tools/_fmt_phases.py "nc $1 overlap $2 gpc $3:" | tee -a $O/op_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --overlap 1 --iters 2 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/_fmt_trace.py "$f" > $O/overlap_trace.txt
head -40 $O/overlap_trace.txt; tail -2 $O/overlap_trace.txt
