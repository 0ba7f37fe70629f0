#!/bin/bash
# (historical: the --overlap modes / --side-wgs option of tools/op_bench.py that this call exercised were removed with the schedule they tested; results under profiles/)
# round 2, call L: the recompute beside the sweep in launches limited to the CUs the sweep leaves free - tests, kernel trace, A/B
mkdir -p gpurun_out/r2l
O=$GRAFT_REPO_ROOT/gpurun_out/r2l
timeout 600 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q -rf -k "recompute_under_sweep" 2>&1 | tail -5 | cut -c1-400 | tee $O/pytest_new.txt
for cfg in "282 0 0 0" "282 1 0 0" "804 0 0 0" "804 1 0 0" "804 1 32 0" "804 1 48 0" "804 1 96 0" "804 1 0 4" "804 1 0 8" "282 1 0 4"; do
  set -- $cfg
  timeout 300 python tools/op_bench.py --nc $1 --overlap $2 --side-wgs $3 --gpc $4 --iters 5 2>/dev/null | python tools/_fmt_phases.py "nc $1 overlap $2 side_wgs $3 gpc $4:" | tee -a $O/op_ab.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --overlap 1 --iters 2 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/_fmt_trace.py "$f" > $O/overlap_trace.txt
head -60 $O/overlap_trace.txt
