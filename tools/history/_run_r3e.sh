#!/bin/bash
# round 3, call E: recompute beside the sweep (two-stream schedule of revision 4) and the L2 prefetch touches: parity subset, A/B timing
mkdir -p gpurun_out/r3e
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
( timeout 700 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py -x -q -m gpu -k "mfma_mlp_vs_oracle or bwd_cluster or bwd_tail or at_benchmarked_length_vs_oracle or handover or deterministic" 2>&1 | tail -30 ) > $O/pytest.log; tail -8 $O/pytest.log
for cfg in "1 1" "0 1" "1 0" "0 0"; do
  set -- $cfg
  timeout 120 python tools/op_bench.py --nc 804 --overlap $1 --prefetch $2 --iters 8 --phases 2>/dev/null | python tools/_fmt_phases.py "rev4 nc804 overlap=$1 prefetch=$2" | tee -a $O/ab_nc804.txt
done
timeout 120 python tools/op_bench.py --nc 282 --iters 8 2>/dev/null | python tools/_fmt_phases.py "rev4 nc282" | tee -a $O/ab_nc804.txt
timeout 120 python tools/op_bench.py --nc 2630 --iters 3 2>/dev/null | python tools/_fmt_phases.py "rev4 nc2630 (30 s)" | tee -a $O/ab_nc804.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 5 > /tmp/prof_e.log 2>&1
f=$(find /tmp/prof_e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_kernel_stats.csv && head -6 "$f" | cut -c1-160
