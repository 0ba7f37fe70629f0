#!/bin/bash
# round 3, call D: revision 4 with the deriver rework (R4 parking area, no register-held fragments, derivers beside the compute
# waves): determinism, stage cycle stamps, A/B timing, kernel trace of both revisions
mkdir -p gpurun_out/r3d
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
timeout 200 python tools/_det_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > $O/det.txt; cut -c1-700 $O/det.txt
for rev in 4 3; do
  timeout 120 python tools/op_bench.py --nc 804 --bwd-rev $rev --iters 5 --phases 2>/dev/null | python tools/_fmt_phases.py "rev$rev nc804" | tee $O/phases_rev${rev}_nc804.txt
done
timeout 120 python tools/op_bench.py --nc 282 --bwd-rev 4 --iters 5 2>/dev/null | python tools/_fmt_phases.py "rev4 nc282" | tee $O/phases_rev4_nc282.txt
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py -x -q -m gpu -k "mfma_mlp_vs_oracle or bwd_cluster or bwd_tail or at_benchmarked_length_vs_oracle or handover" 2>&1 | tail -30 ) > $O/pytest.log; tail -12 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for rev in 4 3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rev$rev -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --bwd-rev $rev --iters 5 > /tmp/prof_rev$rev.log 2>&1
  f=$(find /tmp/prof_rev$rev -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_rev${rev}_kernel_stats.csv && echo "== rev $rev" && head -8 "$f" | cut -c1-160
done
