#!/bin/bash
# round 3, call F: one-box A/B of the revision-4 schedule (0 one stream / 1 tail beside the sweep / 2 tail + recompute beside),
# the placement of the L2 prefetch touches (0 off / 1 before Bc / 2 owners behind Bc) and non-temporal record stores
mkdir -p gpurun_out/r3f
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
run() { timeout 100 python tools/op_bench.py --nc 804 --overlap $1 --prefetch $2 --rc-nt $3 --iters 8 $4 2>/dev/null | python tools/_fmt_phases.py "ov=$1 pf=$2 nt=$3" | tee -a $O/ab_nc804.txt; }
run 1 0 0 --phases
run 1 2 0 --phases
run 1 1 0
run 2 0 0
run 2 2 0
run 2 2 1 --phases
run 1 2 1
run 0 2 0
timeout 100 python tools/op_bench.py --nc 804 --bwd-rev 3 --iters 8 2>/dev/null | python tools/_fmt_phases.py "rev3" | tee -a $O/ab_nc804.txt
