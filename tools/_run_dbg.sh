#!/bin/bash
# DEBUG helper for gpurun: revision-2 backward correctness (vs revision 1 / oracle) + per-stage timing
mkdir -p gpurun_out/dbg
timeout 100 python tools/debug_bwd_v2.py > gpurun_out/dbg/debug_bwd2.log 2>&1
grep -E "===|dXK|dW1 |dlast_eta|NaN" gpurun_out/dbg/debug_bwd2.log | head -30
timeout 120 python tools/op_bench.py --phases --iters 3 > gpurun_out/dbg/op_phases.json 2>&1
tail -1 gpurun_out/dbg/op_phases.json
