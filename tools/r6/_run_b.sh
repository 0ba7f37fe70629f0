#!/bin/bash
# round 6, call B: the new parity tests (default pipelined path vs the reference fixture, 48-head pipelined regime, split sweep same
# bits), then the deriver-split A/B of the TTT-MLP backward (barrier Bc inside the reverse step) interleaved in one process, stamps
cd /root/repo; mkdir -p gpurun_out/r6b; O=$GRAFT_REPO_ROOT/gpurun_out/r6b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_r6_gpu.py -x -q -m gpu -s > $O/r6_tests.log 2>&1; echo "r6 tests rc=$?"; grep -E "passed|failed|error|vs reference|Error" $O/r6_tests.log | tail -12
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --ab deriver_split > $O/op_nc804_ab_split.json 2>$O/op_ab.err; echo "ab rc=$?"; tail -c 900 $O/op_nc804_ab_split.json
for ov in 1 2; do
timeout 200 python tools/op_bench.py --nc 804 --iters 8 --overlap $ov --phases > $O/op_nc804_split_ov${ov}_phases.json 2>/dev/null
python - <<PY
import json
d=[json.loads(l) for l in open("$O/op_nc804_split_ov${ov}_phases.json") if l.startswith("{")][0]
print("split on, overlap $ov: bwd avg %.3f min %.3f" % (d["bwd"]["avg_ms"], d["bwd"]["min_ms"]))
ph=d["phase_cycles_per_step"]; print({k: round(ph[k]) for k in range(16,36)})
PY
done
timeout 200 python tools/op_bench.py --nc 282 --iters 8 --ab deriver_split 2>/dev/null | tail -c 600
ls $O
