#!/bin/bash
# round 6, call C: r6 parity tests again; where barrier Bc falls inside the derivers' reverse step (1: behind the W2 update, 2: between the
# token tiles) interleaved in one process under both schedules; stage stamps at position 1
cd /root/repo; mkdir -p gpurun_out/r6c; O=$GRAFT_REPO_ROOT/gpurun_out/r6c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_r6_gpu.py -x -q -m gpu -s > $O/r6_tests.log 2>&1; echo "r6 tests rc=$?"; grep -E "passed|failed|error|vs reference|Error" $O/r6_tests.log | tail -12
for ov in 1 2; do
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --overlap $ov --ab deriver_split --ab-values 1,2 --ab-restore 1 --phases > $O/op_nc804_ab_split12_ov$ov.json 2>$O/op_ab.err
python - <<PY
import json
d=[json.loads(l) for l in open("$O/op_nc804_ab_split12_ov$ov.json") if l.startswith("{")][0]
print("overlap $ov", d["ab"])
ph=d["phase_cycles_per_step"]; print("stamps at split=1:", {k: round(ph[k]) for k in range(16,36)})
PY
done
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --overlap 2 --ab deriver_split --ab-values 0,1 2>/dev/null | python -c "import sys,json; print('ov2 split 0 vs 1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ab'])"
