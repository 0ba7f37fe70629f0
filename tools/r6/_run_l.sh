#!/bin/bash
# round 6, call L: the group-sequential tail (tail5) - oracle / determinism / schedule tests, interleaved A/B against the per-step tail
cd /root/repo; mkdir -p gpurun_out/r6l; O=$GRAFT_REPO_ROOT/gpurun_out/r6l
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py tests/test_parity_r6_gpu.py -x -q -m gpu -k "mlp or mfma or bwd or backward or sweep or regime or pipelined or handover" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for nc in 804 282; do
timeout 300 python tools/op_bench.py --nc $nc --iters 16 --ab tail5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nc $nc tail5 0/1', {k: round(v['bwd_avg_ms'],3) for k,v in d['ab'].items() if isinstance(v, dict)})"
done
timeout 300 python tools/op_bench.py --nc 804 --iters 8 --phases 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ph=d['phase_cycles_per_step']; print('tail5 bwd', round(d['bwd']['avg_ms'],3), {k: round(ph[k]) for k in range(16,36)})"
