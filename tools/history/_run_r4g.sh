#!/bin/bash
# Round 4, call G: bf16 inner-LayerNorm owner rows (own_bf16): tests + A/B; deriver placement re-check.
cd /root/repo; mkdir -p gpurun_out/r4g; O=$GRAFT_REPO_ROOT/gpurun_out/r4g
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity_r4_gpu.py -q -m gpu -s > $O/tests_variants.log 2>&1; echo "variant tests rc=$?"; tail -2 $O/tests_variants.log | cut -c1-400
for nc in 804 282; do
  timeout 120 python tools/op_bench.py --nc $nc --iters 12 --ab own_bf16 > $O/op_ab_own_bf16_nc$nc.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_own_bf16_nc$nc.json').read().strip().splitlines()[-1]);print('nc$nc own_bf16',d['ab'])"
done
timeout 120 python tools/op_bench.py --nc 804 --iters 12 --ab sweep_deriver_wave0 > $O/op_ab_dw_nc804.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_dw_nc804.json').read().strip().splitlines()[-1]);print('nc804 deriver_wave0 (0 = waves 4,5; 1 = waves 2,3)',d['ab'])"
