#!/bin/bash
# Round 4, call N: LDS-only barriers (debug option "light_barriers": bit 0 forward scan, 1 recompute, 2 sweep compute waves):
# bit-identity test, then the op at NC = 804 / 282 per setting (one box, sequential, two rounds), phase stamps of the forward.
cd /root/repo; mkdir -p gpurun_out/r4n; O=$GRAFT_REPO_ROOT/gpurun_out/r4n
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_r4_gpu.py -m gpu -x -q -s -k "lds_only_barriers or half_chunk_swap" > $O/test_light.log 2>&1; echo "test rc=$?"; grep -h "passed\|failed\|Error\|assert" $O/test_light.log | tail -5
for r in 1 2; do for v in 0 1 2 4 7; do
  timeout 120 python tools/op_bench.py --nc 804 --iters 10 --ab-fixed light_barriers=$v > $O/op_nc804_light${v}_$r.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/op_nc804_light${v}_$r.json').read().strip().splitlines()[-1]); print('nc804 light=$v fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3))"
done; done
for v in 0 7; do
  timeout 120 python tools/op_bench.py --nc 282 --iters 10 --ab-fixed light_barriers=$v > $O/op_nc282_light${v}.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/op_nc282_light${v}.json').read().strip().splitlines()[-1]); print('nc282 light=$v fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3))"
done
for v in 0 1; do
  timeout 120 python tools/op_bench.py --nc 804 --fwd-only --iters 4 --ab-fixed light_barriers=$v --phases > $O/fwd_phases_light$v.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/fwd_phases_light$v.json').read().strip().splitlines()[-1]); print('phases light=$v', d['fwd']['avg_ms'], d['phase_cycles_per_step'][:16])"
done
for v in 0 7; do
  timeout 120 python tools/op_bench.py --nc 804 --iters 4 --ab-fixed light_barriers=$v --phases > $O/op_phases_light$v.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/op_phases_light$v.json').read().strip().splitlines()[-1]); print('bwd phases light=$v', d['bwd']['avg_ms'], d['phase_cycles_per_step'])"
done
