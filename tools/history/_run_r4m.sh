#!/bin/bash
# Round 4, call M: forward scan with the half-chunk swap of its LDS tiles (debug option "scan_swap"): correctness, interleaved
# A/B, phase stamps of both variants; and the backward's schedule 2 (recompute beside the sweep) once more on the round-4 sweep.
cd /root/repo; mkdir -p gpurun_out/r4m; O=$GRAFT_REPO_ROOT/gpurun_out/r4m
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_r4_gpu.py -m gpu -x -q -s -k "half_chunk_swap" > $O/test_scan_swap.log 2>&1; echo "test rc=$?"; grep -h "scan_swap 1 vs 0\|passed\|failed\|Error" $O/test_scan_swap.log | tail -5
for r in 1 2; do
  timeout 120 python tools/op_bench.py --nc 804 --fwd-only --iters 30 --ab scan_swap > $O/fwd_ab_nc804_$r.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/fwd_ab_nc804_$r.json').read().strip().splitlines()[-1]); print('fwd A/B nc804', d['ab'])"
done
timeout 120 python tools/op_bench.py --nc 282 --fwd-only --iters 30 --ab scan_swap > $O/fwd_ab_nc282.json 2>&1
python -c "import json,sys; d=json.loads(open('$O/fwd_ab_nc282.json').read().strip().splitlines()[-1]); print('fwd A/B nc282', d['ab'])"
for v in 0 1; do
  timeout 120 python tools/op_bench.py --nc 804 --fwd-only --iters 4 --ab-fixed scan_swap=$v --phases > $O/fwd_phases_swap$v.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/fwd_phases_swap$v.json').read().strip().splitlines()[-1]); print('phases swap=$v', d['fwd']['avg_ms'], d['phase_cycles_per_step'][:16])"
done
# full op (forward + backward) with the swap on: the recompute kernel is unchanged, the backward must not move
timeout 120 python tools/op_bench.py --nc 804 --iters 10 --ab scan_swap > $O/op_ab_nc804.json 2>&1
python -c "import json,sys; d=json.loads(open('$O/op_ab_nc804.json').read().strip().splitlines()[-1]); print('op A/B nc804', d['ab'])"
for ov in 1 2 1 2; do
  timeout 120 python tools/op_bench.py --nc 804 --iters 10 --overlap $ov > $O/bwd_overlap${ov}.json 2>&1
  python -c "import json,sys; d=json.loads(open('$O/bwd_overlap${ov}.json').read().strip().splitlines()[-1]); print('bwd overlap $ov', round(d['bwd']['avg_ms'],3), round(d['bwd']['min_ms'],3))"
done
