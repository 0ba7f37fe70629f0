#!/bin/bash
# Round 4, call O: the eta prefetch of the forward scan / recompute without its early conversion (no vmcnt(0) behind the tile
# loads): library before (lib/libttt_hip_prev.so, built from the parent commit) against after, same box, alternating; then
# correctness of the new build (oracle parity tests of the TTT-MLP op) and the forward's phase stamps.
cd /root/repo; mkdir -p gpurun_out/r4o; O=$GRAFT_REPO_ROOT/gpurun_out/r4o
export TMPDIR=/tmp
L=ttt-video-dit_amd/lib
cp $L/libttt_hip.so /tmp/new.so; cp $L/libttt_hip_prev.so /tmp/prev.so
for r in 1 2; do for which in prev new; do
  cp /tmp/$which.so $L/libttt_hip.so
  for sw in 0 1; do
    timeout 120 python tools/op_bench.py --nc 804 --iters 10 --ab-fixed scan_swap=$sw > $O/op_nc804_${which}_swap${sw}_$r.json 2>&1
    python -c "import json,sys; d=json.loads(open('$O/op_nc804_${which}_swap${sw}_$r.json').read().strip().splitlines()[-1]); print('nc804 $which swap=$sw fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3))"
  done
done; done
cp /tmp/new.so $L/libttt_hip.so
timeout 120 python tools/op_bench.py --nc 282 --iters 10 --ab-fixed scan_swap=1 > $O/op_nc282_new.json 2>&1
python -c "import json,sys; d=json.loads(open('$O/op_nc282_new.json').read().strip().splitlines()[-1]); print('nc282 new swap=1 fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3))"
timeout 120 python tools/op_bench.py --nc 804 --fwd-only --iters 4 --ab-fixed scan_swap=1 --phases > $O/fwd_phases_new.json 2>&1
python -c "import json,sys; d=json.loads(open('$O/fwd_phases_new.json').read().strip().splitlines()[-1]); print('phases new swap=1', d['fwd']['avg_ms'], d['phase_cycles_per_step'][:16])"
timeout 900 python -m pytest tests/test_parity_r4_gpu.py tests/test_parity_r2_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
