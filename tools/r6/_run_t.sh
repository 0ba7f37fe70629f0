#!/bin/bash
# round 6, call T: the forward scan as a pair of workgroups - bit identity against the one-workgroup kernel, op-level A/B, stage stamps
cd /root/repo
mkdir -p gpurun_out/r6t
timeout 900 python -m pytest tests/test_scan_pair_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6t/tests.log
cat gpurun_out/r6t/tests.log
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --fwd-only --ab scan_pair --phases > gpurun_out/r6t/op_nc804_ab_scan_pair.json 2>gpurun_out/r6t/op.err
tail -1 gpurun_out/r6t/op_nc804_ab_scan_pair.json
timeout 300 python tools/op_bench.py --nc 804 --iters 6 --fwd-only --ab-fixed scan_pair=0 --phases > gpurun_out/r6t/op_nc804_single_phases.json 2>>gpurun_out/r6t/op.err
tail -1 gpurun_out/r6t/op_nc804_single_phases.json
timeout 300 python tools/op_bench.py --nc 282 --iters 12 --fwd-only --ab scan_pair > gpurun_out/r6t/op_nc282_ab_scan_pair.json 2>>gpurun_out/r6t/op.err
tail -1 gpurun_out/r6t/op_nc282_ab_scan_pair.json
tail -5 gpurun_out/r6t/op.err
