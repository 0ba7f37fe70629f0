#!/bin/bash
# round 6, call Y: pair scan version 3 (version 2 + the next tiles parked behind B2)
cd /root/repo; mkdir -p gpurun_out/r6y; O=gpurun_out/r6y
timeout 600 python -X faulthandler -m pytest tests/test_scan_pair_gpu.py tests/test_parity_r5_gpu.py -x -q -m gpu > $O/pair_tests.log 2>&1; echo "pair tests rc=$?"; tail -3 $O/pair_tests.log
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --fwd-only --ab scan_pair --phases > $O/op_nc804_ab_scan_pair.json 2>$O/op.err; tail -1 $O/op_nc804_ab_scan_pair.json
timeout 300 python tools/op_bench.py --nc 804 --iters 6 --fwd-only --ab-fixed scan_pair=0 --phases > $O/op_nc804_single_phases.json 2>>$O/op.err; tail -1 $O/op_nc804_single_phases.json
