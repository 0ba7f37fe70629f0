#!/bin/bash
# round 2, call A: the whole GPU suite incl. the new reference-pinned parity tests, gelu_pk parity, op-level numbers at the
# 9 s scan length with per-stage cycle stamps, and the new default bench line (9 s config) with a short step count.
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -q -rf -x --deselect tests/test_kernels_gpu.py::test_variant_scan8_gelu_pk_is_bit_identical 2>&1 | tail -40 | tee $O/pytest_gpu.txt
TTT_TEST_VARIANTS=1 timeout 300 python -m pytest tests/test_parity_r2_gpu.py -q -rf -s -k "benchmarked or lastrow or dit_on" 2>&1 | grep -v "^$" | tail -120 > $O/parity_r2_verbose.txt
timeout 200 python tools/op_bench.py --nc 804 --iters 5 --phases 2>/dev/null | tail -1 | tee $O/op_nc804.json
timeout 200 python tools/op_bench.py --nc 282 --iters 5 --phases 2>/dev/null | tail -1 | tee $O/op_nc282.json
timeout 900 python bench.py --steps 2 --warmup 1 2>$O/bench_9s.err | tail -1 > $O/bench_9s.json
tail -c 1500 $O/bench_9s.json; tail -5 $O/bench_9s.err
