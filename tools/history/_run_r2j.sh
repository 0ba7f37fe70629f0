#!/bin/bash
# (historical: the --overlap modes / --side-wgs option of tools/op_bench.py that this call exercised were removed with the schedule they tested; results under profiles/)
# round 2, call J: A/B of the backward's recompute-under-sweep schedule (op level, 3 s and 9 s scan lengths), the new GPU tests,
# and the bench lines with the schedule + the 9 s GEMM selections + the 0.92 memory cap
mkdir -p gpurun_out/r2j
O=gpurun_out/r2j
timeout 600 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q -rf -s -k "recompute_under_sweep or head_sharded or handover" 2>&1 | grep -v "^$" | tail -22 | cut -c1-400 | tee $O/pytest_new.txt
for nc in 282 804; do
  for ov in 0 1 2; do
    echo "== nc $nc overlap $ov" | tee -a $O/op_ab.txt
    timeout 300 python tools/op_bench.py --nc $nc --overlap $ov --iters 6 2>/dev/null | python tools/_fmt_phases.py "  " | tee -a $O/op_ab.txt
  done
done
for gpc in 3 8; do
  echo "== nc 804 overlap 1 gpc $gpc" | tee -a $O/op_ab.txt
  timeout 300 python tools/op_bench.py --nc 804 --overlap 1 --gpc $gpc --iters 6 2>/dev/null | python tools/_fmt_phases.py "  " | tee -a $O/op_ab.txt
done
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -6; cut -c1-700 $O/bench_9s.json
timeout 600 python bench.py --video-length 3sec --steps 5 --warmup 2 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_3s.err | grep '^{"metric"' > $O/bench_3s.json
cut -c1-400 $O/bench_3s.json
ls -la $O
