#!/bin/bash
# gpurun helper: full GPU validation = pytest -m gpu, smoke, bench line, rocprof stats of the bench command, PMC passes;
# outputs under gpurun_out/full/
mkdir -p gpurun_out/full
R=$PWD
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest_gpu.log 2>&1
tail -3 gpurun_out/full/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/full/smoke.log 2>&1
tail -1 gpurun_out/full/smoke.log
timeout 600 python bench.py --steps 2 --warmup 1 2> gpurun_out/full/bench.err | grep '^{"metric' > gpurun_out/full/bench.json
cat gpurun_out/full/bench.json
tail -2 gpurun_out/full/bench.err
if [ "$1" = "prof" ]; then
  export TMPDIR=/tmp
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> $R/gpurun_out/full/bench_prof.err | grep '^{"metric' > $R/gpurun_out/full/bench_prof.json
  cp $(find /tmp/prof -name "*kernel_stats.csv") $R/gpurun_out/full/bench_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_$c -o op -- python $R/tools/op_bench.py --iters 2 > $R/gpurun_out/full/op_$c.log 2>&1
    cp $(find /tmp/pmc_$c -name "*counter_collection.csv") $R/gpurun_out/full/op_pmc_$c.csv
    timeout 200 rocprofv3 --pmc $c --kernel-include-regex "attn_" --output-format csv -d /tmp/pmca_$c -o op -- python $R/tools/attn_bench.py --iters 2 --no-sdpa > $R/gpurun_out/full/attn_$c.log 2>&1
    cp $(find /tmp/pmca_$c -name "*counter_collection.csv") $R/gpurun_out/full/attn_pmc_$c.csv
  done
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "mlp_|attn_" --output-format csv -d /tmp/pmc_sq -o op -- python $R/tools/op_bench.py --iters 2 > $R/gpurun_out/full/op_sq.log 2>&1
  cp $(find /tmp/pmc_sq -name "*counter_collection.csv") $R/gpurun_out/full/op_pmc_sq.csv
  cd $R
  ls gpurun_out/full
fi
