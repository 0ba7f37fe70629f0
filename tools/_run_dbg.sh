#!/bin/bash
# gpurun helper: hipBLASLt / rocBLAS GEMM selection through PyTorch TunableOp on the bench's GEMM shapes
mkdir -p gpurun_out/dbg
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_TUNING=1
export PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/dbg/tunableop_results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=10
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/dbg/tune.err | grep '^{"metric' > gpurun_out/dbg/bench_tuned.json
python -c "import json; d=json.loads(open('gpurun_out/dbg/bench_tuned.json').read().split('\n')[0]); print('tuned run', d['value'], d['ms_per_step'], d['config']['remat_free_layers'])"
ls -la gpurun_out/dbg/ | head; wc -l gpurun_out/dbg/tunableop_results*.csv; tail -3 gpurun_out/dbg/tune.err | cut -c1-200
