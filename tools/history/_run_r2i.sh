#!/bin/bash
# round 2, call I: hipBLASLt / rocBLAS solution selection for the GEMM shapes of the 9 s configuration (PyTorch TunableOp, one layer
# is enough: every distinct shape of the model occurs in it), as round 1 did for the 3 s shapes
mkdir -p gpurun_out/r2i
O=gpurun_out/r2i
export PYTORCH_TUNABLEOP_ENABLED=1
export PYTORCH_TUNABLEOP_TUNING=1
export PYTORCH_TUNABLEOP_FILENAME=$PWD/$O/tunableop_9s.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
timeout 1100 python bench.py --layers 1 --remat-free-layers 1 --steps 1 --warmup 0 --no-cpu-baseline --no-fsdp1-compare --no-tuned-gemms 2> $O/tune.err | grep '^{"metric' | cut -c1-200
ls -la $O; wc -l $O/tunableop_9s*.csv; tail -3 $O/tune.err
