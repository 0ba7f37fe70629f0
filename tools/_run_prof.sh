#!/bin/bash
# gpurun helper: rocprofv3 kernel-trace stats of the default bench command; summary CSV under gpurun_out/prof/
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/bench_prof.json 2> $R/gpurun_out/prof/bench_prof.err
cp $(find /tmp/prof -name "*kernel_stats.csv") $R/gpurun_out/prof/bench_kernel_stats.csv
head -c 700 $R/gpurun_out/prof/bench_prof.json
