#!/bin/bash
# round 2, call Z: the 3 s line (BASELINE configs[1]) on the final tree
mkdir -p gpurun_out/r2z
timeout 300 python bench.py --video-length 3sec --steps 5 --warmup 2 --no-cpu-baseline --no-fsdp1-compare 2>/dev/null | grep '^{"metric"' > gpurun_out/r2z/bench_3s.json
cut -c1-260 gpurun_out/r2z/bench_3s.json
