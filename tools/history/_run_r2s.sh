#!/bin/bash
# round 2, call S: the driver's command on the final tree
mkdir -p gpurun_out/r2s
O=gpurun_out/r2s
timeout 1200 python bench.py --gpus 1 --steps 3 --warmup 1 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -6; cut -c1-300 $O/bench_9s.json
