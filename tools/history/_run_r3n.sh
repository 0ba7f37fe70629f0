#!/bin/bash
# call N: the whole GPU suite on the final tree (incl. the one-rank TP test), then the 30 s line on the replica path
cd /root/repo; mkdir -p gpurun_out/r3n; O=gpurun_out/r3n
timeout 420 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite.log
timeout 60 python -m pytest tests/test_zz_replica_gpu.py -q -m gpu -s -k tensor_parallel 2>&1 | grep -E "LAYOUT|median" > $O/tp_layouts.txt
timeout 400 python bench.py --video-length 30sec --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > $O/bench_30s.json 2> $O/bench_30s.err; echo "30s rc=$?"
tail -12 $O/gpu_suite.log; cut -c1-400 $O/tp_layouts.txt; cut -c1-700 $O/bench_30s.json; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/bench_30s.err | tail -8
