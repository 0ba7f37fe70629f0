#!/bin/bash
# call O: the 30 s line again (allocator headroom)
cd /root/repo; mkdir -p gpurun_out/r3o; O=gpurun_out/r3o
timeout 420 python bench.py --video-length 30sec --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > $O/bench_30s.json 2> $O/bench_30s.err; echo "30s rc=$?"
cut -c1-900 $O/bench_30s.json; grep "^\[bench\|bench.py:" $O/bench_30s.err | tail -12
