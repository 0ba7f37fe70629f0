#!/bin/bash
# gpurun helper: sampling-path measurement (tools/sample_bench.py).  Usage: _run_sample.sh [video-length ...]
# default: 2-layer check, then the 42-layer model at 3 s with the guidance pair batched / sequential;
# with arguments: one un-warmed denoising step per listed video length (long videos)
mkdir -p gpurun_out/sample
if [ $# -gt 0 ]; then
  for v in "$@"; do
    timeout 400 python tools/sample_bench.py --video-length $v --steps 1 --no-warmup 2> gpurun_out/sample/$v.err | tee gpurun_out/sample/$v.json
    tail -3 gpurun_out/sample/$v.err
  done
  exit 0
fi
timeout 150 python tools/sample_bench.py --layers 2 --steps 1 > gpurun_out/sample/check.json 2> gpurun_out/sample/check.err || { tail -15 gpurun_out/sample/check.err; exit 1; }
cat gpurun_out/sample/check.json
timeout 250 python tools/sample_bench.py --steps 3 2> gpurun_out/sample/batched.err | tee gpurun_out/sample/batched.json
tail -2 gpurun_out/sample/batched.err
timeout 250 python tools/sample_bench.py --steps 3 --sequential 2> gpurun_out/sample/sequential.err | tee gpurun_out/sample/sequential.json
tail -2 gpurun_out/sample/sequential.err
