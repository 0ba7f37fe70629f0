#!/bin/bash
# gpurun helper: sampling-path measurement (tools/sample_bench.py): 2-layer check first, then the 42-layer model with the
# guidance pair batched / sequential
mkdir -p gpurun_out/sample
timeout 150 python tools/sample_bench.py --layers 2 --steps 1 > gpurun_out/sample/check.json 2> gpurun_out/sample/check.err || { tail -15 gpurun_out/sample/check.err; exit 1; }
cat gpurun_out/sample/check.json
timeout 250 python tools/sample_bench.py --steps 3 2> gpurun_out/sample/batched.err | tee gpurun_out/sample/batched.json
tail -2 gpurun_out/sample/batched.err
timeout 250 python tools/sample_bench.py --steps 3 --sequential 2> gpurun_out/sample/sequential.err | tee gpurun_out/sample/sequential.json
tail -2 gpurun_out/sample/sequential.err
