#!/bin/bash
# gpurun helper: full GPU validation = pytest -m gpu, smoke, bench line; outputs under gpurun_out/full/
mkdir -p gpurun_out/full
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest_gpu.log 2>&1
tail -4 gpurun_out/full/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/full/smoke.log 2>&1
tail -2 gpurun_out/full/smoke.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
tail -c 2500 gpurun_out/full/bench.json
tail -3 gpurun_out/full/bench.err
