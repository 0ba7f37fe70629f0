#!/bin/bash
# gpurun helper: last validation of a round - the whole GPU suite and smoke(), nothing else
mkdir -p gpurun_out/final
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1
tail -3 gpurun_out/final/pytest_gpu.log
timeout 100 python __graft_entry__.py smoke > gpurun_out/final/smoke.log 2>&1
tail -3 gpurun_out/final/smoke.log
