#!/bin/bash
# round 2, call O: which PyTorch operators own the elementwise / copy kernels of the 9 s step (torch.profiler, 4 layers, one step)
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout 900 python bench.py --layers 4 --remat-free-layers 2 --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --torch-profile $O/torch_profile_9s_4layers.txt 2>$O/err.txt | cut -c1-300
tail -3 $O/err.txt; wc -l $O/torch_profile_9s_4layers.txt
