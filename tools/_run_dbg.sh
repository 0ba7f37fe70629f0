#!/bin/bash
# gpurun helper: the bench's reference-settings flags and the 9 s configuration still work
mkdir -p gpurun_out/dbg
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --remat-free-layers 0 --reshard-after-forward --no-tuned-gemms 2> gpurun_out/dbg/ref.err | grep '^{"metric' > gpurun_out/dbg/bench_refsettings.json
python -c "import json; d=json.loads(open('gpurun_out/dbg/bench_refsettings.json').read()); print('reference settings:', round(d['value'],1), round(d['ms_per_step'],1), d['config']['remat_free_layers'], d['config']['fsdp_reshard_after_forward'], d['config']['tuned_gemm_selections'], round(d['peak_mem_gib'],1), d['loss'])"
timeout 700 python bench.py --video-length 9sec --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/dbg/b9.err | grep '^{"metric' > gpurun_out/dbg/bench9.json
python -c "import json; d=json.loads(open('gpurun_out/dbg/bench9.json').read()); print('9 s:', round(d['value'],1), round(d['ms_per_step'],1), d['config']['remat_free_layers'], round(d['peak_mem_gib'],1), d['loss'])"
