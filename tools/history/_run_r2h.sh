#!/bin/bash
# round 2, call H: BASELINE configs 4 (30 s stage) and 5 (63 s sampling, warmed), overlap-wgrad and local-batch-2 A/B at 3 s
mkdir -p gpurun_out/r2h
O=gpurun_out/r2h
timeout 200 python -m pytest tests/test_parity_r2_gpu.py -q -k "dit_on" 2>&1 | tail -3 | tee $O/dit_test.txt
timeout 900 python bench.py --video-length 30sec --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_30s.err | grep '^{"metric"' > $O/bench_30s.json
grep "bench " $O/bench_30s.err | tail -3; cut -c1-700 $O/bench_30s.json
timeout 600 python tools/sample_bench.py --video-length 63sec --steps 3 2>$O/sample_63s.err | tail -1 > $O/sample_63s.json; cut -c1-600 $O/sample_63s.json
timeout 400 python bench.py --video-length 3sec --steps 4 --warmup 2 --no-cpu-baseline --no-fsdp1-compare --overlap-wgrad 2>$O/bench_3s_wgrad.err | grep '^{"metric"' > $O/bench_3s_wgrad.json
python -c "
import json; d=json.loads(open('$O/bench_3s_wgrad.json').read()); print('overlap-wgrad 3s:', d['value'], d['ms_per_step'], 'bwd scan', d['roofline']['avg_launch_ms'])"
timeout 600 python bench.py --video-length 3sec --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --local-batch 2 2>$O/bench_3s_lb2.err | grep '^{"metric"' > $O/bench_3s_lb2.json
python -c "
import json; d=json.loads(open('$O/bench_3s_lb2.json').read()); print('local batch 2, 3s:', d['value'], d['ms_per_step'], 'remat-free', d['config']['remat_free_layers'], 'mem', d['peak_mem_gib'], 'bwd scan', d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
tail -3 $O/bench_3s_lb2.err
