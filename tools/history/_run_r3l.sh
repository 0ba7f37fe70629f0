#!/bin/bash
# round 3, call L: the remaining BASELINE configurations on the final tree: 18 s stage (first run), 30 s stage, 63 s sampling; the
# fsdp1 point (FSDP2 over a one-rank mesh = the code path of N > 1) beside the replica value at 9 s
mkdir -p gpurun_out/r3l
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
for v in 18sec 30sec; do
  timeout 900 python bench.py --video-length $v --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_$v.err | grep '^{"metric"' > $O/bench_$v.json
  grep "timed region" $O/bench_$v.err | tail -1; python -c "
import json; d=json.load(open('$O/bench_$v.json')); r=d['roofline']; print('$v', round(d['value'],1), round(d['ms_per_step'],1), 'L', d['config']['seq_len'], 'free', d['config']['remat_free_layers'], 'mem', round(d['peak_mem_gib'],1), 'bwd ms', round(r['avg_launch_ms'],2), 'valid', d['config']['valid'])"
done
timeout 600 python tools/sample_bench.py --video-length 63sec --steps 3 2>$O/sample_63s.err | tail -1 > $O/sample_63s.json; cut -c1-500 $O/sample_63s.json
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>$O/bench_9s_fsdp1.err | grep '^{"metric"' > $O/bench_9s_fsdp1.json
python -c "
import json; d=json.load(open('$O/bench_9s_fsdp1.json')); print('9s replica1', round(d['value'],1), round(d['ms_per_step'],1), 'fsdp1', d.get('fsdp1'))"
