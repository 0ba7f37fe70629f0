#!/bin/bash
# round 6, call HO12: with the copies out batched per layer (56 GB/s): the 9 s step with 2 / 3 GiB per free layer parked, interleaved with the baseline; then the 63 s step with parked kernel outputs
cd /root/repo; mkdir -p gpurun_out/r6ho12; O=gpurun_out/r6ho12
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; h=c.get('host_offload') or {}; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'keep', c['remat_keep'], c.get('remat_keep_limits'), 'gib', h.get('gib_per_step'), 'trace', h.get('trace'), 'host_s', h.get('host_s_per_step'), 'waits', h.get('throttle_waits'), 'dom', round(r['avg_launch_ms'],3), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || grep -h "OutOfMemoryError: HIP" ${1%.json}.err | tail -1 | cut -c1-300; }
run() { timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
run base1
run off2 --offload-trace --offload-gib-per-layer 2 --offload-backlog-gib 24
run off3 --offload-trace --offload-gib-per-layer 3 --offload-backlog-gib 24
run base2
run off4 --offload-trace --offload-gib-per-layer 4 --offload-backlog-gib 32
run63() { timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 "${@:2}" > $O/bench63_$1.json 2> $O/bench63_$1.err; show $O/bench63_$1.json 63$1; }
run63 parkattn --remat-keep attn --offload-park-kept --offload-trace --offload-lookahead 2
run63 base --remat-keep attn --remat-keep-layers 10
run63 parkattnscan16 --remat-keep attn,scan:16 --offload-park-kept --offload-trace --offload-lookahead 1
