#!/bin/bash
# round 6, call HO14: one copy stream, no host wait at the end of the forward: 63 s with attention + scan (+ MLP) outputs parked, 30 s with everything kept parked, 9 s with 3 GiB per free layer
cd /root/repo; mkdir -p gpurun_out/r6ho14; O=gpurun_out/r6ho14
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; h=c.get('host_offload') or {}; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'keep', c['remat_keep'], c.get('remat_keep_limits'), 'gib', h.get('gib_per_step'), 'host_s', h.get('host_s_per_step'), 'waits', h.get('throttle_waits'), 'late', h.get('late_fetches'), 'dom', round(r['avg_launch_ms'],3), 'W', c.get('power_w_avg'), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || grep -h "OutOfMemoryError: HIP" ${1%.json}.err | tail -1 | cut -c1-300; }
OS="--offload-one-stream --offload-nonblocking-end --offload-backlog-gib 64"
run63() { timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 $OS "${@:2}" > $O/bench63_$1.json 2> $O/bench63_$1.err; show $O/bench63_$1.json 63$1; }
run63 attnscan20 --remat-keep attn,scan:20 --offload-park-kept --offload-lookahead 1
# (NOT to be run again: 560 GiB of pinned host memory, took the boxes down) run63 all --remat-keep attn,scan,fc2 --offload-park-kept --offload-lookahead 1
timeout 900 python bench.py --role worker --gpus 1 --video-length 30sec --steps 2 --warmup 1 --no-fsdp1-compare $OS --offload-park-kept --offload-lookahead 1 > $O/bench30_park.json 2> $O/bench30_park.err; show $O/bench30_park.json 30park
timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare $OS --offload-gib-per-layer 3 > $O/bench_off3os.json 2> $O/bench_off3os.err; show $O/bench_off3os.json off3os
