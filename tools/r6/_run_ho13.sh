#!/bin/bash
# round 6, call HO13: last offload A/B - one copy stream (a hardware queue less), no host wait at the end of the forward, backlog cap 64 GiB: 9 s with 2 GiB per layer, 63 s with parked attention outputs
cd /root/repo; mkdir -p gpurun_out/r6ho13; O=gpurun_out/r6ho13
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; h=c.get('host_offload') or {}; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'keep', c['remat_keep'], 'gib', h.get('gib_per_step'), 'host_s', h.get('host_s_per_step'), 'waits', h.get('throttle_waits'), 'dom', round(r['avg_launch_ms'],3), 'W', c.get('power_w_avg'), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || grep -h "OutOfMemoryError: HIP" ${1%.json}.err | tail -1 | cut -c1-300; }
run() { timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
run base1
run off2os --offload-gib-per-layer 2 --offload-backlog-gib 64 --offload-one-stream --offload-nonblocking-end
run off2nb --offload-gib-per-layer 2 --offload-backlog-gib 64 --offload-nonblocking-end
run off1os --offload-gib-per-layer 1 --offload-backlog-gib 64 --offload-one-stream --offload-nonblocking-end
run63() { timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 "${@:2}" > $O/bench63_$1.json 2> $O/bench63_$1.err; show $O/bench63_$1.json 63$1; }
run63 parkattnos --remat-keep attn --offload-park-kept --offload-lookahead 2 --offload-backlog-gib 64 --offload-one-stream --offload-nonblocking-end
