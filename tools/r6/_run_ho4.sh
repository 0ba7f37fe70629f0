#!/bin/bash
# round 6, call HO4: is the slow D2H (15 - 22 GB/s beside the forward) a hardware-queue alias?  baseline / 8 layers x 2 GiB with 4 (default) and 8 hardware queues, one box
cd /root/repo; mkdir -p gpurun_out/r6ho4; O=gpurun_out/r6ho4
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'off', c.get('host_offload'), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || tail -5 ${1%.json}.err; }
run() { timeout 900 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
run base --remat-free-layers 14
run l8q4 --offload-trace --offload-gib-per-layer 2 --offload-backlog-gib 64 --offload-layers 8 --remat-free-layers 16
export GPU_MAX_HW_QUEUES=8
run l8q8 --offload-trace --offload-gib-per-layer 2 --offload-backlog-gib 64 --offload-layers 8 --remat-free-layers 16
run baseq8 --remat-free-layers 14
run a2q8 --offload-trace --offload-gib-per-layer 2 --offload-backlog-gib 24 --remat-free-layers 19
