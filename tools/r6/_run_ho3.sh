#!/bin/bash
# round 6, call HO3: host offload at 2 GiB per free layer - where does the time go?  timed events around every copy / compute-stream wait; backlog 16 / 64 GiB; lookahead 2 / 4
cd /root/repo; mkdir -p gpurun_out/r6ho3; O=gpurun_out/r6ho3
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'off', c.get('host_offload'), 'bwd', round(r['avg_launch_ms'],3), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || tail -5 ${1%.json}.err; }
run() { timeout 900 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --offload-trace "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
run b64   --offload-gib-per-layer 2 --offload-backlog-gib 64 --remat-free-layers 19
run b16la4 --offload-gib-per-layer 2 --offload-backlog-gib 16 --offload-lookahead 4 --remat-free-layers 19
run l8    --offload-gib-per-layer 2 --offload-backlog-gib 64 --offload-layers 8 --remat-free-layers 16
