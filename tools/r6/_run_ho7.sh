#!/bin/bash
# round 6, call HO7: the 9 s step - 4 (default) against 8 hardware queues, interleaved twice; then two modest offload settings with 8 queues
cd /root/repo; mkdir -p gpurun_out/r6ho7; O=gpurun_out/r6ho7
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'off', c.get('host_offload'), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || grep -h "OutOfMemoryError: HIP" ${1%.json}.err | tail -1 | cut -c1-300; }
run() { timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
for rep in 1 2; do
run q4_$rep --remat-free-layers 14
GPU_MAX_HW_QUEUES=8 run q8_$rep --remat-free-layers 14
done
export GPU_MAX_HW_QUEUES=8
run off1 --offload-trace --offload-gib-per-layer 1 --offload-backlog-gib 24
run off2l8 --offload-trace --offload-gib-per-layer 2 --offload-layers 8 --offload-backlog-gib 24
