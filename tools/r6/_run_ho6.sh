#!/bin/bash
# round 6, call HO6 (HO5 again with permuted kernel outputs parked too): the 63 s training step with kept kernel outputs parked in host memory; D2H beside kinds of compute
cd /root/repo; mkdir -p gpurun_out/r6ho6; O=gpurun_out/r6ho6
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'keep', c['remat_keep'], c.get('remat_keep_limits'), 'off', c.get('host_offload'), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || grep -h "OutOfMemoryError: HIP" ${1%.json}.err | tail -1 | cut -c1-300; }
run() { timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 "${@:2}" > $O/bench63_$1.json 2> $O/bench63_$1.err; show $O/bench63_$1.json $1; }
timeout 200 python tools/pcie_probe2.py > $O/probe2.json 2> $O/probe2.err; cat $O/probe2.json
export GPU_MAX_HW_QUEUES=8
run parkattn --remat-keep attn --offload-park-kept --offload-trace --offload-lookahead 2
run parkattnscan16 --remat-keep attn,scan:16 --offload-park-kept --offload-trace --offload-lookahead 1
unset GPU_MAX_HW_QUEUES
run base --remat-keep attn --remat-keep-layers 10
