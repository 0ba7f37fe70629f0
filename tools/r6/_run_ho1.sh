#!/bin/bash
# round 6, call HO1: host offload of saved activations - the device test, then the 9 s step with 0 / 3 / 5 GiB per free layer parked in host memory (one box)
cd /root/repo; mkdir -p gpurun_out/r6ho1; O=gpurun_out/r6ho1
timeout 600 python -m pytest tests/test_host_offload_gpu.py -x -q -s > $O/test.log 2>&1; tail -5 $O/test.log
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'off', c.get('host_offload'), 'bwd', round(r['avg_launch_ms'],3), 'peak', round(d['peak_mem_gib'],1), 'retries', d['alloc_retries_total'])" || tail -5 ${1%.json}.err; }
for g in 0 3 5; do
timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare --offload-gib-per-layer $g > $O/bench_off$g.json 2> $O/bench_off$g.err; show $O/bench_off$g.json off$g
done
grep -h "sizing" $O/bench_off*.err | tail -20
