#!/bin/bash
# round 6, call HO11: copies out on the 6-layer debug model (15 s per run): default / batch per layer / HSA_ENABLE_SDMA_RECOMMENDED_ENG 0, 1 / AMD_SERIALIZE_COPY; device test
cd /root/repo; mkdir -p gpurun_out/r6ho11; O=gpurun_out/r6ho11
timeout 600 python -m pytest tests/test_host_offload_gpu.py -q > $O/test.log 2>&1; tail -3 $O/test.log
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); h=d['config']['host_offload']; print('$2', round(d['ms_per_step'],1), 'ms', 'gib', h['gib_per_step'], 'd2h', h['trace']['d2h']['gbps'], 'h2d', h['trace']['h2d']['gbps'], 'wait_ms', h['trace']['wait']['ms'], h['host_s_per_step'])" || tail -3 ${1%.json}.err; }
run() { timeout 600 python bench.py --role worker --gpus 1 --layers 6 --steps 3 --warmup 1 --no-fsdp1-compare --offload-trace --offload-gib-per-layer 3 --remat-free-layers 6 "${@:2}" > $O/bench_$1.json 2> $O/bench_$1.err; show $O/bench_$1.json $1; }
timeout 600 python bench.py --role worker --gpus 1 --layers 6 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 6 > $O/bench_base.json 2> $O/bench_base.err; grep -h "^{" $O/bench_base.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', round(d['ms_per_step'],1))"
run default
run batch --offload-batch
HSA_ENABLE_SDMA_RECOMMENDED_ENG=1 run rec1
HSA_ENABLE_SDMA_RECOMMENDED_ENG=0 run rec0
HSA_ENABLE_SDMA_RECOMMENDED_ENG=1 run rec1batch --offload-batch
HSA_ENABLE_SDMA_GANG=0 run gang0
