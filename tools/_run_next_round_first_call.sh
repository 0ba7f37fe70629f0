#!/bin/bash
# What round 6 left unmeasured when call HO14 closed its GPU (DESIGN.md sections 6 / 7).  Nothing in here changes a default.  ONE box, no retry loop around it:
# a call that loses its box must be looked at, not repeated (three repeats of HO14 cost the round its GPU).
cd /root/repo; mkdir -p gpurun_out/next; O=gpurun_out/next
# 1. the full device suite on the final tree (146 collected; the last file holds the host-offload test with the shipped defaults, never run on a device)
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
# 2. the driver's command (the 63 s leg now parks the attention outputs of all 42 layers: expect ~43 s/step; `fallback_after` in its entry says if it did not)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
# 3. the 63 s step with scan outputs parked too, inside the pinned cap (160 GiB: attention 92 GiB + 8 layers' scan outputs); same box: compare with the leg of (2)
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; h=c.get('host_offload') or {}; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], c['remat_keep'], c.get('remat_keep_limits'), 'gib', h.get('gib_per_step'), 'pinned', h.get('pinned_gib'), 'peak', round(d['peak_mem_gib'],1))" || tail -3 ${1%.json}.err; }
timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn,scan:8 --offload-park-kept --offload-lookahead 1 > $O/bench63_scan8.json 2> $O/bench63_scan8.err; show $O/bench63_scan8.json 63scan8
# 4. the 30 s stage with everything it keeps parked (6.4 GB per layer against ~8 GB of host link per layer-forward)
timeout 900 python bench.py --role worker --gpus 1 --video-length 30sec --steps 2 --warmup 1 --no-fsdp1-compare --offload-park-kept --offload-lookahead 1 > $O/bench30_park.json 2> $O/bench30_park.err; show $O/bench30_park.json 30park
timeout 900 python bench.py --role worker --gpus 1 --video-length 30sec --steps 2 --warmup 1 --no-fsdp1-compare > $O/bench30_base.json 2> $O/bench30_base.err; show $O/bench30_base.json 30base
# 5. the 9 s step with 3 GiB per free layer (2 GiB: +0.5 %)
timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare --offload-gib-per-layer 3 > $O/bench_off3.json 2> $O/bench_off3.err; show $O/bench_off3.json off3
timeout 900 python bench.py --role worker --gpus 1 --steps 4 --warmup 1 --no-fsdp1-compare > $O/bench_base.json 2> $O/bench_base.err; show $O/bench_base.json base
