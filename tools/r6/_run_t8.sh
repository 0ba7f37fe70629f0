#!/bin/bash
# round 6, call T8: the 63 s training step - 8 tapered parts (generic taper) vs 8 equal parts, one box
cd /root/repo; mkdir -p gpurun_out/r6t8; O=gpurun_out/r6t8
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'parts', c.get('ttt_pipeline_parts'), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn --remat-keep-layers 10 > $O/bench63_taper_$rep.json 2> $O/bench63_taper_$rep.err; show $O/bench63_taper_$rep.json taper8
TTT_PIPELINE_WEIGHTS=equal timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn --remat-keep-layers 10 > $O/bench63_equal_$rep.json 2> $O/bench63_equal_$rep.err; show $O/bench63_equal_$rep.json equal8
done
