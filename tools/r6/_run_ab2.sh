#!/bin/bash
# round 6, call AB2: the pipelined TTT layer forward - tapered parts vs equal parts, scan stream priority; layer level, then in-step
cd /root/repo; mkdir -p gpurun_out/r6ab2; O=gpurun_out/r6ab2
lay() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', {k:[round(x,3) for x in (v['median_ms']['fwd'], v['median_ms']['fwd_rev'], v['median_ms']['fwd_bwd'])] for k,v in d['by_parts'].items()})"; }
for rep in 1 2; do
TTT_PIPELINE_WEIGHTS=equal timeout 300 python tools/ttt_layer_bench.py --parts 4,5,6 --rounds 3 > $O/layer_equal_$rep.json 2>$O/layer.err; lay $O/layer_equal_$rep.json equal
timeout 300 python tools/ttt_layer_bench.py --parts 4,5,6 --rounds 3 > $O/layer_taper_$rep.json 2>>$O/layer.err; lay $O/layer_taper_$rep.json taper
TTT_SCAN_STREAM_PRIORITY=-1 timeout 300 python tools/ttt_layer_bench.py --parts 4,5 --rounds 3 > $O/layer_taper_prio_$rep.json 2>>$O/layer.err; lay $O/layer_taper_prio_$rep.json taper+prio
done
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_taper_$rep.json 2> $O/bench_taper_$rep.err; show $O/bench_taper_$rep.json taper4
TTT_PIPELINE_WEIGHTS=equal timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_equal_$rep.json 2> $O/bench_equal_$rep.err; show $O/bench_equal_$rep.json equal4
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --pipeline-parts 5 > $O/bench_taper5_$rep.json 2> $O/bench_taper5_$rep.err; show $O/bench_taper5_$rep.json taper5
done
