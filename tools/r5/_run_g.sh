#!/bin/bash
# round 5, call G: the 63 s leg with the first ten re-materialised layers keeping their attention outputs; the 9 s line with a third
# sizing refinement (14 instead of 13 remat-free layers?)
cd /root/repo; mkdir -p gpurun_out/r5g; O=$GRAFT_REPO_ROOT/gpurun_out/r5g
export TMPDIR=/tmp
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'keep', d['config']['remat_keep'], d['config'].get('remat_keep_layers'), 'peak', round(d['peak_mem_gib'],1), round(d['peak_reserved_gib'],1), 'retries', d['alloc_retries_total'], 'bwd', round(r['avg_launch_ms'],2), {k: round(v['avg_ms'],2) for k,v in r['other'].items()})" || tail -5 ${1%.json}.err; }
timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn --remat-keep-layers 10 > $O/bench_63s_keep10.json 2> $O/bench_63s_keep10.err; echo "63s rc=$?"; show $O/bench_63s_keep10.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-fsdp1-compare > $O/bench_9s_refine3.json 2> $O/bench_9s_refine3.err; echo "9s rc=$?"; show $O/bench_9s_refine3.json; grep sizing $O/bench_9s_refine3.err
