#!/bin/bash
# round 5, call H: two same-box A/Bs that call G left open (its box was a slow one): 13 vs 14 remat-free layers at 9 s; the 63 s step
# with nothing kept vs the attention outputs of the first ten layers kept
cd /root/repo; mkdir -p gpurun_out/r5h; O=$GRAFT_REPO_ROOT/gpurun_out/r5h
export TMPDIR=/tmp
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'keep', d['config']['remat_keep'], d['config'].get('remat_keep_layers'), 'peak', round(d['peak_mem_gib'],1), round(d['peak_reserved_gib'],1), 'retries', d['alloc_retries_total'], 'bwd', round(r['avg_launch_ms'],2), {k: round(v['avg_ms'],2) for k,v in r['other'].items()})" || tail -5 ${1%.json}.err; }
for n in 13 14 13 14; do
  i=$((i+1))
  timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers $n > $O/bench_9s_free${n}_$i.json 2> $O/bench_9s_free${n}_$i.err; echo "9s free=$n rc=$?"; show $O/bench_9s_free${n}_$i.json
done
timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep none > $O/bench_63s_none.json 2> $O/bench_63s_none.err; echo "63s none rc=$?"; show $O/bench_63s_none.json
timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn --remat-keep-layers 10 > $O/bench_63s_keep10.json 2> $O/bench_63s_keep10.err; echo "63s keep10 rc=$?"; show $O/bench_63s_keep10.json
