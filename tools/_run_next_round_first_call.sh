#!/bin/bash
# Prepared at the end of round 5 (GPU budget spent): measurements that were left open, for the first gpurun call of a next round
# (~14 GPU-minutes on one box).  Nothing here changes a default; each line says what a result would decide.
#   1. the 63 s step with 16 parts in the pipelined TTT layer forward (8 gave 7 122 against 7 001 video-tok/s for 4, profiles/r5i_*):
#      if 16 wins, raise the cap of `pipeline_parts_auto` (ttt_amd/models/ssm/ttt_layer.py: min(8, groups // 40)).
#   2. the 18 s and 30 s stages (not re-run since round 3: 7 240 / 6 742 video-tok/s) on the round-5 tree.
#   3. on an 8-GPU node instead: `python bench.py --gpus 8` (the parent retries once with safe memory settings; `rccl` in the line
#      summarises RCCL's rings / trees) and a `rocprofv3 --kernel-trace` of two ranks read with tools/sweep_launches.py - does the
#      cluster sweep keep its 0.8 ms per launch beside RCCL's reduce-scatter kernels?
cd /root/repo; mkdir -p gpurun_out/r6a; O=$GRAFT_REPO_ROOT/gpurun_out/r6a
export TMPDIR=/tmp
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'parts', d['config'].get('ttt_pipeline_parts'), 'peak', round(d['peak_mem_gib'],1), {k: round(v['avg_ms'],2) for k,v in r['other'].items()})" || tail -5 ${1%.json}.err; }
for parts in 16 8; do
  timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn --remat-keep-layers 10 --pipeline-parts $parts > $O/bench_63s_parts$parts.json 2> $O/bench_63s_parts$parts.err; echo "63s parts=$parts rc=$?"; show $O/bench_63s_parts$parts.json
done
for vl in 18sec 30sec; do
  timeout 900 python bench.py --role worker --gpus 1 --video-length $vl --steps 2 --warmup 1 --no-fsdp1-compare > $O/bench_$vl.json 2> $O/bench_$vl.err; echo "$vl rc=$?"; show $O/bench_$vl.json
done
