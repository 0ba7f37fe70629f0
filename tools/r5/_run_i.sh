#!/bin/bash
# round 5, call I: the 63 s step with 8 instead of 4 parts in the pipelined TTT layer forward (same box, the leg's settings)
cd /root/repo; mkdir -p gpurun_out/r5i; O=$GRAFT_REPO_ROOT/gpurun_out/r5i
export TMPDIR=/tmp
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'parts', d['config'].get('ttt_pipeline_parts'), 'peak', round(d['peak_mem_gib'],1), round(d['peak_reserved_gib'],1), 'bwd', round(r['avg_launch_ms'],2), {k: round(v['avg_ms'],2) for k,v in r['other'].items()})" || tail -5 ${1%.json}.err; }
for parts in 8 4; do
  timeout 900 python bench.py --role worker --gpus 1 --video-length 63sec --steps 2 --warmup 1 --no-fsdp1-compare --remat-free-layers 0 --remat-keep attn --remat-keep-layers 10 --pipeline-parts $parts > $O/bench_63s_parts$parts.json 2> $O/bench_63s_parts$parts.err; echo "63s parts=$parts rc=$?"; show $O/bench_63s_parts$parts.json
done
