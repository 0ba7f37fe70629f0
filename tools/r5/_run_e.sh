#!/bin/bash
# round 5, call E: how many parts for the pipelined TTT layer forward, with tuned GEMM selections for the parts' shapes
cd /root/repo; mkdir -p gpurun_out/r5e; O=$GRAFT_REPO_ROOT/gpurun_out/r5e
export TMPDIR=/tmp
timeout 900 python tools/tune_pipeline_gemms.py $O/tunableop_parts.csv --video-length 9sec,3sec --parts 4,6 > $O/tune_parts.log 2>&1; echo "tune rc=$?"; tail -3 $O/tune_parts.log
python - <<'PY'
import os
base='ttt-video-dit_amd/ttt_amd/infra/gemm_tuning_gfx950.csv'; new=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r5e/tunableop_parts.csv'
b=open(base).read().rstrip('\n').split('\n'); have={','.join(l.split(',')[:2]) for l in b if not l.startswith('Validator')}
add=[l for l in open(new).read().split('\n') if l and not l.startswith('Validator') and ','.join(l.split(',')[:2]) not in have]
open(base,'w').write('\n'.join(b+add)+'\n'); print('merged', len(add)); print('\n'.join(add))
PY
cp ttt-video-dit_amd/ttt_amd/infra/gemm_tuning_gfx950.csv $O/gemm_tuning_gfx950_merged.csv
timeout 600 python tools/ttt_layer_bench.py --parts 0,4,5,6,8 --rounds 3 > $O/ttt_layer_pipeline_ab.json 2> $O/ttt_layer.err; echo "layer rc=$?"; python -c "
import json; d=json.load(open('$O/ttt_layer_pipeline_ab.json'))
for k,v in d['by_parts'].items(): print(k, {a: round(b,2) for a,b in v['median_ms'].items()}, v.get('forward',{}).get('out_rel_l2'))"
for parts in 4 6 0 4 6; do
  i=$((i+1))
  timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-fsdp1-compare --pipeline-parts $parts > $O/bench_p${parts}_$i.json 2> $O/bench_p${parts}_$i.err; echo "bench parts=$parts rc=$?"
  grep -h "^{" $O/bench_p${parts}_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'ttt bwd', round(r['avg_launch_ms'],3), {k: (round(v['avg_ms'],3), v.get('parts_per_scan')) for k,v in r['other'].items()}, 'parts', d['config'].get('ttt_pipeline_parts'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 $O/bench_p${parts}_$i.err
done
