#!/bin/bash
# round 5, call J (last): the tree as committed - the full GPU suite and the driver's entry point end to end (orchestrator, both legs)
cd /root/repo; mkdir -p gpurun_out/r5j; O=$GRAFT_REPO_ROOT/gpurun_out/r5j
export TMPDIR=/tmp
timeout 500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_suite.log
timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > $O/bench_short_with_legs.json 2> $O/bench_short_with_legs.err; echo "bench rc=$?"
grep -h "^{" $O/bench_short_with_legs.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'parts', d['config']['ttt_pipeline_parts'], 'bwd ms', round(r['avg_launch_ms'],3), {k: (round(v['avg_ms'],3), v.get('parts_per_scan')) for k,v in r['other'].items()})
for k in ('ctx3s','ctx63s'): print(k, {a: d[k].get(a) for a in ('value','ms_per_step','remat_keep','remat_keep_layers','peak_mem_gib','scan_fwd_ms','error','skipped')})
print('wall', d.get('bench_wall_s'))" || tail -20 $O/bench_short_with_legs.err
