#!/bin/bash
# round 6, call O: full GPU suite with the group-sequential tail as default; in-step A/B on one box: defaults vs tail5=0 vs attn_prio=0
cd /root/repo; mkdir -p gpurun_out/r6o; O=$GRAFT_REPO_ROOT/gpurun_out/r6o
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err; show $O/bench_new_$rep.json new
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option tail5=0 > $O/bench_tail4_$rep.json 2> $O/bench_tail4_$rep.err; show $O/bench_tail4_$rep.json tail5=0
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option attn_prio=0 > $O/bench_noprio_$rep.json 2> $O/bench_noprio_$rep.err; show $O/bench_noprio_$rep.json attn_prio=0
done
