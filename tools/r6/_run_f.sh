#!/bin/bash
# round 6, call F: the full GPU suite on the tree with schedule 2 + the split sweep as defaults, then in-step A/B (short bench runs, same box):
# defaults vs the round-5 settings (overlap_tail=1, deriver_split=0)
cd /root/repo; mkdir -p gpurun_out/r6f; O=$GRAFT_REPO_ROOT/gpurun_out/r6f
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err; show $O/bench_new_$rep.json new
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option overlap_tail=1 --debug-option deriver_split=0 > $O/bench_r5_$rep.json 2> $O/bench_r5_$rep.err; show $O/bench_r5_$rep.json r5settings
done
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option overlap_tail=1 > $O/bench_ov1.json 2> $O/bench_ov1.err; show $O/bench_ov1.json ov1+split
