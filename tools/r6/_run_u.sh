#!/bin/bash
# round 6, call U: the pair scan as default - its tests, the full GPU suite, in-step A/B on one box (scan_pair 1 vs 0)
cd /root/repo; mkdir -p gpurun_out/r6u; O=$GRAFT_REPO_ROOT/gpurun_out/r6u
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_scan_pair_gpu.py -x -q -m gpu > $O/pair_tests.log 2>&1; echo "pair tests rc=$?"; tail -3 $O/pair_tests.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_pair_$rep.json 2> $O/bench_pair_$rep.err; show $O/bench_pair_$rep.json pair
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option scan_pair=0 > $O/bench_single_$rep.json 2> $O/bench_single_$rep.err; show $O/bench_single_$rep.json single
done
