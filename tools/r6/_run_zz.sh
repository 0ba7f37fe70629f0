#!/bin/bash
# round 6, call ZZ (final tree of round 6: pair scan, 5 tapered parts, ragged backward chunk first in the sequence): smoke, the full GPU suite, the driver's bench command (main + fsdp1 + four legs + cpu_baseline),
# rocprofv3 kernel statistics of a short worker run, PMC passes (traffic, MFMA busy) of the TTT-MLP kernels at NC = 804
cd /root/repo; mkdir -p gpurun_out/r6zz; O=$GRAFT_REPO_ROOT/gpurun_out/r6zz
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -h "^{" $O/bench_default.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'frac', round(r['frac'],4), 'bwd ms', round(r['avg_launch_ms'],3), {k: v for k,v in r.items() if k.endswith('_ms')}, 'clk', c.get('clock_mhz_avg'), c.get('power_w_avg'))
for k,v in c.get('legs',{}).items(): print(k, {kk: vv for kk,vv in v.items() if kk in ('value','ms_per_step','peak_mem_gib','latent_frames_per_s','projected_50_step_video_s','error','skipped','leg_wall_s')})
print('fsdp1', c.get('fsdp1')); print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')} if 'cpu_baseline' in d else None, 'wall', d.get('bench_wall_s'))" || tail -20 $O/bench_default.err
cd /tmp
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --role worker --gpus 1 --steps 2 --warmup 1 --no-fsdp1-compare > $O/bench_rocprof.json 2> $O/bench_rocprof.err; echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_9s_kernel_stats.csv && head -12 "$f" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_804_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_804_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_$c.csv
done
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_sq.csv || tail -5 /tmp/pmc_sq.log
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "mlp_scan" --output-format csv -d /tmp/pmc_tcc -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 --fwd-only > /tmp/pmc_tcc.log 2>&1
f=$(find /tmp/pmc_tcc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_fwd_pmc_tcc.csv || tail -5 /tmp/pmc_tcc.log
timeout 120 python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 8 > $O/op_bench_nc804.json 2>/dev/null; tail -1 $O/op_bench_nc804.json | cut -c1-600
ls -la $O
