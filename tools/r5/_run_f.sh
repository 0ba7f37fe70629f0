#!/bin/bash
# round 5, call F (final tree): smoke, the full GPU suite, the driver's bench command (with the ctx3s / ctx63s legs, fsdp1, cpu_baseline),
# rocprofv3 kernel statistics of a short bench run, PMC passes (HBM traffic, MFMA busy) of the TTT-MLP scans and the attention kernels
cd /root/repo; mkdir -p gpurun_out/r5f; O=$GRAFT_REPO_ROOT/gpurun_out/r5f
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 700 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -h "^{" $O/bench_default.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'frac', round(r['frac'],4), 'bwd ms', round(r['avg_launch_ms'],3), {k: round(v['avg_ms'],3) for k,v in r['other'].items()})
print('fsdp1', d.get('fsdp1')); print('ctx3s', d.get('ctx3s')); print('ctx63s', d.get('ctx63s')); print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')} if 'cpu_baseline' in d else None, 'wall', d.get('bench_wall_s'))" || tail -20 $O/bench_default.err
cd /tmp
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --role worker --gpus 1 --steps 2 --warmup 1 --no-fsdp1-compare > $O/bench_rocprof.json 2> $O/bench_rocprof.err; echo "rocprof rc=$?"
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_9s_kernel_stats.csv && head -12 "$f" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_804_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_804_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_$c.csv
done
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_sq.csv || tail -5 /tmp/pmc_sq.log
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "attn_" --output-format csv -d /tmp/pmc_sq_attn -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --no-sdpa > /tmp/pmc_sq_attn.log 2>&1
f=$(find /tmp/pmc_sq_attn -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_pmc_sq.csv || tail -5 /tmp/pmc_sq_attn.log
timeout 120 python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 5 > $O/op_bench_nc804.txt 2>&1; tail -6 $O/op_bench_nc804.txt
ls -la $O | head -30
