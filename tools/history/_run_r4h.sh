#!/bin/bash
# Round 4, call H (state after the sweep / attention changes): the full GPU suite, the driver's default bench command, kernel stats
# of the bench command, PMC traffic + MFMA-utilisation + wait / LDS counters of the TTT kernels, the other configurations.
cd /root/repo; mkdir -p gpurun_out/r4h; O=$GRAFT_REPO_ROOT/gpurun_out/r4h
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], 'bwd', r['avg_launch_ms'], r['frac'], {k: round(v['avg_ms'],2) for k,v in r['other'].items()}, 'fsdp1', d.get('fsdp1'), 'cpu', {k: v for k, v in d.get('cpu_baseline', {}).items() if k != 'sample'})"
cd /tmp
for nc in 804 282; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_${nc}_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc $nc --iters 2 > /dev/null 2>&1
    f=$(find /tmp/pmc_${nc}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc${nc}_pmc_$c.csv
  done
done
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_sq.csv || tail -5 /tmp/pmc_sq.log
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "attn_" --output-format csv -d /tmp/pmc_sq_attn -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --no-sdpa > /tmp/pmc_sq_attn.log 2>&1
f=$(find /tmp/pmc_sq_attn -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_pmc_sq.csv || tail -5 /tmp/pmc_sq_attn.log
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
timeout 150 rocprofv3 --pmc $C --kernel-include-regex "mlp_" --output-format csv -d /tmp/pm -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pm.log 2>&1
f=$(find /tmp/pm -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_wait_lds.csv || tail -5 /tmp/pm.log
timeout 150 rocprofv3 --pmc $C --kernel-include-regex "attn_" --output-format csv -d /tmp/pa -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --no-sdpa > /tmp/pa.log 2>&1
f=$(find /tmp/pa -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_pmc_wait_lds.csv || tail -5 /tmp/pa.log
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > /tmp/prof_bench.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_9s_kernel_stats.csv && head -12 "$f" | cut -c1-150
cd /root/repo
timeout 400 python bench.py --video-length 3sec --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > $O/bench_3s.json 2> $O/bench_3s.err; echo "3s rc=$?"; grep -h "^{" $O/bench_3s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['peak_mem_gib'])"
timeout 1200 python bench.py --video-length 63sec --steps 1 --warmup 1 --remat-free-layers 0 --remat-keep none --no-cpu-baseline --no-fsdp1-compare > $O/bench_63s.json 2> $O/bench_63s.err; echo "63s rc=$?"; grep -h "^{" $O/bench_63s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['peak_mem_gib'], 'bwd', r['avg_launch_ms'], {k: round(v['avg_ms'],2) for k,v in r['other'].items()})"
ls $O
