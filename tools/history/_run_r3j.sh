#!/bin/bash
# round 3, call J (final state): PMC traffic of the TTT-MLP scans (separate FETCH_SIZE / WRITE_SIZE passes), MFMA-utilisation
# counters of the dominant kernels, rocprofv3 kernel stats of the driver's bench command on the final tree
mkdir -p gpurun_out/r3j
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
cd /tmp && export TMPDIR=/tmp
for nc in 804 282; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_${nc}_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc $nc --iters 2 > /dev/null 2>&1
    f=$(find /tmp/pmc_${nc}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc${nc}_pmc_$c.csv
  done
done
# MFMA utilisation: busy cycles of the MFMA pipe vs the SQ's busy cycles, and the wait share, TTT kernels and attention kernels
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_sq.csv || tail -5 /tmp/pmc_sq.log
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "attn_" --output-format csv -d /tmp/pmc_sq_attn -- python $GRAFT_REPO_ROOT/tools/attn_bench.py > /tmp/pmc_sq_attn.log 2>&1
f=$(find /tmp/pmc_sq_attn -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_pmc_sq.csv || tail -5 /tmp/pmc_sq_attn.log
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > /tmp/prof_bench.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_9s_kernel_stats.csv && head -14 "$f" | cut -c1-150
ls -la $O
