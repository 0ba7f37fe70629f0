#!/bin/bash
# round 2, call Q (final state): full GPU suite, smoke, the driver's bench command (9 s default) + the 3 s line, rocprofv3 kernel
# stats of the bench command, PMC traffic of the TTT-MLP scans at both scan lengths
mkdir -p gpurun_out/r2q
O=$GRAFT_REPO_ROOT/gpurun_out/r2q
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -12 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 3 --warmup 1 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -8; cut -c1-600 $O/bench_9s.json
timeout 600 python bench.py --video-length 3sec --steps 5 --warmup 2 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_3s.err | grep '^{"metric"' > $O/bench_3s.json
cut -c1-300 $O/bench_3s.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > /tmp/prof_bench.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_9s_kernel_stats.csv && head -12 "$f" | cut -c1-150
for nc in 804 282; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_${nc}_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc $nc --iters 2 > /dev/null 2>&1
    f=$(find /tmp/pmc_${nc}_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc${nc}_pmc_$c.csv
  done
done
timeout 200 python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 5 2>/dev/null | python $GRAFT_REPO_ROOT/tools/_fmt_phases.py "nc 804:" | tee $O/op_final.txt
timeout 200 python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 282 --iters 5 2>/dev/null | python $GRAFT_REPO_ROOT/tools/_fmt_phases.py "nc 282:" | tee -a $O/op_final.txt
ls -la $O
