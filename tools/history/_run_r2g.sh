#!/bin/bash
# round 2, call G: full GPU suite after the variant clean-up, the bench lines (9 s default, 3 s), rocprof kernel stats of the
# bench command, PMC traffic of the TTT-MLP backward at the 9 s scan length
mkdir -p gpurun_out/r2g
O=gpurun_out/r2g
timeout 1200 python -m pytest tests -m gpu -q -rf 2>&1 | tail -25 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 1500 python bench.py --steps 3 --warmup 1 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -8; cut -c1-900 $O/bench_9s.json
timeout 600 python bench.py --video-length 3sec --steps 5 --warmup 2 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_3s.err | grep '^{"metric"' > $O/bench_3s.json
cut -c1-400 $O/bench_3s.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > /tmp/prof_bench.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/bench_9s_kernel_stats.csv && head -14 "$f" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/op_nc804_pmc_$c.csv
done
cd $GRAFT_REPO_ROOT; ls -la $O
