#!/bin/bash
# round 3, call C: where does revision 4 spend its time (kernel trace of both revisions, stage cycle stamps), and is it deterministic?
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
timeout 200 python tools/_det_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3c/det.txt; cat gpurun_out/r3c/det.txt
for rev in 4 3; do
  timeout 120 python tools/op_bench.py --nc 804 --bwd-rev $rev --iters 3 --phases 2>/dev/null | python tools/_fmt_phases.py "rev$rev nc804" | tee gpurun_out/r3c/phases_rev$rev.txt
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r3c/prof_rev$rev -o rev$rev -- python tools/op_bench.py --nc 804 --bwd-rev $rev --iters 5 > /dev/null 2>&1
  f=$(find gpurun_out/r3c/prof_rev$rev -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats rev $rev"; head -8 "$f" | cut -c1-200
  cp "$f" gpurun_out/r3c/kernel_stats_rev$rev.csv
  rm -rf gpurun_out/r3c/prof_rev$rev
done
