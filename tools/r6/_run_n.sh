#!/bin/bash
# round 6, call N: counter traffic of the TTT-MLP backward with the group-sequential tail (FETCH_SIZE / WRITE_SIZE, separate passes)
cd /root/repo; mkdir -p gpurun_out/r6n; O=$GRAFT_REPO_ROOT/gpurun_out/r6n
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_804_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_804_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_$c.csv
done
ls -la $O
