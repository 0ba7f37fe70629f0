#!/bin/bash
# Round 4, call S (one box): sizing log of the flat path without collectives against the replica path (why 7 free layers?)
cd /root/repo; mkdir -p gpurun_out/r4s; O=$GRAFT_REPO_ROOT/gpurun_out/r4s
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; grep "sizing\|timed region" $O/bench_$name.err | cut -c1-250
  grep -h "^{" $O/bench_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', round(d['value'],1), round(d['ms_per_step'],1), d['config']['remat_free_layers'], d['peak_mem_gib'], 'ttt bwd', round(r['avg_launch_ms'],3))"; }
run flat --fsdp flat
run replica --fsdp off
