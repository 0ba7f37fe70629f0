#!/bin/bash
# Round 4, the last GPU minute: 6-layer model at 9 s (DEBUG --layers: invalid as a bench line, fine as an A/B of kernel times):
# main (replica) + fsdp1 point, the backward's schedule options.
cd /root/repo; mkdir -p gpurun_out/r4zd; O=$GRAFT_REPO_ROOT/gpurun_out/r4zd
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 30 python bench.py --layers 6 --remat-free-layers 6 --steps 2 --warmup 1 --fsdp1-steps 2 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  grep -h "^{" $O/bench_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', round(d['ms_per_step'],1), 'ttt bwd', round(r['avg_launch_ms'],3), 'fsdp1', d.get('fsdp1'))"; }
run base X=0
run early TTT_FLAGS_MEMSET_EARLY=1
run delay TTT_TAIL_DELAY_US=25
