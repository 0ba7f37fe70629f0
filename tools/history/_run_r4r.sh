#!/bin/bash
# Round 4, call R (one box): why does the cluster sweep run 0.95 instead of 0.82 ms under FlatFSDP (and FSDP2) on one rank?
# replica | flat without a side stream | the same with every unit reduced at the end of the backward (FLAT_FSDP_DEFER_REDUCE=1)
cd /root/repo; mkdir -p gpurun_out/r4r; O=$GRAFT_REPO_ROOT/gpurun_out/r4r
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --remat-free-layers 12 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  grep -h "^{" $O/bench_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', round(d['value'],1), round(d['ms_per_step'],1), 'ttt bwd', round(r['avg_launch_ms'],3), 'attn bwd', round(r['other']['attn_bwd']['avg_ms'],3))"; }
run replica --fsdp off
run flat_nostream --fsdp flat
FLAT_FSDP_DEFER_REDUCE=1 run flat_defer --fsdp flat
