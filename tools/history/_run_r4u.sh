#!/bin/bash
# Round 4, call U (one box): repeatability of the replica path's kernel times (same setting twice), then the default line with its
# fsdp1 point (FlatFSDP with collectives over one rank, persistent reduce buffers) and that point's own kernel times.
cd /root/repo; mkdir -p gpurun_out/r4u; O=$GRAFT_REPO_ROOT/gpurun_out/r4u
export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"
  grep -h "^{" $O/bench_$name.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', round(d['value'],1), round(d['ms_per_step'],1), d['config']['remat_free_layers'], d['peak_mem_gib'], 'ttt bwd', round(r['avg_launch_ms'],3), 'attn bwd', round(r['other']['attn_bwd']['avg_ms'],3), 'fsdp1', d.get('fsdp1'))"; }
run replica_a --fsdp off --no-fsdp1-compare --remat-free-layers 12
run replica_b --fsdp off --no-fsdp1-compare --remat-free-layers 12
run default
