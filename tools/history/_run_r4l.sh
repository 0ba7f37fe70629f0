#!/bin/bash
# Round 4, call L: ONE box: ReplicaMixedPrecision (--fsdp off) against the flat path without collectives (default) + its fsdp1 point.
cd /root/repo; mkdir -p gpurun_out/r4l; O=$GRAFT_REPO_ROOT/gpurun_out/r4l
export TMPDIR=/tmp
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fsdp off > $O/bench_replica.json 2> $O/bench_replica.err; echo "replica rc=$?"
grep -h "^{" $O/bench_replica.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('replica', d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'])"
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -2 $O/bench_default.err | cut -c1-300
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat1', d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], d['config']['parallelism'], 'fsdp1', d.get('fsdp1'))"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --remat-free-layers 11 > $O/bench_flat_11.json 2> $O/bench_flat_11.err; echo "flat 11 rc=$?"
grep -h "^{" $O/bench_flat_11.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat1 @11 layers', d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'])"
