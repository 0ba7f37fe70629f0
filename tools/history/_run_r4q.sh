#!/bin/bash
# Round 4, call Q (one box): torch profiler tables of one 9 s step, replica path against FlatFSDP without collectives, same
# re-materialisation setting: where the flat path's 2 % go.
cd /root/repo; mkdir -p gpurun_out/r4q; O=$GRAFT_REPO_ROOT/gpurun_out/r4q
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --fsdp off --remat-free-layers 12 --torch-profile $O/torch_profile_replica.txt > $O/bench_replica.json 2> $O/bench_replica.err; echo "replica rc=$?"
grep -h "^{" $O/bench_replica.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('replica', d['value'], d['ms_per_step'])"
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --fsdp flat --remat-free-layers 12 --torch-profile $O/torch_profile_flat.txt > $O/bench_flat.json 2> $O/bench_flat.err; echo "flat rc=$?"
grep -h "^{" $O/bench_flat.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat', d['value'], d['ms_per_step'])"
