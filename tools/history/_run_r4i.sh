#!/bin/bash
# Round 4, call I: where does the FSDP2 one-rank path lose 7 % against the replica path?  Kernel stats + torch profiler of ONE step each.
cd /root/repo; mkdir -p gpurun_out/r4i; O=$GRAFT_REPO_ROOT/gpurun_out/r4i
export TMPDIR=/tmp
cd /tmp
timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fsdp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --fsdp on --remat-free-layers 10 --no-cpu-baseline --no-fsdp1-compare > $O/bench_fsdp_prof.log 2>&1
f=$(find /tmp/prof_fsdp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_9s_fsdp1_kernel_stats.csv
grep -h "^{" $O/bench_fsdp_prof.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fsdp1 under rocprof', d['value'], d['ms_per_step'])"
cd /root/repo
timeout 500 python bench.py --steps 1 --warmup 1 --fsdp on --remat-free-layers 10 --no-cpu-baseline --no-fsdp1-compare --torch-profile $O/torch_profile_fsdp1.txt > $O/bench_fsdp_tp.json 2> $O/bench_fsdp_tp.err; echo "fsdp torch-profile rc=$?"
timeout 500 python bench.py --steps 1 --warmup 1 --fsdp off --remat-free-layers 10 --no-cpu-baseline --no-fsdp1-compare --torch-profile $O/torch_profile_replica.txt > $O/bench_replica_tp.json 2> $O/bench_replica_tp.err; echo "replica torch-profile rc=$?"
for f in fsdp replica; do grep -h "^{" $O/bench_${f}_tp.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['peak_mem_gib'])"; done
head -45 $O/torch_profile_fsdp1.txt | cut -c1-200
