#!/bin/bash
# Round 4, call K: the default bench line on the flat path (one GPU: collectives skipped), fsdp1 = the same with collectives.
cd /root/repo; mkdir -p gpurun_out/r4k; O=$GRAFT_REPO_ROOT/gpurun_out/r4k
export TMPDIR=/tmp

timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err | cut -c1-300
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], 'fsdp1', d.get('fsdp1'))"
