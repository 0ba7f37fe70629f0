#!/bin/bash
# Round 4, call P (one box): FlatFSDP with the gradient cast on the compute stream: device test, then the default line (replica path,
# no fp32 masters of frozen parameters) with its fsdp1 point (flat + collectives over one rank) and the flat path without collectives.
cd /root/repo; mkdir -p gpurun_out/r4p; O=$GRAFT_REPO_ROOT/gpurun_out/r4p
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_replica_gpu.py -m gpu -x -q > $O/test_replica_flat.log 2>&1; echo "test rc=$?"; tail -2 $O/test_replica_flat.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -2 $O/bench_default.err | cut -c1-300
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], d['config']['parallelism'], 'fsdp1', d.get('fsdp1')); print('roofline', d['roofline'])"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --fsdp flat > $O/bench_flat.json 2> $O/bench_flat.err; echo "flat rc=$?"
grep -h "^{" $O/bench_flat.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat (no collectives)', d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'])"
