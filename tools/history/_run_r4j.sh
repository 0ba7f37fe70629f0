#!/bin/bash
# Round 4, call J: FlatFSDP on the device (one-rank RCCL group with its collectives) + the default bench line with the flat fsdp1 point.
cd /root/repo; mkdir -p gpurun_out/r4j; O=$GRAFT_REPO_ROOT/gpurun_out/r4j
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_replica_gpu.py -q -m gpu -s -k "flat or replica_equals" > $O/tests_flat.log 2>&1; echo "flat tests rc=$?"; grep -h "TRACE\|passed\|failed\|Error" $O/tests_flat.log | cut -c1-400 | tail -8
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err | cut -c1-300
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], 'fsdp1', d.get('fsdp1'))"
