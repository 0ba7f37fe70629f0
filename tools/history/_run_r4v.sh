#!/bin/bash
# Round 4, call V: default line + fsdp1 point with FlatFSDP holding a unit's bf16 gradients until the end of the backward
# (persistent reduce buffers, nothing allocated and nothing freed inside the backward).
cd /root/repo; mkdir -p gpurun_out/r4v; O=$GRAFT_REPO_ROOT/gpurun_out/r4v
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"
grep -h "^{" $O/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('default', round(d['value'],1), round(d['ms_per_step'],1), d['config']['remat_free_layers'], d['peak_mem_gib'], 'ttt bwd', round(r['avg_launch_ms'],3), 'attn bwd', round(r['other']['attn_bwd']['avg_ms'],3), 'fsdp1', d.get('fsdp1'))"
