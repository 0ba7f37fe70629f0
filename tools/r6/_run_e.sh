#!/bin/bash
# round 6, call E: issue priorities by wave role in the sweep (sweep_prio) x where barrier Bc falls in the reverse step, schedule 2
cd /root/repo; mkdir -p gpurun_out/r6e; O=$GRAFT_REPO_ROOT/gpurun_out/r6e
for sp in 1 2 3 5; do
timeout 300 python tools/op_bench.py --nc 804 --iters 16 --overlap 2 --ab-fixed deriver_split=$sp --ab sweep_prio --ab-restore 0 2>/dev/null | python -c "import sys,json; print('split $sp prio 0/1', {k: round(v['bwd_avg_ms'],3) for k,v in json.loads(sys.stdin.read().strip().splitlines()[-1])['ab'].items() if isinstance(v, dict)})"
done
