#!/bin/bash
# round 6, call OV: backward schedules 0 / 1 / 2 on the current kernels (op level, NC = 804), and per-kernel times under each
cd /root/repo; mkdir -p gpurun_out/r6ov; O=gpurun_out/r6ov
for rep in 1 2; do for ov in 2 1 0; do
timeout 200 python tools/op_bench.py --nc 804 --iters 10 --overlap $ov 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $ov fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3), 'min', round(d['bwd']['min_ms'],3))" | tee -a $O/overlap_ab.txt
done; done
