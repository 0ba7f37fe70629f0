#!/bin/bash
# round 2, call Y: share of the sweep's cluster workgroups that proved same-XCD placement, replica path vs FSDP2 path (4 layers)
mkdir -p gpurun_out/r2y
O=gpurun_out/r2y
for mode in off on; do
  timeout 300 python bench.py --fsdp $mode --layers 4 --remat-free-layers 2 --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fsdp $mode:', round(d['ms_per_step'],1), 'ms/step, backward', round(r['avg_launch_ms'],2), 'ms, sweep_same_xcd_frac', r['sweep_same_xcd_frac'])" | tee -a $O/same_xcd.txt
done
