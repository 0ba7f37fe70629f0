#!/bin/bash
# round 3, call I: duration of the round-3 GPU tests with the thread cap; the 9 s line with and without kept kernel outputs in
# re-materialised layers (same box)
mkdir -p gpurun_out/r3i
O=$GRAFT_REPO_ROOT/gpurun_out/r3i
timeout 900 python -m pytest tests/test_parity_r3_gpu.py -q -m gpu --durations=12 2>&1 | tail -24 | cut -c1-200 | tee $O/pytest_r3.txt
for keep in attn,scan none attn; do
  timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --remat-keep $keep 2>$O/bench_$keep.err | grep '^{"metric"' > $O/bench_9s_keep_$keep.json
  echo "keep=$keep"; grep "timed region" $O/bench_$keep.err | tail -1; python - "$O/bench_9s_keep_$keep.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print({k:d[k] for k in ("value","ms_per_step","peak_mem_gib")}, d["config"]["remat_free_layers"], d["config"].get("remat_keep"), "bwd ms", round(r["avg_launch_ms"],2), "frac", round(r["frac"],4), {k:round(v["avg_ms"],2) for k,v in r["other"].items()})
P
done
