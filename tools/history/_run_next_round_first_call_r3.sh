#!/bin/bash
# Prepared at the end of round 3 (no GPU-minutes left): the A/Bs that decide round 3's opt-in items, for the FIRST gpurun call of
# the next round (~12 GPU-minutes).  Same box, back to back, so that the lines are comparable (boxes differ by ~6 %).
#   1. the 9 s line with re-materialised layers keeping attn,scan (default) vs attn,scan,fc2 (DESIGN.md section 8, item 4:
#      +1 % by arithmetic); adopt as the default of bench.py --remat-keep if it wins
#   2. the full GPU suite (incl. the stolen-CU stress test, the one-rank TP layouts, remat_keep with fc2)
#   3. (second call, after `git merge attn-staged-device` and a rebuild) the staged / swizzled attention backward kernels: first
#      find out why their first device run aborted (AMD_LOG_LEVEL=3 shows the runtime's own message):
#        AMD_LOG_LEVEL=3 timeout 60 python -m pytest tests/test_attention_gpu.py -x -q -m gpu -k two_tiles 2>&1 | tail -40
#      then the interleaved A/B at the training geometry:
#        python tools/attn_bench.py --no-sdpa --stages 1,2,3,4 --rounds 5
#   4. allocator fragmentation: the 30 s line reserves 275 GiB for 216 GiB allocated (9 s: 252 for 241) - every GiB of that gap
#      is a kept kernel output or a remat-free layer that does not fit.  One try:
#        PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python bench.py --video-length 30sec --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare
cd /root/repo; mkdir -p gpurun_out/r4a; O=gpurun_out/r4a
timeout 300 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_suite.log
for keep in attn,scan attn,scan,fc2; do
  timeout 330 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --remat-keep $keep > $O/bench_9s_keep_${keep//,/_}.json 2> $O/bench_9s_keep_${keep//,/_}.err
  echo "keep=$keep rc=$?"; grep -h "^{" $O/bench_9s_keep_${keep//,/_}.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'])"
done
