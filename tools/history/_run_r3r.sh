#!/bin/bash
# call R: FSDP x TP on one rank at 3 s again: the timed region must no longer measure allocator retries
cd /root/repo; mkdir -p gpurun_out/r3r; O=gpurun_out/r3r
timeout 230 python bench.py --video-length 3sec --tp 1 --steps 2 --warmup 1 > $O/bench_3s_fsdp1xtp1.json 2> $O/bench_3s_fsdp1xtp1.err; echo "rc=$?"
grep -h "^{" $O/*.json | cut -c1-300; grep -h "^\[bench\|bench.py:\|Error" $O/*.err | tail -12
