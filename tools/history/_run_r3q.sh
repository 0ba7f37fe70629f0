#!/bin/bash
# call Q: the TP code path after the re-wiring (layers through __call__, checkpointing on token shards, TP x FSDP) on one rank
cd /root/repo; mkdir -p gpurun_out/r3q; O=gpurun_out/r3q
timeout 120 python -m pytest tests/test_zz_replica_gpu.py -x -q -m gpu > $O/tp_test.log 2>&1; echo "tp test rc=$?"; tail -2 $O/tp_test.log
timeout 170 python bench.py --video-length 3sec --tp 1 --steps 2 --warmup 1 > $O/bench_3s_fsdp1xtp1.json 2> $O/bench_3s_fsdp1xtp1.err; echo "3s fsdp x tp rc=$?"
timeout 240 python bench.py --video-length 9sec --tp 1 --fsdp off --steps 1 --warmup 1 > $O/bench_9s_tp1.json 2> $O/bench_9s_tp1.err; echo "9s tp1 rc=$?"
grep -h "^{" $O/*.json | cut -c1-760
grep -h "^\[bench\|bench.py:\|Error" $O/*.err | tail -12
