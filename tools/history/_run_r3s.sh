#!/bin/bash
# call S: fused glue kernels on token shards - parity (one-rank TP test) and the 3 s line of the code path
cd /root/repo; mkdir -p gpurun_out/r3s; O=gpurun_out/r3s
timeout 50 python -m pytest tests/test_zz_replica_gpu.py -x -q -m gpu -s -k tensor_parallel > $O/tp_test.log 2>&1; echo "tp test rc=$?"; grep -E "LAYOUT|passed|failed|Error" $O/tp_test.log | cut -c1-200
timeout 115 python bench.py --video-length 3sec --tp 1 --fsdp off --steps 2 --warmup 1 > $O/bench_3s_tp1.json 2> $O/bench_3s_tp1.err; echo "rc=$?"
grep -h "^{" $O/*.json | cut -c1-260; grep -h "^\[bench\|Error" $O/*.err | tail -5
