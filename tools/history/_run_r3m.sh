#!/bin/bash
# call M: one-rank TP test on the device, the 18 s / 30 s lines after the fixes of call L, the TP code path on one GPU
cd /root/repo; mkdir -p gpurun_out/r3m; O=gpurun_out/r3m
timeout 240 python -m pytest tests/test_zz_replica_gpu.py -x -q -m gpu -s > $O/tp_test.log 2>&1; echo "tp test rc=$?" >> $O/tp_test.log
timeout 330 python bench.py --video-length 18sec --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > $O/bench_18s.json 2> $O/bench_18s.err; echo "18s rc=$?"
timeout 200 python bench.py --video-length 3sec --tp 1 --steps 2 --warmup 1 > $O/bench_3s_tp1.json 2> $O/bench_3s_tp1.err; echo "tp1 rc=$?"
timeout 400 python bench.py --video-length 30sec --steps 1 --warmup 1 --no-cpu-baseline --no-fsdp1-compare > $O/bench_30s.json 2> $O/bench_30s.err; echo "30s rc=$?"
tail -5 $O/tp_test.log; cat $O/bench_18s.json $O/bench_30s.json $O/bench_3s_tp1.json | cut -c1-600
tail -3 $O/bench_18s.err $O/bench_30s.err $O/bench_3s_tp1.err
