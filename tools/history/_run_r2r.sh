#!/bin/bash
# round 2, call R: SegmentSplit / SegmentMerge on the GPU - the multi-scene DiT tests and the 9 s bench line
mkdir -p gpurun_out/r2r
O=gpurun_out/r2r
timeout 900 python -m pytest tests/test_parity_r2_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -k "dit_on_hip or transformer_layer or cogvideox or multiscene" 2>&1 | tail -6 | cut -c1-300 | tee $O/pytest_dit.txt
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -4; cut -c1-500 $O/bench_9s.json
for i in 1 2 3; do timeout 200 python tools/op_bench.py --nc 804 --iters 5 2>/dev/null | python tools/_fmt_phases.py "nc 804 overlap 1 run $i:" | tee -a $O/op_repeat.txt; done
timeout 200 python tools/op_bench.py --nc 804 --iters 5 --overlap 0 2>/dev/null | python tools/_fmt_phases.py "nc 804 overlap 0:" | tee -a $O/op_repeat.txt
