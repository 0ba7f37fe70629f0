#!/bin/bash
# round 3, call K: full GPU suite + smoke on the cleaned tree (revision 3 removed), the 9 s line with the default policy, the 3 s line
mkdir -p gpurun_out/r3k
O=$GRAFT_REPO_ROOT/gpurun_out/r3k
timeout 1500 python -m pytest tests -m gpu -q -rf -x --durations=8 2>&1 | tail -22 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -3; cut -c1-700 $O/bench_9s.json
timeout 600 python bench.py --video-length 3sec --steps 5 --warmup 2 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_3s.err | grep '^{"metric"' > $O/bench_3s.json
cut -c1-400 $O/bench_3s.json
