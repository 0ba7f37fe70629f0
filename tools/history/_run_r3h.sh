#!/bin/bash
# round 3, call H: full GPU suite on revision 4 (default), smoke, op-level timing, the 9 s bench line (short)
mkdir -p gpurun_out/r3h
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
timeout 1500 python -m pytest tests -m gpu -q -rf -x 2>&1 | tail -15 | cut -c1-300 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
for nc in 804 282; do timeout 100 python tools/op_bench.py --nc $nc --iters 8 2>/dev/null | python tools/_fmt_phases.py "rev4 nc$nc" | tee -a $O/op_final.txt; done
timeout 100 python tools/op_bench.py --nc 804 --bwd-rev 3 --iters 8 2>/dev/null | python tools/_fmt_phases.py "rev3 nc804" | tee -a $O/op_final.txt
timeout 900 python bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-fsdp1-compare 2>$O/bench_9s.err | grep '^{"metric"' > $O/bench_9s.json
grep "bench " $O/bench_9s.err | tail -4; cut -c1-900 $O/bench_9s.json
