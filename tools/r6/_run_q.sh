#!/bin/bash
# round 6, call Q: the 8-wave group-sequential tail - TTT tests, op-level and in-step A/B against the per-step tail, kernel statistics
cd /root/repo; mkdir -p gpurun_out/r6q; O=$GRAFT_REPO_ROOT/gpurun_out/r6q
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py tests/test_parity_r6_gpu.py -x -q -m gpu -k "mlp or mfma or bwd or backward or sweep or regime or pipelined or handover" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for nc in 804 282; do
timeout 300 python tools/op_bench.py --nc $nc --iters 16 --ab tail5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nc $nc tail5 0/1', {k: round(v['bwd_avg_ms'],3) for k,v in d['ab'].items() if isinstance(v, dict)})"
done
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_tail5_$rep.json 2> $O/bench_tail5_$rep.err; show $O/bench_tail5_$rep.json tail5
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option tail5=0 > $O/bench_tail4_$rep.json 2> $O/bench_tail4_$rep.err; show $O/bench_tail4_$rep.json tail5=0
done
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 6 > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_tail5_kernel_stats.csv && head -6 "$f" | cut -c1-200
