#!/bin/bash
# round 6, call OE2: own_early with the owners' step loads requested behind Bc of the previous iteration - TTT tests, interleaved A/B, stamps, in-step
cd /root/repo; mkdir -p gpurun_out/r6oe2; O=gpurun_out/r6oe2
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r3_gpu.py tests/test_parity_r4_gpu.py tests/test_parity_r2_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --ab own_early > $O/op_nc804_ab_own_early_$rep.json 2>$O/op.err; tail -1 $O/op_nc804_ab_own_early_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ab'])"
done
timeout 300 python tools/op_bench.py --nc 804 --iters 4 --phases > $O/op_nc804_phases_early.json 2>>$O/op.err; tail -1 $O/op_nc804_phases_early.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); ph=d['phase_cycles_per_step']; print('early', {k: round(ph[k]) for k in range(16,36)})"
timeout 300 python tools/op_bench.py --nc 804 --iters 4 --phases --ab-fixed own_early=0 > $O/op_nc804_phases_late.json 2>>$O/op.err; tail -1 $O/op_nc804_phases_late.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); ph=d['phase_cycles_per_step']; print('late ', {k: round(ph[k]) for k in range(16,36)})"
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'parts', c.get('ttt_pipeline_parts'), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_early_$rep.json 2> $O/bench_early_$rep.err; show $O/bench_early_$rep.json early
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option own_early=0 > $O/bench_late_$rep.json 2> $O/bench_late_$rep.err; show $O/bench_late_$rep.json late
done
