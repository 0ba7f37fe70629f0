#!/bin/bash
# round 6, call S3: backward schedule 3 (recompute beside the sweep, tail BEHIND its sweep on the whole chip): bit identity, op level, in-step
cd /root/repo; mkdir -p gpurun_out/r6s3; O=gpurun_out/r6s3
timeout 900 python -m pytest tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py -x -q -m gpu -k "tail_under_next_sweep or run_to_run or stolen" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for ov in 3 2; do
timeout 200 python tools/op_bench.py --nc 804 --iters 10 --overlap $ov 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $ov fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3), 'min', round(d['bwd']['min_ms'],3))" | tee -a $O/overlap_ab.txt
done; done
for nc in 282 2630; do for ov in 3 2; do
timeout 300 python tools/op_bench.py --nc $nc --iters 6 --overlap $ov 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nc $nc overlap $ov fwd', round(d['fwd']['avg_ms'],3), 'bwd', round(d['bwd']['avg_ms'],3))" | tee -a $O/overlap_ab.txt
done; done
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'parts', c.get('ttt_pipeline_parts'), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 --debug-option overlap_tail=3 > $O/bench_ov3_$rep.json 2> $O/bench_ov3_$rep.err; show $O/bench_ov3_$rep.json schedule3
timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 > $O/bench_ov2_$rep.json 2> $O/bench_ov2_$rep.err; show $O/bench_ov2_$rep.json schedule2
done
