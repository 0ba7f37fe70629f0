#!/bin/bash
# round 6, call P: schedule 3 (tails on their own low-priority stream) and the stage-major reverse step - tests, op-level and in-step A/Bs
cd /root/repo; mkdir -p gpurun_out/r6p; O=$GRAFT_REPO_ROOT/gpurun_out/r6p
timeout 900 python -m pytest tests/test_parity_r2_gpu.py tests/test_parity_r3_gpu.py tests/test_parity_r6_gpu.py -x -q -m gpu -k "tail_under or deterministic or stolen or barrier_inside" 2>&1 | tail -3
ab() { timeout 300 python tools/op_bench.py --nc 804 --iters 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', {k: round(v['bwd_avg_ms'],3) for k,v in d['ab'].items() if isinstance(v, dict)})"; }
ab --ab overlap_tail --ab-values 2,3 --ab-restore 2
ab --ab overlap_tail --ab-values 2,3 --ab-restore 2 --ab-fixed tail5=0
ab --ab deriver_il --overlap 3
ab --ab tail5 --overlap 3
show() { grep -h "^{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'bwd', round(r['avg_launch_ms'],3), 'fwd', r.get('scan_fwd_ms'), 'attn', r.get('attn_fwd_ms'), r.get('attn_bwd_ms'), 'clk', c.get('clock_mhz_avg'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 ${1%.json}.err; }
for rep in 1 2; do
for cfg in "overlap_tail=3" "overlap_tail=3,tail5=0" "overlap_tail=3,deriver_il=1" "overlap_tail=2,tail5=0"; do
  opts=""; for kv in ${cfg//,/ }; do opts="$opts --debug-option $kv"; done
  timeout 600 python bench.py --role worker --gpus 1 --steps 3 --warmup 1 --no-fsdp1-compare --remat-free-layers 13 $opts > $O/bench_${cfg//,/_}_$rep.json 2> $O/bench_${cfg//,/_}_$rep.err; show $O/bench_${cfg//,/_}_$rep.json "$cfg"
done; done
