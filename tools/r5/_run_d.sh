#!/bin/bash
# round 5, call D: the pipelined TTT layer forward - parity on a 2-layer DiT, op-level A/B at the 9 s geometry, in-step A/B - on the
# pruned library (five debug options left)
cd /root/repo; mkdir -p gpurun_out/r5d; O=$GRAFT_REPO_ROOT/gpurun_out/r5d
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_r5_gpu.py tests/test_parity_r4_gpu.py tests/test_attention_gpu.py -x -q -m gpu -s > $O/r5_tests.log 2>&1; echo "tests rc=$?"; grep -h "pipelined forward vs\|passed\|failed\|Error" $O/r5_tests.log | tail -8
timeout 600 python tools/ttt_layer_bench.py --parts 0,2,3,4 --rounds 3 > $O/ttt_layer_pipeline_ab.json 2> $O/ttt_layer.err; echo "layer rc=$?"; cat $O/ttt_layer_pipeline_ab.json; tail -3 $O/ttt_layer.err
for parts in 0 3 4 0 3; do
  i=$((i+1))
  timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-fsdp1-compare --pipeline-parts $parts > $O/bench_p${parts}_$i.json 2> $O/bench_p${parts}_$i.err; echo "bench parts=$parts rc=$?"
  grep -h "^{" $O/bench_p${parts}_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'ttt bwd', round(r['avg_launch_ms'],3), {k: (round(v['avg_ms'],3), v.get('parts_per_scan')) for k,v in r['other'].items()}, 'parts', d['config'].get('ttt_pipeline_parts'), 'peak', round(d['peak_mem_gib'],1))" || tail -5 $O/bench_p${parts}_$i.err
done
