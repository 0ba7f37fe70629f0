#!/bin/bash
# round 5, call C: the forward in parts + the pipelined layer forward on the device (parity), TunableOp for the strided per-projection
# weight gradients, op-level A/Bs (q/k/v backward: three variants; TTT layer: pipeline parts), in-step A/Bs (short bench runs)
cd /root/repo; mkdir -p gpurun_out/r5c; O=$GRAFT_REPO_ROOT/gpurun_out/r5c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_r5_gpu.py tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py -x -q -m gpu -s > $O/r5_tests.log 2>&1; echo "tests rc=$?"; grep -h "pipelined forward vs\|passed\|failed\|Error" $O/r5_tests.log | tail -8
timeout 600 python tools/qkv_backward_bench.py --tune $O/tunableop_qkv2.csv > $O/qkv_backward_ab2.json 2> $O/qkv_tune2.err; echo "tune rc=$?"; cat $O/qkv_backward_ab2.json; tail -2 $O/qkv_tune2.err
CSV=ttt-video-dit_amd/ttt_amd/infra/gemm_tuning_gfx950.csv
python - <<'PY'
import os
base='ttt-video-dit_amd/ttt_amd/infra/gemm_tuning_gfx950.csv'; new=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r5c/tunableop_qkv2.csv'
b=open(base).read().rstrip('\n').split('\n'); have={','.join(l.split(',')[:2]) for l in b if not l.startswith('Validator')}
add=[l for l in open(new).read().split('\n') if l and not l.startswith('Validator') and ','.join(l.split(',')[:2]) not in have]
open(base,'w').write('\n'.join(b+add)+'\n'); print('merged', len(add)); print('\n'.join(add))
PY
cp $CSV $O/gemm_tuning_gfx950_merged.csv
timeout 300 python tools/qkv_backward_bench.py > $O/qkv_backward_ab3.json 2> $O/qkv_ab3.err; echo "ab3 rc=$?"; cat $O/qkv_backward_ab3.json
timeout 600 python tools/ttt_layer_bench.py --parts 0,2,3,4 --rounds 3 > $O/ttt_layer_pipeline_ab.json 2> $O/ttt_layer.err; echo "layer rc=$?"; cat $O/ttt_layer_pipeline_ab.json; tail -3 $O/ttt_layer.err
for cfg in "0 dgrad" "3 dgrad" "0 0" "4 dgrad"; do
  set -- $cfg
  TTT_FUSE_QKV_BACKWARD=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-fsdp1-compare --pipeline-parts $1 > $O/bench_p$1_f$2.json 2> $O/bench_p$1_f$2.err; echo "bench parts=$1 fuse=$2 rc=$?"
  grep -h "^{" $O/bench_p$1_f$2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'ttt bwd', round(r['avg_launch_ms'],3), {k: (round(v['avg_ms'],3), v.get('parts_per_scan')) for k,v in r['other'].items()}, 'parts', d['config'].get('ttt_pipeline_parts'))"
done
