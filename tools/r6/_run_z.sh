#!/bin/bash
# round 6, call Z2: GEMM selections for the parts' projections searched BESIDE a kernel that holds 96 CUs (what the pair scan leaves: 160),
# then the TTT layer forward with the committed selections vs the merged ones
cd /root/repo; mkdir -p gpurun_out/r6z2; O=gpurun_out/r6z2
timeout 900 python tools/tune_pipeline_gemms.py $O/tunableop_parts_beside96.csv --video-length 9sec --parts 4 --min-rows 1000 --occupy 96 > $O/tune.log 2>&1; tail -8 $O/tune.log
cat $O/tunableop_parts_beside96.csv
python tools/merge_tuning.py ttt-video-dit_amd/ttt_amd/infra/gemm_tuning_gfx950.csv $O/tunableop_parts_beside96.csv $O/merged.csv
for rep in 1 2; do
timeout 300 python tools/ttt_layer_bench.py --parts 0,4 --rounds 3 > $O/layer_committed_$rep.json 2>$O/layer.err; tail -1 $O/layer_committed_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('committed', {k:v['median_ms'] for k,v in d['by_parts'].items()})"
timeout 300 python tools/ttt_layer_bench.py --parts 0,4 --rounds 3 --tuning-file $O/merged.csv > $O/layer_merged_$rep.json 2>>$O/layer.err; tail -1 $O/layer_merged_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merged   ', {k:v['median_ms'] for k,v in d['by_parts'].items()})"
done
# pair scan version 4 (B0 as a wave-pair hand-off): tests + op-level A/B + stamps
timeout 600 python -X faulthandler -m pytest tests/test_scan_pair_gpu.py tests/test_parity_r5_gpu.py -x -q -m gpu > $O/pair_tests.log 2>&1; echo "pair tests rc=$?"; tail -3 $O/pair_tests.log
timeout 300 python tools/op_bench.py --nc 804 --iters 12 --fwd-only --ab scan_pair --phases > $O/op_nc804_ab_scan_pair.json 2>$O/op.err; tail -1 $O/op_nc804_ab_scan_pair.json
