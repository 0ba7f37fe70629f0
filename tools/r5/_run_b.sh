#!/bin/bash
# round 5, call B: full GPU suite on the tree with FlatFSDP's optimizer classes, the ld_out pre-backward kernels and the fused q/k/v
# backward; TunableOp search for the fused GEMM shapes; op-level A/B; in-step A/B (two short bench runs through the new orchestrator)
cd /root/repo; mkdir -p gpurun_out/r5b; O=$GRAFT_REPO_ROOT/gpurun_out/r5b
export TMPDIR=/tmp
timeout 700 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/gpu_suite.log
timeout 600 python tools/qkv_backward_bench.py --tune $O/tunableop_qkv.csv > $O/qkv_backward_ab_tuning.json 2> $O/qkv_tune.err; echo "tune rc=$?"; cat $O/qkv_backward_ab_tuning.json
# merge the new selections into the committed file (on this box; the merged file comes back for committing)
CSV=ttt-video-dit_amd/ttt_amd/infra/gemm_tuning_gfx950.csv
ls $O/tunableop_qkv*.csv
for f in $O/tunableop_qkv*.csv; do grep -v "^Validator" $f | grep "9216" >> $CSV; done
cp $CSV $O/gemm_tuning_gfx950_merged.csv; tail -8 $CSV | cut -c1-150
timeout 300 python tools/qkv_backward_bench.py > $O/qkv_backward_ab.json 2> $O/qkv_ab.err; echo "ab rc=$?"; cat $O/qkv_backward_ab.json
for fuse in 1 0; do
  TTT_FUSE_QKV_BACKWARD=$fuse timeout 600 python bench.py --steps 3 --warmup 1 --no-legs --no-cpu-baseline --no-fsdp1-compare > $O/bench_fuse$fuse.json 2> $O/bench_fuse$fuse.err; echo "bench fuse=$fuse rc=$?"
  grep -h "^{" $O/bench_fuse$fuse.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), 'free', d['config']['remat_free_layers'], 'ttt bwd', round(r['avg_launch_ms'],3), 'wall', d.get('bench_wall_s'), d.get('optimizer'))"
done
