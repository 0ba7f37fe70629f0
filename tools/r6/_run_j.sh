#!/bin/bash
# round 6, call J: the driver's command on the tree so far (main 9 s measurement + fsdp1 + the four legs + cpu_baseline), then the PMC passes
# for roofline.traffic (FETCH_SIZE / WRITE_SIZE over the TTT-MLP kernels at NC = 804) and MFMA busy
cd /root/repo; mkdir -p gpurun_out/r6j; O=$GRAFT_REPO_ROOT/gpurun_out/r6j
export TMPDIR=/tmp
timeout 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
grep -h "^{" $O/bench_default.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'frac', round(r['frac'],4), 'bwd ms', round(r['avg_launch_ms'],3), {k: v for k,v in r.items() if k.endswith('_ms')}, 'clk', c.get('clock_mhz_avg'), c.get('power_w_avg'))
for k,v in c.get('legs',{}).items(): print(k, {kk: vv for kk,vv in v.items() if kk in ('value','ms_per_step','peak_mem_gib','latent_frames_per_s','projected_50_step_video_s','error','skipped','leg_wall_s')})
print('fsdp1', c.get('fsdp1')); print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')} if 'cpu_baseline' in d else None, 'wall', d.get('bench_wall_s'))" || tail -20 $O/bench_default.err
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_804_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_804_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_$c.csv
done
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_sq.csv || tail -5 /tmp/pmc_sq.log
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-include-regex "attn_" --output-format csv -d /tmp/pmc_sq_attn -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --no-sdpa > /tmp/pmc_sq_attn.log 2>&1
f=$(find /tmp/pmc_sq_attn -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/attn_pmc_sq.csv || tail -5 /tmp/pmc_sq_attn.log
ls -la $O
