#!/bin/bash
# round 6, call A (measurement on the round-5 kernels + schedule 2): fresh stage stamps of the sweep, checkpoint groups per chunk
# 1 / 2 / 3 / 5 under schedules 1 and 2 (the next recompute beside the sweep), L2 hit / miss of the three backward kernels,
# the large GEMMs alone vs inside the step (torch profiler by shape), attention baseline, the schedule test
cd /root/repo; mkdir -p gpurun_out/r6a; O=$GRAFT_REPO_ROOT/gpurun_out/r6a
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_r2_gpu.py -x -q -m gpu -k "tail_under_next_sweep" > $O/sched_test.log 2>&1; echo "sched test rc=$?"; tail -3 $O/sched_test.log
timeout 200 python tools/op_bench.py --nc 804 --iters 10 --phases > $O/op_nc804_phases.json 2>$O/op_nc804_phases.err; echo "phases rc=$?"
for ov in 1 2; do for gpc in 1 2 3 5; do
  timeout 200 python tools/op_bench.py --nc 804 --iters 6 --overlap $ov --gpc $gpc > $O/op_nc804_ov${ov}_gpc$gpc.json 2>/dev/null
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("$O/op_nc804_ov${ov}_gpc$gpc.json") if l.startswith("{")][0]
    print("overlap $ov gpc $gpc bwd avg %.3f min %.3f fwd %.3f" % (d["bwd"]["avg_ms"], d["bwd"]["min_ms"], d["fwd"]["avg_ms"]))
except Exception as ex: print("overlap $ov gpc $gpc failed", ex)
PY
done; done
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "TCC_(HIT|MISS|EA0_RDREQ|EA0_WRREQ|EA0_RD_UNCACHED|REQ|READ|WRITE|BUBBLE|TAG_STALL|MALL)|MALL" | head -60 > $O/counters_available.txt
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_$n -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /tmp/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_$n.csv || { echo "pmc $n failed"; tail -3 /tmp/pmc_$n.log; }
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/op_nc804_pmc_*.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(f.split("pmc_")[-1][:-4], k, {c: "%.3e per launch (%d)" % (v / n[(k, c)], n[(k, c)]) for c, v in d.items()})
PY
cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_isolated.py > $O/gemm_isolated.json 2> $O/gemm_isolated.err; echo "gemm rc=$?"; cat $O/gemm_isolated.err | tail -10
timeout 200 python tools/attn_bench.py --no-sdpa > $O/attn_bench.txt 2>&1; tail -4 $O/attn_bench.txt
timeout 600 python bench.py --role worker --gpus 1 --steps 2 --warmup 1 --no-fsdp1-compare --torch-profile $O/bench_torch_profile.txt > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"
grep -h "^{" $O/bench_short.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],1), 'free', c['remat_free_layers'], 'bwd ms', round(r['avg_launch_ms'],3), 'clk', c.get('clock_mhz_avg'), 'W', c.get('power_w_avg'), c.get('clocks'), {k: v for k,v in r.items() if k.endswith('_ms')})" || tail -20 $O/bench_short.err
grep -E "aten::(mm|addmm|linear|matmul)" $O/bench_torch_profile.txt | head -40 | cut -c1-250
ls -la $O | head -40
