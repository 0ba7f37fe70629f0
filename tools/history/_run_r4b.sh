#!/bin/bash
# Round 4, call B: the new sweep / attention variants on the device (tests), interleaved A/Bs on one box (owner overlap, bf16
# records, attention tiles per stage), the 9 s line with the new defaults and the fc2 keep A/B, PMC traffic of the new build.
cd /root/repo; mkdir -p gpurun_out/r4b; O=$GRAFT_REPO_ROOT/gpurun_out/r4b
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity_r4_gpu.py tests/test_attention_gpu.py "tests/test_parity_r3_gpu.py::test_backward_is_run_to_run_deterministic_at_the_benchmarked_head_count" -q -m gpu -s -x > $O/tests_variants.log 2>&1; echo "variant tests rc=$?"; tail -3 $O/tests_variants.log | cut -c1-300
for nc in 804 282; do
  timeout 120 python tools/op_bench.py --nc $nc --iters 12 --ab sweep_owner_overlap > $O/op_ab_owner_overlap_nc$nc.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_owner_overlap_nc$nc.json').read().strip().splitlines()[-1]);print('nc$nc owner_overlap',d['ab'], 'fwd', d['fwd']['avg_ms'])"
  timeout 120 python tools/op_bench.py --nc $nc --iters 12 --ab sweep_records_bf16 > $O/op_ab_records_bf16_nc$nc.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_records_bf16_nc$nc.json').read().strip().splitlines()[-1]);print('nc$nc records_bf16',d['ab'])"
done
timeout 120 python tools/op_bench.py --nc 804 --iters 4 --phases > $O/op_phases_nc804.json 2>/dev/null; tail -1 $O/op_phases_nc804.json | cut -c1-600
timeout 300 python tools/attn_bench.py --no-sdpa --stages 1:1,2:2,2:1,1:2,2:3,2:4 --rounds 5 > $O/attn_bench_stages.log 2>&1; echo "attn_bench rc=$?"; grep "^{" $O/attn_bench_stages.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_identical_to_first'], {k: round(v['median_ms'],3) for k,v in d['bwd_by_stage'].items()})"
for keep in attn,scan attn,scan,fc2; do
  timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --remat-keep $keep > $O/bench_9s_keep_${keep//,/_}.json 2> $O/bench_9s_keep_${keep//,/_}.err
  echo "keep=$keep rc=$?"; grep -h "^{" $O/bench_9s_keep_${keep//,/_}.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], 'bwd', r['avg_launch_ms'], {k: round(v['avg_ms'],2) for k,v in r['other'].items()})"
done
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "mlp_" --output-format csv -d /tmp/pmc_804_$c -- python $GRAFT_REPO_ROOT/tools/op_bench.py --nc 804 --iters 2 > /dev/null 2>&1
  f=$(find /tmp/pmc_804_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/op_nc804_pmc_$c.csv
done
ls -la $O | head -30
