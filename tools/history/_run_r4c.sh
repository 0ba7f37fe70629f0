#!/bin/bash
# Round 4, call C: sweep variants (bf16 records x owner order x deriver wave placement), dQ with 64 query rows per wave,
# stage stamps with the owners' pre-Bb phases, the 9 s line on the new defaults.
cd /root/repo; mkdir -p gpurun_out/r4c; O=$GRAFT_REPO_ROOT/gpurun_out/r4c
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity_r4_gpu.py tests/test_attention_gpu.py "tests/test_parity_r3_gpu.py::test_backward_is_run_to_run_deterministic_at_the_benchmarked_head_count" "tests/test_parity_r3_gpu.py::test_dit_multiscene_on_hip_path_vs_reference_lastrow" -q -m gpu -s > $O/tests_variants.log 2>&1; echo "variant tests rc=$?"; tail -3 $O/tests_variants.log | cut -c1-300; grep -h "lr_gate" $O/tests_variants.log | cut -c1-300
for opt in sweep_owner_overlap sweep_deriver_wave0 sweep_records_bf16; do
  timeout 120 python tools/op_bench.py --nc 804 --iters 12 --ab $opt > $O/op_ab_${opt}_nc804.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_${opt}_nc804.json').read().strip().splitlines()[-1]);print('nc804',d['ab'])"
done
python - <<'PY'
import subprocess, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4c"
# phases: default, derivers on waves 2/3, fp32 records
for tag, extra in (("default", []), ("dw2", ["--ab-fixed", "sweep_deriver_wave0=2"]), ("fp32rec", ["--ab-fixed", "sweep_records_bf16=0"])):
    r = subprocess.run(["python", "tools/op_bench.py", "--nc", "804", "--iters", "4", "--phases"] + extra, capture_output=True, text=True, timeout=200)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        open(f"{O}/op_phases_{tag}.json", "w").write(line[-1] + "\n")
        d = json.loads(line[-1]); ph = d["phase_cycles_per_step"]
        print(tag, "bwd", round(d["bwd"]["avg_ms"], 3), "compute", ph[16:24], "owners", ph[24:28], "derivers", ph[28:32], "owners pre-Bb", ph[32:36])
    else:
        print(tag, "failed", r.stderr[-300:])
PY
timeout 300 python tools/attn_bench.py --no-sdpa --stages 2:2,2w:2,1w:2,2:3,2w:3 --rounds 5 > $O/attn_bench_dq_wide.log 2>&1; echo "attn_bench rc=$?"; grep "^{" $O/attn_bench_dq_wide.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bit_identical_to_first'], {k: round(v['median_ms'],3) for k,v in d['bwd_by_stage'].items()})"
timeout 420 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fsdp1-compare --remat-keep attn,scan,fc2 > $O/bench_9s.json 2> $O/bench_9s.err; echo "bench rc=$?"; grep -h "^{" $O/bench_9s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['config']['remat_free_layers'], d['peak_mem_gib'], 'bwd', r['avg_launch_ms'], {k: round(v['avg_ms'],2) for k,v in r['other'].items()})"
