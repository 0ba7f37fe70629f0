#!/bin/bash
# Round 4, call F: the two-part reverse step (sweep_prederive) on the device: parity at chunk / group edges, determinism, A/B, stamps.
cd /root/repo; mkdir -p gpurun_out/r4f; O=$GRAFT_REPO_ROOT/gpurun_out/r4f
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity_r4_gpu.py -q -m gpu -s > $O/tests_variants.log 2>&1; echo "variant tests rc=$?"; tail -4 $O/tests_variants.log | cut -c1-400
for nc in 804 282; do
  timeout 120 python tools/op_bench.py --nc $nc --iters 12 --ab sweep_prederive > $O/op_ab_prederive_nc$nc.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_prederive_nc$nc.json').read().strip().splitlines()[-1]);print('nc$nc prederive',d['ab'])"
done
timeout 120 python tools/op_bench.py --nc 804 --iters 12 --ab-fixed sweep_deriver_wave0=2 --ab sweep_prederive > $O/op_ab_prederive_dw2_nc804.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/op_ab_prederive_dw2_nc804.json').read().strip().splitlines()[-1]);print('nc804 dw2 prederive',d['ab'])"
python - <<'PY'
import subprocess, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r4f"
for tag, extra in (("pre", ["--ab-fixed", "sweep_prederive=1"]), ("nopre", [])):
    r = subprocess.run(["python", "tools/op_bench.py", "--nc", "804", "--iters", "4", "--phases"] + extra, capture_output=True, text=True, timeout=200)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line:
        open(f"{O}/op_phases_{tag}.json", "w").write(line[-1] + "\n")
        d = json.loads(line[-1]); ph = d["phase_cycles_per_step"]
        print(tag, "bwd", round(d["bwd"]["avg_ms"], 3), "compute", ph[16:24], "owners", ph[24:28], "derivers", ph[28:32], "owners pre-Bb", ph[32:36], "window", ph[36])
    else:
        print(tag, "failed", r.stderr[-300:])
PY
timeout 100 python tools/_det_check.py > $O/det_check.log 2>&1; tail -5 $O/det_check.log | cut -c1-300
