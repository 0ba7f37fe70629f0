#!/bin/bash
# gpurun helper (next round's FIRST call): parity of every opt-in variant, then their A/B timings, all in one call.
#   bash tools/_run_ab.sh            # ~10 GPU-minutes: tests + op-level A/Bs
#   bash tools/_run_ab.sh bench      # + the full-model bench lines (default, each flag alone, all flags): ~25 GPU-minutes more
# Everything lands in gpurun_out/ab/ (copy what should be judged into profiles/).
mkdir -p gpurun_out/ab
O=gpurun_out/ab
# ---- parity first: kernel variants, attention revision 2 == revision 1, side-stream weight gradients ----------------------
TTT_TEST_VARIANTS=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -rf -k "variant or wgrad_overlap" 2>&1 | tail -20 | tee $O/variant_tests.txt
TTT_TEST_VARIANTS=1 timeout 400 python -m pytest tests/test_attention_gpu.py -q -rf -k "v2_equals_v1 or dkdv_variants" 2>&1 | tail -25 | tee -a $O/variant_tests.txt
TTT_TEST_VARIANTS=1 timeout 400 python -m pytest tests/test_zz_replica_gpu.py -q -rf 2>&1 | tail -25 | tee -a $O/variant_tests.txt
# ---- design parameter of the planned multi-CU backward sweep: cost of a workgroup-to-workgroup rendezvous --------------------
mkdir -p tools/_build
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_rendezvous.hip -o tools/_build/probe_rendezvous 2>/dev/null; timeout 60 tools/_build/probe_rendezvous) 2>&1 | tee $O/rendezvous.txt
# ---- attention: revision 1 vs revision 2 (forward unchanged in substance, dQ without the per-tile tail mask) ----------------
for v in 1 2; do
  timeout 120 python tools/attn_bench.py --no-sdpa --iters 10 --variant $v 2>/dev/null | tail -1 | tee -a $O/attn_ab.txt
done
TTT_ATTN_DQ_OCC=2 timeout 120 python tools/attn_bench.py --no-sdpa --iters 10 --variant 2 2>/dev/null | tail -1 | tee -a $O/attn_ab.txt
for d in 2 3 4; do      # dK / dV through the body: same arithmetic / accumulator-initialised row scalars with 8 / 12 waves
  timeout 120 python tools/attn_bench.py --no-sdpa --iters 10 --variant 2 --dkdv-variant $d 2>/dev/null | tail -1 | tee -a $O/attn_ab.txt
done
# ---- TTT-MLP forward scan (CS = 64): default vs packed-f32 gelu on aligned register pairs ------------------------------------
for f in "" "--gelu-pk"; do
  timeout 120 python tools/op_bench.py --fwd-only --iters 20 $f 2>/dev/null | tail -1 | sed "s/^/scan8 '$f' /" | cut -c1-400 | tee -a $O/ab.txt
done
# ---- CS = 16 kernels ---------------------------------------------------------------------------------------------------
for body in "" "--body"; do
  timeout 100 python tools/cs16_bench.py --no-generic --batch 2 $body 2>&1 | grep "^mfma" | sed "s/^/mlp16 body='$body' /" | tee -a $O/ab.txt
done
for slots in 0 4; do
  timeout 100 python tools/cs16_bench.py --linear --no-generic --lds-slots $slots 2>&1 | grep "bwd" | sed "s/^/lds_slots=$slots /" | tee -a $O/ab.txt
done
# ---- full model: one flag at a time against the default line --------------------------------------------------------------
if [ "$1" = "bench" ]; then
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
  timeout 400 $B                                   2>$O/bench_default.err  | tail -1 > $O/bench_default.json
  timeout 400 $B --attn-variant 2                  2>$O/bench_attn2.err    | tail -1 > $O/bench_attn2.json
  timeout 400 $B --overlap-wgrad                   2>$O/bench_wgrad.err    | tail -1 > $O/bench_wgrad.json
  timeout 400 $B --fsdp on                         2>$O/bench_fsdp1.err    | tail -1 > $O/bench_fsdp1.json     # FSDP2 over a one-rank mesh (the default is the replica path)
  timeout 400 $B --attn-variant 2 --attn-dkdv-variant 4 --overlap-wgrad --scan-gelu-pk 2>$O/bench_all.err | tail -1 > $O/bench_all.json
  timeout 600 $B --local-batch 2                   2>$O/bench_lb2.err      | tail -1 > $O/bench_lb2.json
  for f in default fsdp1 attn2 wgrad all lb2; do
    python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    o = r.get("other", {})
    print(f"{sys.argv[2]:8s} {d['value']:8.1f} video-tok/s  {d['ms_per_step']:7.1f} ms/step  mem {d['peak_mem_gib']:.0f} GiB  "
          f"scan bwd {r.get('avg_launch_ms', 0):.2f} ms  attn fwd {o.get('attn_fwd', {}).get('avg_ms', 0):.2f}  attn bwd {o.get('attn_bwd', {}).get('avg_ms', 0):.2f}")
except Exception as ex:
    print(sys.argv[2], "FAILED", repr(ex))
PY
  done | tee $O/bench_summary.txt
fi
