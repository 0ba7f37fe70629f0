"""Round-2 GPU parity tests (-m gpu), all through the C ABI (ctypes binding ``test_time_training``):

  1. the BENCHMARKED scan lengths against the fp64 oracle, head by head, no head excluded: TTT-MLP MFMA forward + backward
     at NC = 282 (3 s) and NC = 804 (9 s) with the chunking of the 48-head launch (5 checkpoint groups per chunk, prefetch
     helpers on); the CS = 16 scans (TTT-MLP forward, TTT-Linear forward + backward) at NC = 1128 (3 s) / 3216 (9 s);
  2. the fused HIP module path in multi-scene mode against the reference's module run on last-row eta tiles (the kernel
     contract, tests/golden/gen_golden_r2.py), forward and time-reversed;
  3. the assembled DiffusionTransformer through the bf16 HIP path against the reference's fp32 DiT goldens;
  4. ``GeluLinear`` and ``CogVideoX.forward`` on the device.
Tolerances: bf16 activations vs fp64 arithmetic on the same rounded inputs: outputs rel-L2 <= 1e-2, gradients
<= 3e-2; module / model level vs the reference's fp32 run (adds bf16 projections and parameters): 2e-2 / 8e-2.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, rel_l2
from oracle import ttt_oracle as O
from test_kernels_gpu import DEV, ext, oracle_on, round_acts, run_lin, run_mlp

pytestmark = pytest.mark.gpu


def per_head(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    nh = a.shape[1]
    a, b = a.transpose(0, 1).reshape(nh, -1), b.transpose(0, 1).reshape(nh, -1)
    return (a - b).norm(dim=1) / b.norm(dim=1).clamp_min(1e-30)


def check_per_head(what, out, cks, g, ro, rc, rg, tol_out, tol_g):
    rows = {"XQW": (per_head(out, ro), tol_out)}
    for i, (c, r) in enumerate(zip(cks, rc)):
        rows[f"ck{i}"] = (per_head(c, r), tol_out)
    for k, r in rg.items():
        if k in g:
            rows[k] = (per_head(g[k], r), tol_g)
    print(f"{what}: per-head rel-L2 vs fp64 oracle (median / max over heads)")
    bad = {}
    for k, (e, tol) in rows.items():
        print(f"   {k:10s} {float(e.median()):.2e} / {float(e.max()):.2e}   (tol {tol:.0e})")
        if not bool((e < tol).all()):
            bad[k] = [round(float(x), 5) for x in e]
    assert not bad, f"{what}: heads out of tolerance: {bad}"


# ---------------------------------------------------------------------------------- 1. benchmarked scan lengths vs the oracle
@pytest.mark.parametrize("NC", [282, 804])
def test_mfma_mlp_at_benchmarked_length_vs_oracle(NC):
    """CS = 64, G = 16, 8 heads (helpers on: heads % 8 == 0), chunked like the 48-head launch (5 groups per chunk)."""
    e = ext()
    NH, G = 8, 16
    assert e.resolved_impl(1, NH, NC, 64, 64, G, torch.bfloat16, mlp=True, backward=True) == "mfma"
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 64, 64, seed=1000 + NC), torch.bfloat16)
    e.debug_groups_per_chunk(5)
    try:
        out, cks, g = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
    finally:
        e.debug_groups_per_chunk(0)
    ro, rc, rg = oracle_on(d, G, "mlp")
    check_per_head(f"TTT-MLP MFMA NC={NC}", out, cks, g, ro, rc, rg, 1e-2, 3e-2)


@pytest.mark.parametrize("NC", [1128, 3216])
def test_mfma_mlp_cs16_at_benchmarked_length_vs_oracle(NC):
    """Evaluation geometry: mini-batches of 16, no scan checkpoints (one group), forward only."""
    e = ext()
    NH = 4
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 16, 64, seed=2000 + NC), torch.bfloat16)
    G = NC
    assert e.resolved_impl(1, NH, NC, 16, 64, G, torch.bfloat16, mlp=True, backward=False) == "mfma"
    dev = DEV
    XQ, XK, XV = (d[k].to(dev, torch.bfloat16).contiguous() for k in ("XQ", "XK", "XV"))
    le = d["eta"][:, :, :, -1, :, None].to(dev, torch.bfloat16).contiguous()
    lw, lb = d["ln_w"].reshape(1, NH, 1, 64).to(dev), d["ln_b"].reshape(1, NH, 1, 64).to(dev)
    st = [d[k].unsqueeze(0).to(dev, torch.float32).contiguous() for k in ("W1", "b1", "W2", "b2")]
    f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    out, cks = torch.empty_like(XQ), (f32(1, NH, 1, 64, 256), f32(1, NH, 1, 1, 256), f32(1, NH, 1, 256, 64), f32(1, NH, 1, 1, 64))
    e.ttt_forward(XQ, XK, XV, le, lw, lb, *st, *cks, out, G)
    torch.cuda.synchronize()
    d64 = {k: v.double() for k, v in d.items()}
    ro, rc, _ = O.mlp_forward(d64["XQ"], d64["XK"], d64["XV"], d64["eta"][:, :, :, -1, :, None], d64["ln_w"], d64["ln_b"],
                              *[d64[k].unsqueeze(0) for k in ("W1", "b1", "W2", "b2")], G)
    check_per_head(f"TTT-MLP CS=16 MFMA forward NC={NC}", out, (), {}, ro, (), {}, 1e-2, 3e-2)


@pytest.mark.parametrize("NC", [1128, 3216])
def test_mfma_linear_cs16_at_benchmarked_length_vs_oracle(NC):
    """TTT-Linear training geometry (mini_batch_size 16, G = 4): one-wave MFMA scan and reverse sweep."""
    e = ext()
    NH, G = 4, 4
    assert e.resolved_impl(1, NH, NC, 16, 64, G, torch.bfloat16, mlp=False, backward=True) == "mfma"
    d = round_acts(O.make_inputs("linear", 1, NH, NC, 16, 64, seed=3000 + NC), torch.bfloat16)
    out, cks, g = run_lin(e, d, G, torch.bfloat16, impl="auto")
    ro, rc, rg = oracle_on(d, G, "linear")
    check_per_head(f"TTT-Linear CS=16 MFMA NC={NC}", out, cks, g, ro, rc, rg, 1e-2, 3e-2)


# ---------------------------------------------------------------------------------- 2. multi-scene kernel contract
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("name", ["mod_lin_multi_lastrow.pt", "mod_mlp_multi_lastrow.pt", "mod_mlp_multi64_lastrow.pt"])
def test_fused_module_multiscene_vs_reference_lastrow(name, reverse):
    """The fused HIP module path (pre kernel with token maps, scan kernels on last-row eta, post kernel) against the
    reference module executed on last-row eta tiles - the reference-pinned target every >= 9 s configuration needs."""
    from ttt_amd.models.cogvideo.utils import SequenceMetadata
    from ttt_amd.models.configs import ModelConfig
    from ttt_amd.models.ssm.ttt_layer import TTTWrapper
    ext()
    g = load_golden(name)
    ref = g["rev" if reverse else "fwd"]
    m = TTTWrapper(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(torch.bfloat16)          # bf16 parameters as under FSDP mixed precision (parallelisms.py:155-175)
    assert m.ttt.use_kernel and m.ttt.use_fused
    meta = SequenceMetadata(t_emb=torch.zeros(1, 512, device=DEV), **g["meta"])
    meta.init_multiscene_offsets()
    x = g["x"].to(DEV, torch.bfloat16).requires_grad_(True)
    y = m(x, meta, reverse)
    y.backward(g["dy"].to(DEV, y.dtype))
    errs = {"y": rel_l2(y, ref["y"]), "dx": rel_l2(x.grad, ref["dx"])}
    params = dict(m.named_parameters())
    for k, r in ref["grads"].items():
        errs[k] = rel_l2(params[k].grad, r)
    print(name, "reverse" if reverse else "forward", "fused HIP path vs reference (last-row eta):", {k: round(v, 4) for k, v in errs.items()})
    assert errs["y"] < 2e-2, errs
    bad = {k: v for k, v in errs.items() if k != "y" and not v < 8e-2}
    assert not bad, (bad, errs)
    assert rel_l2(g["dual_form_full_tile_y"], ref["y"]) > 1e-3       # and it is NOT the dual form on the full tiles


@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("name", ["mod_lin_multi_lastrow.pt", "mod_mlp_multi64_lastrow.pt"])
def test_head_sharded_layer_equals_unsharded_hip_path(name, reverse):
    """Head sharding on ONE rank (the per-rank work of tensor / sequence parallelism, `forward_heads`): running the layer for
    heads [0, NH/2) and [NH/2, NH) on the HIP path and concatenating must give the unsharded HIP layer - forward, input
    gradient and every parameter gradient (a rank produces its own heads' rows; the sum over the shards is what
    `tp_sync_gradients` forms).  Reference: head-local DTensor placements, ttt_layer.py:114-131, mlp_tk.py:297-343."""
    from ttt_amd.models.cogvideo.utils import SequenceMetadata
    from ttt_amd.models.configs import ModelConfig
    from ttt_amd.models.ssm.ttt_layer import TTTWrapper
    ext()
    g = load_golden(name)
    m = TTTWrapper(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    meta = SequenceMetadata(t_emb=torch.zeros(1, 512, device=DEV), **g["meta"])
    meta.init_multiscene_offsets()
    dy = g["dy"].to(DEV, torch.bfloat16)
    NH = m.ttt.num_heads
    assert NH % 2 == 0

    def run(sharded):
        m.zero_grad(set_to_none=True)
        x = g["x"].to(DEV, torch.bfloat16).requires_grad_(True)
        if sharded:
            parts = [m.forward_heads(x, meta, reverse, h0, h0 + NH // 2) for h0 in (0, NH // 2)]
            y = m.ttt.wo(m.ttt.post_norm(torch.cat(parts, dim=-1)))
        else:
            y = m(x, meta, reverse)
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach(), x.grad, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    errs = {"y": rel_l2(y1, y0), "dx": rel_l2(dx1, dx0)}
    assert set(g0) == set(g1), set(g0) ^ set(g1)
    for k in g0:
        errs[k] = rel_l2(g1[k], g0[k])
    print(name, "reverse" if reverse else "forward", "two head shards vs unsharded HIP layer:", {k: round(v, 5) for k, v in errs.items()})
    # same kernels per head; the projections / the post-norm are GEMMs and reductions of other shapes: bf16 rounding only
    bad = {k: v for k, v in errs.items() if not v < 2e-2}
    assert not bad, (bad, errs)


# ---------------------------------------------------------------------------------- 3. assembled DiT on the HIP path
@pytest.mark.parametrize("name", ["dit_mlp64_1scene.pt", "dit_lin_1scene.pt", "dit_mlp_3scene.pt"])
def test_dit_on_hip_path_vs_reference_golden(name):
    """DiffusionTransformer (patch embedding, AdaLN glue, local attention kernels, bidirectional TTT on the scan kernels,
    GeluLinear MLP, final layer) in bf16 on the GPU against the reference's fp32 run of its own model code.  The 3-scene
    fixture is a dual-form (full eta tile) run: compared only where the kernel contract coincides with it - it does not
    (hazard C2), so there the test asserts the documented difference stays small at this geometry instead."""
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    ext()
    g = load_golden(name)
    m = DiffusionTransformer(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
    out = m(g["video"].to(DEV, torch.bfloat16), g["text"].to(DEV, torch.bfloat16), g["timesteps"].to(DEV))
    out.backward(g["dout"].to(DEV, out.dtype))
    torch.cuda.synchronize()
    errs = {"out": rel_l2(out, g["out"])}
    params = dict(m.named_parameters())
    for k, r in g["grads"].items():
        if params[k].grad is not None:
            errs[k] = rel_l2(params[k].grad, r)
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(name, "bf16 HIP DiT vs reference fp32:", {"out": round(errs["out"], 4), "n_grads": len(errs) - 1, "worst": (worst[0], round(worst[1], 4))})
    multi = g["scenes"] > 1
    assert errs["out"] < (6e-2 if multi else 2e-2), errs
    # Gradient tolerance 8e-2 (bf16 end to end vs the reference's fp32 run); the reference's own bf16-autocast run of these
    # models is off by up to 2.3 - 3.0e-2 (tests/golden/dit_bf16_yardstick.pt).  The two learning-rate-gate parameters (token
    # sums of d(eta)): round 2 / 3 measured 0.12 - 0.15 with the MFMA kernels against 0.02 - 0.03 with the fp32-arithmetic
    # generic kernels and bounded them by 0.25; round 4 found the cause (db2's column sums taken from a bf16 tile,
    # tests/test_rounding_budget_cpu.py; the TTT-Linear backward had the same path for db1) and the single-scene fixtures are held
    # to the common 8e-2 (measured: TTT-MLP <= 3.0e-2, TTT-Linear <= 7.6e-3, profiles/r4a_parity_tests.log).  The dual-form
    # 3-scene fixture is a different function of the eta tile (hazard C2) and keeps the 0.25 bound (measured 2.0e-2).
    tol = 0.25 if multi else 8e-2
    lr_gate = ("learnable_ttt_lr_bias", "learnable_ttt_lr_weight")
    lr_tol = 0.25 if multi else 8e-2
    print(name, "learning-rate-gate gradients:", {k.split("layers.")[1][:2] + k.rsplit("_", 1)[1]: round(v, 4) for k, v in errs.items() if k.endswith(lr_gate)})
    bad = {k: v for k, v in errs.items() if k != "out" and not v < (lr_tol if k.endswith(lr_gate) else tol)}
    assert not bad, bad


# ---------------------------------------------------------------------------------- 4. GeluLinear, CogVideoX.forward
@pytest.mark.parametrize("train_w", [True, False])
def test_gelu_linear_on_gpu_vs_fp32_statements(train_w):
    from ttt_amd.models.cogvideo.dit import GeluLinear
    gen = torch.Generator().manual_seed(3)
    z0 = torch.randn(2, 333, 1024, generator=gen).bfloat16().to(DEV)
    w0 = (0.03 * torch.randn(256, 1024, generator=gen)).bfloat16().to(DEV)
    b0 = (0.1 * torch.randn(256, generator=gen)).bfloat16().to(DEV)
    dy = torch.randn(2, 333, 256, generator=gen).bfloat16().to(DEV)
    z, w, b = z0.clone().requires_grad_(True), w0.clone().requires_grad_(train_w), b0.clone().requires_grad_(train_w)
    y = GeluLinear.apply(z, w, b)
    y.backward(dy)
    zr, wr, br = z0.float().requires_grad_(True), w0.float().requires_grad_(True), b0.float().requires_grad_(True)
    yr = F.linear(F.gelu(zr, approximate="tanh"), wr, br)
    yr.backward(dy.float())
    errs = {"y": rel_l2(y, yr), "dz": rel_l2(z.grad, zr.grad)}
    if train_w:
        errs.update(dw=rel_l2(w.grad, wr.grad), db=rel_l2(b.grad, br.grad))
    else:
        assert w.grad is None and b.grad is None
    assert all(v < 1e-2 for v in errs.values()), errs


def test_cogvideox_forward_on_gpu_vs_reference():
    """Loss of the reference's CogVideoX.forward (fixture generated under a 1-rank gloo group) with the fixture's own random
    draws fed to the bf16 HIP model."""
    from ttt_amd.models.cogvideo.model import CogVideoX
    from ttt_amd.models.configs import ModelConfig
    ext()
    g = load_golden("cogvideox_loss.pt")
    m = CogVideoX(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    # weights come from the fixture; only the RoPE tables are rebuilt after the dtype cast
    for layer in m.dit.layers:
        layer.seq_modeling_block.rotary.init_freqs()
        layer.seq_modeling_block.ssm.init_freqs()
    loss = m(g["vid"].to(DEV, torch.bfloat16), g["text"].to(DEV, torch.bfloat16), noise_idx=g["idx"], noise=g["noise"].to(DEV, torch.bfloat16))
    loss.sum().backward()
    torch.cuda.synchronize()
    assert rel_l2(loss, g["loss"]) < 2e-2, (loss, g["loss"])
    params = dict(m.named_parameters())
    errs = {k: rel_l2(params[k].grad, r) for k, r in g["grads"].items() if params[k].grad is not None}
    bad = {k: v for k, v in errs.items() if not v < 0.1}
    assert not bad, bad


# ---------------------------------------------------------------------------------- 5. backward sweep: cluster hand-over
@pytest.mark.parametrize("shape", [(1, 2, 11, 2, 0), (1, 8, 40, 16, 5), (2, 3, 7, 3, 0), (1, 4, 33, 16, 1), (2, 40, 9, 4, 0)])
def test_bwd_cluster_sweep_handover_forms_and_oracle(shape):
    """The TTT-MLP backward sweep runs on four workgroups per (b,h) that exchange partial d(gZ2) tiles inside the launch
    (csrc/ttt_mfma_bwd3.hip, Guideline-16 hand-over).  (1) Records published write-through only (placement-independent form)
    and plain records on a proven common XCD must give IDENTICAL bits - the protocol moves the same values; (2) head by head
    against the fp64 oracle; (3) no hand-over poll may have timed out; repeated launches: stale flags / records of the previous
    call must not matter.  Shapes: 2 heads (the four workgroups land on four XCDs), 8 heads (one XCD per cluster, chunked),
    batch 2, 4 heads (two XCDs per cluster), 80 (b,h) (two sweep launches per chunk: more than 64 clusters do not fit the chip)."""
    e = ext()
    B, NH, NC, G, gpc = shape
    d = round_acts(O.make_inputs("mlp", B, NH, NC, 64, 64, seed=500 + NC), torch.bfloat16)
    res = {}
    for mode in (0, 1, 1):
        e.debug_option("fast_records", mode)
        e.debug_groups_per_chunk(gpc)
        try:
            res[mode] = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
        finally:
            e.debug_option("fast_records", 1)
            e.debug_groups_per_chunk(0)
    assert e.sweep_error() == 0
    print("cluster workgroup launches that published plain (same-XCD) records so far:", e.sweep_fast_count())
    (o0, _, g0), (o1, _, g1) = res[0], res[1]
    assert torch.equal(o0, o1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), (k, rel_l2(g1[k], g0[k]))
    if B * NH <= 16:
        ro, rc, rg = oracle_on(d, G, "mlp")
        check_per_head(f"cluster backward {shape}", o1, (), g1, ro, (), rg, 1e-2, 3e-2)


@pytest.mark.parametrize("shape", [(1, 8, 40, 16, 1), (2, 3, 23, 3, 2), (1, 48, 96, 16, 0), (1, 48, 130, 16, 2), (2, 40, 9, 2, 1)])
def test_bwd_tail_under_next_sweep_same_bits(shape):
    """The backward walks the sequence in chunks (recompute A, sweep B, tail C); the tail of chunk c runs on a side stream
    underneath the sweep of chunk c-1, in two alternating slot buffers (csrc/ttt_mfma_bwd2.hip:mlp_backward).  Same
    kernels on the same data: every output must equal the one-stream schedule bit for bit - also when the call is repeated
    (buffers and events are reused) and when other work sits on the stream before and after the call.  2 - 5 chunks each;
    (1, 48, ...) is the benchmarked head count (automatic chunking, and 2 groups per chunk = 5 chunks); 80 (b,h) leave no CU
    free beside the sweep: there the schedule falls back to one stream by itself."""
    e = ext()
    B, NH, NC, G, gpc = shape
    d = round_acts(O.make_inputs("mlp", B, NH, NC, 64, 64, seed=900 + NC), torch.bfloat16)
    res = {}
    for mode in (0, 1, 1, 2, 2):          # (2 = round 6: the NEXT chunk's recompute beside the sweep too, three record buffers)
        e.debug_option("overlap_tail", mode)
        e.debug_groups_per_chunk(gpc)
        try:
            junk = torch.randn(2048, 2048, device=DEV) @ torch.randn(2048, 2048, device=DEV)      # work queued in front
            res[mode] = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
            junk = junk @ junk                                                                     # and behind
        finally:
            e.debug_option("overlap_tail", 2)
            e.debug_groups_per_chunk(0)
    torch.cuda.synchronize()
    assert e.sweep_error() == 0
    o0, _, g0 = res[0]
    for mode in (1, 2):
        o1, _, g1 = res[mode]
        assert torch.equal(o0, o1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (mode, k, rel_l2(g1[k], g0[k]))
