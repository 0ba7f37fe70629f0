"""Round-3 GPU parity tests (-m gpu), all through the C ABI (ctypes binding ``test_time_training``):

  1. the MODEL REGIME at the benchmarked launch shape: the TTT-MLP op inputs that occur inside a DiffusionTransformer layer at
     the CogVideoX-5B head geometry (48 heads x 64, the 192-workgroup cluster launch, two chunks of checkpoint groups) are
     captured at the extension boundary and the MFMA forward / backward are compared with the fp64 oracle on sampled heads -
     the regime in which round 1's sweep was wrong by 20 - 50 % while every random-input test passed;
  2. the scan lengths of BASELINE configs 4 and 5 against the fp64 oracle: NC = 2 630 (30 s, mini-batches of 64, G = 16,
     forward + backward) and NC = 21 948 (63 s, mini-batches of 16, one checkpoint group, forward: TTT-MLP and TTT-Linear);
  3. the assembled 3-scene DiT on the kernel path against the reference's model code run on last-row eta tiles;
  4. the hand-over failure path: a cluster that cannot complete poisons its outputs and the next call raises.
Tolerances: bf16 activations vs fp64 arithmetic on the same rounded inputs: outputs rel-L2 <= 1e-2, gradients
<= 3e-2; model level vs the reference's fp32 run: 2e-2 / 8e-2.
"""
import math

import pytest
import torch

from helpers import load_golden, rel_l2
from oracle import ttt_oracle as O
from test_kernels_gpu import DEV, ext, oracle_on, round_acts, run_mlp
from test_parity_r2_gpu import check_per_head

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------- 1. model regime, 48-head launch
def _capture_layer_calls(frames=4, text_tokens=104, seed=7):
    """One TransformerLayer-deep DiT at the 5B width (model_dim 3072 = 48 heads x 64; latent 60 x 90 -> 1 350 tokens per frame)
    in bf16 on the device; returns the positional argument lists of every ttt_forward / ttt_backward call it makes."""
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    e = ext()
    torch.manual_seed(seed)
    cfg = ModelConfig(model_dim=3072, num_heads=48, num_layers=1, ssm_layer="ttt_mlp", mini_batch_size=64, text_dim=256,
                      compressed_num_frames=frames, adapter_method="sft", scan_checkpoint_group_size=16, remat_free_layers=1)
    m = DiffusionTransformer(cfg)
    for layer in m.layers:
        layer.seq_modeling_block.ssm.ttt.init_weights()
    m = m.to(DEV).to(torch.bfloat16)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
    gen = torch.Generator().manual_seed(seed)
    video = torch.randn(1, frames, 16, 60, 90, generator=gen).to(DEV, torch.bfloat16)
    text = torch.randn(1, 1, text_tokens, 256, generator=gen).to(DEV, torch.bfloat16)
    fwd, bwd = [], []
    of, ob = e.ttt_forward, e.ttt_backward
    keep = lambda a: [t.detach().clone() if isinstance(t, torch.Tensor) else t for t in a]

    def rec_f(*a):
        r = of(*a)
        fwd.append(keep(a))            # after the call: XQW and the checkpoints are filled in
        return r

    def rec_b(*a):
        r = ob(*a)
        bwd.append(keep(a))
        return r

    e.ttt_forward, e.ttt_backward = rec_f, rec_b
    try:
        out = m(video, text, torch.tensor([417], device=DEV))
        out.backward(torch.randn(out.shape, generator=gen).to(DEV, out.dtype))
        torch.cuda.synchronize()
    finally:
        e.ttt_forward, e.ttt_backward = of, ob
    return fwd, bwd


def test_model_regime_5b_head_geometry_vs_oracle():
    e = ext()
    fwd, bwd = _capture_layer_calls()
    assert len(fwd) == 2 and len(bwd) == 2                       # forward and time-reversed pass of the one layer
    assert e.sweep_error() == 0
    heads = [0, 5, 11, 17, 23, 30, 38, 47]                        # 8 of the 48: every XCD residue class of bh % 8 but two
    f64 = lambda t: t.detach().double().cpu()
    names = ["dln_w", "dln_b", "dW1", "db1", "dW2", "db2", "dlast_eta", "dXQ", "dXK", "dXV"]
    for ci, (fa, ba) in enumerate(zip(fwd, reversed(bwd))):       # backward calls come in reverse order of the forward calls
        XQ, XK, XV, le, lnw, lnb, W1, b1, W2, b2, W1c, b1c, W2c, b2c, XQW, G = fa
        B, NH, NC, CS, F = XQ.shape
        assert (NH, CS, F, G) == (48, 64, 64, 16) and NC > 80      # two chunks of 5 checkpoint groups at 48 heads
        assert e.resolved_impl(B, NH, NC, CS, F, G, torch.bfloat16, mlp=True, backward=True) == "mfma"
        assert torch.equal(ba[0], XQ) and torch.equal(ba[6], W1c)  # the pair belongs together
        rest = ba[11:-1]
        ups, gout, outs = rest[16:20], rest[20], rest[21:]
        hs = torch.tensor(heads)
        sel = lambda t: f64(t)[:, hs] if t.shape[1] == NH else f64(t)
        ro, rc, _ = O.mlp_forward(sel(XQ), sel(XK), sel(XV), sel(le), sel(lnw), sel(lnb), sel(W1), sel(b1), sel(W2), sel(b2), G)
        rg = O.mlp_backward(sel(XQ), sel(XK), sel(XV), sel(le), sel(lnw), sel(lnb), tuple(sel(c) for c in (W1c, b1c, W2c, b2c)), G,
                            sel(gout), dst_last=tuple(sel(u) for u in ups))
        got = {n: o[:, hs] for n, o in zip(names, outs)}
        check_per_head(f"model-regime op call {ci} (NC={NC}, 48-head launch, heads {heads})", XQW[:, hs],
                       tuple(c[:, hs] for c in (W1c, b1c, W2c, b2c)), got, ro, rc, rg, 1e-2, 3e-2)


# ---------------------------------------------------------------------------------- 2. 30 s / 63 s scan lengths
def test_mfma_mlp_at_30s_length_vs_oracle():
    """BASELINE config 4 (configs/train/ttt-mlp/30s.toml): 121 latent frames + 10 x 497 text tokens = 168 320 = 2 630
    mini-batches of 64, checkpoint groups of 16; chunked like the 48-head launch (5 groups per chunk -> 33 chunks)."""
    e = ext()
    NH, NC, G = 2, 2630, 16
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 64, 64, seed=4000 + NC), torch.bfloat16)
    e.debug_groups_per_chunk(5)
    try:
        out, cks, g = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
    finally:
        e.debug_groups_per_chunk(0)
    assert e.sweep_error() == 0
    ro, rc, rg = oracle_on(d, G, "mlp")
    check_per_head(f"TTT-MLP MFMA NC={NC}", out, cks, g, ro, rc, rg, 1e-2, 3e-2)


def test_mfma_mlp_at_63s_train_length_vs_oracle():
    """The metric's SECOND context (BASELINE.json: "at 9s & 63s"; configs/train/ttt-mlp/63s.toml, ttt/models/configs.py:71-87):
    253 latent frames + 21 x 458 text tokens = 351 168 tokens = 5 487 mini-batches of 64, checkpoint groups of 16 (343 groups,
    the last one of 15 steps), forward + backward, chunked like the 48-head launch (5 groups per chunk -> 69 chunks).  The
    carried state and state gradient walk 5 487 steps: the first and the last tenth of the sequence are compared separately
    (drift of either would show at one end)."""
    e = ext()
    NH, NC, G = 2, 5487, 16
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 64, 64, seed=6000 + NC), torch.bfloat16)
    e.debug_groups_per_chunk(5)
    try:
        out, cks, g = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
    finally:
        e.debug_groups_per_chunk(0)
    assert e.sweep_error() == 0
    ro, rc, rg = oracle_on(d, G, "mlp")
    check_per_head(f"TTT-MLP MFMA NC={NC}", out, cks, g, ro, rc, rg, 1e-2, 3e-2)
    tenth = NC // 10
    for tag, sl in (("first tenth", slice(0, tenth)), ("last tenth", slice(NC - tenth, NC))):
        sub = lambda t: t[:, :, sl]
        gs = {k: sub(v) for k, v in g.items() if k in ("dXQ", "dXK", "dXV", "dlast_eta")}
        rs = {k: sub(v) for k, v in rg.items() if k in gs}
        check_per_head(f"TTT-MLP MFMA NC={NC}, {tag}", sub(out), (), gs, sub(ro), (), rs, 1e-2, 3e-2)


@pytest.mark.parametrize("kind", ["mlp", "linear"])
def test_mfma_cs16_at_63s_length_vs_oracle(kind):
    """BASELINE config 5 (configs/eval/ttt-mlp/63s.toml:9,45): 253 latent frames + 21 x 458 text tokens = 351 168 tokens =
    21 948 mini-batches of 16, no scan checkpoints (one group), forward only (sampling)."""
    e = ext()
    NH, NC = 2, 21948
    G = NC
    d = round_acts(O.make_inputs(kind, 1, NH, NC, 16, 64, seed=5000 + NC), torch.bfloat16)
    assert e.resolved_impl(1, NH, NC, 16, 64, G, torch.bfloat16, mlp=(kind == "mlp"), backward=False) == "mfma"
    XQ, XK, XV = (d[k].to(DEV, torch.bfloat16).contiguous() for k in ("XQ", "XK", "XV"))
    le = d["eta"][:, :, :, -1, :, None].to(DEV, torch.bfloat16).contiguous()
    f32 = lambda *s: torch.empty(s, device=DEV, dtype=torch.float32)
    out = torch.full_like(XQ, float("nan"))
    d64 = {k: v.double() for k, v in d.items()}
    le64 = d64["eta"][:, :, :, -1, :, None]
    if kind == "mlp":
        lw, lb = d["ln_w"].reshape(1, NH, 1, 64).to(DEV), d["ln_b"].reshape(1, NH, 1, 64).to(DEV)
        st = [d[k].unsqueeze(0).to(DEV, torch.float32).contiguous() for k in ("W1", "b1", "W2", "b2")]
        cks = (f32(1, NH, 1, 64, 256), f32(1, NH, 1, 1, 256), f32(1, NH, 1, 256, 64), f32(1, NH, 1, 1, 64))
        e.ttt_forward(XQ, XK, XV, le, lw, lb, *st, *cks, out, G)
        ro, _, _ = O.mlp_forward(d64["XQ"], d64["XK"], d64["XV"], le64, d64["ln_w"], d64["ln_b"],
                                 *[d64[k].unsqueeze(0) for k in ("W1", "b1", "W2", "b2")], G)
    else:
        lw, lb = d["ln_w"].to(DEV, torch.float32).contiguous(), d["ln_b"].to(DEV, torch.float32).contiguous()
        st = [d[k].unsqueeze(0).to(DEV, torch.float32).contiguous() for k in ("W1", "b1")]
        cks = (f32(1, NH, 1, 64, 64), f32(1, NH, 1, 1, 64))
        e.ttt_linear_forward(XQ, XK, XV, le, lw, lb, *st, *cks, out, G)
        ro, _, _ = O.linear_forward(d64["XQ"], d64["XK"], d64["XV"], le64, d64["ln_w"], d64["ln_b"],
                                    *[d64[k].unsqueeze(0) for k in ("W1", "b1")], G)
    torch.cuda.synchronize()
    # the last tenth of the sequence separately: drift of the carried state over 21 948 steps would show there first
    tail = NC - NC // 10
    print(f"{kind} CS=16 NC={NC}: whole {rel_l2(out, ro):.2e}, last tenth {rel_l2(out[:, :, tail:], ro[:, :, tail:]):.2e}")
    check_per_head(f"TTT-{kind} CS=16 MFMA forward NC={NC}", out, (), {}, ro, (), {}, 1e-2, 3e-2)
    check_per_head(f"TTT-{kind} CS=16 MFMA forward NC={NC}, last tenth", out[:, :, tail:], (), {}, ro[:, :, tail:], (), {}, 1e-2, 3e-2)


# ---------------------------------------------------------------------------------- 3. multi-scene DiT, kernel contract
def test_dit_multiscene_on_hip_path_vs_reference_lastrow():
    """The driver-benchmarked case in miniature: 3 interleaved scenes, TTT-MLP at mini-batches of 64, bf16 on the HIP path,
    against the reference's own model code run on last-row eta tiles (tests/golden/gen_golden_r2.py:
    dit_mlp64_multiscene_lastrow_case).  2e-2 on the output, 8e-2 on every gradient; the learning-rate-gate parameters (token
    sums of d(eta) at a base learning rate 10x the default - the fixture's choice, so that the inner loop matters) are bounded
    by what the REFERENCE's own bf16-autocast run of this model loses on the same parameter where that is more than 8e-2
    (0.12 on layers.1 lr_bias, dit_bf16_yardstick_r3.pt) - SURVEY 8c's rule, no factor.  Round 3 measured 0.32 there and
    carried a 3x exception; the cause was the column sum of the bf16 dZ2b tile feeding db2 (tests/test_rounding_budget_cpu.py),
    summed in fp32 by the sweep's owner waves since round 4."""
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    e = ext()
    g = load_golden("dit_mlp64_3scene_lastrow.pt")
    m = DiffusionTransformer(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
    out = m(g["video"].to(DEV, torch.bfloat16), g["text"].to(DEV, torch.bfloat16), g["timesteps"].to(DEV))
    out.backward(g["dout"].to(DEV, out.dtype))
    torch.cuda.synchronize()
    assert e.sweep_error() == 0
    errs = {"out": rel_l2(out, g["out"])}
    params = dict(m.named_parameters())
    for k, r in g["grads"].items():
        if params[k].grad is not None:
            errs[k] = rel_l2(params[k].grad, r)
    worst = max(((k, v) for k, v in errs.items() if k != "out"), key=lambda kv: kv[1])
    lr_gate = ("learnable_ttt_lr_bias", "learnable_ttt_lr_weight")
    print("3-scene bf16 HIP DiT vs reference (last-row eta):", {"out": round(errs["out"], 4), "n_grads": len(errs) - 1,
                                                                 "worst": (worst[0], round(worst[1], 4)),
                                                                 "lr_gate": {k.split("layers.")[1][:2] + k.rsplit("_", 1)[1]: round(v, 4)
                                                                             for k, v in errs.items() if k.endswith(lr_gate)}})
    assert errs["out"] < 2e-2, errs
    yard = load_golden("dit_bf16_yardstick_r3.pt")["dit_mlp64_3scene_lastrow.pt"]
    tol = lambda k: max(8e-2, yard.get(k, 0.0)) if k.endswith(lr_gate) else 8e-2
    bad = {k: (v, tol(k)) for k, v in errs.items() if k != "out" and not v < tol(k)}
    assert not bad, bad


# ---------------------------------------------------------------------------------- 3a. selective re-materialisation
@pytest.mark.parametrize("name", ["dit_mlp64_1scene.pt", "dit_mlp64_3scene_lastrow.pt"])
def test_remat_keep_is_bit_identical(name):
    """A re-materialised layer that KEEPS its attention / scan kernel outputs (ttt_amd/infra/remat_cache.py, bench.py's
    default) must give the bits of the reference's behaviour (the whole layer recomputed): same kernels, run once instead
    of twice."""
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
    m.remat_free_layers = 0
    res = {}
    for keep in ((), ("attn", "scan"), ("attn",), ("attn", "scan", "fc2")):
        m.remat_keep = keep
        m.zero_grad(set_to_none=True)
        out = m(g["video"].to(DEV, torch.bfloat16), g["text"].to(DEV, torch.bfloat16), g["timesteps"].to(DEV))
        out.backward(g["dout"].to(DEV, out.dtype))
        torch.cuda.synchronize()
        res[keep] = (out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    o0, g0 = res[()]
    for keep in (("attn", "scan"), ("attn",), ("attn", "scan", "fc2")):
        o1, g1 = res[keep]
        assert torch.equal(o0, o1)
        assert set(g0) == set(g1)
        bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
        assert not bad, (keep, bad[:5])


# ---------------------------------------------------------------------------------- 3b. run-to-run determinism
@pytest.mark.parametrize("overlap", [0, 1, 2])
def test_backward_is_run_to_run_deterministic_at_the_benchmarked_head_count(overlap):
    """48 heads (192 cluster workgroups, recompute and tail beside the sweep on the 64 free CUs), two chunks: the same call
    repeated must give the same bits.  (The first cut of revision 4 - fragments held in registers across the workgroup barriers,
    ~500 spilled dwords - passed every oracle test and was NOT deterministic: a few heads per call differed from the step at which
    a stale fragment was staged, profiles/r3c_determinism_first_cut_FAILED.txt.)"""
    e = ext()
    d = round_acts(O.make_inputs("mlp", 1, 48, 96, 64, 64, seed=996), torch.bfloat16)
    e.debug_option("overlap_tail", overlap)
    try:
        ref = None
        for rep in range(4):
            junk = torch.randn(2048, 2048, device=DEV) @ torch.randn(2048, 2048, device=DEV)      # shifts the timing a little
            out, cks, g = run_mlp(e, d, 16, torch.bfloat16, impl="mfma")
            torch.cuda.synchronize()
            if ref is None:
                ref = {k: v.clone() for k, v in g.items()}
                continue
            bad = [k for k, v in g.items() if not torch.equal(v, ref[k])]
            assert not bad, (rep, bad)
    finally:
        e.debug_option("overlap_tail", 2)
    assert e.sweep_error() == 0


# ---------------------------------------------------------------------------------- 3c. CUs stolen under the cluster launch
@pytest.mark.parametrize("stolen", [96, 160])
def test_backward_survives_stolen_cus_with_the_same_bits(stolen):
    """What a collective's kernels do to a cluster launch at N > 1 (reduce-scatter of the previous layer beside this layer's
    backward): another stream holds `stolen` CUs - a kernel whose workgroups keep 150 KiB of LDS each, so no sweep workgroup
    (157 KiB) fits beside one - for 30 ms, launched ahead of a 48-head backward (192 sweep workgroups, 160 / 96 CUs free).  The
    clusters whose fourth workgroup cannot be placed wait in their bounded polls (seconds) until the CUs come back: no
    hand-over gives up, and the gradients are bit-identical to the undisturbed call."""
    e = ext()
    d = round_acts(O.make_inputs("mlp", 1, 48, 96, 64, 64, seed=997), torch.bfloat16)
    out0, _, ref = run_mlp(e, d, 16, torch.bfloat16, impl="mfma")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        t0[0].record()
        e.debug_occupy_cus(stolen, 150 * 1024, 30000, stream=side)
        out, _, g = run_mlp(e, d, 16, torch.bfloat16, impl="mfma")
        t0[1].record()
        torch.cuda.synchronize()
        assert e.sweep_error() == 0
        bad = [k for k, v in g.items() if not torch.equal(v, ref[k])]
        assert not bad and torch.equal(out, out0), (rep, bad)
    print(f"stolen={stolen}: disturbed call {t0[0].elapsed_time(t0[1]):.1f} ms")


# ---------------------------------------------------------------------------------- 4. hand-over failure is loud
def test_handover_timeout_poisons_outputs_and_raises():
    """A cluster workgroup whose partners never arrive (forced: the debug option makes workgroup 3 of every cluster leave before
    its first hand-over) must not return plausible numbers: the partners' bounded polls give up, the kernel fills its outputs
    with NaN, and the NEXT extension call raises (the error word is read at entry: no synchronisation on the fast path)."""
    e = ext()
    d = round_acts(O.make_inputs("mlp", 1, 2, 3, 64, 64, seed=91), torch.bfloat16)
    e.debug_option("sweep_fault", 1)
    try:
        out, cks, g = run_mlp(e, d, 2, torch.bfloat16, impl="mfma")
    finally:
        e.debug_option("sweep_fault", 0)
    torch.cuda.synchronize()
    assert e.sweep_error() != 0
    assert not bool(torch.isfinite(g["dW1"]).all()) and not bool(torch.isfinite(g["dXV"].float()).all())
    with pytest.raises(RuntimeError, match="hand-over"):
        run_mlp(e, d, 2, torch.bfloat16, impl="mfma")
    # the error is sticky until acknowledged; after that the same call works again
    e.sweep_error_clear()
    out2, _, g2 = run_mlp(e, d, 2, torch.bfloat16, impl="mfma")
    assert e.sweep_error() == 0
    ro, rc, rg = oracle_on(d, 2, "mlp")
    assert rel_l2(g2["dW1"], rg["dW1"]) < 3e-2 and rel_l2(out2, ro) < 1e-2
