"""Round-6 tests on the device (through the C ABI of libttt_hip.so): the DEFAULT layer forward - the pipeline over parts of the
sequence that bench.py times (ttt_amd/models/ssm/pipeline.py) - directly against reference-executed numbers and the fp64 oracle,
and the schedules / options of the TTT-MLP backward added in round 6."""
import pytest
import torch

from helpers import load_golden, rel_l2
from oracle import ttt_oracle as O
from test_kernels_gpu import DEV, ext, oracle_on, round_acts, run_mlp
from test_parity_r2_gpu import check_per_head

pytestmark = pytest.mark.gpu

LR_GATE = ("learnable_ttt_lr_bias", "learnable_ttt_lr_weight")


def _long_dit(g):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    m = DiffusionTransformer(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(torch.bfloat16)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
    return m


def _run_and_compare(m, g, e):
    m.zero_grad(set_to_none=True)
    out = m(g["video"].to(DEV, torch.bfloat16), g["text"].to(DEV, torch.bfloat16), g["timesteps"].to(DEV))
    out.backward(g["dout"].to(DEV, out.dtype))
    torch.cuda.synchronize()
    assert e.sweep_error() == 0
    errs = {"out": rel_l2(out, g["out"])}
    params = dict(m.named_parameters())
    for k, r in g["grads"].items():
        if params[k].grad is not None:
            errs[k] = rel_l2(params[k].grad, r)
    return errs, out.detach().clone(), {k: p.grad.clone() for k, p in params.items() if p.grad is not None}


@pytest.mark.parametrize("remat", ["free", "remat_keep_scan"])
def test_default_pipelined_forward_vs_reference_long_fixture(remat, monkeypatch):
    """the path bench.py times - TTTBase.pipeline_parts at its DEFAULT, i.e. the layer forward as a
    pipeline over four parts (row-block GEMMs, injected Linear3 / post-norm nodes, the scan in parts on a side stream) - against
    numbers the REFERENCE produced (tests/golden/gen_golden_r6.py: its own 2-layer DiT, TTT-MLP, mini-batches of 64, 3 interleaved
    scenes, 15 mini-batches = 8 checkpoint groups, last-row eta; what ttt_layer.py:314-334 computes inside dit.py:224-266).  The
    test asserts that the pipelined pre-pass really ran (2 layers x 2 directions, with ttt_forward_chunk launches and NO one-call
    forward) and holds the module-level tolerances of the other DiT fixtures: 2e-2 on the output, 8e-2 on every gradient (the
    learning-rate-gate parameters bounded by the reference's own bf16-autocast error where that is larger).
    `remat_keep_scan`: every layer re-materialised with its scan / attention / MLP outputs kept - the recomputation
    forms q / k / v with ONE whole-sequence GEMM where the forward used per-part GEMMs; the gradients must still hold the tolerance."""
    from ttt_amd.models.ssm import pipeline
    from ttt_amd.models.ssm.ttt_layer import TTTBase
    e = ext()
    g = load_golden("dit_mlp64_3scene_long_lastrow.pt")
    assert g["checkpoint_groups"] >= 8
    m = _long_dit(g)
    ttts = [mod for mod in m.modules() if isinstance(mod, TTTBase)]
    assert ttts and all(t.pipeline_parts >= 2 for t in ttts), "the library default must be the pipelined forward"
    if remat == "remat_keep_scan":
        m.remat_free_layers = 0
        m.remat_keep = ("attn", "scan", "fc2")
    else:
        m.remat_free_layers = len(m.layers)                   # (the config default re-materialises every layer: the pre-pass would run twice)
    calls = {"prepass": 0, "chunk": 0, "one_call": 0}
    orig_prepass, orig_chunk, orig_fwd = pipeline.prepass, e.ttt_forward_chunk, e.ttt_forward

    def prepass(*a, **k):
        calls["prepass"] += 1
        return orig_prepass(*a, **k)

    def chunk(*a):
        calls["chunk"] += 1
        return orig_chunk(*a)

    def one_call(*a):
        calls["one_call"] += 1
        return orig_fwd(*a)

    monkeypatch.setattr(pipeline, "prepass", prepass)
    monkeypatch.setattr(e, "ttt_forward_chunk", chunk)
    monkeypatch.setattr(e, "ttt_forward", one_call)
    errs, _, _ = _run_and_compare(m, g, e)
    assert calls["prepass"] == 4 and calls["chunk"] >= 4 * 4 and calls["one_call"] == 0, calls
    worst = max(((k, v) for k, v in errs.items() if k != "out"), key=lambda kv: kv[1])
    print(f"default (pipelined, {remat}) bf16 HIP DiT vs reference, {g['mini_batches']} mini-batches:",
          {"out": round(errs["out"], 4), "n_grads": len(errs) - 1, "worst": (worst[0], round(worst[1], 4)), "calls": calls})
    assert errs["out"] < 2e-2, errs
    yard = load_golden("dit_bf16_yardstick_r6.pt")["dit_mlp64_3scene_long_lastrow.pt"]
    tol = lambda k: max(8e-2, yard.get(k, 0.0)) if k.endswith(LR_GATE) else 8e-2
    bad = {k: (v, tol(k)) for k, v in errs.items() if k != "out" and not v < tol(k)}
    assert not bad, bad


def test_pipelined_and_one_piece_forward_agree_on_the_reference_fixture():
    """The same fixture as one piece (pipeline_parts = 0): both paths hold the reference tolerances, and their difference is bf16
    rounding (1e-2 / 2e-2) - the advisor's case "piped forward, non-piped recomputation" is the remat variant of the test above."""
    e = ext()
    g = load_golden("dit_mlp64_3scene_long_lastrow.pt")
    m = _long_dit(g)
    res = {}
    for parts in (None, 0):
        if parts is not None:
            for mod in m.modules():
                if hasattr(mod, "pipeline_parts"):
                    mod.pipeline_parts, mod.pipeline_parts_auto = parts, False
        res[parts] = _run_and_compare(m, g, e)
    (e1, o1, g1), (e0, o0, g0) = res[None], res[0]
    assert e1["out"] < 2e-2 and e0["out"] < 2e-2
    assert rel_l2(o1, o0.double()) < 1e-2
    bad = {k: rel_l2(g1[k], g0[k].double()) for k in g0 if float(g0[k].float().norm()) > 0 and not rel_l2(g1[k], g0[k].double()) < 2e-2}
    assert not bad, bad


def test_model_regime_5b_head_geometry_pipelined_vs_oracle():
    """The 48-head model regime of tests/test_parity_r3_gpu.py at a length the pipelined forward takes (7 frames x 1 350 + 86 text
    tokens = 149 mini-batches = 10 checkpoint groups of 16): the forward scans arrive as ttt_forward_chunk launches (captured), the
    backward gets the tensors those parts filled; outputs, checkpoints and all ten gradients of 8 heads against the fp64 oracle."""
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    e = ext()
    frames, text_tokens, seed = 7, 86, 11
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
    chunks, one_call, bwd = [], [], []
    oc, of, ob = e.ttt_forward_chunk, e.ttt_forward, e.ttt_backward
    keep = lambda a: [t.detach().clone() if isinstance(t, torch.Tensor) else t for t in a]

    outs_of = {}                                                  # XQ storage -> the tensors the parts fill (checkpoints, XQW)

    def rec_c(*a):
        chunks.append((int(a[-2]), int(a[-1])))
        outs_of[a[0].data_ptr()] = a[10:15]
        return oc(*a)

    def rec_f(*a):
        one_call.append(1)
        return of(*a)

    def rec_b(*a):
        r = ob(*a)
        bwd.append(keep(a))
        return r

    e.ttt_forward_chunk, e.ttt_forward, e.ttt_backward = rec_c, rec_f, rec_b
    try:
        out = m(video, text, torch.tensor([417], device=DEV))
        out.backward(torch.randn(out.shape, generator=gen).to(DEV, out.dtype))
        torch.cuda.synchronize()
    finally:
        e.ttt_forward_chunk, e.ttt_forward, e.ttt_backward = oc, of, ob
    assert e.sweep_error() == 0
    assert not one_call and len(bwd) == 2 and len(chunks) >= 8, (len(one_call), len(bwd), chunks)
    NCs = (frames * 1350 + text_tokens) // 64
    for d in range(2):                                            # each direction's parts tile [0, NC) exactly, in order
        part = chunks[d * (len(chunks) // 2):(d + 1) * (len(chunks) // 2)]
        assert part[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(part, part[1:])) and part[-1][0] + part[-1][1] == NCs, part
    heads = [0, 5, 11, 17, 23, 30, 38, 47]
    hs = torch.tensor(heads)
    f64 = lambda t: t.detach().double().cpu()
    names = ["dln_w", "dln_b", "dW1", "db1", "dW2", "db2", "dlast_eta", "dXQ", "dXK", "dXV"]
    for ci, ba in enumerate(bwd):
        XQ, XK, XV, le, lnw, lnb, W1c, b1c, W2c, b2c = ba[:10]     # (the ABI's XQW slot of the backward is a placeholder: never read)
        G = ba[-1]
        hit = [v for v in outs_of.values() if v[0].shape == W1c.shape and torch.equal(v[0], W1c)]
        assert len(hit) == 1, "the backward's checkpoints must be the ones the forward parts wrote"
        XQW = hit[0][4]
        B, NH, NC, CS, F = XQ.shape
        assert (NH, CS, F, G, NC) == (48, 64, 64, 16, NCs)
        rest = ba[11:-1]
        ups, gout, outs = rest[16:20], rest[20], rest[21:]
        sel = lambda t: f64(t)[:, hs] if t.shape[1] == NH else f64(t)
        st = [sel(c)[:, :, 0] for c in (W1c, b1c, W2c, b2c)]      # the initial state = the first checkpoint
        ro, rc, _ = O.mlp_forward(sel(XQ), sel(XK), sel(XV), sel(le), sel(lnw), sel(lnb), *st, G)
        rg = O.mlp_backward(sel(XQ), sel(XK), sel(XV), sel(le), sel(lnw), sel(lnb), tuple(sel(c) for c in (W1c, b1c, W2c, b2c)), G,
                            sel(gout), dst_last=tuple(sel(u) for u in ups))
        got = {n: o[:, hs] for n, o in zip(names, outs)}
        check_per_head(f"pipelined model-regime op call {ci} (NC={NC}, parts {chunks[:len(chunks) // 2]})", XQW[:, hs],
                       tuple(c[:, hs] for c in (W1c, b1c, W2c, b2c)), got, ro, rc, rg, 1e-2, 3e-2)


@pytest.mark.parametrize("shape", [(1, 8, 40, 16, 1), (1, 48, 96, 16, 2), (2, 3, 23, 3, 2)])
def test_sweep_with_barrier_inside_the_reverse_step_has_the_same_bits(shape):
    """Round 6: the sweep's barrier Bc falls between the two token tiles of the derivers' reverse step (debug option `deriver_split`,
    default 1) - a re-timing only: every output must equal the round-5 placement (0) bit for bit, repeatedly, and match the oracle."""
    e = ext()
    B, NH, NC, G, gpc = shape
    d = round_acts(O.make_inputs("mlp", B, NH, NC, 64, 64, seed=1600 + NC), torch.bfloat16)
    res = {}
    for mode in (0, 1, 1, 0):
        e.debug_option("deriver_split", mode)
        e.debug_groups_per_chunk(gpc)
        try:
            res.setdefault(mode, []).append(run_mlp(e, d, G, torch.bfloat16, impl="mfma"))
        finally:
            e.debug_option("deriver_split", 1)
            e.debug_groups_per_chunk(0)
    torch.cuda.synchronize()
    assert e.sweep_error() == 0
    o0, _, g0 = res[0][0]
    for o1, _, g1 in res[1] + res[0][1:]:
        assert torch.equal(o0, o1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), (k, rel_l2(g1[k], g0[k]))
    if B * NH <= 16:
        ro, rc, rg = oracle_on(d, G, "mlp")
        check_per_head(f"split sweep {shape}", res[1][0][0], (), res[1][0][2], ro, (), rg, 1e-2, 3e-2)


def test_sweep_beside_a_saturating_masked_stream_gemm_has_the_same_bits():
    """a side stream confined to a quarter of the compute units (``ttt_hip_stream_create_masked`` =
    hipExtStreamCreateWithCUMask, wrapped as a torch ExternalStream by ``test_time_training.masked_stream``) runs a saturating bf16 GEMM
    loop while the TTT-MLP backward of the benchmarked head count (48 heads: 192 cluster workgroups + recompute / tail beside them) runs:
    every gradient must equal the run without the neighbour bit for bit, no hand-over may time out, and the placement probe must show
    the masked stream's workgroups on exactly the masked CUs (8 of every XCD).  The backward runs on a NON-default stream here: a
    CU-masked stream is a blocking stream and would otherwise serialise with torch's default stream instead of running beside it
    (tools/cu_mask_probe.py, profiles/r6k_*: what the neighbour costs - the 64 CUs the sweep leaves are not idle, the backward's own
    recompute / tail kernels live there)."""
    e = ext()
    d = round_acts(O.make_inputs("mlp", 1, 48, 96, 64, 64, seed=1800), torch.bfloat16)
    work = torch.cuda.Stream()
    work.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(work):
        o0, _, g0 = run_mlp(e, d, 16, torch.bfloat16, impl="mfma")
        torch.cuda.synchronize()
        # mask bit i is a CU of XCD i % 8 (profiles/r6k_cu_mask_probe.json): the low byte of every word = 8 CUs of every XCD = 64 of 256.
        # (A mask that leaves an XCD without a CU is silently ignored by the runtime: the stream then runs everywhere.)
        side = e.masked_stream([0xFF] * 8)
        cus = set(e.placement_probe(256, stream=side))
        assert len(cus) == 64 and all(sum(1 for c in cus if c[0] == x) == 8 for x in range(8)), sorted(cus)[:16]
        a = torch.randn(4096, 4096, device=DEV).bfloat16()
        b = torch.randn(4096, 4096, device=DEV).bfloat16()
        for rep in range(3):
            side.wait_stream(work)
            with torch.cuda.stream(side):
                for _ in range(60):
                    torch.mm(a, b)
            o1, _, g1 = run_mlp(e, d, 16, torch.bfloat16, impl="mfma")
            torch.cuda.synchronize()
            assert e.sweep_error() == 0
            assert torch.equal(o0, o1)
            bad = [k for k in g0 if not torch.equal(g0[k], g1[k])]
            assert not bad, (rep, bad)
    torch.cuda.current_stream().wait_stream(work)
