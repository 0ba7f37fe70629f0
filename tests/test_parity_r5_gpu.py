"""Round-5 tests on the device (through the C ABI of libttt_hip.so): the TTT-MLP forward in parts and the pipelined layer forward."""
import math

import pytest
import torch

from helpers import rel_l2
from test_kernels_gpu import DEV, ext

pytestmark = pytest.mark.gpu


def _scan_inputs(B, NH, NC, seed):
    g = torch.Generator().manual_seed(seed)
    F, H = 64, 256
    mk = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    l2 = lambda t: torch.nn.functional.normalize(t, dim=-1)
    d = dict(XQ=l2(mk(B, NH, NC, 64, F)).bfloat16(), XK=l2(mk(B, NH, NC, 64, F)).bfloat16(), XV=mk(B, NH, NC, 64, F, scale=0.5).bfloat16(),
             eta=(torch.rand(B, NH, NC, 64, 1, generator=g) * 0.02 + 0.005).bfloat16(),
             ln_w=(1 + 0.1 * mk(1, NH, 1, F)).float(), ln_b=(0.1 * mk(1, NH, 1, F)).float(),
             W1=mk(B, NH, F, H, scale=0.02).float(), b1=mk(B, NH, 1, H, scale=0.02).float(),
             W2=mk(B, NH, H, F, scale=0.02).float(), b2=mk(B, NH, 1, F, scale=0.02).float())
    return {k: v.to(DEV).contiguous() for k, v in d.items()}


@pytest.mark.parametrize("B,NH,NC,G,cuts", [(1, 8, 40, 4, (0, 12, 28, 40)), (2, 3, 37, 8, (0, 16, 37)), (1, 48, 96, 16, (0, 32, 64, 96))])
def test_forward_in_parts_has_the_bits_of_the_one_call_forward(B, NH, NC, G, cuts):
    """``ttt_hip_mlp_forward_chunk`` (round 5): the scan over parts of the sequence, each started from the fp32 state the previous
    one left, writes the SAME output tiles and state checkpoints as the one-call forward - bit for bit (the state is handed on
    exactly as the kernel holds it) - incl. a ragged last checkpoint group (37 = 4 groups of 8 + 5) and the 48-head launch."""
    e = ext()
    d = _scan_inputs(B, NH, NC, 5 + NC)
    K = math.ceil(NC / G)
    F, H = 64, 256

    def bufs():
        cks = (torch.empty(B, NH, K, F, H, device=DEV), torch.empty(B, NH, K, 1, H, device=DEV),
               torch.empty(B, NH, K, H, F, device=DEV), torch.empty(B, NH, K, 1, F, device=DEV))
        for t in cks:
            t.fill_(float("nan"))
        return cks, torch.full((B, NH, NC, 64, F), float("nan"), device=DEV, dtype=torch.bfloat16)

    e.set_impl("mfma")
    try:
        cks0, out0 = bufs()
        e.ttt_forward(d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"], d["W1"], d["b1"], d["W2"], d["b2"], *cks0, out0, G)
        cks1, out1 = bufs()
        state = [d[k].clone() for k in ("W1", "b1", "W2", "b2")]
        for s0, s1 in zip(cuts[:-1], cuts[1:]):
            e.ttt_forward_chunk(d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"], *state, *cks1, out1, G, s0, s1 - s0)
        torch.cuda.synchronize()
    finally:
        e.set_impl("auto")
    assert not torch.isnan(out1.float()).any() and not any(torch.isnan(t).any() for t in cks1)
    assert torch.equal(out0, out1)
    for a, b in zip(cks0, cks1):
        assert torch.equal(a, b)
    # a part that does not start at a checkpoint-group boundary is refused
    with pytest.raises(RuntimeError, match="checkpoint-group"):
        e.ttt_forward_chunk(d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"], *state, *cks1, out1, G, 1, G)


def _dit(adapter="qkvo"):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    cfg = ModelConfig(model_dim=512, num_heads=8, num_layers=2, mini_batch_size=64, latent_height=8, latent_width=16, compressed_num_frames=25,
                      ssm_layer="ttt_mlp", adapter_method=adapter, time_embed_dim=512, text_dim=64, remat_free_layers=1,
                      scan_checkpoint_group_size=2, ttt_base_lr=1.0)
    torch.manual_seed(0)
    m = DiffusionTransformer(cfg)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.02)
    m = m.to(DEV).to(torch.bfloat16)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
    return m


@pytest.mark.parametrize("keep", [(), ("attn", "scan", "fc2")])
def test_pipelined_layer_forward_against_the_one_piece_forward(keep):
    """The TTT-MLP layer's forward as a pipeline over parts of the sequence (ttt_amd/models/ssm/pipeline.py: the scan of one part on
    a side stream beside the projections of the next and the post-norm / output projection of the previous) against the one-piece
    forward on a 2-layer DiT: 2 interleaved scenes (a part covers several token runs), both scan directions, one layer
    re-materialised (with and without kept kernel outputs), one not.  The scan parts reproduce the one-call scan bit for bit (test
    above); the projections are hipBLASLt GEMMs over row blocks of the same matrices, whose bits may depend on the solution picked for
    a shape - so: outputs and every gradient to bf16 rounding level (1e-2 / 2e-2 rel-L2), and the log says whether they were equal."""
    ext()
    m = _dit()
    g = torch.Generator(device=DEV).manual_seed(5)
    vid = torch.randn(1, 25, 16, 16, 32, device=DEV, generator=g).bfloat16()
    text = torch.randn(1, 2, 32, 64, device=DEV, generator=g).bfloat16()      # 25 frames x 128 tokens + 2 x 32 = 51 mini-batches of 64
    ts = torch.tensor([300], device=DEV)
    dout = None
    res = {}
    m.remat_keep = keep
    for parts in (0, 3):
        for mod in m.modules():
            if hasattr(mod, "pipeline_parts"):
                mod.pipeline_parts = parts
        m.zero_grad(set_to_none=True)
        out = m(vid, text, ts)
        if dout is None:
            dout = torch.randn(out.shape, device=DEV, generator=g).to(out.dtype)
        out.backward(dout)
        torch.cuda.synchronize()
        res[parts] = (out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    (o0, g0), (o1, g1) = res[0], res[3]
    assert set(g0) == set(g1) and not torch.isnan(o1.float()).any()
    same = torch.equal(o0, o1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    worst = max([rel_l2(o1, o0.double())] + [rel_l2(g1[k], g0[k].double()) for k in g0 if float(g0[k].float().norm()) > 0])
    print("pipelined forward vs one piece: bit-identical" if same else f"pipelined forward vs one piece: worst rel-L2 {worst:.2e}")
    assert rel_l2(o1, o0.double()) < 1e-2
    bad = {k: rel_l2(g1[k], g0[k].double()) for k in g0 if float(g0[k].float().norm()) > 0 and not rel_l2(g1[k], g0[k].double()) < 2e-2}
    assert not bad, bad
