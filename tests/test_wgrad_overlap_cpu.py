"""Weight-gradient deferral (ttt_amd/infra/wgrad_overlap.py) on the CPU: the deferred work runs inline there, so these
tests pin the bookkeeping - every projection gradient is produced exactly once, weights used by both scan directions
accumulate, re-materialised layers and layers whose inputs carry no gradient are handled, FSDP2 (gloo, world size 2)
reduces the published gradients - against the plain autograd path of the same model.  The stream choreography itself
is a device matter (tests/test_kernels_gpu.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import cpu_ext
from ttt_amd.infra import wgrad_overlap as wgrad
from ttt_amd.models.cogvideo.dit import DiffusionTransformer
from ttt_amd.models.configs import ModelConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(adapter="sft", remat_free=0, ssm="ttt_linear", seed=0):
    torch.manual_seed(seed)
    cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=2, mini_batch_size=16, latent_height=8, latent_width=8,
                      compressed_num_frames=3, ssm_layer=ssm, text_dim=32, time_embed_dim=64, attn_length=2,
                      prefix_temporal_length=1, adapter_method=adapter, scan_checkpoint_group_size=2, remat_free_layers=remat_free)
    m = DiffusionTransformer(cfg)
    if ssm == "ttt_mlp":        # the TTT-MLP kernel boundary is bf16-only (mlp_tk.py:89); fp32 here -> the dual form
        for layer in m.layers:
            layer.seq_modeling_block.ssm.ttt.use_kernel = False
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.02)
    return m


def _inputs(seed=100):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 3, 16, 8, 8, generator=g), torch.randn(1, 1, 16, 32, generator=g), torch.tensor([300])


def _grads(m, steps=1):
    for _ in range(steps):
        m(*_inputs()).square().mean().backward()
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.fixture
def ext():
    cpu_ext.install()
    yield
    wgrad.enable(False)
    cpu_ext.uninstall()


@pytest.mark.parametrize("adapter,remat_free,ssm", [("sft", 0, "ttt_linear"), ("sft", 2, "ttt_mlp"), ("qkvo", 1, "ttt_linear")])
def test_deferred_gradients_equal_autograd(ext, adapter, remat_free, ssm):
    ref = _grads(_build(adapter, remat_free, ssm))
    wgrad.enable(True)
    before = wgrad.stats()
    m = _build(adapter, remat_free, ssm)
    got = _grads(m)
    after = wgrad.stats()
    assert set(got) == set(ref)
    for k, v in ref.items():
        assert torch.allclose(got[k], v, rtol=2e-5, atol=1e-8), k
    if adapter == "sft":
        per_layer = 4 + 2 * 4 + 2          # attention q,k,v,o; wq,wk,wv,wo twice; MLP layer1, layer2
        assert after["submitted"] - before["submitted"] == 2 * per_layer
        assert after["joined"] - before["joined"] == 2                 # one join per layer, none left for the end-of-backward net
    else:                    # frozen patch embedding: layer 0 has no input gradient -> plain autograd there, deferral in layer 1 only
        assert after["submitted"] - before["submitted"] == 4 + 2 * 4


def test_gradient_accumulation_over_two_backwards(ext):
    ref = _grads(_build(), steps=2)
    wgrad.enable(True)
    got = _grads(_build(), steps=2)
    for k, v in ref.items():
        assert torch.allclose(got[k], v, rtol=2e-5, atol=1e-8), k


def test_disabled_is_the_plain_path(ext):
    wgrad.enable(False)
    before = wgrad.stats()
    _grads(_build())
    assert wgrad.stats() == before


# ---- FSDP2, world size 2, gloo ------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import cpu_ext as ce
    from ttt_amd.infra import wgrad_overlap as wg
    from ttt_amd.infra.parallelisms import apply_fsdp, end_distributed, get_dp_mesh, init_distributed
    ce.install()
    init_distributed("gloo")
    out = {}
    for mode in ("plain", "deferred"):
        wg.enable(mode == "deferred")
        m = _build(remat_free=1)
        apply_fsdp(m, get_dp_mesh(), param_dtype=torch.float32, reshard_after_forward=(mode == "plain"))
        v, t, ts = _inputs(100 + rank)
        m(v, t, ts).square().mean().backward()
        out[mode] = {n: p.grad.full_tensor().clone() for n, p in m.named_parameters() if p.grad is not None}
    assert wg.stats()["submitted"] == 2 * 14 and wg.stats()["joined"] == 2
    wg.enable(False)
    if rank == 0:
        torch.save(out, os.path.join(out_dir, "grads.pt"))
    end_distributed()


@pytest.mark.timeout(600)
def test_fsdp2_world2_reduces_the_deferred_gradients(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    out = torch.load(os.path.join(tmp_path, "grads.pt"))
    assert set(out["plain"]) == set(out["deferred"]) and len(out["plain"]) > 20
    for k, v in out["plain"].items():
        assert torch.allclose(out["deferred"][k], v, rtol=2e-5, atol=1e-8), k


def test_linear3_equals_three_linears():
    """The fused q/k/v node: outputs bit-equal to three F.linear calls, input gradient equal to the sum of the three
    (accumulated in the GEMM instead of by two additions), weight / bias gradients equal; unused outputs are handled."""
    torch.manual_seed(0)
    mods = [torch.nn.Linear(48, 32) for _ in range(3)]
    x = torch.randn(2, 7, 48, requires_grad=True)
    ys = wgrad.linear3(*mods, x)
    ref = [m(x) for m in mods]
    for a, b in zip(ys, ref):
        assert torch.equal(a, b)
    gs = [torch.randn_like(r) for r in ref]
    (ys[0] * gs[0]).sum().add((ys[2] * gs[2]).sum()).backward()          # the k output stays unused
    got = [x.grad.clone()] + [m.weight.grad.clone() if m.weight.grad is not None else None for m in mods]
    x.grad = None
    for m in mods:
        m.zero_grad()
    (ref[0] * gs[0]).sum().add((ref[2] * gs[2]).sum()).backward()
    want = [x.grad.clone()] + [m.weight.grad.clone() if m.weight.grad is not None else None for m in mods]
    assert torch.allclose(got[0], want[0], rtol=1e-5, atol=1e-6)
    assert got[2] is None and want[2] is None
    for g, w in ((got[1], want[1]), (got[3], want[3])):
        assert torch.allclose(g, w, rtol=1e-5, atol=1e-6)
