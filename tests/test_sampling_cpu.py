"""Sampling-time classes vs golden vectors produced by the reference's own sampler
(tests/golden/gen_sampling_golden.py; reference ``ttt/models/cogvideo/utils.py``:312-711) on CPU, fp32.

Tolerances: the schedule tables and the denoiser are elementwise fp32 arithmetic in the same order -> 1e-6 absolute;
full sampling runs chain 8..50 such steps (with exp/log of the log-SNR) -> 1e-4 relative L2.
"""
import types

import pytest
import torch

from helpers import ToyNet, load_golden, rel_l2
from ttt_amd.models.cogvideo import sampling as S


@pytest.fixture(scope="module")
def gold():
    return load_golden("sampling.pt")


def test_zero_snr_tables(gold):
    t = S.ZeroSNRDDPMDiscretization()(1000, flip=True)
    assert torch.allclose(t, gold["table_1000_flip"], atol=1e-7, rtol=0)
    assert t[-1] == 0 and t[0] > 0.99                                       # exactly zero terminal SNR
    s, idx = S.ZeroSNRDDPMDiscretization()(50, return_idx=True)
    assert torch.equal(torch.tensor(list(idx)), gold["idx_50"])
    assert torch.allclose(s, gold["table_50"], atol=1e-7, rtol=0)
    s, idx = S.ZeroSNRDDPMDiscretization(shift_scale=3.0)(17, return_idx=True, do_append_zero=True)
    assert torch.equal(torch.tensor(list(idx)), gold["idx_17"])
    assert torch.allclose(s, gold["table_17_shift3_zero"], atol=1e-7, rtol=0)
    with pytest.raises(ValueError):
        S.ZeroSNRDDPMDiscretization()(1001)


def test_training_table_is_the_same_schedule(gold):
    from ttt_amd.models.cogvideo.model import zero_snr_alphas_cumprod_sqrt
    assert torch.allclose(zero_snr_alphas_cumprod_sqrt(1000), gold["table_1000_flip"], atol=1e-7, rtol=0)


@pytest.mark.parametrize("batch_samples", [True, False])
def test_denoiser(gold, batch_samples):
    i = gold["den_in"]
    den = S.DiscreteDenoiser(ToyNet(), num_idx=1000, quantize_c_noise=False, dtype=torch.float32, batch_samples=batch_samples)
    out = den(i["x"], i["sigma"], {"crossattn": i["text"]}, idx=i["idx"])
    assert torch.allclose(out, gold["den_out"], atol=1e-6, rtol=0)
    denq = S.DiscreteDenoiser(ToyNet(), num_idx=1000, quantize_c_noise=True, dtype=torch.float32, batch_samples=batch_samples)
    out = denq(i["x"], i["sigma"], {"crossattn": i["text"]}, idx=torch.tensor([0.5, 0.9]))
    assert torch.allclose(out, gold["den_out_quantized"], atol=1e-6, rtol=0)


def test_dynamic_cfg_weights():
    g = S.DynamicCFG(scale=6, exp=5, num_steps=50)
    assert g.scale_schedule(None, 0) == 1.0
    assert abs(g.scale_schedule(None, 50) - 7.0) < 1e-12
    u, c = torch.zeros(1, 3), torch.ones(1, 3)
    assert torch.allclose(g(torch.cat([u, c]), None, torch.tensor(50)), torch.full((1, 3), 7.0))


def _sampler(steps, shift, batch_samples):
    return S.VPSDEDPMPP2MSampler(
        denoiser=S.DiscreteDenoiser(ToyNet(), num_idx=1000, quantize_c_noise=False, dtype=torch.float32, batch_samples=batch_samples),
        discretization_config={"shift_scale": shift}, guider_config={"scale": 6, "exp": 5, "num_steps": steps},
        use_wandb=False, verbose=False, device="cpu", num_steps=steps)


@pytest.mark.parametrize("batch_samples", [True, False])
@pytest.mark.parametrize("name", ["s8", "s50", "s20_shift"])
def test_sampler_matches_reference(gold, name, batch_samples):
    c = gold["cases"][name]
    torch.manual_seed(c["seed"])
    noise = torch.randn(1, 3, 4, 6, 5)
    text, neg = torch.randn(1, 2, 7, 16), torch.randn(1, 2, 7, 16)
    with torch.no_grad():
        res = _sampler(c["steps"], c["shift"], batch_samples)(noise, {"crossattn": text}, {"crossattn": neg})
    assert torch.isfinite(res).all()
    assert rel_l2(res, c["result"]) < 1e-4


def test_guidance_pair_is_one_network_call():
    """The MI355X-first change: cond + uncond go through the network as one batch of two (96 concurrent scans)."""
    calls = []

    class Spy(ToyNet):
        def forward(self, x, text, t):
            calls.append(x.shape[0])
            return super().forward(x, text, t)

    for batch_samples, expect in ((True, [2] * 4), (False, [1] * 8)):
        calls.clear()
        smp = S.VPSDEDPMPP2MSampler(
            denoiser=S.DiscreteDenoiser(Spy(), num_idx=1000, quantize_c_noise=False, dtype=torch.float32, batch_samples=batch_samples),
            discretization_config={}, guider_config={"scale": 6, "exp": 5, "num_steps": 4}, device="cpu", num_steps=4)
        smp(torch.randn(1, 2, 2, 4, 4), {"crossattn": torch.randn(1, 1, 3, 8)}, {"crossattn": torch.randn(1, 1, 3, 8)})
        assert calls == expect


def test_denoiser_sampler_glue():
    ns = types.SimpleNamespace
    cfg = ns(denoiser=ns(num_idx=1000, quantize_c_noise=False), discretization=ns(shift_scale=1.0),
             guider=ns(scale=6, exp=5), eval=ns(num_denoising_steps=5))
    a = S.DenoiserSampler(ToyNet(), cfg, torch.float32, effective_rank=0, seed=3, device="cpu")
    b = S.DenoiserSampler(ToyNet(), cfg, torch.float32, effective_rank=0, seed=3, device="cpu")
    text, neg = torch.randn(1, 1, 3, 8), torch.randn(1, 1, 3, 8)
    torch.manual_seed(0)
    ra = a.sample(text, neg, (2, 2, 4, 4), batch_size=1)
    torch.manual_seed(0)
    rb = b.sample(text, neg, (2, 2, 4, 4), batch_size=1)
    assert ra.shape == (1, 2, 2, 4, 4) and torch.equal(ra, rb) and torch.isfinite(ra).all()


def test_sampling_through_the_dit_batched_equals_sequential():
    """The real network: a 3-scene DiT with TTT-MLP layers (weights of the reference golden), 3 denoising steps under
    no_grad.  One batch of two (cond + uncond) must give the sequential result - samples are independent."""
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    g = load_golden("dit_mlp_3scene.pt")
    net = DiffusionTransformer(ModelConfig(**g["cfg"]))
    net.load_state_dict(g["state_dict"], strict=True)
    for layer in net.layers:
        layer.seq_modeling_block.ssm.ttt.use_kernel = False           # CPU: the dual-form path
    net.eval()
    res = []
    for batch_samples in (True, False):
        smp = S.VPSDEDPMPP2MSampler(
            denoiser=S.DiscreteDenoiser(net, num_idx=1000, quantize_c_noise=False, dtype=torch.float32, batch_samples=batch_samples),
            discretization_config={}, guider_config={"scale": 6, "exp": 5, "num_steps": 3}, device="cpu", num_steps=3)
        torch.manual_seed(1)
        noise = torch.randn(1, *g["video"].shape[1:])
        neg = torch.randn(g["text"].shape)
        with torch.no_grad():
            res.append(smp(noise, {"crossattn": g["text"]}, {"crossattn": neg}))
    assert torch.isfinite(res[0]).all()
    assert rel_l2(res[0], res[1]) < 1e-5
