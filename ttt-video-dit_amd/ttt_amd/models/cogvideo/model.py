"""``CogVideoX``: diffusion training loss around the DiT (reference ``ttt/models/cogvideo/model.py``
:9-66): draw a discrete noise level, noise the latent, v-prediction scaling, weighted L2.

The zero-terminal-SNR schedule is the training half of the reference's
``ZeroSNRDDPMDiscretization`` / ``DiscreteSampler`` (``cogvideo/utils.py``:262-358), rebuilt here as
a small tensor table; the sampling-time classes of that file live in ``sampling.py``.
Unlike the reference's sampler this one also works without an initialised process group
(SURVEY.md hazard C8): rank/world default to 0/1.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ttt_amd.models.cogvideo.dit import DiffusionTransformer


def zero_snr_alphas_cumprod_sqrt(n: int = 1000, linear_start=0.00085, linear_end=0.0120) -> torch.Tensor:
    """sqrt(alpha_bar_t), shifted/rescaled so the last step has exactly zero SNR, index 0 = least noise."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=np.float64) ** 2
    acp = np.cumprod(1.0 - betas)
    s = torch.tensor(acp, dtype=torch.float32).sqrt()
    s0, sT = s[0].clone(), s[-1].clone()
    return (s - sT) * (s0 / (s0 - sT))


class DiscreteSampler:
    """Uniform draw of a noise index; ranks are striped over index intervals like the reference (:262-310)."""

    def __init__(self, config, effective_rank: int = 0, effective_world_size: int = 1):
        self.sigma_interval = config.sigma_interval
        self.effective_rank = effective_rank
        i = 1
        while effective_world_size % i != 0 or self.sigma_interval % (effective_world_size // i) != 0:
            i += 1
        self.group_num = effective_world_size // i
        self.group_width = effective_world_size // self.group_num
        self.group_sigma_interval = self.sigma_interval // self.group_num
        self.sigmas = None

    def __call__(self, n_samples, rand=None, generator=None, device="cuda"):
        if self.sigmas is None:
            # the reference sub-samples the fixed 1000-step schedule and THEN rescales (cogvideo/utils.py:296-298 ->
            # ZeroSNRDDPMDiscretization()(sigma_interval, flip=True)); identical to a fresh table only at 1000
            from ttt_amd.models.cogvideo.sampling import ZeroSNRDDPMDiscretization
            self.sigmas = ZeroSNRDDPMDiscretization()(self.sigma_interval, device=device, flip=True)
        g = self.effective_rank // self.group_width
        lo, hi = g * self.group_sigma_interval, (g + 1) * self.group_sigma_interval
        if rand is None:
            idx = torch.randint(lo, hi, (n_samples,), generator=generator, device=device)
        else:
            idx = torch.full((n_samples,), rand, dtype=torch.long, device=device)
        return self.sigmas[idx], idx


class CogVideoX(nn.Module):
    def __init__(self, config, effective_rank: int = 0, effective_world_size: int = 1):
        super().__init__()
        self.config = config
        self.sigma_sampler = DiscreteSampler(config, effective_rank, effective_world_size)
        self.dit = DiffusionTransformer(config)
        self.effective_rank = effective_rank
        self.noise_generator = None

    def init_ssm_weights(self):
        for layer in self.dit.layers:
            layer.seq_modeling_block.rotary.init_freqs()
            layer.seq_modeling_block.ssm.init_freqs()
            layer.seq_modeling_block.ssm.ttt.init_weights()

    def setup_generator(self, seed: int, device="cuda"):
        self.noise_generator = torch.Generator(device=device)
        self.noise_generator.manual_seed(seed)

    @staticmethod
    def get_l2_loss(model_output, target, w):
        return torch.mean((w * (model_output - target) ** 2).reshape(target.shape[0], -1), 1)

    def forward(self, vid, text, *, noise_idx=None, noise=None):
        """vid [B,T,16,H,W] latent, text [B,n_scenes,S,text_dim] -> per-sample loss [B].

        ``noise_idx`` / ``noise`` (keyword-only, not in the reference's signature) replace the two random draws; parity tests
        use them to feed a GPU run the draws of a CPU-generated reference fixture."""
        if noise_idx is None:
            a, idx = self.sigma_sampler(vid.shape[0], generator=self.noise_generator, device=vid.device)
        else:
            a, idx = self.sigma_sampler(vid.shape[0], rand=0, device=vid.device)
            idx = noise_idx.to(vid.device)
            a = self.sigma_sampler.sigmas[idx]
        a = a.view(-1, *([1] * (vid.ndim - 1)))
        if noise is None:
            noise = torch.randn(vid.shape, dtype=vid.dtype, device=vid.device, generator=self.noise_generator)
        noised = vid.float() * a + noise * (1 - a ** 2) ** 0.5
        # v-prediction scaling (VideoScaling, cogvideo/utils.py:252-258): c_skip = a, c_out = -sqrt(1-a^2), c_in = 1
        out = self.dit(noised.to(vid.dtype), text, idx) * (-((1 - a ** 2) ** 0.5)) + noised * a
        return self.get_l2_loss(out, vid, 1 / (1 - a ** 2))
