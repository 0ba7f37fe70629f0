"""Sampling-time half of the diffusion wrapper (reference ``ttt/models/cogvideo/utils.py``
:252-258 ``VideoScaling``, :312-359 ``ZeroSNRDDPMDiscretization``, :441-509 ``DiscreteDenoiser``, :512-543
``NoDynamicThresholding`` / ``DynamicCFG``, :547-711 ``VPSDEDPMPP2MSampler`` and ``cogvideo/sampler.py``:197-246
``DenoiserSampler``).  Same class names, constructor arguments and call signatures, so ``sample.py`` of the reference
can import them from here.

What is different on purpose:

* **The classifier-free-guidance pair runs as ONE batch of two.**  The reference's denoiser walks over the batch and calls
  the network once per sample (``utils.py``:478-492), which leaves the TTT scan at 48 concurrent workgroups on a 256-CU
  part.  Here the conditional and unconditional halves go through the network together (96 independent scans, twice the
  rows in every GEMM) unless ``batch_samples=False`` is passed, which restores the one-by-one order.  Every sample is
  still computed independently of the others, so the two modes agree to GEMM-selection rounding (tested on CPU in fp32).
* ``device`` defaults to the tensor's / current default device instead of the literal ``"cuda"``, so the classes also
  work in CPU tests.
* The per-step multipliers of the DPM-Solver++(2M) SDE update are computed once per step from scalars of the schedule
  (they do not depend on the sample), not re-derived per batch element.

The random-number protocol is kept: the reference draws ``randn_like(x)`` for the first-order update and, when the
second-order correction applies, draws AGAIN for the update it actually returns (``utils.py``:671,677).  Skipping the
unused first draw would change every later sample for a given seed, so it is drawn and discarded here too.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
from torch import nn


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"Cannot reduce dimensions: input has {x.ndim} dims but target_dims is {target_dims}")
    return x.reshape(*x.shape, *([1] * extra))


class VideoScaling:
    """v-prediction: c_skip = a, c_out = -sqrt(1 - a^2), c_in = 1, c_noise = timestep index (reference :252-258)."""

    def __call__(self, sigma, idx):
        return sigma, -((1 - sigma ** 2) ** 0.5), torch.ones_like(sigma), idx.clone()


class ZeroSNRDDPMDiscretization:
    """sqrt(alpha_bar) table with exactly zero terminal SNR (reference :312-359).

    ``get_sigmas(n)`` returns the table in order of DECREASING noise index (entry 0 = step 999 = value 0); ``__call__``
    optionally appends a zero and/or flips it back (``flip=True``: entry t = noise index t, as the training sampler and
    the denoiser use it)."""

    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000, shift_scale=1.0):
        self.num_timesteps = num_timesteps
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2).numpy()
        acp = np.cumprod(1.0 - betas, axis=0)
        self.alphas_cumprod = acp / (shift_scale + (1 - shift_scale) * acp)

    def get_sigmas(self, n, device=None, return_idx=False):
        if n > self.num_timesteps:
            raise ValueError(f"{n} steps requested from a {self.num_timesteps}-step schedule")
        timesteps = None
        acp = self.alphas_cumprod
        if n < self.num_timesteps:
            # n roughly equally spaced indices, always ending at the last (zero-SNR) one
            timesteps = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
            acp = acp[timesteps]
        s = torch.tensor(acp, dtype=torch.float32, device=device).sqrt()
        first, last = s[0].clone(), s[-1].clone()
        s -= last
        s *= first / (first - last)
        s = torch.flip(s, (0,))
        return (s, timesteps) if return_idx else s

    def __call__(self, n, do_append_zero=False, device=None, flip=False, return_idx=False):
        got = self.get_sigmas(n, device=device, return_idx=return_idx)
        sigmas, idx = got if return_idx else (got, None)
        if do_append_zero:
            sigmas = torch.cat([sigmas, sigmas.new_zeros([1])])
        if flip:
            sigmas = torch.flip(sigmas, (0,))
        return (sigmas, idx) if return_idx else sigmas


class DiscreteDenoiser(nn.Module):
    """x0 estimate  D(x; a) = c_out * network(c_in * x, text, c_noise) + c_skip * x  with the noise level snapped to the
    discrete table (reference :441-509)."""

    def __init__(self, network: nn.Module, num_idx: int, dtype, do_append_zero=False, quantize_c_noise=True, flip=True,
                 batch_samples: bool = True):
        super().__init__()
        self.scaling = VideoScaling()
        self.sigmas = ZeroSNRDDPMDiscretization()(num_idx, do_append_zero=do_append_zero, device="cpu", flip=flip)
        self.quantize_c_noise = quantize_c_noise
        self.network = network
        self.dtype = dtype
        self.batch_samples = batch_samples

    def sigma_to_idx(self, sigma):
        table = self.sigmas.to(sigma.device)
        return (sigma.reshape(1, -1) - table[:, None]).abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self.sigmas.to(idx.device)[idx]

    def possibly_quantize_sigma(self, sigma):
        return self.idx_to_sigma(self.sigma_to_idx(sigma))

    def possibly_quantize_c_noise(self, c_noise):
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise

    def forward(self, input: torch.Tensor, sigma: torch.Tensor, cond: Dict, **additional_model_inputs) -> torch.Tensor:
        sigma = self.possibly_quantize_sigma(sigma)
        shape = sigma.shape
        c_skip, c_out, c_in, c_noise = self.scaling(append_dims(sigma, input.ndim), **additional_model_inputs)
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(shape))
        scaled = (input * c_in).to(dtype=self.dtype)
        text = cond["crossattn"]
        if self.batch_samples or input.shape[0] == 1:
            net = self.network(scaled, text, c_noise)
        else:
            net = torch.cat([self.network(scaled[i:i + 1], text[i:i + 1], c_noise[i:i + 1]) for i in range(input.shape[0])], dim=0)
        return net * c_out + input * c_skip


class NoDynamicThresholding:
    def __call__(self, uncond, cond, scale):
        if isinstance(scale, torch.Tensor):
            scale = append_dims(scale, cond.ndim)
        return uncond + scale * (cond - uncond)


class DynamicCFG:
    """Guidance weight 1 + scale * (1 - cos(pi * (step / num_steps)^exp)) / 2: weak at the first steps, ``1 + scale`` at
    the last (reference :519-543)."""

    def __init__(self, scale, exp, num_steps):
        self.scale, self.exp, self.num_steps = scale, exp, num_steps
        self.dyn_thresh = NoDynamicThresholding()

    def scale_schedule(self, sigma, step_index):
        return 1 + self.scale * (1 - math.cos(math.pi * (step_index / self.num_steps) ** self.exp)) / 2

    def prepare_inputs(self, x, s, c, uc):
        """Stack (unconditional, conditional) along the batch."""
        both = {}
        for k in c:
            if k in ("vector", "crossattn", "concat"):
                both[k] = torch.cat((uc[k], c[k]), 0)
            else:
                assert c[k] == uc[k]
                both[k] = c[k]
        return torch.cat([x, x]), torch.cat([s, s]), both

    def __call__(self, x, sigma, step_index, scale=None):
        x_u, x_c = x.chunk(2)
        step = step_index.item() if hasattr(step_index, "item") else step_index
        return self.dyn_thresh(x_u, x_c, self.scale_schedule(sigma, step))


def _log_snr_half(a):
    """lambda = log(alpha / sigma) of the VP process at sqrt(alpha_bar) = a  (= -inf at a = 0, +inf at a = 1)."""
    a2 = a ** 2
    return ((a2 / (1 - a2)) ** 0.5).log()


class VPSDEDPMPP2MSampler:
    """DPM-Solver++(2M), SDE variant, on the variance-preserving schedule, with classifier-free guidance
    (reference :547-711).  ``__call__(x, cond, uc)`` runs ``num_steps`` network evaluations (each on the guidance pair)
    from pure noise ``x`` to the clean latent."""

    def __init__(self, denoiser: nn.Module, discretization_config: Dict, num_steps: int, guider_config: Dict,
                 use_wandb: bool = False, verbose: bool = False, device: Optional[str] = None):
        self.denoiser = denoiser
        self.num_steps = num_steps
        self.discretization = ZeroSNRDDPMDiscretization(**discretization_config)
        self.guider = DynamicCFG(**guider_config)
        self.verbose = verbose
        self.device = device
        self.use_wandb = use_wandb          # accepted for signature compatibility; logging is out of scope

    # ---- network evaluation --------------------------------------------------------------------------------------------
    def denoise(self, x, alpha_cumprod_sqrt, cond, uc, timestep=None, idx=None, scale=None, scale_emb=None):
        ts = x.new_ones([x.shape[0]]) * timestep
        if not isinstance(scale, torch.Tensor) and scale == 1:           # guidance disabled: conditional branch only
            extra = {"idx": ts}
            if scale_emb is not None:
                extra["scale_emb"] = scale_emb
            return self.denoiser(x, alpha_cumprod_sqrt, cond, **extra).to(torch.float32)
        xs, sig, both = self.guider.prepare_inputs(x, alpha_cumprod_sqrt, cond, uc)
        pair = self.denoiser(xs, sig, both, idx=torch.cat([ts, ts])).to(dtype=torch.float32)
        return self.guider(pair, (1 - alpha_cumprod_sqrt ** 2) ** 0.5, step_index=self.num_steps - timestep, scale=scale)

    # ---- solver coefficients ---------------------------------------------------------------------------------------------
    def get_variables(self, alpha_cumprod_sqrt, next_alpha_cumprod_sqrt, previous_alpha_cumprod_sqrt=None):
        lamb, lamb_next = _log_snr_half(alpha_cumprod_sqrt), _log_snr_half(next_alpha_cumprod_sqrt)
        h = lamb_next - lamb
        r = None
        if previous_alpha_cumprod_sqrt is not None:
            r = (lamb - _log_snr_half(previous_alpha_cumprod_sqrt)) / h
        return h, r, lamb, lamb_next

    def get_mult(self, h, r, alpha_cumprod_sqrt, next_alpha_cumprod_sqrt, previous_alpha_cumprod_sqrt):
        keep = ((1 - next_alpha_cumprod_sqrt ** 2) / (1 - alpha_cumprod_sqrt ** 2)) ** 0.5 * (-h).exp()
        data = (-2 * h).expm1() * next_alpha_cumprod_sqrt
        if previous_alpha_cumprod_sqrt is None:
            return keep, data
        return keep, data, 1 + 1 / (2 * r), 1 / (2 * r)

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        n = self.num_steps if num_steps is None else num_steps
        table, timesteps = self.discretization(n, device=self.device if self.device is not None else x.device,
                                               return_idx=True, do_append_zero=False)
        table = torch.cat([table, table.new_ones([1])])                  # ends at a = 1: the clean sample
        steps = torch.tensor(list(timesteps))
        timesteps = torch.cat([steps.new_zeros([1]) - 1, steps])
        return x, x.new_ones([x.shape[0]]), table, len(table), cond, (uc or cond), timesteps

    def sampler_step(self, old_denoised, previous_alpha_cumprod_sqrt, alpha_cumprod_sqrt, next_alpha_cumprod_sqrt, x, cond,
                     uc=None, idx=None, timestep=None):
        denoised = self.denoise(x, alpha_cumprod_sqrt, cond, uc, timestep, idx).to(torch.float32)
        if idx == 1:                                                     # last step lands on the x0 estimate itself
            return denoised, denoised
        h, r, _, _ = self.get_variables(alpha_cumprod_sqrt, next_alpha_cumprod_sqrt, previous_alpha_cumprod_sqrt)
        mult = [append_dims(m, x.ndim) for m in
                self.get_mult(h, r, alpha_cumprod_sqrt, next_alpha_cumprod_sqrt, previous_alpha_cumprod_sqrt)]
        mult_noise = append_dims((1 - next_alpha_cumprod_sqrt ** 2) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5, x.ndim)
        first_order = mult[0] * x - mult[1] * denoised + mult_noise * torch.randn_like(x)
        if old_denoised is None or torch.sum(next_alpha_cumprod_sqrt) < 1e-14:
            return first_order, denoised
        extrapolated = mult[2] * denoised - mult[3] * old_denoised
        # second draw: see the module docstring (random-number protocol of the reference)
        return mult[0] * x - mult[1] * extrapolated + mult_noise * torch.randn_like(x), denoised

    def __call__(self, x, cond, uc=None, num_steps=None, scale=None, **kwargs):
        x, s_in, table, num_sigmas, cond, uc, timesteps = self.prepare_sampling_loop(x, cond, uc, num_steps)
        old_denoised = None
        for i in range(num_sigmas - 1):
            x, old_denoised = self.sampler_step(
                old_denoised,
                None if i == 0 else s_in * table[i - 1], s_in * table[i], s_in * table[i + 1],
                x, cond, uc=uc, idx=self.num_steps - i, timestep=timesteps[-(i + 1)])
        return x


class DenoiserSampler:
    """Glue of ``sample.py``: build the sampler from the job configuration's ``denoiser`` / ``discretization`` /
    ``guider`` / ``eval`` sections and draw latents (reference ``cogvideo/sampler.py``:197-246)."""

    def __init__(self, model: nn.Module, config, dtype, effective_rank: int, seed: int = 0, device: str = "cuda",
                 use_wandb: bool = False, batch_samples: bool = True):
        self.sampler = VPSDEDPMPP2MSampler(
            denoiser=DiscreteDenoiser(model, num_idx=config.denoiser.num_idx, quantize_c_noise=config.denoiser.quantize_c_noise,
                                      dtype=dtype, batch_samples=batch_samples),
            discretization_config={"shift_scale": config.discretization.shift_scale},
            guider_config={"scale": config.guider.scale, "exp": config.guider.exp, "num_steps": config.eval.num_denoising_steps},
            verbose=False, device=device, num_steps=config.eval.num_denoising_steps, use_wandb=use_wandb)
        self.noise_generator = torch.Generator(device=device)
        self.noise_generator.manual_seed(effective_rank + seed)
        self.device = device
        self.dtype = dtype

    @torch.no_grad()
    def sample(self, text_emb: torch.Tensor, neg_emb: torch.Tensor, shape: tuple, batch_size: int):
        if torch.cuda.is_available() and str(self.device).startswith("cuda"):
            # a training run in the same process leaves the TTT-MLP backward's step records cached (2.2 GB per stream at 48 heads,
            # test_time_training._workspace): sampling never runs a backward, give the memory back
            import test_time_training
            test_time_training.release_workspaces()
        noise = torch.randn(batch_size, *shape, device=self.device, generator=self.noise_generator, dtype=torch.float32)
        return self.sampler(noise, {"crossattn": text_emb}, {"crossattn": neg_emb}).to(self.dtype)
