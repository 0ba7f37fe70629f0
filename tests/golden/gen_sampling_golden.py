"""Golden vectors for the sampling-time classes, made by EXECUTING THE REFERENCE (build container only):

    python tests/golden/gen_sampling_golden.py

Runs the reference's ZeroSNRDDPMDiscretization / DiscreteDenoiser / DynamicCFG / VPSDEDPMPP2MSampler
(/root/reference/ttt/models/cogvideo/utils.py:312-711) on CPU around tests.helpers.ToyNet and stores tables, denoiser
outputs and final samples in sampling.pt.  Only numbers are saved.
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.modules["wandb"] = types.ModuleType("wandb")
import tomli  # noqa: E402

sys.modules["tomllib"] = tomli
sys.path.insert(0, "/root/reference")

from ttt.models.cogvideo import utils as R  # noqa: E402

from tests.helpers import ToyNet  # noqa: E402

# the reference hard-codes device="cuda" as the default of these two methods; there is no GPU in this container
R.ZeroSNRDDPMDiscretization.get_sigmas.__defaults__ = ("cpu", False)
R.ZeroSNRDDPMDiscretization.__call__.__defaults__ = (False, "cpu", False, False)

out = {}
out["table_1000_flip"] = R.ZeroSNRDDPMDiscretization()(1000, flip=True)
s, idx = R.ZeroSNRDDPMDiscretization()(50, return_idx=True)
out["table_50"], out["idx_50"] = s, torch.tensor(list(idx))
s, idx = R.ZeroSNRDDPMDiscretization(shift_scale=3.0)(17, return_idx=True, do_append_zero=True)
out["table_17_shift3_zero"], out["idx_17"] = s, torch.tensor(list(idx))

den = R.DiscreteDenoiser(ToyNet(), num_idx=1000, quantize_c_noise=False, dtype=torch.float32)
torch.manual_seed(5)
x = torch.randn(2, 3, 4, 6, 5)
text = torch.randn(2, 2, 7, 16)
sig = torch.tensor([0.31, 0.87])
out["den_in"] = dict(x=x, text=text, sigma=sig, idx=torch.tensor([700.0, 120.0]))
out["den_out"] = den(x, sig, {"crossattn": text}, idx=out["den_in"]["idx"])
denq = R.DiscreteDenoiser(ToyNet(), num_idx=1000, quantize_c_noise=True, dtype=torch.float32)
out["den_out_quantized"] = denq(x, sig, {"crossattn": text}, idx=torch.tensor([0.5, 0.9]))

cases = {}
for name, steps, seed, shift in (("s8", 8, 11, 1.0), ("s50", 50, 12, 1.0), ("s20_shift", 20, 13, 2.5)):
    sampler = R.VPSDEDPMPP2MSampler(denoiser=R.DiscreteDenoiser(ToyNet(), num_idx=1000, quantize_c_noise=False, dtype=torch.float32),
                                    discretization_config={"shift_scale": shift},
                                    guider_config={"scale": 6, "exp": 5, "num_steps": steps},
                                    use_wandb=False, verbose=False, device="cpu", num_steps=steps)
    torch.manual_seed(seed)
    noise = torch.randn(1, 3, 4, 6, 5)
    text, neg = torch.randn(1, 2, 7, 16), torch.randn(1, 2, 7, 16)
    with torch.no_grad():
        res = sampler(noise, {"crossattn": text}, {"crossattn": neg})
    cases[name] = dict(steps=steps, seed=seed, shift=shift, result=res)
out["cases"] = cases
torch.save(out, os.path.join(HERE, "sampling.pt"))
print({k: (v["result"].abs().mean().item(), torch.isfinite(v["result"]).all().item()) for k, v in cases.items()})
