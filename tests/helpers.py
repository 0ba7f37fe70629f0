"""Shared test helpers (golden loading, error metrics)."""
import os

import torch

from oracle import ttt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def op_inputs(g):
    """Regenerate the seeded inputs of an op-level golden and verify their checksums."""
    dtype = getattr(torch, g["dtype"])
    d = O.make_inputs(dtype=dtype, **g["gen"])
    for k, ref in g["input_checksums"].items():
        got = float(d[k].double().abs().sum())
        assert abs(got - ref) <= 1e-9 * max(1.0, abs(ref)), f"input RNG drift in {k}: {got} vs {ref}"
    return d


def rel_l2(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def tile_states(d, B):
    t = lambda w: torch.tile(w.unsqueeze(0), dims=(B, 1, 1, 1)).contiguous()
    return {k: t(d[k]) for k in ("W1", "b1", "W2", "b2") if k in d}


class ToyNet(torch.nn.Module):
    """Stand-in for the DiT in sampler tests: nonlinear in x, depends on text and timestep, independent per sample."""

    def forward(self, x, text, t):
        g = torch.tanh(text.float().mean(dim=(1, 2, 3))).view(-1, 1, 1, 1, 1)
        tt = torch.sin(t.float() / 100).view(-1, 1, 1, 1, 1)
        xf = x.float()
        return (0.6 * torch.tanh(xf.roll(1, -1)) + 0.3 * g * xf.roll(1, 1) + 0.1 * tt).to(x.dtype)
