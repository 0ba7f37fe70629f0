"""Projection helpers of the DiT block (host side, PyTorch-ROCm; the GEMMs go to hipBLASLt).

``Linear3``: the q / k / v projections of the attention block and the wq / wk / wv projections of the TTT layer read the SAME
input; as one autograd node their input gradient is one GEMM accumulation chain instead of three GEMMs and two full-size additions.

(Round 1 also carried a side-stream queue that ran the weight-gradient GEMMs beside the backward scans, opt-in until timed.  Timed
in round 2 on an MI355X - 6 612 vs 6 619 video-tok/s at the 3 s configuration, the backward scan 6.94 vs 6.81 ms,
profiles/r2h_bench_3s_overlap_wgrad.json: the cluster sweep leaves 64 CUs idle, not 208 - it lost and was removed.)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _plain(p) -> bool:
    return p is None or type(p) in (torch.Tensor, torch.nn.Parameter)


class Linear3(torch.autograd.Function):
    """Three projections of the SAME input (q / k / v of the attention block, wq / wk / wv of the TTT layer) as one autograd
    node: ``(F.linear(x, w0, b0), F.linear(x, w1, b1), F.linear(x, w2, b2))``.  Its backward forms the input gradient as ONE
    accumulation chain - ``dX = dY0 W0`` then two GEMMs with ``beta = 1`` into the same buffer - instead of three GEMM outputs
    and the two full-size additions autograd inserts for a tensor with three consumers (2 x 333 MB of traffic per group at
    the 3 s geometry, 6 such additions per layer; accumulated in the GEMM's fp32 epilogue, so also rounded once instead of
    three times)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2):
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)             # an unused output arrives as None, not as a tensor of zeros
        ctx.ws, ctx.bs = (w0, w1, w2), (b0, b1, b2)
        return F.linear(x, w0, b0), F.linear(x, w1, b1), F.linear(x, w2, b2)

    @staticmethod
    def backward(ctx, *dys):
        (x,) = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1])
        dx2 = None
        gw, gb = [None] * 3, [None] * 3
        for i, (dy, w, b) in enumerate(zip(dys, ctx.ws, ctx.bs)):
            if dy is None:
                continue
            dy2 = dy.reshape(-1, dy.shape[-1])
            if ctx.needs_input_grad[0]:
                if dx2 is None:
                    dx2 = dy2.mm(w)
                else:
                    dx2.addmm_(dy2, w)
            if w.requires_grad:
                gw[i] = dy2.t().mm(x2)
            if b is not None and b.requires_grad:
                gb[i] = dy2.sum(0)
        dx = None if dx2 is None else dx2.view(x.shape)
        return dx, gw[0], gb[0], gw[1], gb[1], gw[2], gb[2]


def linear3(m0: torch.nn.Linear, m1: torch.nn.Linear, m2: torch.nn.Linear, x: torch.Tensor):
    """``(m0(x), m1(x), m2(x))`` through ``Linear3`` when that is an exact substitute (plain-tensor parameters of x's dtype, no
    autocast, gradients enabled); the three module calls otherwise."""
    ok = torch.is_grad_enabled() and not torch.is_autocast_enabled(x.device.type) and x.dim() >= 2
    for m in (m0, m1, m2):
        ok = ok and _plain(m.weight) and _plain(m.bias) and m.weight.dtype == x.dtype and (m.bias is None or m.bias.dtype == x.dtype)
    if ok and (x.requires_grad or any(m.weight.requires_grad for m in (m0, m1, m2))):
        return Linear3.apply(x, m0.weight, m0.bias, m1.weight, m1.bias, m2.weight, m2.bias)
    return m0(x), m1(x), m2(x)
