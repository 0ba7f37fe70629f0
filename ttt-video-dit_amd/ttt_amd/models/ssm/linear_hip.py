"""``HipLinear`` (alias ``TritonLinear``): autograd boundary around the TTT-Linear scan kernels.

Signature of the reference's ``TritonLinear.apply(ttt_norm_weight, ttt_norm_bias, W1, b1, XQ, XV,
XK, eta, checkpoint_group_size)`` (``ttt/models/ssm/linear_triton.py:14-26``; call site
``ttt_layer.py:371-381``).  The Triton kernels it replaces (``kernels/linear_forward.py``,
``linear_backward.py``) are re-implemented as HIP for gfx950 behind ``test_time_training``.
Activations may be bf16 or fp32 (the Triton path accepts both); state and checkpoints are fp32.
"""
from __future__ import annotations

import math

import torch

_F32 = torch.float32


def _ext():
    import test_time_training
    return test_time_training


class HipLinear(torch.autograd.Function):
    sharded_mode = False

    @staticmethod
    def forward(ctx, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, XQ_batch, XV_batch, XK_batch, eta_batch,
                checkpoint_group_size):
        ext = _ext()
        B, NH, NC, CS, F = XQ_batch.shape
        G = int(checkpoint_group_size)
        K = math.ceil(NC / G)
        dev, act = XQ_batch.device, XQ_batch.dtype
        XQ, XV, XK = XQ_batch.contiguous(), XV_batch.contiguous(), XK_batch.contiguous()
        last_eta = eta_batch.to(act)[:, :, :, -1, :, None].contiguous()   # kernels/linear_forward.py:90-101
        ln_w = ttt_norm_weight.reshape(NH, F).to(_F32).contiguous()
        ln_b = ttt_norm_bias.reshape(NH, F).to(_F32).contiguous()
        W1, b1 = W1_init.to(_F32).contiguous(), b1_init.to(_F32).contiguous()
        out = torch.empty(B, NH, NC, CS, F, device=dev, dtype=act)
        W1c = torch.empty(B, NH, K, F, F, device=dev, dtype=_F32)
        b1c = torch.empty(B, NH, K, 1, F, device=dev, dtype=_F32)
        ext.ttt_linear_forward(XQ, XK, XV, last_eta, ln_w, ln_b, W1, b1, W1c, b1c, out, G)
        ctx.save_for_backward(XQ, XV, XK, last_eta, ln_w, ln_b, W1c, b1c)
        ctx.G = G
        ctx.eta_shape = tuple(eta_batch.shape)
        ctx.param_dtypes = (ttt_norm_weight.dtype, W1_init.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ext = _ext()
        XQ, XV, XK, last_eta, ln_w, ln_b, W1c, b1c = ctx.saved_tensors
        B, NH, NC, CS, F = XQ.shape
        G, dev, act = ctx.G, XQ.device, XQ.dtype
        e32 = lambda *s: torch.empty(*s, device=dev, dtype=_F32)
        up = (torch.zeros(B, NH, F, F, device=dev, dtype=_F32), torch.zeros(B, NH, 1, F, device=dev, dtype=_F32))
        grp = (e32(B, NH, G, F, F), e32(B, NH, G, 1, F))
        d_lnw, d_lnb = e32(B, NH, 1, F), e32(B, NH, 1, F)
        dW1, db1 = e32(B, NH, F, F), e32(B, NH, 1, F)
        d_eta = torch.empty(B, NH, NC, CS, 1, device=dev, dtype=act)
        dQ, dK, dV = (torch.empty_like(XQ) for _ in range(3))
        ext.ttt_linear_backward(XQ, XK, XV, last_eta, ln_w, ln_b, W1c, b1c, *up, grad_out.to(act).contiguous(), *grp,
                                d_lnw, d_lnb, dW1, db1, d_eta, dQ, dK, dV, G)
        ln_dt, st_dt = ctx.param_dtypes
        row = d_eta.transpose(-2, -1)
        rows = ctx.eta_shape[-2]
        d_eta_full = row if rows == 1 else torch.nn.functional.pad(row, (0, 0, rows - 1, 0))  # linear_backward.py:134-135
        return (d_lnw.sum(0).squeeze(1).to(ln_dt), d_lnb.sum(0).squeeze(1).to(ln_dt), dW1.to(st_dt), db1.to(st_dt),
                dQ, dV, dK, d_eta_full.to(act), None)


TritonLinear = HipLinear  # reference class name (ttt/models/ssm/linear_triton.py:12)
