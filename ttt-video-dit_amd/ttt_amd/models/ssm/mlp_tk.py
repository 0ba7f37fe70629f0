"""``TkMLP``: autograd boundary around the TTT-MLP scan kernels.

Same signature and return contract as the reference's ``ttt/models/ssm/mlp_tk.py`` (``TkMLP.apply(
ttt_norm_weight, ttt_norm_bias, W1, b1, W2, b2, XQ, XV, XK, eta, checkpoint_group_size)`` ->
``[B,NH,NC,CS,F]``; 11 gradients, last None), and the same division of labour: Python allocates
every buffer, the extension ``test_time_training`` (here: HIP for gfx950) only computes.

Differences from the reference wrapper, on purpose:
  * the eta gradient is padded with CS-1 rows instead of a hard-coded 63 (mlp_tk.py:280 assumes
    CS=64 although every eval config uses 16 - SURVEY.md hazard C3);
  * ``eta`` may also be given directly as its last row ``[B,NH,NC,1,CS]`` so callers need not
    materialise the 64x redundant tile; the gradient then has that shape too.
"""
from __future__ import annotations

import math

import torch

_BF16, _F32 = torch.bfloat16, torch.float32


def _ext():
    import test_time_training  # the HIP extension (raises if its library is missing)
    return test_time_training


def _last_row(eta: torch.Tensor, act_dtype) -> torch.Tensor:
    # mlp_tk.py:104-105 : cast, take the last row of the [CS,CS] tile, as a column [.., CS, 1]
    return eta.to(act_dtype)[:, :, :, -1, :, None].contiguous()


class TkMLP(torch.autograd.Function):
    sharded_mode = False  # head-sharded tensor parallel (next row): op is head-local

    @staticmethod
    def forward(ctx, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, W2_init, b2_init,
                XQ_batch, XV_batch, XK_batch, eta_batch, checkpoint_group_size):
        ext = _ext()
        B, NH, NC, CS, F = XQ_batch.shape
        G = int(checkpoint_group_size)
        K = math.ceil(NC / G)
        dev, act = XQ_batch.device, XQ_batch.dtype
        if act != _BF16:
            raise AssertionError("TTT-MLP kernel must run in mixed-precision bfloat16.")  # mlp_tk.py:89

        XQ, XV, XK = XQ_batch.contiguous(), XV_batch.contiguous(), XK_batch.contiguous()
        last_eta = _last_row(eta_batch, act)
        ln_w = ttt_norm_weight.reshape(1, NH, 1, F).to(_F32).contiguous()
        ln_b = ttt_norm_bias.reshape(1, NH, 1, F).to(_F32).contiguous()
        state = [t.to(_F32).contiguous() for t in (W1_init, b1_init, W2_init, b2_init)]

        out = torch.empty(B, NH, NC, CS, F, device=dev, dtype=act)
        cks = (torch.empty(B, NH, K, F, 4 * F, device=dev, dtype=_F32), torch.empty(B, NH, K, 1, 4 * F, device=dev, dtype=_F32),
               torch.empty(B, NH, K, 4 * F, F, device=dev, dtype=_F32), torch.empty(B, NH, K, 1, F, device=dev, dtype=_F32))
        ext.ttt_forward(XQ, XK, XV, last_eta, ln_w, ln_b, *state, *cks, out, G)

        ctx.save_for_backward(XQ, XV, XK, last_eta, ln_w, ln_b, *cks)   # XQW is not an input of the backward arithmetic
        ctx.G = G
        ctx.eta_shape = tuple(eta_batch.shape)
        ctx.param_dtypes = (ttt_norm_weight.dtype, W1_init.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ext = _ext()
        XQ, XV, XK, last_eta, ln_w, ln_b, W1c, b1c, W2c, b2c = ctx.saved_tensors
        B, NH, NC, CS, F = XQ.shape
        G, H, dev = ctx.G, 4 * XQ.shape[-1], XQ.device
        z32 = lambda *s: torch.zeros(*s, device=dev, dtype=_F32)
        e32 = lambda *s: torch.empty(*s, device=dev, dtype=_F32)
        e16 = lambda *s: torch.empty(*s, device=dev, dtype=_BF16)

        up = (z32(B, NH, F, H), z32(B, NH, 1, H), z32(B, NH, H, F), z32(B, NH, 1, F))  # final state is not an output
        g_out = grad_out.to(_BF16).contiguous()
        out = g_out                 # placeholder for the ABI's XQW slot (mlp_tk.py:236): same shape / dtype, never read
        # re-materialisation scratch, shapes/dtypes of mlp_tk.py:192-210
        remat = (e32(B, NH, G, F, H), e32(B, NH, G, 1, H), e32(B, NH, G, H, F), e32(B, NH, G, 1, F),
                 e16(B, NH, G, CS, F), e32(B, NH, G, CS, 1),
                 e16(B, NH, G, CS, H), e16(B, NH, G, CS, H), e16(B, NH, G, CS, H), e16(B, NH, G, CS, H),
                 e16(B, NH, G, CS, F), e16(B, NH, G, CS, H), e16(B, NH, G, CS, F), e16(B, NH, G, CS, F), e16(B, NH, G, CS, F),
                 e32(B, NH, G, CS, 1))
        d_lnw, d_lnb = e32(B, NH, 1, F), e32(B, NH, 1, F)       # per batch element, summed below
        d_state = (e32(B, NH, F, H), e32(B, NH, 1, H), e32(B, NH, H, F), e32(B, NH, 1, F))
        d_eta = torch.empty(B, NH, NC, CS, 1, device=dev, dtype=_BF16)
        dQ, dK, dV = (torch.empty_like(XQ) for _ in range(3))

        ext.ttt_backward(XQ, XK, XV, last_eta, ln_w, ln_b, W1c, b1c, W2c, b2c, out, *remat, *up, g_out,
                         d_lnw, d_lnb, *d_state, d_eta, dQ, dK, dV, G)

        ln_dt, st_dt = ctx.param_dtypes
        d_lnw = d_lnw.sum(dim=0).squeeze(1).to(ln_dt)
        d_lnb = d_lnb.sum(dim=0).squeeze(1).to(ln_dt)
        row = d_eta.transpose(-2, -1)                              # [B,NH,NC,1,CS]
        rows = ctx.eta_shape[-2]
        d_eta_full = row if rows == 1 else torch.nn.functional.pad(row, (0, 0, rows - 1, 0))
        act = XQ.dtype
        return (d_lnw, d_lnb, *(g.to(st_dt) for g in d_state), dQ.to(act), dV.to(act), dK.to(act), d_eta_full.to(act), None)
