"""Autograd boundaries around the fused pre- / post-processing HIP kernels of the TTT layer
(``csrc/ttt_prepost.hip`` behind ``include/ttt_hip.h``): the MI355X replacement for the chains of
elementwise PyTorch ops in the reference's ``TTTBase.process_input`` (``ttt_layer.py:252-306``), its
``post_norm`` + re-layout tail (``:327-334``) and the gated residual of ``SeqModelingBlock._gate``
(``cogvideo/dit.py:219-222``).  Used only for bf16 activations on a HIP device; every other case keeps the
plain PyTorch path in ``ttt_layer.py`` (which is also the parity reference of these kernels in the tests)."""
from __future__ import annotations

import torch

_BF16, _F32 = torch.bfloat16, torch.float32


from ttt_amd.infra.remat_cache import kernel_result


def _ext():
    import test_time_training
    return test_time_training


def fused_available(x: torch.Tensor, head_dim: int) -> bool:
    return x.is_cuda and x.dtype == _BF16 and head_dim == 64


def qkv_grad_blocks(like: torch.Tensor):
    """Three ``[B, L, D]`` gradient tensors (d q_raw, d k_raw, d v_raw of projections that read the SAME input) as the column blocks
    of ONE ``[B, L, 3 D]`` buffer: ``Linear3.backward`` (``ttt_amd/infra/fused_linear.py``) recognises the layout and forms the
    three weight gradients as one ``[3 D, L] x [L, D]`` GEMM and the input gradient as one ``[L, 3 D] x [3 D, D]`` GEMM instead of
    three each (the 3072 x 3072 x L weight-gradient shape fills 144 of 256 CUs' worth of tiles and runs at 43 % of the MFMA roof,
    profiles/r4y_bench_default_kernel_stats.csv).  The kernels that write them take the row stride (``ld_out``)."""
    B, L, D = like.shape
    buf = torch.empty(B, L, 3 * D, device=like.device, dtype=like.dtype)
    return buf[..., :D], buf[..., D:2 * D], buf[..., 2 * D:]


class FusedPre(torch.autograd.Function):
    """(XQ_raw, XK_raw, XV_raw [B,L,NH*64], ln_w, ln_b [NH,64], rope [n,32,2] | None, src, pos [L] int32 | None)
    -> XQ, XK, XV [B,NH,L,64] in scan order (token permutation, L2-norm, RoPE, LayerNorm target fused)."""

    @staticmethod
    def forward(ctx, XQ_raw, XK_raw, XV_raw, ln_w, ln_b, rope, src, pos, NH):
        ext = _ext()
        B, L, D = XQ_raw.shape
        q, k, v = XQ_raw.contiguous(), XK_raw.contiguous(), XV_raw.contiguous()
        w32, b32 = ln_w.detach().to(_F32).contiguous(), ln_b.detach().to(_F32).contiguous()
        outs = [torch.empty(B, NH, L, D // NH, device=q.device, dtype=_BF16) for _ in range(3)]
        ext.pre_forward(q, k, v, rope, src, pos, w32, b32, *outs, NH, n_pos=getattr(pos, "_ttt_max_pos", None))
        ctx.save_for_backward(q, k, v, w32, rope, src, pos)
        ctx.NH, ctx.param_dtype = NH, ln_w.dtype
        return tuple(outs)

    @staticmethod
    def backward(ctx, dXQ, dXK, dXV):
        ext = _ext()
        q, k, v, w32, rope, src, pos = ctx.saved_tensors
        NH = ctx.NH
        P = ext.pre_backward_partials(NH)
        dq, dk, dv = qkv_grad_blocks(q)
        pw = torch.empty(P, q.shape[-1], device=q.device, dtype=_F32)
        pb = torch.empty_like(pw)
        ext.pre_backward(q, k, v, rope, src, pos, w32, dXQ.contiguous(), dXK.contiguous(), dXV.contiguous(), dq, dk, dv, pw, pb, NH,
                         ld_out=dq.stride(1))
        dw = pw.sum(0).view(NH, -1).to(ctx.param_dtype)
        db = pb.sum(0).view(NH, -1).to(ctx.param_dtype)
        return dq, dk, dv, dw, db, None, None, None, None


class FusedPost(torch.autograd.Function):
    """(Y [B,NH,L,64] scan order, weight, bias [D], src | None, eps) -> LayerNorm_D(Y) as [B,L,D] in token order."""

    @staticmethod
    def forward(ctx, Y, weight, bias, src, eps):
        ext = _ext()
        B, NH, L, F = Y.shape
        y = Y.contiguous()
        w32, b32 = weight.detach().to(_F32).contiguous(), bias.detach().to(_F32).contiguous()
        from ttt_amd.models.ssm.pipeline import injected
        out = injected("post")
        if out is None:
            out = torch.empty(B, L, NH * F, device=y.device, dtype=_BF16)
            ext.post_forward(y, src, w32, b32, out, float(eps))
        ctx.save_for_backward(y, w32, src)
        ctx.eps, ctx.param_dtype = float(eps), weight.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        ext = _ext()
        y, w32, src = ctx.saved_tensors
        B, NH, L, F = y.shape
        P = ext.post_partials(B, L)
        dY = torch.empty_like(y)
        pw = torch.empty(P, NH * F, device=y.device, dtype=_F32)
        pb = torch.empty_like(pw)
        ext.post_backward(y, g.contiguous(), src, w32, dY, pw, pb, ctx.eps)
        return dY, pw.sum(0).to(ctx.param_dtype), pb.sum(0).to(ctx.param_dtype), None, None


class FusedGate(torch.autograd.Function):
    """residual + tanh(alpha_text | alpha_video) * y with the first ``n_text`` tokens using the text gate."""

    @staticmethod
    def forward(ctx, res, y, alpha_text, alpha_video, n_text):
        ext = _ext()
        r, yy = res.contiguous(), y.contiguous()
        tt, tv = torch.tanh(alpha_text.detach().to(_F32)).contiguous(), torch.tanh(alpha_video.detach().to(_F32)).contiguous()
        out = torch.empty_like(r)
        ext.gate_forward(r, yy, tt, tv, out, n_text)
        ctx.save_for_backward(yy, tt, tv)
        ctx.n_text, ctx.param_dtype = n_text, alpha_text.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        ext = _ext()
        yy, tt, tv = ctx.saved_tensors
        g = g.contiguous()
        D = g.shape[-1]
        P = ext.gate_backward_partials(D)
        dy = torch.empty_like(g)
        part = torch.empty(P, 2, D, device=g.device, dtype=_F32)
        ext.gate_backward(g, yy, tt, tv, dy, part, ctx.n_text)
        dt = part.sum(0)                                   # d/d tanh(alpha)
        da_t = (dt[0] * (1 - tt * tt)).to(ctx.param_dtype)
        da_v = (dt[1] * (1 - tv * tv)).to(ctx.param_dtype)
        return g, dy, da_t, da_v, None


class FusedPreScanMLP(torch.autograd.Function):
    """``FusedPre`` followed by the TTT-MLP scan (``TkMLP``) as ONE autograd node that keeps only the raw projections:
    the processed XQ / XK / XV (3 x [B,L,D] per call) are re-derived in the backward by re-running the pre kernel
    (~0.2 ms) instead of living from forward to backward.  Activation memory is what bounds how many transformer
    layers can skip re-materialisation on a 288-GB MI355X (``remat_free_layers``).

    (XQ_raw, XK_raw, XV_raw [B,L,NH*64], ln_w, ln_b [NH,64], rope, src, pos, NH, W1, b1, W2, b2 [B,NH,..] fp32 views,
     eta [B,NH,NC,1,CS], G) -> XQW [B,NH,NC,CS,64]."""

    @staticmethod
    def forward(ctx, XQ_raw, XK_raw, XV_raw, ln_w, ln_b, rope, src, pos, NH, W1, b1, W2, b2, eta, G):
        import math
        ext = _ext()
        B, L, D = XQ_raw.shape
        Fh = D // NH
        q, k, v = XQ_raw.contiguous(), XK_raw.contiguous(), XV_raw.contiguous()
        w32, b32 = ln_w.detach().to(_F32).contiguous(), ln_b.detach().to(_F32).contiguous()
        CS = eta.shape[-1]
        NC = L // CS
        K = math.ceil(NC / G)
        last_eta = eta.to(_BF16)[:, :, :, -1, :, None].contiguous()

        def run():
            from ttt_amd.models.ssm.pipeline import injected
            pre = injected("scan")           # (a pipelined forward: the scan has walked the sequence part by part already)
            if pre is not None:
                return pre
            XQ, XK, XV = (torch.empty(B, NH, L, Fh, device=q.device, dtype=_BF16) for _ in range(3))
            ext.pre_forward(q, k, v, rope, src, pos, w32, b32, XQ, XK, XV, NH, n_pos=getattr(pos, "_ttt_max_pos", None))
            mb = lambda t: t.view(B, NH, NC, CS, Fh)
            lw, lb = w32.reshape(1, NH, 1, Fh), b32.reshape(1, NH, 1, Fh)
            state = [t.to(_F32).contiguous() for t in (W1, b1, W2, b2)]
            out = torch.empty(B, NH, NC, CS, Fh, device=q.device, dtype=_BF16)
            cks = (torch.empty(B, NH, K, Fh, 4 * Fh, device=q.device, dtype=_F32), torch.empty(B, NH, K, 1, 4 * Fh, device=q.device, dtype=_F32),
                   torch.empty(B, NH, K, 4 * Fh, Fh, device=q.device, dtype=_F32), torch.empty(B, NH, K, 1, Fh, device=q.device, dtype=_F32))
            ext.ttt_forward(mb(XQ), mb(XK), mb(XV), last_eta, lw, lb, *state, *cks, out, G)
            return (out, *cks)

        # (inside a checkpointed region that keeps "scan" the recomputation gets the scan output and the state checkpoints back)
        out, *cks = kernel_result("scan", run)
        ctx.save_for_backward(q, k, v, w32, b32, rope, src, pos, last_eta, *cks)
        ctx.meta = (NH, G, tuple(eta.shape), ln_w.dtype, W1.dtype, eta.dtype)
        ctx.n_pos = getattr(pos, "_ttt_max_pos", None) if pos is not None else None      # (the backward's pre_forward must not fall into the binding's synchronising pos.max())
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ext = _ext()
        q, k, v, w32, b32, rope, src, pos, last_eta, W1c, b1c, W2c, b2c = ctx.saved_tensors
        NH, G, eta_shape, ln_dt, st_dt, eta_dt = ctx.meta
        B, L, D = q.shape
        Fh = D // NH
        CS = eta_shape[-1]
        NC = L // CS
        H, dev = 4 * Fh, q.device
        XQ, XK, XV = (torch.empty(B, NH, L, Fh, device=dev, dtype=_BF16) for _ in range(3))
        ext.pre_forward(q, k, v, rope, src, pos, w32, b32, XQ, XK, XV, NH, n_pos=ctx.n_pos)      # re-derive the scan inputs
        mb = lambda t: t.view(B, NH, NC, CS, Fh)
        z32 = lambda *s: torch.zeros(*s, device=dev, dtype=_F32)
        e32 = lambda *s: torch.empty(*s, device=dev, dtype=_F32)
        e16 = lambda *s: torch.empty(*s, device=dev, dtype=_BF16)
        up = (z32(B, NH, Fh, H), z32(B, NH, 1, H), z32(B, NH, H, Fh), z32(B, NH, 1, Fh))
        g_out = grad_out.to(_BF16).contiguous()
        # the reference's sixteen caller-allocated re-materialisation buffers (mlp_tk.py:192-210): the MFMA backward touches none
        # of them (0.55 GB per call at 48 heads that nobody reads - round-3 verdict), the generic kernels four; the reference-
        # shaped wrapper (mlp_tk.py here) keeps allocating all of them, as the ABI it mirrors demands
        if ext.resolved_impl(B, NH, NC, CS, Fh, G, _BF16, mlp=True, backward=True) == "generic":
            remat = (e32(B, NH, G, Fh, H), e32(B, NH, G, 1, H), e32(B, NH, G, H, Fh), e32(B, NH, G, 1, Fh)) + (None,) * 12
        else:
            remat = (None,) * 16
        d_lnw, d_lnb = e32(B, NH, 1, Fh), e32(B, NH, 1, Fh)
        d_state = (e32(B, NH, Fh, H), e32(B, NH, 1, H), e32(B, NH, H, Fh), e32(B, NH, 1, Fh))
        d_eta = torch.empty(B, NH, NC, CS, 1, device=dev, dtype=_BF16)
        dQ, dK, dV = (torch.empty(B, NH, NC, CS, Fh, device=dev, dtype=_BF16) for _ in range(3))
        lw, lb = w32.reshape(1, NH, 1, Fh), b32.reshape(1, NH, 1, Fh)
        ext.ttt_backward(mb(XQ), mb(XK), mb(XV), last_eta, lw, lb, W1c, b1c, W2c, b2c, g_out, *remat, *up, g_out,
                         d_lnw, d_lnb, *d_state, d_eta, dQ, dK, dV, G)
        del XQ, XK, XV
        # pre backward (re-uses the raw projections)
        P = ext.pre_backward_partials(NH)
        dq, dk, dv = qkv_grad_blocks(q)
        pw = torch.empty(P, D, device=dev, dtype=_F32)
        pb = torch.empty_like(pw)
        flat = lambda t: t.view(B, NH, L, Fh)
        ext.pre_backward(q, k, v, rope, src, pos, w32, flat(dQ), flat(dK), flat(dV), dq, dk, dv, pw, pb, NH, ld_out=dq.stride(1))
        g_w = (pw.sum(0).view(NH, Fh) + d_lnw.sum(dim=0).squeeze(1)).to(ln_dt)
        g_b = (pb.sum(0).view(NH, Fh) + d_lnb.sum(dim=0).squeeze(1)).to(ln_dt)
        row = d_eta.transpose(-2, -1)
        rows = eta_shape[-2]
        d_eta_full = row if rows == 1 else torch.nn.functional.pad(row, (0, 0, rows - 1, 0))
        return (dq, dk, dv, g_w, g_b, None, None, None, None, *(g.to(st_dt) for g in d_state), d_eta_full.to(eta_dt), None)


class FusedAdaLN(torch.autograd.Function):
    """(vid [B,Lv,D], text [B,Lt,D], LayerNorm weight / bias [D], shift_v, scale_v, shift_t, scale_t [B,D], eps) ->
    ``cat(modulate(LN(text), shift_t, scale_t), modulate(LN(vid), shift_v, scale_v))`` as one ``[B, Lt+Lv, D]`` tensor:
    the TransformerLayer's layernorm + modulate + concat (reference ``cogvideo/dit.py:353-357, 366-371``) in one HIP pass
    per direction; keeps only its inputs for backward."""

    @staticmethod
    def forward(ctx, vid, text, w, b, shift_v, scale_v, shift_t, scale_t, eps):
        ext = _ext()
        v, t = vid.contiguous(), text.contiguous()
        B, Lv, D = v.shape
        w32, b32 = w.detach().to(_F32).contiguous(), b.detach().to(_F32).contiguous()
        # [B, 2, D], group 0 = text, 1 = video; "1 + scale" is formed in the activation dtype like the unfused modulate
        shift = torch.stack((shift_t, shift_v), dim=1).detach().to(_F32).contiguous()
        scale1p = torch.stack((1 + scale_t, 1 + scale_v), dim=1).detach().to(_F32).contiguous()
        out = torch.empty(B, t.shape[1] + Lv, D, device=v.device, dtype=_BF16)
        ext.adaln_forward(v, t, w32, b32, shift, scale1p, out, float(eps))
        ctx.save_for_backward(v, t, w32, b32, scale1p)
        ctx.meta = (float(eps), w.dtype, shift_v.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        ext = _ext()
        v, t, w32, b32, scale1p = ctx.saved_tensors
        eps, pdt, mdt = ctx.meta
        B, Lv, D = v.shape
        P = ext.adaln_backward_partials()
        dv, dt = torch.empty_like(v), torch.empty_like(t)
        part = torch.empty(B * 2 * P, 4, D, device=v.device, dtype=_F32)
        ext.adaln_backward(v, t, g.contiguous(), w32, b32, scale1p, dv, dt, part, eps)
        part = part.view(B, 2, P, 4, D).sum(2)                     # [B, group, 4, D]
        dw = part[:, :, 0].sum((0, 1)).to(pdt)
        db = part[:, :, 1].sum((0, 1)).to(pdt)
        dsc, dsh = part[:, :, 2].to(mdt), part[:, :, 3].to(mdt)    # [B, group, D]
        return dv, dt, dw, db, dsh[:, 1], dsc[:, 1], dsh[:, 0], dsc[:, 0], None


class FusedResGate(torch.autograd.Function):
    """(vid [B,Lv,D], text [B,Lt,D], y [B,Lt+Lv,D] = [text | video], gate_v, gate_t [B,D]) ->
    (vid + gate_v * y[:, Lt:], text + gate_t * y[:, :Lt]): the TransformerLayer's gated residuals
    (reference ``cogvideo/dit.py:358-359, 372-373``) in one pass; the backward writes dy once instead of two padded slices."""

    @staticmethod
    def forward(ctx, vid, text, y, gate_v, gate_t):
        ext = _ext()
        v, t, yy = vid.contiguous(), text.contiguous(), y.contiguous()
        gate = torch.stack((gate_t, gate_v), dim=1).detach().to(_F32).contiguous()
        ov, ot = torch.empty_like(v), torch.empty_like(t)
        ext.resgate_forward(v, t, yy, gate, ov, ot)
        ctx.save_for_backward(yy, gate)
        ctx.gdt = gate_v.dtype
        return ov, ot

    @staticmethod
    def backward(ctx, dv, dt):
        ext = _ext()
        yy, gate = ctx.saved_tensors
        dv, dt = dv.contiguous(), dt.contiguous()
        B, L, D = yy.shape
        P = ext.resgate_backward_partials(D)
        dy = torch.empty_like(yy)
        part = torch.empty(P, B, 2, D, device=yy.device, dtype=_F32)
        ext.resgate_backward(dv, dt, yy, gate, dy, part)
        dg = part.sum(0).to(ctx.gdt)                               # [B, 2, D]
        return dv, dt, dy, dg[:, 1], dg[:, 0]
