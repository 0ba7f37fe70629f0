"""Local (per-segment) self-attention of the CogVideoX block on the hand-written gfx950 kernels
(``csrc/attn_fwd.hip``, ``csrc/attn_bwd.hip`` behind ``ttt_hip_attn_forward/backward``).

``segment_attention(q, k, v)`` is the drop-in for the reference's
``F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)``
(``ttt/models/cogvideo/dit.py:196-198``) on ``[B, NH, S, 64]`` tensors.  bf16 tensors on a HIP device go to the
HIP kernels (strided views are consumed as they are; the output is laid out ``[B, S, NH, 64]`` in memory so the
following ``transpose(1, 2).reshape(B, S, NH*64)`` of the block is free); there is no silent fallback on a GPU.  CPU
tensors (host-side tests of the module logic) use PyTorch's SDPA.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ttt_amd.infra.remat_cache import kernel_result


def _ext():
    import test_time_training
    return test_time_training


class SegmentAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        ext = _ext()
        B, NH, S, D = q.shape
        scale = 1.0 / math.sqrt(D)
        out = torch.empty(B, S, NH, D, device=q.device, dtype=q.dtype).transpose(1, 2)     # [B,NH,S,D] view of [B,S,NH,D]
        lse = torch.empty(B, NH, S, device=q.device, dtype=torch.float32)
        ext.attn_forward(q, k, v, out, lse, scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        ext = _ext()
        q, k, v, out, lse = ctx.saved_tensors
        B, NH, S, D = q.shape
        if dout.stride(3) != 1:
            dout = dout.contiguous()
        mk = lambda: torch.empty(B, S, NH, D, device=q.device, dtype=q.dtype).transpose(1, 2)
        dq, dk, dv = mk(), mk(), mk()
        delta = torch.empty(B, NH, S, device=q.device, dtype=torch.float32)
        ext.attn_backward(q, k, v, out, dout, lse, delta, dq, dk, dv, ctx.scale)
        return dq, dk, dv


def segment_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Non-causal self-attention over one 3-second segment, [B, NH, S, D] (reference dit.py:196-198)."""
    if q.is_cuda:
        if q.dtype != torch.bfloat16 or q.shape[-1] != 64:
            raise RuntimeError("segment_attention: the HIP kernels need bf16 activations and head_dim 64 "
                               f"(got {q.dtype}, head_dim {q.shape[-1]}); no fallback on a GPU")
        fix = lambda t: t if t.stride(3) == 1 else t.contiguous()
        return SegmentAttention.apply(fix(q), fix(k), fix(v))
    return F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)


class AttnPre(torch.autograd.Function):
    """(q_raw, k_raw [B,S,NH*64] bf16, q_norm weight/bias, k_norm weight/bias [64], cos, sin [n_pos,64] fp32, n_text, eps)
    -> q, k as [B, NH, S, 64] views of [B, S, NH, 64] buffers: per-head LayerNorm + RoPE on tokens >= n_text, one HIP
    pass per direction (``csrc/attn_pre.hip``) instead of the LayerNorm / slice / rotate / cat chain of the unfused path."""

    @staticmethod
    def forward(ctx, q_raw, k_raw, wq, bq, wk, bk, cos, sin, NH, n_text, eps):
        ext = _ext()
        B, S, D = q_raw.shape
        qr, kr = q_raw.contiguous(), k_raw.contiguous()
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        wq32, bq32, wk32, bk32 = f32(wq), f32(bq), f32(wk), f32(bk)
        q, k = torch.empty_like(qr), torch.empty_like(kr)
        ext.attn_pre_forward(qr, kr, wq32, bq32, wk32, bk32, cos, sin, q, k, NH, n_text, float(eps))
        ctx.save_for_backward(qr, kr, wq32, wk32, cos, sin)
        ctx.meta = (NH, n_text, float(eps), wq.dtype)
        view = lambda t: t.view(B, S, NH, D // NH).transpose(1, 2)
        return view(q), view(k)

    @staticmethod
    def backward(ctx, dq, dk):
        ext = _ext()
        qr, kr, wq32, wk32, cos, sin = ctx.saved_tensors
        NH, n_text, eps, pdt = ctx.meta
        B, S, D = qr.shape
        fix = lambda t: t if t.stride(3) == 1 else t.contiguous()
        dq, dk = fix(dq), fix(dk)
        dq_raw, dk_raw = torch.empty_like(qr), torch.empty_like(kr)
        P = ext.attn_pre_partials(B, S, NH)
        part = torch.empty(P, 4, 64, device=qr.device, dtype=torch.float32)
        ext.attn_pre_backward(qr, kr, dq, dk, wq32, wk32, cos, sin, dq_raw, dk_raw, part, NH, n_text, eps)
        g = part.sum(0).to(pdt)
        return dq_raw, dk_raw, g[0], g[1], g[2], g[3], None, None, None, None, None


def attn_pre_available(x: torch.Tensor, head_dim: int) -> bool:
    return x.is_cuda and x.dtype == torch.bfloat16 and head_dim == 64


class FusedSegmentAttention(torch.autograd.Function):
    """``AttnPre`` + ``SegmentAttention`` as ONE autograd node that keeps only the raw q / k projections (plus v, the
    output and the log-sum-exp): the normalised / rotated q, k are re-derived in the backward (one ~0.2 ms pass) instead
    of living from forward to backward - activation memory bounds ``remat_free_layers`` on the 288-GB MI355X.

    (q_raw, k_raw [B,S,NH*64], v [B,NH,S,64] view, q_norm w/b, k_norm w/b [64], cos, sin, NH, n_text, eps) -> out [B,NH,S,64]
    (a view of a [B,S,NH,64] buffer)."""

    @staticmethod
    def forward(ctx, q_raw, k_raw, v, wq, bq, wk, bk, cos, sin, NH, n_text, eps):
        ext = _ext()
        B, S, D = q_raw.shape
        Dh = D // NH
        qr, kr = q_raw.contiguous(), k_raw.contiguous()
        v = v if v.stride(3) == 1 else v.contiguous()
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        p32 = (f32(wq), f32(bq), f32(wk), f32(bk))
        scale = 1.0 / math.sqrt(Dh)

        def run():
            q, k = torch.empty_like(qr), torch.empty_like(kr)
            ext.attn_pre_forward(qr, kr, *p32, cos, sin, q, k, NH, n_text, float(eps))
            view = lambda t: t.view(B, S, NH, Dh).transpose(1, 2)
            out = torch.empty(B, S, NH, Dh, device=qr.device, dtype=qr.dtype).transpose(1, 2)
            lse = torch.empty(B, NH, S, device=qr.device, dtype=torch.float32)
            ext.attn_forward(view(q), view(k), v, out, lse, scale)
            return out, lse

        # (inside a checkpointed region that keeps "attn" the recomputation gets the remembered output back: no kernel runs)
        out, lse = kernel_result("attn", run)
        ctx.save_for_backward(qr, kr, v, out, lse, *p32, cos, sin)
        ctx.meta = (NH, n_text, float(eps), scale, wq.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        ext = _ext()
        qr, kr, v, out, lse, wq32, bq32, wk32, bk32, cos, sin = ctx.saved_tensors
        NH, n_text, eps, scale, pdt = ctx.meta
        B, S, D = qr.shape
        Dh = D // NH
        q, k = torch.empty_like(qr), torch.empty_like(kr)
        ext.attn_pre_forward(qr, kr, wq32, bq32, wk32, bk32, cos, sin, q, k, NH, n_text, eps)      # re-derive q, k
        view = lambda t: t.view(B, S, NH, Dh).transpose(1, 2)
        if dout.stride(3) != 1:
            dout = dout.contiguous()
        mk = lambda: torch.empty(B, S, NH, Dh, device=qr.device, dtype=qr.dtype).transpose(1, 2)
        dq, dk = mk(), mk()
        # d q_raw, d k_raw and dV as the column blocks of ONE [B, S, 3 D] buffer: the q / k / v projections' weight gradients become one
        # GEMM over the concatenated output gradient (Linear3.backward); dV is written there directly through its strides
        from ttt_amd.models.ssm.fused import qkv_grad_blocks
        dq_raw, dk_raw, dv_blk = qkv_grad_blocks(qr)
        dv = dv_blk.view(B, S, NH, Dh).transpose(1, 2)                      # [B, NH, S, 64] view, token stride 3 D
        delta = torch.empty(B, NH, S, device=qr.device, dtype=torch.float32)
        ext.attn_backward(view(q), view(k), v, out, dout, lse, delta, dq, dk, dv, scale)
        del q, k
        P = ext.attn_pre_partials(B, S, NH)
        part = torch.empty(P, 4, 64, device=qr.device, dtype=torch.float32)
        ext.attn_pre_backward(qr, kr, dq, dk, wq32, wk32, cos, sin, dq_raw, dk_raw, part, NH, n_text, eps, ld_out=dq_raw.stride(1))
        g = part.sum(0).to(pdt)
        return dq_raw, dk_raw, dv, g[0], g[1], g[2], g[3], None, None, None, None, None
