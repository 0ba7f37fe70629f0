"""Projection helpers of the DiT block (host side, PyTorch-ROCm; the GEMMs go to hipBLASLt).

``Linear3``: the q / k / v projections of the attention block and the wq / wk / wv projections of the TTT layer read the SAME
input; as one autograd node their input gradient is one GEMM accumulation chain instead of three GEMMs and two full-size additions.

(Round 1 also carried a side-stream queue that ran the weight-gradient GEMMs beside the backward scans, opt-in until timed.  Timed
in round 2 on an MI355X - 6 612 vs 6 619 video-tok/s at the 3 s configuration, the backward scan 6.94 vs 6.81 ms,
profiles/r2h_bench_3s_overlap_wgrad.json: the cluster sweep leaves 64 CUs idle, not 208 - it lost and was removed.)
"""
from __future__ import annotations

import os

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
    three times).  Round 5: when the three output gradients arrive as the column blocks of one buffer (the consumers' backward kernels
    write them that way), the input gradient is ONE GEMM with the contraction over all three (see ``backward``)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2):
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)             # an unused output arrives as None, not as a tensor of zeros
        ctx.ws, ctx.bs = (w0, w1, w2), (b0, b1, b2)
        from ttt_amd.models.ssm.pipeline import injected
        pre = injected("linear3")            # a pipelined TTT forward has formed the three projections part by part already
        if pre is not None:
            return pre
        return F.linear(x, w0, b0), F.linear(x, w1, b1), F.linear(x, w2, b2)

    @staticmethod
    def backward(ctx, *dys):
        (x,) = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1])
        cat = _column_blocks(dys) if FUSE_QKV_BACKWARD else None
        if cat is not None:
            # the three output gradients are the column blocks of ONE [rows, n0 + n1 + n2] buffer (the backward kernels of the
            # consumers write them that way, ttt_amd/models/ssm/fused.py: qkv_grad_blocks): the input gradient is ONE GEMM with the
            # contraction over all three (measured at the 5B / 9 s shapes, profiles/r5b_*: 1.83 ms against 0.79 + 2 x 0.85 for the
            # accumulation chain at 51 456 rows, 0.62 against 0.82 at 18 052).  The weight gradients stay one GEMM per projection
            # over the strided blocks: the concatenated 9216 x 3072 x L product is no faster than three of 3072 x 3072 x L
            # (2.85 against 3 x 0.90 ms - that shape is bound by its K-major operands, not by tile quantisation); "both" keeps
            # the one-GEMM form selectable.
            ns = [w.shape[0] for w in ctx.ws]
            dx = cat.mm(_stacked_weights(ctx.ws)).view(x.shape) if ctx.needs_input_grad[0] else None
            blocks = cat.split(ns, dim=1)
            if FUSE_QKV_BACKWARD == "both":
                gwc = cat.t().mm(x2) if any(w.requires_grad for w in ctx.ws) else None
                gw = [None if gwc is None or not w.requires_grad else g for w, g in zip(ctx.ws, gwc.split(ns, dim=0) if gwc is not None else [None] * 3)]
            else:
                gw = [blk.t().mm(x2) if w.requires_grad else None for w, blk in zip(ctx.ws, blocks)]
            gbc = cat.sum(0) if any(b is not None and b.requires_grad for b in ctx.bs) else None
            gb = [None if gbc is None or b is None or not b.requires_grad else g for b, g in zip(ctx.bs, gbc.split(ns) if gbc is not None else [None] * 3)]
            return dx, gw[0], gb[0], gw[1], gb[1], gw[2], gb[2]
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


def _fuse_mode(value: str) -> str:
    """TTT_FUSE_QKV_BACKWARD: "dgrad" (default) / "1": one input-gradient GEMM over the concatenation, one weight-gradient GEMM per
    projection; "both": the weight gradients as one GEMM too; "" / "0": three GEMMs each (A/B switch: tools/qkv_backward_bench.py).
    Anything else is an error (a typo such as "off" used to select the fusion silently)."""
    modes = {"": "", "0": "", "1": "dgrad", "dgrad": "dgrad", "both": "both"}
    if value not in modes:
        raise ValueError(f"TTT_FUSE_QKV_BACKWARD={value!r}: expected one of {sorted(modes)}")
    return modes[value]


FUSE_QKV_BACKWARD = _fuse_mode(os.environ.get("TTT_FUSE_QKV_BACKWARD", "dgrad"))

# The [n0 + n1 + n2, K] stack of a group's three weights for the one-GEMM input gradient.  The two scan directions of a TTT layer
# share wq / wk / wv and their backward nodes run back to back, so the stack of the LAST group is kept (one entry: 56 MB at D = 3072;
# keeping every group's would be 4.7 GB of the 5B model) and re-used while the same three tensors are unchanged (`_version` moves
# with every in-place update, e.g. the publish of new parameters after the optimizer step).
_stack_cache = {"ws": None, "key": None, "value": None}


def _stacked_weights(ws):
    # identity of the three tensor OBJECTS (held here, so that their ids cannot be re-used by other tensors) + storage + version
    key = tuple((w.data_ptr(), w._version, tuple(w.shape), w.dtype) for w in ws)
    held = _stack_cache["ws"]
    if held is None or len(held) != len(ws) or any(a is not b for a, b in zip(held, ws)) or _stack_cache["key"] != key:
        _stack_cache["value"] = None                       # (free the old stack before the new one is allocated)
        _stack_cache["value"] = torch.cat([w.detach() for w in ws], dim=0)
        _stack_cache["ws"], _stack_cache["key"] = tuple(ws), key
    return _stack_cache["value"]



def _column_blocks(dys):
    """The 2-D ``[rows, n0 + n1 + n2]`` view that the three gradients are consecutive column blocks of, or None."""
    if any(d is None for d in dys):
        return None
    d0 = dys[0]
    if d0.dim() < 2 or any(d.dim() != d0.dim() or d.shape[:-1] != d0.shape[:-1] or d.dtype != d0.dtype or d.stride(-1) != 1 for d in dys):
        return None
    ld = sum(d.shape[-1] for d in dys)
    base = d0.untyped_storage().data_ptr()
    off = d0.storage_offset()
    rows = 1
    for n in d0.shape[:-1]:
        rows *= n
    want = tuple(ld * (rows // _prod(d0.shape[:k + 1])) for k in range(d0.dim() - 1)) + (1,)       # a contiguous [..., ld] buffer's strides
    for d in dys:
        if d.untyped_storage().data_ptr() != base or d.storage_offset() != off or tuple(d.stride()) != want:
            return None
        off += d.shape[-1]
    return torch.as_strided(d0, (rows, ld), (ld, 1), d0.storage_offset())


def _prod(xs):
    r = 1
    for v in xs:
        r *= v
    return r


def linear3_applies(m0, m1, m2, x: torch.Tensor) -> bool:
    """``Linear3`` is an exact substitute for the three module calls: plain-tensor parameters of x's dtype, no autocast"""
    ok = not torch.is_autocast_enabled(x.device.type) and x.dim() >= 2
    for m in (m0, m1, m2):
        ok = ok and _plain(m.weight) and _plain(m.bias) and m.weight.dtype == x.dtype and (m.bias is None or m.bias.dtype == x.dtype)
    return ok


def linear3(m0: torch.nn.Linear, m1: torch.nn.Linear, m2: torch.nn.Linear, x: torch.Tensor, always: bool = False):
    """``(m0(x), m1(x), m2(x))`` through ``Linear3`` when that is an exact substitute (``linear3_applies``) and gradients are enabled;
    the three module calls otherwise.  ``always``: through the node whenever it applies (a pipelined TTT forward hands it the
    projections it has formed already - also under ``no_grad``)."""
    ok = linear3_applies(m0, m1, m2, x)
    if ok and (always or (torch.is_grad_enabled() and (x.requires_grad or any(m.weight.requires_grad for m in (m0, m1, m2))))):
        return Linear3.apply(x, m0.weight, m0.bias, m1.weight, m1.bias, m2.weight, m2.bias)
    return m0(x), m1(x), m2(x)
