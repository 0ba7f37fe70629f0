"""Weight-gradient GEMMs on a side HIP stream, under the TTT scans.

Why (MI355X-first; measured in ``profiles/r1f_bench_kernel_stats.csv``): the backward scans of a layer hold 48 (+ 96
light prefetch-helper) of the 256 CUs for 2 x 8.6 ms while the rest of the chip idles, and the backward of every dense
projection contains a GEMM nobody waits for - ``dW = dY^T X`` feeds only the optimizer.  PyTorch's autograd enqueues
it on the compute stream between the critical-path kernels; here it goes to a second stream, so the hardware runs it
beside whichever kernel the compute stream has in flight.  Backward order inside one ``TransformerLayer`` (reference
``cogvideo/dit.py:321-382`` read bottom-up): MLP, reverse-direction TTT (``wo``, scan, ``wq/wk/wv``), forward-direction
TTT, attention.  The MLP's two weight gradients (8 of the layer's 20 ``L x D x D`` GEMM units) and ``wo``'s are
submitted before the first backward scan, the reverse direction's ``wq/wk/wv`` before the second.

Contract
  * ``linear(mod, x)`` is ``mod(x)``; with the queue enabled its backward returns ``dX`` (computed on the current stream)
    and NO gradient for ``mod.weight`` / ``mod.bias`` to autograd: those are accumulated into per-parameter buffers by
    work enqueued on the side stream (all deferred work is ordered on that one stream, so a weight used twice per layer
    - the TTT projections serve both scan directions - accumulates race-free, through the GEMM's ``beta = 1`` epilogue).
  * ``join()`` makes the current stream wait for the side stream and publishes the buffers as ``param.grad``
    (``+=`` if a gradient is already there).  ``JoinWgrad.apply(*layer_inputs)`` does that in backward when the
    gradients of a layer's inputs are produced, i.e. after every deferred submission of the layer and before FSDP2's
    post-backward hook reads ``unsharded_param.grad`` for the reduce-scatter (its ``RegisterPostBackwardFunction`` sits
    on the layer inputs OUTSIDE the module's forward, so it runs after this node).
  * memory: everything is allocated on the current (compute) stream's pool; tensors the side stream reads are kept
    referenced until their event has completed (polled at each submission) or until ``join()``.
  * without a GPU (CPU tensors) the deferred work runs inline, same arithmetic - that is what the CPU tests cover; the
    stream choreography itself can only be exercised on the device (``tests/test_kernels_gpu.py::test_wgrad_overlap_*``).

Opt-in (``enable(True)`` / ``bench.py --overlap-wgrad``) until it has been timed on an MI355X.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import torch
import torch.nn.functional as F


class _Queue:
    def __init__(self):
        self.enabled = False
        self.armed = False            # inside a layer whose input gradients will trigger the join (see layer_entry)
        self._cb_queued = False
        self._side: Dict[int, torch.cuda.Stream] = {}
        self._pending: List[Tuple[object, tuple]] = []          # (event, tensors the side stream still reads)
        self._grads: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}   # id(param) -> (param, buffer)
        self.stats = {"submitted": 0, "joined": 0}

    # -- stream plumbing ------------------------------------------------------------------------------------------------
    def _stream(self, device: torch.device) -> torch.cuda.Stream:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        s = self._side.get(idx)
        if s is None:
            s = self._side[idx] = torch.cuda.Stream(device=idx)
        return s

    def buffer(self, param: torch.Tensor) -> Tuple[torch.Tensor, bool]:
        """Accumulation buffer of ``param`` (allocated on the current stream) and whether it is fresh (unwritten)."""
        hit = self._grads.get(id(param))
        if hit is not None:
            return hit[1], False
        buf = torch.empty_like(param, memory_format=torch.contiguous_format)
        self._grads[id(param)] = (param, buf)
        return buf, True

    def submit(self, work: Callable[[], None], keep: tuple, device: torch.device):
        self.stats["submitted"] += 1
        if not self._cb_queued:        # safety net: whatever no JoinWgrad node picked up is published when backward() ends
            self._cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        if device.type != "cuda":
            work()
            return
        main = torch.cuda.current_stream(device)
        side = self._stream(device)
        side.wait_stream(main)                     # operands (and freshly allocated buffers) are ready / owned
        with torch.cuda.stream(side):
            work()
            ev = torch.cuda.Event()
            ev.record(side)
        self._pending = [(e, k) for e, k in self._pending if not e.query()]
        self._pending.append((ev, keep))

    def _end_of_backward(self):
        self._cb_queued = False
        self.join()

    def join(self):
        if not self._grads and not self._pending:
            return
        self.stats["joined"] += 1
        for idx, side in self._side.items():       # unconditionally: the buffers below were written on the side stream
            torch.cuda.current_stream(idx).wait_stream(side)
        self._pending.clear()
        grads, self._grads = self._grads, {}
        for param, buf in grads.values():
            if param.grad is None:
                param.grad = buf
            else:
                param.grad += buf


    def reset(self):
        """Drop whatever an aborted backward left behind (an exception inside ``backward()`` skips the end-of-backward
        callback: ``_cb_queued`` would stay set and the half-accumulated buffers would leak into the next step's gradients)."""
        for idx, side in self._side.items():
            torch.cuda.current_stream(idx).wait_stream(side)
        self._pending.clear()
        self._grads.clear()
        self._cb_queued = False
        self.armed = False


_Q = _Queue()


def enable(on: bool = True):
    """Switch the deferral on / off (process-wide).  Turning it off joins whatever is outstanding; turning it on starts
    from a clean queue."""
    if not on:
        _Q.join()
    elif not _Q.enabled:
        _Q.reset()
    _Q.enabled = bool(on)


def reset():
    """Call after a ``backward()`` that raised (e.g. an out-of-memory probe step) before stepping again."""
    _Q.reset()


def enabled() -> bool:
    return _Q.enabled


def join():
    _Q.join()


def stats() -> dict:
    return dict(_Q.stats)


def _plain(p) -> bool:
    return p is None or type(p) in (torch.Tensor, torch.nn.Parameter)


def _defer_weight_grads(weight, bias, dy2, x2_fn, keep, device):
    """Enqueue ``weight.grad += dy2^T @ x2_fn()`` and ``bias.grad += dy2.sum(0)`` on the side stream."""
    wbuf, wfresh = _Q.buffer(weight) if weight.requires_grad else (None, False)
    bbuf, bfresh = _Q.buffer(bias) if bias is not None and bias.requires_grad else (None, False)
    if wbuf is None and bbuf is None:
        return

    def work():
        if wbuf is not None:
            x2 = x2_fn()
            if wfresh:
                torch.mm(dy2.t(), x2, out=wbuf)
            else:
                wbuf.addmm_(dy2.t(), x2)
        if bbuf is not None:
            if bfresh:
                torch.sum(dy2, dim=0, out=bbuf)
            else:
                bbuf.add_(dy2.sum(0))

    _Q.submit(work, keep, device)


class OverlapLinear(torch.autograd.Function):
    """``F.linear(x, weight, bias)`` whose weight / bias gradients are produced on the side stream (see module docstring)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x)
        ctx.weight, ctx.bias = weight, bias          # the parameter OBJECTS: their .grad is set at join()
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dx = dy.matmul(weight) if ctx.needs_input_grad[0] else None
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        _defer_weight_grads(weight, bias, dy2, lambda: x2, (dy2, x2), dy.device)
        return dx, None, None


class OverlapGeluLinear(torch.autograd.Function):
    """``F.linear(gelu_tanh(z), weight, bias)`` keeping only ``z`` (as ``cogvideo/dit.py:GeluLinear``); the GELU
    re-evaluation that feeds the weight gradient moves to the side stream together with that GEMM."""

    @staticmethod
    def forward(ctx, z, weight, bias):
        ctx.save_for_backward(z)
        ctx.weight, ctx.bias = weight, bias
        return F.linear(F.gelu(z, approximate="tanh"), weight, bias)

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy2, z2 = dy.reshape(-1, dy.shape[-1]), z.reshape(-1, z.shape[-1])
        _defer_weight_grads(weight, bias, dy2, lambda: F.gelu(z2, approximate="tanh"), (dy2, z2), dy.device)
        dz = torch.ops.aten.gelu_backward(dy.matmul(weight), z, approximate="tanh")
        return dz, None, None


class Linear3(torch.autograd.Function):
    """Three projections of the SAME input (q / k / v of the attention block, wq / wk / wv of the TTT layer) as one autograd
    node: ``(F.linear(x, w0, b0), F.linear(x, w1, b1), F.linear(x, w2, b2))``.  Its backward forms the input gradient as ONE
    accumulation chain - ``dX = dY0 W0`` then two GEMMs with ``beta = 1`` into the same buffer - instead of three GEMM outputs
    and the two full-size additions autograd inserts for a tensor with three consumers (2 x 333 MB of traffic per group at
    the 3 s geometry, 6 such additions per layer; accumulated in the GEMM's fp32 epilogue, so also rounded once instead of
    three times).  Weight / bias gradients: deferred to the side stream when the queue is armed, inline otherwise."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2):
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)             # an unused output arrives as None, not as a tensor of zeros
        ctx.ws, ctx.bs = (w0, w1, w2), (b0, b1, b2)
        ctx.defer = _Q.enabled and _Q.armed
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
            if ctx.defer:
                _defer_weight_grads(w, b, dy2, lambda: x2, (dy2, x2), dy.device)
            else:
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


class JoinWgrad(torch.autograd.Function):
    """Identity on a layer's inputs; in backward (= when the layer's input gradients exist) it joins the side stream."""

    @staticmethod
    def forward(ctx, *xs):
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        _Q.join()
        return gs


def usable(mod: torch.nn.Linear, x: torch.Tensor) -> bool:
    return (_Q.enabled and _Q.armed and torch.is_grad_enabled() and mod.weight.requires_grad and _plain(mod.weight)
            and _plain(mod.bias) and x.dtype == mod.weight.dtype and not torch.is_autocast_enabled(x.device.type))


def linear(mod: torch.nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """``mod(x)``; with the queue enabled and a trainable plain-tensor weight, through ``OverlapLinear``."""
    if usable(mod, x):
        return OverlapLinear.apply(x, mod.weight, mod.bias)
    return mod(x)


def gelu_linear_usable(z: torch.Tensor, weight: torch.Tensor, bias) -> bool:
    return (_Q.enabled and _Q.armed and torch.is_grad_enabled() and weight.requires_grad and _plain(weight) and _plain(bias)
            and z.dtype == weight.dtype and not torch.is_autocast_enabled(z.device.type))


def gelu_linear(z: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    return OverlapGeluLinear.apply(z, weight, bias)


def layer_entry(*xs):
    """Call on a layer's inputs at the top of its forward; marks where the layer's deferred gradients are joined.
    A layer none of whose inputs requires a gradient has no such point (FSDP2 then reduces it from its end-of-backward
    callback, which runs before ours): its projections stay on the plain autograd path."""
    _Q.armed = bool(_Q.enabled and torch.is_grad_enabled() and any(x.requires_grad for x in xs))
    if _Q.armed:
        return JoinWgrad.apply(*xs)
    return xs
