"""Selective re-materialisation: inside a checkpointed region the RESULTS of the expensive sequence kernels are kept.

The reference checkpoints every transformer layer (``configs/train/ttt-mlp/*.toml: remat_transformer_layer_group_size = 1``,
``ttt/models/cogvideo/dit.py:493-499``): in backward the layer's whole forward runs again - projections, attention, both TTT
scans, MLP.  On a 288-GB MI355X the memory that decides how many layers must do so is better spent per byte: a layer's
local-attention outputs (+ log-sum-exp) are 0.34 GB at the 9 s geometry and save 12.8 ms of the 41 ms a re-materialised layer
costs (37 ms per GB), its two TTT scan outputs with their state checkpoints 1.3 GB for 12.4 ms (9.7 ms per GB) - against 5 ms per
GB for keeping a whole layer's activations (``remat_free_layers``).  So a checkpointed region may KEEP these kernel outputs:
the autograd nodes around the kernels (``FusedSegmentAttention``, ``FusedPreScanMLP``) ask ``kernel_result(kind, compute)``;
in the region's forward pass the outputs are remembered, in its recomputation they are handed back instead of launching the
kernel again.  Same tensors, same bits; the cheap elementwise / projection work around them is still re-materialised, so the
node's own saved inputs are rebuilt as before.

``DiffusionTransformer.remat_keep`` selects the kinds (``("attn", "scan")`` by default in bench.py; ``()`` = the reference's
behaviour)."""
from __future__ import annotations

import threading

_state = threading.local()


class _Region:
    """The kept kernel outputs of ONE ``checkpoint`` call, in call order.  Entries stay until the region is freed with its
    checkpoint frame (they share storage with what the recomputed graph saves anyway), so a second recomputation of the same
    region - ``retain_graph=True``, or a double backward through it - finds them again instead of an empty queue."""
    __slots__ = ("kinds", "entries", "cursor", "park")

    def __init__(self, kinds, park=None):
        self.kinds, self.entries, self.cursor, self.park = frozenset(kinds), [], 0, park


class _Scope:
    """Re-usable context manager (``torch.utils.checkpoint`` may enter the contexts it was handed more than once)."""

    def __init__(self, region, mode):
        self.region, self.mode, self._prev = region, mode, []

    def __enter__(self):
        self._prev.append(getattr(_state, "cur", None))
        _state.cur = None if self.region is None else (self.region, self.mode)
        if self.region is not None and self.mode == "recompute":
            self.region.cursor = 0                 # every recomputation walks the region's calls from the start
        return self

    def __exit__(self, *exc):
        _state.cur = self._prev.pop()
        return False


def context_fn(kinds, park=None):
    """``context_fn`` for ``torch.utils.checkpoint.checkpoint(..., use_reentrant=False)``: one region per checkpoint call.
    ``park = (HostOffload, layer index)``: the kept outputs wait in pinned host memory (``ttt_amd/infra/host_offload.py``: copied out
    behind the kernel, fetched back when the backward approaches the layer) instead of in HBM - the same bits either way."""
    region = _Region(kinds, park)
    return lambda: (_Scope(region, "forward"), _Scope(region, "recompute"))


def suspended():
    """Context in which nothing is kept (nested checkpoints recompute in an order of their own)."""
    return _Scope(None, None)


def replaying(kind: str) -> bool:
    """True inside the RECOMPUTATION of a region that kept ``kind``: ``kernel_result(kind, ...)`` will hand the kept tensors back."""
    cur = getattr(_state, "cur", None)
    return cur is not None and cur[1] == "recompute" and kind in cur[0].kinds


def kernel_result(kind: str, compute):
    """``compute() -> tuple of tensors`` (the kernel's outputs).  Outside a keeping region: just ``compute()``.  In the forward
    pass of a region that keeps ``kind``: compute and remember.  In its recomputation: hand the remembered tuple back (the
    calls of a region recur in the same order; the kind is checked, and so is that nobody wrote into a kept tensor in between:
    the kept copies alias the forward pass's outputs)."""
    cur = getattr(_state, "cur", None)
    if cur is None or kind not in cur[0].kinds:
        return compute()
    region, mode = cur
    # (detached aliases on both sides: the tensor object a Function returns gets that call's autograd identity stamped on it,
    #  and the recomputation's node must not be handed an object that already carries the forward pass's)
    if mode == "forward":
        out = compute()
        kept = tuple(t.detach() for t in out)
        versions = tuple(t._version for t in kept)
        if region.park is not None:
            kept = region.park[0].park(region.park[1], kept)       # (handles; small tensors and views stay as they are)
        region.entries.append((kind, kept, versions))
        return out
    if region.cursor >= len(region.entries):
        raise RuntimeError(f"remat_cache: the recomputation asks for a {kind!r} result the forward pass of this region did not produce "
                           f"({len(region.entries)} kept): the region's calls must recur in the same order")
    got, out, versions = region.entries[region.cursor]
    region.cursor += 1
    if got != kind:
        raise RuntimeError(f"remat_cache: recomputation asked for {kind!r} where the forward pass produced {got!r}")
    if region.park is not None:
        out = region.park[0].unpark(out)
    elif any(t._version != v for t, v in zip(out, versions)):
        raise RuntimeError(f"remat_cache: a kept {kind!r} output was modified in place after the forward pass (the kept copy aliases it); "
                           "clone before writing into a kernel output inside a region that keeps it")
    return tuple(t.detach() for t in out)
