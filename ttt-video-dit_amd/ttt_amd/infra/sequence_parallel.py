"""Sequence-parallel inference across the GPUs of a node (BASELINE config 4: one long video sampled on 8
GPUs).  The reference reaches long contexts with DTensor tensor parallelism - head-sharded q/k/v and TTT op, sequence-
parallel norms and MLP (``ttt/infra/parallelisms.py``:106-152, ``mlp_tk.py``:297-343 ``local_map``) - because a 63 s sequence
does not fit an 80-GB GPU.  On MI355X it fits (81 GiB measured for the batched guidance pair), so the same two layouts are
used here for a different reason, latency, and with explicit RCCL collectives instead of DTensor redistribution:

  * token-wise work (AdaLN, q/k/v/o and wq/wk/wv/wo projections' output side, post-norm, gates, MLP) runs on a rank's
    TOKEN SHARD: 1/T of the GEMM and elementwise time;
  * sequence-mixing work (local attention, the TTT scan with its RoPE / interleave / time reversal) runs on a rank's HEAD
    SHARD over the full sequence: NH/T heads.  The scan is latency-bound (one workgroup per head), so its time does not
    shrink - it bounds the speed-up.

Per sequence-mixing op: one all-gather of the token shards (its input is needed for every head) and one all-to-all back
(heads -> tokens).  Token shards are padded to equal size (63 s: 341 550 video tokens are not a multiple of 8); pad rows
are carried through the token-wise ops and dropped at every gather.  Weights are replicated (14.5 GB in bf16 for
sampling).  The collectives are autograd functions (all-gather <-> reduce-scatter, all-to-all <-> all-to-all), so the same
layout also trains: every rank then holds partial parameter gradients (its tokens' share of the token-wise parameters,
its heads' slices of the per-head ones) and ``sum_gradients`` adds them up over the group.  ``nccl`` = RCCL on ROCm; the
CPU tests run it over ``gloo``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class SeqParallel:
    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    # ---- layouts -------------------------------------------------------------------------------------------------------
    def head_range(self, num_heads: int) -> Tuple[int, int]:
        if num_heads % self.size:
            raise ValueError(f"{num_heads} heads cannot be split over {self.size} ranks")
        n = num_heads // self.size
        return self.rank * n, (self.rank + 1) * n

    def shard_len(self, length: int) -> int:
        return -(-length // self.size)

    def _pad(self, x: torch.Tensor, dim: int = 1) -> torch.Tensor:
        n = self.shard_len(x.shape[dim]) * self.size - x.shape[dim]
        if n == 0:
            return x
        shape = list(x.shape)
        shape[dim] = n
        return torch.cat((x, x.new_zeros(shape)), dim=dim)

    def shard_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """[B, L, ...] (identical on every rank) -> this rank's [B, ceil(L/T), ...] shard (zero rows pad the last one).
        A slice: under autograd the gradient lands in this rank's rows of ``x`` (see ``sum_gradients``)."""
        n = self.shard_len(x.shape[1])
        return self._pad(x)[:, self.rank * n:(self.rank + 1) * n].contiguous()

    # ---- collectives (raw) -----------------------------------------------------------------------------------------------
    def _all_gather(self, x_loc: torch.Tensor) -> torch.Tensor:          # [B, n, D] -> [B, T*n, D]
        x_loc = x_loc.contiguous()
        parts = [torch.empty_like(x_loc) for _ in range(self.size)]
        dist.all_gather(parts, x_loc, group=self.group)
        return torch.cat(parts, dim=1)

    def _reduce_scatter(self, g: torch.Tensor) -> torch.Tensor:          # [B, T*n, D] (differs per rank) -> sum, my [B, n, D]
        n = g.shape[1] // self.size
        if dist.get_backend(self.group) == "gloo":          # no reduce_scatter in gloo (CPU tests): all-reduce and slice
            g = g.clone()
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            return g[:, self.rank * n:(self.rank + 1) * n].contiguous()
        parts = [p.contiguous() for p in g.split(n, dim=1)]
        out = torch.empty_like(parts[0])
        dist.reduce_scatter(out, parts, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def _a2a_heads_to_tokens(self, x: torch.Tensor) -> torch.Tensor:     # [B, T*n, d] -> [B, n, T*d]
        B, Lp, d = x.shape
        n = Lp // self.size
        send = x.reshape(B, self.size, n, d).transpose(0, 1).contiguous()                    # [T, B, n, d]: chunk j -> rank j
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)                                # recv[r] = rank r's heads, my tokens
        return recv.permute(1, 2, 0, 3).reshape(B, n, self.size * d)                        # heads in rank order = global order

    def _a2a_tokens_to_heads(self, y: torch.Tensor) -> torch.Tensor:     # [B, n, T*d] -> [B, T*n, d]   (inverse of the above)
        B, n, D = y.shape
        d = D // self.size
        send = y.reshape(B, n, self.size, d).permute(2, 0, 1, 3).contiguous()               # [T, B, n, d]: head block r -> rank r
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)                                # recv[j] = token chunk j, my heads
        return recv.transpose(0, 1).reshape(B, self.size * n, d)

    # ---- collectives (differentiable) -------------------------------------------------------------------------------------
    def gather_tokens(self, x_loc: torch.Tensor, length: int, replicated_consumer: bool = False) -> torch.Tensor:
        """token shards [B, n, D] -> the full [B, length, D] on every rank.  Backward: the consumers on different ranks (head
        shards) each produce a gradient for the full tensor -> reduce-scatter; with ``replicated_consumer=True`` every rank
        runs the SAME computation on the result (the final layer / loss), so its own rows of its own gradient are the
        answer."""
        return _Gather.apply(x_loc, self, length, replicated_consumer)

    def heads_to_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """head shard over the full sequence [B, L, (NH/T)*F] -> token shard with every head [B, ceil(L/T), NH*F]."""
        return _HeadsToTokens.apply(x, self)

    def sum_gradients(self, module: torch.nn.Module, bucket_bytes: int = 512 << 20) -> None:
        """After backward: add up the partial parameter gradients of the ranks, in flat buckets of ``bucket_bytes`` (few,
        large all-reduces: the xGMI links are point-to-point, a ring all-reduce is per-link bound)."""
        bucket, size = [], 0

        def flush():
            nonlocal bucket, size
            if not bucket:
                return
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            o = 0
            for g in bucket:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()
            bucket, size = [], 0

        by_dtype = {}
        for p in module.parameters():
            if p.grad is not None:
                by_dtype.setdefault(p.grad.dtype, []).append(p.grad)
        for grads in by_dtype.values():
            for g in grads:
                bucket.append(g)
                size += g.numel() * g.element_size()
                if size >= bucket_bytes:
                    flush()
            flush()


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_loc, sp, length, replicated_consumer):
        ctx.sp, ctx.n, ctx.length, ctx.rep = sp, x_loc.shape[1], length, replicated_consumer
        return sp._all_gather(x_loc)[:, :length]

    @staticmethod
    def backward(ctx, g):
        sp, n = ctx.sp, ctx.n
        pad = sp.size * n - ctx.length
        if pad:
            g = torch.cat((g, g.new_zeros(g.shape[0], pad, *g.shape[2:])), dim=1)
        if ctx.rep:
            return g[:, sp.rank * n:(sp.rank + 1) * n].contiguous(), None, None, None
        return sp._reduce_scatter(g.contiguous()), None, None, None


class _HeadsToTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sp):
        ctx.sp, ctx.length = sp, x.shape[1]
        return sp._a2a_heads_to_tokens(sp._pad(x))

    @staticmethod
    def backward(ctx, g):
        return ctx.sp._a2a_tokens_to_heads(g.contiguous())[:, :ctx.length], None


class _GatherFeatures(torch.autograd.Function):
    """head shards [B, L, d] -> [B, L, T*d] on every rank (feature blocks in rank order = head order).  The consumer is
    replicated (every rank applies the same post-norm / output projection to the same gathered tensor), so the gradient of a
    rank's block is that block of its own incoming gradient: no collective in backward."""

    @staticmethod
    def forward(ctx, x, sp):
        ctx.sp, ctx.d = sp, x.shape[-1]
        parts = [torch.empty_like(x) for _ in range(sp.size)]
        dist.all_gather(parts, x.contiguous(), group=sp.group)
        return torch.cat(parts, dim=-1)

    @staticmethod
    def backward(ctx, g):
        sp, d = ctx.sp, ctx.d
        return g[..., sp.rank * d:(sp.rank + 1) * d].contiguous(), None


def gather_features(x: torch.Tensor, sp: SeqParallel) -> torch.Tensor:
    return _GatherFeatures.apply(x, sp)


class _ReplicatedInput(torch.autograd.Function):
    """Identity on a tensor that is replicated over the group and consumed by rank-local (head-sharded) work: each rank's
    backward yields only its heads' share of the input gradient, so the shares are summed (all-reduce) - Megatron's "f"."""

    @staticmethod
    def forward(ctx, x, sp):
        ctx.sp = sp
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.sp.group)
        return g, None


def replicated_input(x: torch.Tensor, sp: SeqParallel) -> torch.Tensor:
    return _ReplicatedInput.apply(x, sp)
