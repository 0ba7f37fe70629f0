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
are carried through the token-wise ops and dropped at every gather.  Inference only (no autograd through the
collectives); weights are replicated (14.5 GB in bf16).  ``nccl`` = RCCL on ROCm; the CPU tests run it over ``gloo``.
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
        """[B, L, ...] (identical on every rank) -> this rank's [B, ceil(L/T), ...] shard (zero rows pad the last one)."""
        n = self.shard_len(x.shape[1])
        return self._pad(x)[:, self.rank * n:(self.rank + 1) * n].contiguous()

    # ---- collectives ---------------------------------------------------------------------------------------------------
    def gather_tokens(self, x_loc: torch.Tensor, length: int) -> torch.Tensor:
        """token shards [B, n, D] -> the full [B, length, D] on every rank."""
        parts = [torch.empty_like(x_loc) for _ in range(self.size)]
        dist.all_gather(parts, x_loc.contiguous(), group=self.group)
        return torch.cat(parts, dim=1)[:, :length]

    def heads_to_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """head shard over the full sequence [B, L, (NH/T)*F] -> token shard with every head [B, ceil(L/T), NH*F]."""
        B, L, d = x.shape
        n = self.shard_len(L)
        send = self._pad(x).view(B, self.size, n, d).transpose(0, 1).contiguous()          # [T, B, n, d]: chunk j -> rank j
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)                                # recv[r] = rank r's heads, my tokens
        return recv.permute(1, 2, 0, 3).reshape(B, n, self.size * d)                        # heads in rank order = global order
