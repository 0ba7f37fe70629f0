"""Sequence metadata, AdaLN modulation, timestep embedding, 3-D RoPE for the local attention and
DTensor helpers.  Behavioural mirror of the hot-path half of the reference's
``ttt/models/cogvideo/utils.py`` (lines 16-49, 70-75, 102-114, 155-208, 219-248, 363-437); the
sampling classes in the second half of that file are out of scope."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

try:  # DTensor is only needed for the (next-row) tensor-parallel mode
    from torch.distributed.tensor import DTensor, Replicate, Shard
except Exception:  # pragma: no cover
    DTensor = None


def get_interleave_offsets(num_frames: int, num_chunks: int, tokens_per_frame: int, text_length: int):
    """Token offsets of the [text_i, video_i] scenes in the interleaved order (reference :16-26).
    Scene 0 owns the remainder frame(s); every offset includes that scene's text tokens."""
    per_chunk = num_frames // num_chunks
    first = per_chunk + num_frames % per_chunk
    return per_chunk * tokens_per_frame + text_length, first * tokens_per_frame + text_length


def _is_dt(t) -> bool:
    return DTensor is not None and isinstance(t, DTensor)


def to_local(t):
    return t.to_local() if _is_dt(t) else t


def place_into(local, like):
    if not _is_dt(like):
        return local
    return DTensor.from_local(local, device_mesh=like.device_mesh, placements=like.placements,
                              shape=like.shape, stride=like.stride())


def full_tensor(t):
    return t.full_tensor() if _is_dt(t) else t


def replicate_tensor(t, tp_mesh):
    if tp_mesh is None:
        return t
    if not _is_dt(t):
        t = DTensor.from_local(t, tp_mesh, (Replicate(),), run_check=True)
    return t.redistribute(placements=(Replicate(),))


def shard_tensor(t, tp_mesh=None, dim=0):
    if tp_mesh is None:
        return t
    if not _is_dt(t):
        t = DTensor.from_local(t, tp_mesh, (Replicate(),), run_check=True)
    return t.redistribute(placements=(Shard(dim),))


def modulate(x, shift, scale):
    """AdaLN: x * (1 + scale) + shift with [B, D] shift/scale broadcast over tokens (reference :70-75)."""
    extra = x.ndim - shift.ndim
    if extra > 0:
        idx = (slice(None),) + (None,) * extra
        shift, scale = shift[idx], scale[idx]
    return torch.addcmul(shift, x, 1 + scale)


def timestep_embedding(timesteps, dim, max_period=10000, dtype=torch.float32):
    """[cos | sin] sinusoidal embedding of integer timesteps (reference :102-114)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    ang = timesteps[:, None].float() * freqs[None]
    emb = torch.cat((ang.cos(), ang.sin()), dim=-1)
    if dim % 2:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb.to(dtype)


def unpatchify(x, c, p, w, h):
    """[B, T*h*w, c*p*p] -> [B, T, c, h*p, w*p] (reference :155-171)."""
    b, n, _ = x.shape
    t = n // (h * w)
    x = x.view(b, t, h, w, c, p, p).permute(0, 1, 4, 2, 5, 3, 6)
    return x.reshape(b, t, c, h * p, w * p)


def cast_rotary_freqs(model, dtype):
    for m in model.modules():
        if isinstance(m, Rotary3DPositionEmbedding):
            m.freqs_cos.data = m.freqs_cos.data.to(dtype)
            m.freqs_sin.data = m.freqs_sin.data.to(dtype)


@dataclass
class SequenceMetadata:
    """Per-forward sequence geometry handed down the DiT (reference :219-248)."""
    text_length: int
    seq_text_length: int
    num_frames: int
    num_chunks: int
    tokens_per_frame: int
    latent_height: int
    latent_width: int
    t_emb: torch.Tensor
    base_offset: Optional[int] = None
    init_offset: Optional[int] = None

    @property
    def is_multiscene(self) -> bool:
        return self.num_chunks > 1

    def init_multiscene_offsets(self):
        self.base_offset, self.init_offset = get_interleave_offsets(
            self.num_frames, self.num_chunks, self.tokens_per_frame, self.text_length)


class Rotary3DPositionEmbedding(nn.Module):
    """Real-valued 3-D RoPE of the local attention (reference :363-437): rotate-half on adjacent
    pairs, positions restart at 0 for every attention segment (``freqs[:seq_len]``)."""

    def __init__(self, height, width, compressed_num_frames, head_dim, theta=10000):
        super().__init__()
        self.height, self.width, self.compressed_num_frames = height, width, compressed_num_frames
        self.head_dim, self.theta = head_dim, theta
        self.tp_mesh = None
        s, c = self._tables()
        self.register_buffer("freqs_sin", s, persistent=False)
        self.register_buffer("freqs_cos", c, persistent=False)

    def init_device_mesh(self, tp_mesh):
        self.tp_mesh = tp_mesh

    def _tables(self):
        d = self.head_dim
        dims = (d // 4, d // 8 * 3, d // 8 * 3)
        sizes = (self.compressed_num_frames, self.height, self.width)
        parts = []
        for ax, (n, dd) in enumerate(zip(sizes, dims)):
            inv = 1.0 / (self.theta ** (torch.arange(0, dd, 2)[: dd // 2].float() / dd))
            a = torch.outer(torch.arange(n, dtype=torch.float32), inv).repeat_interleave(2, dim=-1)  # [n, dd]
            shape = [1, 1, 1, dd]
            shape[ax] = n
            parts.append(a.view(shape).expand(*sizes, dd))
        ang = torch.cat(parts, dim=-1).reshape(-1, d).contiguous()
        return ang.sin(), ang.cos()

    def init_freqs(self):
        s, c = self._tables()
        self.freqs_sin.copy_(s)
        self.freqs_cos.copy_(c)
        if self.tp_mesh is not None:
            self.freqs_sin = replicate_tensor(self.freqs_sin, self.tp_mesh)
            self.freqs_cos = replicate_tensor(self.freqs_cos, self.tp_mesh)

    def tables_f32(self):
        """(cos, sin) as contiguous fp32 [n_pos, head_dim] for the fused HIP kernel (which applies the bf16 rounding of
        ``forward`` itself); cached per buffer identity."""
        key = (self.freqs_cos.data_ptr(), self.freqs_cos.dtype, self.freqs_cos.device)
        if getattr(self, "_f32_key", None) != key:
            self._f32 = (to_local(self.freqs_cos).float().contiguous(), to_local(self.freqs_sin).float().contiguous())
            self._f32_key = key
        return self._f32

    def forward(self, t):  # t [B, NH, S, D]
        n = t.shape[2]
        # computed in the activation dtype, like the reference after cast_rotary_freqs (train.py:71-72)
        cos, sin = self.freqs_cos[:n].to(t.dtype), self.freqs_sin[:n].to(t.dtype)
        pairs = to_local(t).unflatten(-1, (-1, 2))
        rot = torch.stack((-pairs[..., 1], pairs[..., 0]), dim=-1).flatten(-2)
        return t * cos + place_into(rot, t) * sin
