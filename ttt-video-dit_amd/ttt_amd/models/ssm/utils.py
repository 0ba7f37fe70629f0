"""3-D rotary embedding for the TTT layer.  Behaviour of the reference's
``ttt/models/ssm/utils.py:9-108`` (per-head dim split t/h/w = 1/4, 3/8, 3/8; complex multiply on
adjacent pairs; positions enumerate (t,h,w) over all video tokens), expressed with real cos/sin
tables so no complex dtype is needed on the device."""
from __future__ import annotations

import torch


def _axis_angles(n_pos: int, dim: int, theta: float) -> torch.Tensor:
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    return torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)  # [n_pos, dim/2]


def precompute_freqs_cis_3d(dim: int, height: int, width: int, compressed_num_frames: int, theta: float = 10000.0,
                            as_real: bool = False) -> torch.Tensor:
    """Rotation table for every (t,h,w) position: complex64 [frames*height*width, dim/2] like the
    reference (ssm/utils.py:9-53), or with ``as_real`` the same numbers as fp32 [..., 2] = (cos, sin),
    which survives ``module.to(dtype)`` (a complex buffer silently loses its imaginary part)."""
    dt, dh, dw = dim // 4, dim // 8 * 3, dim // 8 * 3
    at = _axis_angles(compressed_num_frames, dt, theta)[:, None, None, :].expand(-1, height, width, -1)
    ah = _axis_angles(height, dh, theta)[None, :, None, :].expand(compressed_num_frames, -1, width, -1)
    aw = _axis_angles(width, dw, theta)[None, None, :, :].expand(compressed_num_frames, height, -1, -1)
    ang = torch.cat((at, ah, aw), dim=-1).reshape(compressed_num_frames * height * width, -1).contiguous()
    if as_real:
        return torch.stack((ang.cos(), ang.sin()), dim=-1)
    return torch.polar(torch.ones_like(ang), ang)


def apply_rotary_emb(xq: torch.Tensor, xk: torch.Tensor, freqs_cis: torch.Tensor):
    """Rotate adjacent (even, odd) feature pairs of xq/xk [B, S, NH, D] by freqs_cis[:S]
    (reference ssm/utils.py:82-108).  Computed in fp32, returned in the input dtypes."""
    S = xq.shape[1]
    fc = freqs_cis[:S]
    if fc.is_complex():
        cos, sin = fc.real, fc.imag
    else:  # already (cos, sin) stacked on the last dim
        cos, sin = fc[..., 0], fc[..., 1]
    cos = cos.float()[None, :, None, :]
    sin = sin.float()[None, :, None, :]

    def rot(x):
        xf = x.float().unflatten(-1, (-1, 2))
        a, b = xf[..., 0], xf[..., 1]
        return torch.stack((a * cos - b * sin, a * sin + b * cos), dim=-1).flatten(-2).type_as(x)

    return rot(xq), rot(xk)
