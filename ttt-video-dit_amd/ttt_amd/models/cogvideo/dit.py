"""CogVideoX DiT with TTT layers: host-side (PyTorch-ROCm) assembly of the block that surrounds the
TTT scan.  Class names, constructor signatures, parameter names (state-dict keys) and numerical
behaviour follow the reference's ``ttt/models/cogvideo/dit.py`` (PatchEmbedding :17-40, MLP
:43-87, SSMGating :90-103, SeqModelingBlock :106-278, TransformerLayer :281-382, FinalLayer
:385-418, DiffusionTransformer :421-505).

Dense projections / MLP GEMMs go to hipBLASLt through PyTorch; the TTT scan goes to the
hand-written gfx950 kernels (``ttt_amd.models.ssm``); the local attention (q/k LayerNorm + RoPE, QK^T /
softmax / PV and their backward) goes to the MFMA kernels behind ``ttt_amd.models.cogvideo.attention``.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F
from torch import nn
from torch.utils.checkpoint import checkpoint

from ttt_amd.infra import remat_cache
from ttt_amd.infra.fused_linear import linear3
from ttt_amd.models.cogvideo.attention import FusedSegmentAttention, attn_pre_available, segment_attention
from ttt_amd.models.cogvideo.utils import (Rotary3DPositionEmbedding, SequenceMetadata, modulate,
                                           timestep_embedding, unpatchify)
from ttt_amd.models.configs import ModelConfig
from ttt_amd.models.ssm.fused import FusedAdaLN, FusedGate, FusedResGate, fused_available
from ttt_amd.models.ssm.ttt_layer import TTTWrapper


def _ckpt(fn, enabled: bool):
    if not enabled:
        return fn
    # (a checkpoint nested in a region that keeps kernel outputs recomputes in an order of its own: nothing is kept inside it)
    return lambda *a: checkpoint(fn, *a, use_reentrant=False, context_fn=lambda: (remat_cache.suspended(), remat_cache.suspended()))


class GeluLinear(torch.autograd.Function):
    """``F.linear(gelu_tanh(z), W, b)`` as one autograd node that saves the pre-activation ``z`` only (the unfused pair keeps
    both ``z`` and ``gelu(z)``: 2 x [B, L, 4 D] = 0.9 GB per layer at the 3 s geometry); GELU is re-evaluated in backward.
    In a re-materialised region that keeps the kind ``"fc2"`` the output - [B, L, D], a quarter of ``z``, 0.32 GB at the 9 s
    geometry for a 3 ms GEMM + a GELU pass: 10 ms per GB, the best ratio after the sequence kernels' outputs
    (``ttt_amd/infra/remat_cache.py``) - is handed back in the recomputation instead of being formed again."""

    @staticmethod
    def forward(ctx, z, weight, bias):
        ctx.save_for_backward(z, weight)
        ctx.has_bias = bias is not None
        return remat_cache.kernel_result("fc2", lambda: (F.linear(F.gelu(z, approximate="tanh"), weight, bias),))[0]

    @staticmethod
    def backward(ctx, dy):
        z, weight = ctx.saved_tensors
        need_z, need_w, need_b = ctx.needs_input_grad
        dw = db = dz = None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if need_w:         # frozen MLP weights (adapter_method qkvo / none): no GELU pass, no L x 4D x D GEMM
            h = F.gelu(z, approximate="tanh")
            dw = dy2.t().mm(h.reshape(-1, h.shape[-1]))
            del h
        if need_b and ctx.has_bias:
            db = dy2.sum(0)
        if need_z:
            dz = torch.ops.aten.gelu_backward(dy.matmul(weight), z, approximate="tanh")
        return dz, dw, db


def _segment_geometry(meta: SequenceMetadata, n_video: int, attn_length: int, prefix: int):
    """(tl, [(lo, hi)] video-token range of every attention segment, tokens of a shared block): segment i attends over
    ``[text_i, frames attn_length*i .. attn_length*(i+1) + prefix)``; consecutive segments share ``prefix`` frames."""
    tl, tpf = meta.text_length, meta.tokens_per_frame
    rng = [(i * attn_length * tpf, min((prefix + (i + 1) * attn_length) * tpf, n_video)) for i in range(meta.num_chunks)]
    shared = prefix * tpf
    # (a last segment cut short by the end of the video is fine - the reference's slices clip the same way -, an uncovered tail
    # or a segment that consists of its shared block only is not)
    assert 0 < prefix < attn_length and rng[-1][1] == n_video and all(hi - lo > shared for lo, hi in rng), \
        "segments must tile the video, consecutive ones sharing `prefix` frames"
    return tl, rng, shared


class SegmentSplit(torch.autograd.Function):
    """``[text | video]`` sequence -> the per-segment inputs ``[text_i | video[lo_i:hi_i]]`` (the reference builds them with
    slice + cat, dit.py:163-211).  One node instead of 2 n slice nodes: its backward writes the input gradient ONCE (every row
    from its segment, the shared frames as the sum of their two segments) where autograd's slice backward materialises a
    zero-padded full-length tensor per slice and adds them up - 3 GB of traffic per layer at the 9 s geometry.  Same bits."""

    @staticmethod
    def forward(ctx, x, n_text, tl, rng, shared):
        ctx.geo = (n_text, tl, rng, shared, x.shape)
        return tuple(torch.cat((x[:, i * tl:(i + 1) * tl], x[:, n_text + lo:n_text + hi]), dim=1) for i, (lo, hi) in enumerate(rng))

    @staticmethod
    def backward(ctx, *gs):
        n_text, tl, rng, shared, shape = ctx.geo
        ref = next(g for g in gs if g is not None)
        gs = [g if g is not None else ref.new_zeros(shape[0], tl + hi - lo, shape[2]) for g, (lo, hi) in zip(gs, rng)]
        dx = ref.new_empty(shape)
        for i, (lo, hi) in enumerate(rng):
            dx[:, i * tl:(i + 1) * tl] = gs[i][:, :tl]
            dx[:, n_text + lo:n_text + hi] = gs[i][:, tl:]          # a later segment overwrites the block it shares ...
        for i in range(1, len(rng)):                                # ... which then gets the sum of both
            lo = rng[i][0]
            torch.add(gs[i - 1][:, -shared:], gs[i][:, tl:tl + shared], out=dx[:, n_text + lo:n_text + lo + shared])
        return dx, None, None, None, None


class SegmentMerge(torch.autograd.Function):
    """Per-segment outputs -> one ``[text | video]`` sequence, the shared frames averaged (reference dit.py:199-211: accumulate
    into zeros, count, divide, cat).  Written once: every row copied from its segment, the shared blocks as ``(a + b) / 2`` -
    the same roundings as accumulate-then-divide-by-count."""

    @staticmethod
    def forward(ctx, n_text, tl, rng, shared, *outs):
        ctx.geo = (n_text, tl, rng, shared)
        b, _, d = outs[0].shape
        out = outs[0].new_empty(b, n_text + rng[-1][1], d)
        for i, (lo, hi) in enumerate(rng):
            out[:, i * tl:(i + 1) * tl] = outs[i][:, :tl]
            out[:, n_text + lo:n_text + hi] = outs[i][:, tl:]
        for i in range(1, len(rng)):
            lo = rng[i][0]
            blk = out[:, n_text + lo:n_text + lo + shared]
            torch.add(outs[i - 1][:, -shared:], outs[i][:, tl:tl + shared], out=blk)
            blk.div_(2)
        return out

    @staticmethod
    def backward(ctx, g):
        n_text, tl, rng, shared = ctx.geo
        gs = [torch.cat((g[:, i * tl:(i + 1) * tl], g[:, n_text + lo:n_text + hi]), dim=1) for i, (lo, hi) in enumerate(rng)]
        for i in range(1, len(rng)):
            gs[i - 1][:, -shared:].div_(2)
            gs[i][:, tl:tl + shared].div_(2)
        return (None, None, None, None, *gs)


class PatchEmbedding(nn.Module):
    """2x2 patch conv for video latents + Linear for text (reference :17-40)."""

    def __init__(self, config: ModelConfig):
        super().__init__()
        train = config.adapter_method == "sft"
        self.vid_proj = nn.Conv2d(config.in_channels, config.model_dim, config.patch_size, config.patch_size, bias=True).requires_grad_(train)
        self.text_proj = nn.Linear(config.text_dim, config.model_dim, bias=True).requires_grad_(train)

    def forward(self, video, text_encoding):
        b, t, c, H, W = video.shape
        conv = self.vid_proj
        ph, pw = conv.kernel_size
        if conv.stride == conv.kernel_size and conv.padding == (0, 0) and H % ph == 0 and W % pw == 0:
            # a stride = kernel convolution is a GEMM over non-overlapping patches: [(b t h w), c*ph*pw] x [c*ph*pw, D] lands in
            # the [b, t*h*w, D] layout directly (MIOpen's convolution for this shape is a naive kernel, 4.5 ms per call at the
            # 3 s geometry, and its NCHW output needs a transpose copy)
            h, w = H // ph, W // pw
            patches = video.reshape(b * t, c, h, ph, w, pw).permute(0, 2, 4, 1, 3, 5).reshape(b, t * h * w, c * ph * pw)
            x = F.linear(patches, conv.weight.reshape(conv.out_channels, -1), conv.bias)
        else:
            x = conv(video.flatten(0, 1))                                # [(b t), D, h, w]
            x = x.flatten(2).transpose(1, 2).reshape(b, -1, x.shape[1])  # [b, t*h*w, D]
        return self.text_proj(text_encoding).contiguous(), x.contiguous()


class MLP(nn.Module):
    """3072 -> 12288 -> 3072 with tanh-GELU (reference :43-87)."""

    def __init__(self, config: ModelConfig):
        super().__init__()
        train = config.adapter_method == "sft"
        self.do_remat = config.remat_mlp
        self.requires_grad = train
        self.layer1 = nn.Linear(config.model_dim, 4 * config.model_dim, bias=True).requires_grad_(train)
        self.layer2 = nn.Linear(4 * config.model_dim, config.model_dim, bias=True).requires_grad_(train)
        self.tp_mesh = None

    def _run(self, x):
        z = self.layer1(x)
        w2 = self.layer2.weight
        if z.is_cuda and type(w2) in (torch.Tensor, nn.Parameter) and torch.is_grad_enabled():
            return GeluLinear.apply(z, w2, self.layer2.bias)     # keeps z only; GELU(z) is re-derived in backward
        return self.layer2(F.gelu(z, approximate="tanh"))

    def forward(self, x):
        return _ckpt(self._run, self.do_remat)(x)


class SSMGating(nn.Module):
    """``tanh(alpha) * x`` with a learned per-channel alpha (reference :90-103)."""

    def __init__(self, config):
        super().__init__()
        self.gating_alpha = nn.Parameter(torch.ones(config.model_dim) * config.gating_alpha_init)

    def forward(self, x):
        return torch.tanh(self.gating_alpha) * x


class SeqModelingBlock(nn.Module):
    """Local (per 3 s segment) attention followed by a bidirectional, weight-shared TTT pass
    (reference :106-278)."""

    def __init__(self, config: ModelConfig):
        super().__init__()
        train = config.adapter_method in ("sft", "qkvo")
        assert config.adapter_method in ("sft", "qkvo", "none"), f"Invalid adapter method: {config.adapter_method}"
        self.do_attn_remat = config.remat_attention
        self.do_forward_ssm_remat = config.remat_forward_ssm
        self.do_reverse_ssm_remat = config.remat_reverse_ssm
        self.num_heads = config.num_heads
        self.head_dim = config.model_dim // config.num_heads
        self.prefix_temporal_length = config.prefix_temporal_length
        self.attn_length = config.attn_length

        self.q_norm = nn.LayerNorm(self.head_dim, eps=config.layer_norm_eps).requires_grad_(train)
        self.k_norm = nn.LayerNorm(self.head_dim, eps=config.layer_norm_eps).requires_grad_(train)
        self.rotary = Rotary3DPositionEmbedding(config.latent_height, config.latent_width, config.compressed_num_frames,
                                                self.head_dim, config.theta)
        D = config.model_dim
        self.q = nn.Linear(D, D, bias=True)
        self.k = nn.Linear(D, D, bias=True)
        self.v = nn.Linear(D, D, bias=True)
        self.o = nn.Linear(D, D, bias=True)
        self.ssm = TTTWrapper(config)
        self.forward_ssm_gating_video = SSMGating(config)
        self.forward_ssm_gating_text = SSMGating(config)
        self.backward_ssm_gating_video = SSMGating(config)
        self.backward_ssm_gating_text = SSMGating(config)

    # -- local attention ------------------------------------------------------------------------
    def _segment(self, emb, n_text):
        """q/k/v projections, per-head LayerNorm on q,k, 3-D RoPE on the video tokens, SDPA, o."""
        b, s, _ = emb.shape
        heads = lambda t: t.view(b, s, self.num_heads, self.head_dim).transpose(1, 2)   # [b, h, s, d]
        if attn_pre_available(emb, self.head_dim):      # HIP: LayerNorm + RoPE fused, layout kept, strided views downstream
            cos, sin = self.rotary.tables_f32()
            q, k, v = linear3(self.q, self.k, self.v, emb)
            a = FusedSegmentAttention.apply(q, k, heads(v),
                                            self.q_norm.weight, self.q_norm.bias, self.k_norm.weight, self.k_norm.bias, cos, sin,
                                            self.num_heads, n_text, self.q_norm.eps)
            return self.o(a.transpose(1, 2).reshape(b, s, -1))
        else:
            q, k, v = (heads(t) for t in linear3(self.q, self.k, self.v, emb))
            q, k = self.q_norm(q), self.k_norm(k)
            q = torch.cat((q[:, :, :n_text], self.rotary(q[:, :, n_text:])), dim=2)
            k = torch.cat((k[:, :, :n_text], self.rotary(k[:, :, n_text:])), dim=2)
        a = segment_attention(q, k, v)
        return self.o(a.transpose(1, 2).reshape(b, s, -1))

    def _attn_forward(self, vid_emb, text_emb, seq_metadata: SequenceMetadata, cat=None):
        """Each segment i attends over [text_i, frames 12i .. 12(i+1)] (13 frames, 1 shared with its
        neighbour); the shared frame's outputs are averaged (reference :163-211)."""
        tl, tpf = seq_metadata.text_length, seq_metadata.tokens_per_frame
        if seq_metadata.num_chunks == 1:     # one segment: nothing overlaps, the accumulate / average below is the identity
            return self._segment(cat if cat is not None else torch.cat((text_emb, vid_emb), dim=1), tl)
        x = cat if cat is not None else torch.cat((text_emb, vid_emb), dim=1)
        n_text = text_emb.shape[1]
        tl, rng, shared = _segment_geometry(seq_metadata, vid_emb.shape[1], self.attn_length, self.prefix_temporal_length)
        segs = SegmentSplit.apply(x, n_text, tl, rng, shared)
        return SegmentMerge.apply(n_text, tl, rng, shared, *(self._segment(seg, tl) for seg in segs))

    # -- bidirectional TTT ------------------------------------------------------------------------
    def _gate(self, text_gate, video_gate, residual, ssm_output, n_text):
        if fused_available(residual, 64) and residual.shape[-1] % 8 == 0:     # one HIP pass instead of mul, mul, cat, add
            return FusedGate.apply(residual, ssm_output, text_gate.gating_alpha, video_gate.gating_alpha, n_text)
        return residual + torch.cat((text_gate(ssm_output[:, :n_text]), video_gate(ssm_output[:, n_text:])), dim=1)

    def _ssm_forward(self, emb, seq_metadata: SequenceMetadata):
        """forward TTT -> gated residual -> time-reversed TTT with the same weights -> gated residual
        (reference :224-266).  The time reversal is delegated to the TTT layer (``reverse=True``), whose fused
        pre/post kernels fold it into their token maps instead of materialising flipped copies."""
        n_text = seq_metadata.seq_text_length
        fwd = _ckpt(self.ssm, self.do_forward_ssm_remat)
        rev = _ckpt(self.ssm, self.do_reverse_ssm_remat)
        emb = self._gate(self.forward_ssm_gating_text, self.forward_ssm_gating_video, emb, fwd(emb, seq_metadata, False), n_text)
        y = rev(emb, seq_metadata, True)
        return self._gate(self.backward_ssm_gating_text, self.backward_ssm_gating_video, emb, y, n_text)

    # -- sequence-parallel inference (ttt_amd/infra/sequence_parallel.py) ------------------------------------------------
    def _segment_heads(self, emb, n_text, h0, h1):
        """``_segment`` for heads [h0, h1) only and without the output projection: [b, s, D] -> [b, s, (h1-h0)*F]."""
        b, s, _ = emb.shape
        nh, Fh = h1 - h0, self.head_dim
        sl = slice(h0 * Fh, h1 * Fh)
        lin = lambda m: F.linear(emb, m.weight[sl], m.bias[sl])
        heads = lambda t: t.view(b, s, nh, Fh).transpose(1, 2)
        if attn_pre_available(emb, Fh):
            cos, sin = self.rotary.tables_f32()
            a = FusedSegmentAttention.apply(lin(self.q), lin(self.k), heads(lin(self.v)), self.q_norm.weight, self.q_norm.bias,
                                            self.k_norm.weight, self.k_norm.bias, cos, sin, nh, n_text, self.q_norm.eps)
        else:
            q, k, v = heads(lin(self.q)), heads(lin(self.k)), heads(lin(self.v))
            q, k = self.q_norm(q), self.k_norm(k)
            q = torch.cat((q[:, :, :n_text], self.rotary(q[:, :, n_text:])), dim=2)
            k = torch.cat((k[:, :, :n_text], self.rotary(k[:, :, n_text:])), dim=2)
            a = segment_attention(q, k, v)
        return a.transpose(1, 2).reshape(b, s, -1)

    def _attn_heads(self, vid_emb, text_emb, seq_metadata: SequenceMetadata, h0, h1):
        """``_attn_forward`` for heads [h0, h1) before the output projection (which is linear, so it commutes with the
        averaging of the shared frames and is applied afterwards on the token shards)."""
        tl, tpf = seq_metadata.text_length, seq_metadata.tokens_per_frame
        if seq_metadata.num_chunks == 1:
            return self._segment_heads(torch.cat((text_emb, vid_emb), dim=1), tl, h0, h1)
        d = (h1 - h0) * self.head_dim
        out_vid = vid_emb.new_zeros(*vid_emb.shape[:2], d)
        out_txt = text_emb.new_zeros(*text_emb.shape[:2], d)
        count = torch.zeros_like(vid_emb[..., :1])
        for i in range(seq_metadata.num_chunks):
            lo = i * self.attn_length * tpf
            hi = (self.prefix_temporal_length + (i + 1) * self.attn_length) * tpf
            o = self._segment_heads(torch.cat((text_emb[:, i * tl:(i + 1) * tl], vid_emb[:, lo:hi]), dim=1), tl, h0, h1)
            out_txt[:, i * tl:(i + 1) * tl] = o[:, :tl]
            out_vid[:, lo:hi] += o[:, tl:]
            count[:, lo:hi] += 1
        return torch.cat((out_txt, out_vid / count), dim=1)

    def forward_sp(self, vid_loc, text_loc, seq_metadata: SequenceMetadata, sp):
        """The block on token shards ``vid_loc [B, ceil(Lv/T), D]``, ``text_loc [B, ceil(Lt/T), D]``: sequence-mixing parts on
        this rank's heads over the gathered sequence, token-wise parts (o, post_norm, wo, gates) on the shard."""
        n_text = seq_metadata.seq_text_length
        n_vid = seq_metadata.num_frames * seq_metadata.tokens_per_frame
        h0, h1 = sp.head_range(self.num_heads)
        a = self._attn_heads(sp.gather_tokens(vid_loc, n_vid), sp.gather_tokens(text_loc, n_text), seq_metadata, h0, h1)
        text_loc, vid_loc = self.o(sp.heads_to_tokens(a[:, :n_text])), self.o(sp.heads_to_tokens(a[:, n_text:]))
        ttt = self.ssm.ttt
        for g_text, g_vid, reverse in ((self.forward_ssm_gating_text, self.forward_ssm_gating_video, False),
                                       (self.backward_ssm_gating_text, self.backward_ssm_gating_video, True)):
            full = torch.cat((sp.gather_tokens(text_loc, n_text), sp.gather_tokens(vid_loc, n_vid)), dim=1)
            y = self.ssm.forward_heads(full, seq_metadata, reverse, h0, h1)                       # [B, L, (h1-h0)*F], token order
            y_text = ttt.wo(ttt.post_norm(sp.heads_to_tokens(y[:, :n_text])))
            y_vid = ttt.wo(ttt.post_norm(sp.heads_to_tokens(y[:, n_text:])))
            text_loc, vid_loc = text_loc + g_text(y_text), vid_loc + g_vid(y_vid)
        return vid_loc, text_loc

    def forward_cat(self, x, seq_metadata: SequenceMetadata):
        """Same block on the concatenated ``[text | video]`` sequence, returning it concatenated (the fused TransformerLayer
        path produces and consumes that layout directly, without the cat / slice pairs around the block)."""
        n_text = seq_metadata.seq_text_length
        x = _ckpt(self._attn_forward, self.do_attn_remat)(x[:, n_text:], x[:, :n_text], seq_metadata, x)
        return self._ssm_forward(x, seq_metadata)

    def forward(self, vid_emb, text_emb, seq_metadata: SequenceMetadata):
        x = _ckpt(self._attn_forward, self.do_attn_remat)(vid_emb, text_emb, seq_metadata)
        x = self._ssm_forward(x, seq_metadata)
        n_text = seq_metadata.seq_text_length
        return x[:, n_text:], x[:, :n_text]


class TransformerLayer(nn.Module):
    """AdaLN-modulated sequence block + AdaLN-modulated MLP, separate modulation for video and text
    tokens (reference :281-382)."""

    def __init__(self, config):
        super().__init__()
        train = config.adapter_method == "sft"
        self.remat_seq_modeling_block = config.remat_seq_modeling_block
        self.tp_mesh = None
        self.use_fused_glue = True     # HIP AdaLN / gated-residual kernels when the activations are bf16 on a HIP device
        D = config.model_dim
        self.pre_seq_layernorm = nn.LayerNorm(D, eps=config.layer_norm_eps).requires_grad_(train)
        self.pre_seq_adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(config.time_embed_dim, 6 * D, bias=True).requires_grad_(train))
        self.seq_modeling_block = SeqModelingBlock(config)
        self.pre_mlp_layernorm = nn.LayerNorm(D, eps=config.layer_norm_eps).requires_grad_(train)
        self.pre_mlp_adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(config.time_embed_dim, 6 * D, bias=True).requires_grad_(train))
        self.mlp = MLP(config)

    def _forward_fused(self, vid_emb, text_emb, seq_metadata: SequenceMetadata):
        """HIP glue path (bf16 on a HIP device): layernorm + modulate + concat and the gated residuals are one kernel each
        and the [text | video] sequence stays concatenated through the block and the MLP."""
        t = seq_metadata.t_emb
        ln1, ln2 = self.pre_seq_layernorm, self.pre_mlp_layernorm
        sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_seq_adaLN_modulation(t).chunk(6, dim=1)
        x = FusedAdaLN.apply(vid_emb, text_emb, ln1.weight, ln1.bias, sh_v, sc_v, sh_t, sc_t, ln1.eps)
        y = _ckpt(self.seq_modeling_block.forward_cat, self.remat_seq_modeling_block)(x, seq_metadata)
        vid_emb, text_emb = FusedResGate.apply(vid_emb, text_emb, y, g_v, g_t)
        sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_mlp_adaLN_modulation(t).chunk(6, dim=1)
        x = FusedAdaLN.apply(vid_emb, text_emb, ln2.weight, ln2.bias, sh_v, sc_v, sh_t, sc_t, ln2.eps)
        y = self.mlp(x)
        return FusedResGate.apply(vid_emb, text_emb, y, g_v, g_t)

    def forward_sp(self, vid_emb, text_emb, seq_metadata: SequenceMetadata, sp):
        """``forward`` on token shards (sequence parallelism): AdaLN, residual gates and the MLP are token-wise."""
        t = seq_metadata.t_emb
        if self.use_fused_glue and fused_available(vid_emb, 64) and vid_emb.shape[-1] % 8 == 0 and vid_emb.shape[-1] <= 4096:
            # the HIP glue kernels of the fused path on this rank's [text shard | video shard] (token-wise: any row counts)
            ln1, ln2, nt = self.pre_seq_layernorm, self.pre_mlp_layernorm, text_emb.shape[1]
            sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_seq_adaLN_modulation(t).chunk(6, dim=1)
            x = FusedAdaLN.apply(vid_emb, text_emb, ln1.weight, ln1.bias, sh_v, sc_v, sh_t, sc_t, ln1.eps)
            v_out, t_out = self.seq_modeling_block.forward_sp(x[:, nt:], x[:, :nt], seq_metadata, sp)
            vid_emb, text_emb = FusedResGate.apply(vid_emb, text_emb, torch.cat((t_out, v_out), dim=1), g_v, g_t)
            sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_mlp_adaLN_modulation(t).chunk(6, dim=1)
            x = FusedAdaLN.apply(vid_emb, text_emb, ln2.weight, ln2.bias, sh_v, sc_v, sh_t, sc_t, ln2.eps)
            return FusedResGate.apply(vid_emb, text_emb, self.mlp(x), g_v, g_t)
        sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_seq_adaLN_modulation(t).chunk(6, dim=1)
        v_out, t_out = self.seq_modeling_block.forward_sp(modulate(self.pre_seq_layernorm(vid_emb), sh_v, sc_v),
                                                          modulate(self.pre_seq_layernorm(text_emb), sh_t, sc_t), seq_metadata, sp)
        vid_emb = vid_emb + g_v.unsqueeze(1) * v_out
        text_emb = text_emb + g_t.unsqueeze(1) * t_out
        sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_mlp_adaLN_modulation(t).chunk(6, dim=1)
        vid_emb = vid_emb + g_v.unsqueeze(1) * self.mlp(modulate(self.pre_mlp_layernorm(vid_emb), sh_v, sc_v))
        text_emb = text_emb + g_t.unsqueeze(1) * self.mlp(modulate(self.pre_mlp_layernorm(text_emb), sh_t, sc_t))
        return vid_emb, text_emb

    def forward(self, vid_emb, text_emb, seq_metadata: SequenceMetadata, sp=None):
        if sp is not None:       # token shards of one sample (through __call__: FSDP2's unshard / reshard hooks must run)
            return self.forward_sp(vid_emb, text_emb, seq_metadata, sp)
        if self.use_fused_glue and fused_available(vid_emb, 64) and vid_emb.shape[-1] % 8 == 0 and vid_emb.shape[-1] <= 4096:
            return self._forward_fused(vid_emb, text_emb, seq_metadata)
        n_text = seq_metadata.seq_text_length
        t = seq_metadata.t_emb
        sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_seq_adaLN_modulation(t).chunk(6, dim=1)
        block = _ckpt(self.seq_modeling_block, self.remat_seq_modeling_block)
        v_out, t_out = block(modulate(self.pre_seq_layernorm(vid_emb), sh_v, sc_v),
                             modulate(self.pre_seq_layernorm(text_emb), sh_t, sc_t), seq_metadata)
        vid_emb = vid_emb + g_v.unsqueeze(1) * v_out
        text_emb = text_emb + g_t.unsqueeze(1) * t_out

        sh_v, sc_v, g_v, sh_t, sc_t, g_t = self.pre_mlp_adaLN_modulation(t).chunk(6, dim=1)
        x = torch.cat((modulate(self.pre_mlp_layernorm(text_emb), sh_t, sc_t),
                       modulate(self.pre_mlp_layernorm(vid_emb), sh_v, sc_v)), dim=1)
        y = self.mlp(x)
        vid_emb = vid_emb + g_v.unsqueeze(1) * y[:, n_text:]
        text_emb = text_emb + g_t.unsqueeze(1) * y[:, :n_text]
        return vid_emb, text_emb


class FinalLayer(nn.Module):
    """AdaLN + Linear to patch pixels + unpatchify (reference :385-418)."""

    def __init__(self, config: ModelConfig):
        super().__init__()
        train = config.adapter_method == "sft"
        self.out_channels = config.out_channels
        self.patch_size = config.patch_size
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(config.time_embed_dim, 2 * config.model_dim, bias=True).requires_grad_(train))
        self.norm = nn.LayerNorm(config.model_dim, elementwise_affine=True, eps=config.layer_norm_eps).requires_grad_(train)
        self.linear = nn.Linear(config.model_dim, config.patch_size * config.patch_size * self.out_channels, bias=True).requires_grad_(train)

    def forward(self, vid_emb, seq_metadata: SequenceMetadata):
        shift, scale = self.adaLN_modulation(seq_metadata.t_emb).chunk(2, dim=1)
        x = self.linear(modulate(self.norm(vid_emb), shift, scale))
        return unpatchify(x, c=self.out_channels, p=self.patch_size, w=seq_metadata.latent_width // self.patch_size,
                          h=seq_metadata.latent_height // self.patch_size)


class DiffusionTransformer(nn.Module):
    """forward(video [B,T,16,H,W], text [B,n_scenes,S,text_dim], timesteps [B]) -> [B,T,16,H,W]
    (reference :421-505).  Layers are re-materialised in groups of
    ``remat_transformer_layer_group_size`` during backward, except the first ``remat_free_layers`` layers, which keep
    their activations (288 GB of HBM3E per MI355X make that the better trade; same arithmetic, same results)."""

    def __init__(self, config):
        super().__init__()
        train = config.adapter_method == "sft"
        self.frames_per_chunk = config.attn_length
        self.remat_transformer_layer_group_size = config.remat_transformer_layer_group_size
        self.remat_free_layers = getattr(config, "remat_free_layers", 0)
        # kernel outputs a re-materialised layer group keeps instead of recomputing them: any of "attn" (local-attention outputs),
        # "scan" (TTT scan outputs + state checkpoints); () = the reference's behaviour (ttt_amd/infra/remat_cache.py)
        self.remat_keep = tuple(getattr(config, "remat_keep", ()))
        # ... and how many of the re-materialised layers do so (the FIRST ones; None = all of them): at 63 s on one GPU there is room
        # for the kept attention outputs of about ten layers, not of 42
        self.remat_keep_layers = getattr(config, "remat_keep_layers", None)
        # ... per kind: {"scan": 20} = only the first 20 re-materialised layers keep their scan outputs (the other kinds: all that keep)
        self.remat_keep_limits = dict(getattr(config, "remat_keep_limits", None) or {})
        assert config.num_layers % self.remat_transformer_layer_group_size == 0, "Remat group size must be divisible into num layers"
        self.model_dim = config.model_dim
        self.shard_transformer_inputs = config.shard_transformer_inputs
        self.time_embed = nn.Sequential(
            nn.Linear(config.model_dim, config.time_embed_dim, bias=True).requires_grad_(train), nn.SiLU(),
            nn.Linear(config.time_embed_dim, config.time_embed_dim, bias=True).requires_grad_(train))
        self.patch_embedding = PatchEmbedding(config)
        self.layers = nn.ModuleList([TransformerLayer(config) for _ in range(config.num_layers)])
        self.transformer_norm = nn.LayerNorm(config.model_dim, eps=config.layer_norm_eps).requires_grad_(train)
        self.final_layer = FinalLayer(config)
        self.sequence_parallel = None      # a ttt_amd.infra.sequence_parallel.SeqParallel: one sample over the ranks of its group
        # a ttt_amd.infra.host_offload.HostOffload: what the remat-free layers save waits in pinned host memory between forward and backward
        self.host_offload = None

    def _run_group(self, start, vid_emb, text_emb, seq_metadata, sp=None):
        for layer in self.layers[start:start + self.remat_transformer_layer_group_size]:
            vid_emb, text_emb = layer(vid_emb, text_emb, seq_metadata, sp)
        return vid_emb, text_emb

    def forward(self, video, text, timesteps):
        num_frames, height, width = video.shape[1], video.shape[3], video.shape[4]
        t_emb = self.time_embed(timestep_embedding(timesteps, self.model_dim, dtype=video.dtype))
        text_emb, vid_emb = self.patch_embedding(video, text)      # [B,n,S,D], [B,T*h*w,D]
        n_scenes, text_len = text_emb.shape[1], text.shape[-2]
        meta = SequenceMetadata(text_length=text_len, seq_text_length=text_len * n_scenes, num_frames=num_frames,
                                num_chunks=n_scenes, tokens_per_frame=vid_emb.shape[1] // num_frames,
                                latent_height=height, latent_width=width, t_emb=t_emb)
        if meta.is_multiscene:
            meta.init_multiscene_offsets()
        text_emb = text_emb.flatten(1, 2)
        sp = self.sequence_parallel
        if sp is not None:        # every rank got the same inputs; each keeps 1/T of the tokens: what a layer group's checkpoint
            n_vid = vid_emb.shape[1]     # saves is a token shard (reference shard_transformer_inputs, dit.py:494-498)
            vid_emb, text_emb = sp.shard_tokens(vid_emb), sp.shard_tokens(text_emb)
        off = self.host_offload if torch.is_grad_enabled() else None
        if off is not None:
            off.begin_step()
        for i in range(0, len(self.layers), self.remat_transformer_layer_group_size):
            if off is not None and i < self.remat_free_layers:
                with off.layer(i):
                    vid_emb, text_emb = self._run_group(i, vid_emb, text_emb, meta, sp)
            elif torch.is_grad_enabled() and i >= self.remat_free_layers:
                keeps = self.remat_keep_layers is None or (i - self.remat_free_layers) < self.remat_keep_layers
                kinds = tuple(k for k in self.remat_keep
                              if self.remat_keep_limits.get(k) is None or (i - self.remat_free_layers) < self.remat_keep_limits[k]) if keeps else ()
                if kinds:
                    park = (off, i) if off is not None and off.park_kept else None      # the kept outputs wait in host memory
                    with (off.scope() if park else contextlib.nullcontext()):
                        vid_emb, text_emb = checkpoint(self._run_group, i, vid_emb, text_emb, meta, sp, use_reentrant=False,
                                                       context_fn=remat_cache.context_fn(kinds, park))
                else:
                    vid_emb, text_emb = checkpoint(self._run_group, i, vid_emb, text_emb, meta, sp, use_reentrant=False)
            else:
                vid_emb, text_emb = self._run_group(i, vid_emb, text_emb, meta, sp)
            if off is not None and vid_emb.requires_grad:       # the backward announces itself one layer group at a time
                vid_emb.register_hook(lambda g, i=i: off.backward_reaches(i))
        if off is not None:
            off.end_forward()
        if sp is not None:
            # final norm / AdaLN / projection are token-wise too; only the unpatchify reshape needs the whole sequence.  Every
            # rank then evaluates the same loss on the gathered output, hence replicated_consumer.
            fl = self.final_layer
            shift, scale = fl.adaLN_modulation(t_emb).chunk(2, dim=1)
            y = fl.linear(modulate(fl.norm(self.transformer_norm(vid_emb)), shift, scale))
            y = sp.gather_tokens(y, n_vid, replicated_consumer=True)
            return unpatchify(y, c=fl.out_channels, p=fl.patch_size, w=width // fl.patch_size, h=height // fl.patch_size)
        return self.final_layer(self.transformer_norm(vid_emb), meta)
