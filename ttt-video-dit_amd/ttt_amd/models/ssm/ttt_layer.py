"""TTT layer modules: ``TTTWrapper`` -> ``TTTMLP`` | ``TTTLinear`` (on ``TTTBase``).

Public surface, constructor arguments, parameter names/shapes (state-dict keys) and numerical
behaviour follow the reference's ``ttt/models/ssm/ttt_layer.py`` (TTTWrapper :17-50, TTTBase
:53-334, TTTLinear :337-398, TTTMLP :401-473) so reference checkpoints load unchanged.  The scan
itself runs on the gfx950 HIP kernels behind ``TkMLP`` / ``HipLinear`` (``use_kernel=True``, the
default) or, on explicit request, in dual form in PyTorch (``use_kernel=False``).

MI355X-first differences in how the same math is organised:
  * the multi-scene interleave / undo-interleave (reference :157-217) are single gathers through
    a cached permutation instead of chunk/cat chains, and the same permutation yields the
    per-mini-batch eta row directly;
  * on the kernel path eta is never expanded to the ``[CS,CS]`` tile (64x redundant; the kernels
    read one row, mlp_tk.py:105): only ``[B,NH,NC,1,CS]`` is produced.  The full tile is built
    only for ``use_kernel=False``, whose dual form consumes it.
"""
from __future__ import annotations

import os
from typing import Dict

import torch
import torch.nn.functional as F
from torch import nn

from ttt_amd.infra.fused_linear import linear3, linear3_applies
from ttt_amd.models.cogvideo.utils import SequenceMetadata
from ttt_amd.models.configs import ModelConfig
from ttt_amd.models.ssm.fused import FusedPost, FusedPre, FusedPreScanMLP, fused_available
from ttt_amd.models.ssm.linear_hip import HipLinear, TritonLinear  # noqa: F401
from ttt_amd.models.ssm.mlp_tk import TkMLP
from ttt_amd.models.ssm.ops import ttt_linear, ttt_mlp
from ttt_amd.models.ssm.utils import apply_rotary_emb, precompute_freqs_cis_3d


class TTTWrapper(nn.Module):
    """Selects the TTT variant from ``config.ssm_layer`` and owns the 3-D RoPE table (reference :17-50)."""

    def __init__(self, config: ModelConfig):
        super().__init__()
        self.model_dim = config.model_dim
        self.num_heads = config.num_heads
        self.rope_theta = config.rope_theta
        self.latent_height = config.latent_height
        self.latent_width = config.latent_width
        self.compressed_num_frames = config.compressed_num_frames
        if config.ssm_layer == "ttt_linear":
            self.ttt = TTTLinear(config)
        elif config.ssm_layer == "ttt_mlp":
            self.ttt = TTTMLP(config)
        else:
            raise TypeError(f"No ttt layer of type {config.ssm_layer}")
        self.register_buffer("freqs_cis", self._precompute_freqs_cis_3d(), persistent=False)

    def _precompute_freqs_cis_3d(self) -> torch.Tensor:
        # stored as fp32 (cos, sin) pairs rather than complex64: same values, safe under .to(dtype)
        return precompute_freqs_cis_3d(self.model_dim // self.num_heads, self.latent_height, self.latent_width,
                                       self.compressed_num_frames, self.rope_theta, as_real=True)

    def _apply(self, fn, recurse=True):
        table = self.freqs_cis
        super()._apply(fn, recurse)
        if self.freqs_cis.dtype != torch.float32 and not self.freqs_cis.is_meta:   # follow device moves, refuse down-casts
            self.freqs_cis = table.to(self.freqs_cis.device)
        return self

    def init_freqs(self):
        self.freqs_cis.copy_(self._precompute_freqs_cis_3d())

    def forward(self, x: torch.Tensor, seq_metadata: SequenceMetadata, reverse: bool = False):
        """``reverse=True`` runs the layer on the time-reversed sequence and returns the result in the original
        token order - the second half of the bidirectional pass (reference cogvideo/dit.py:247-263 flips around the
        call; here the fused pre/post kernels absorb the flips into their token maps)."""
        return self.ttt(x, self.freqs_cis, seq_metadata, reverse)

    def forward_heads(self, x: torch.Tensor, seq_metadata: SequenceMetadata, reverse: bool, h0: int, h1: int):
        return self.ttt.forward_heads(x, self.freqs_cis, seq_metadata, reverse, h0, h1)


def scene_permutation(meta: SequenceMetadata, seq_len: int) -> torch.Tensor:
    """Index ``p`` with ``interleaved = tokens[p]``: [text_0..text_n, video] -> [text_0 video_0 text_1
    video_1 ...] where scene 0 owns the remainder frame(s) (reference interleave :157-186 with the
    offsets of cogvideo/utils.py:16-26)."""
    n, tl = meta.num_chunks, meta.text_length
    first = meta.init_offset - tl              # video tokens of scene 0
    rest = meta.base_offset - tl               # video tokens of every later scene
    txt = n * tl
    parts, v = [], txt
    for s in range(n):
        parts.append(torch.arange(s * tl, (s + 1) * tl))
        nv = first if s == 0 else rest
        parts.append(torch.arange(v, v + nv))
        v += nv
    p = torch.cat(parts)
    if p.numel() != seq_len or v != seq_len:
        raise ValueError("sequence length does not match the scene layout in SequenceMetadata")
    return p


def flip_sequence(emb: torch.Tensor, meta: SequenceMetadata) -> torch.Tensor:
    """Time reversal of a [texts, video] token sequence: all video tokens flipped, the per-scene text chunks in
    reverse order (reference cogvideo/dit.py:213-217, 247-263).  An involution."""
    n_text = meta.seq_text_length
    txt = emb[:, :n_text]
    if meta.is_multiscene:
        b, n, e = txt.shape
        txt = txt.view(b, meta.num_chunks, n // meta.num_chunks, e).flip(1).reshape(b, n, e)
    return torch.cat((txt, emb[:, n_text:].flip(1)), dim=1)


def reversal_map(meta: SequenceMetadata, seq_len: int) -> torch.Tensor:
    """Index r of the time-reversed sequence -> index of the same token in the original sequence."""
    n_text, tl = meta.seq_text_length, meta.text_length
    r = torch.arange(n_text)
    if meta.is_multiscene:
        r = (meta.num_chunks - 1 - r // tl) * tl + r % tl
    return torch.cat((r, torch.arange(seq_len - 1, n_text - 1, -1)))


class TTTBase(nn.Module):
    def __init__(self, config: ModelConfig):
        super().__init__()
        self.config = config
        self.width = config.model_dim
        self.num_heads = config.num_heads
        self.head_dim = config.model_dim // config.num_heads
        self.mini_batch_size = config.mini_batch_size
        self.ttt_base_lr = config.ttt_base_lr
        self.scan_checkpoint_group_size = config.scan_checkpoint_group_size
        self.tp_mesh = None
        self._tp = None
        self.use_kernel = True
        self.use_fused = True          # fused HIP pre/post-processing when the activations are bf16 on a HIP device
        # TTT-MLP forward as a pipeline over this many parts of the sequence: the scan of one part on a side stream beside the
        # projections of the next and the post-norm / output projection of the previous (ttt_amd/models/ssm/pipeline.py); 0 / 1 = off.
        # Round 5: 4 equal parts (one MI355X, 5B / 9 s: layer forward 10.3 -> 7.8 ms, the training step +3.3 %; profiles/r5d_*, r5e_*).
        # Round 6, with the pair scan (the compute stream became the pipeline's co-bottleneck): 5 parts that TAPER towards the end
        # (pipeline.TAPER: 16 / 16 / 11 / 6 / 2 of 51 checkpoint groups at 9 s) - in-step 8 554 - 8 574 against 8 484 - 8 501 video-tok/s for
        # 4 equal parts on one box (+0.8 %), 4 tapered parts the same as 4 equal (profiles/r6ab2_*).  Fewer parts where the scan has
        # fewer than two checkpoint groups per part; on the MFMA scan at CS = 64 only.
        self.pipeline_parts = int(os.environ.get("TTT_PIPELINE_PARTS", "5"))
        # ... and more of them for long scans (up to 8, one per ~40 checkpoint groups): what stays exposed is the first part's projections
        # and the last part's output projection, which grow with the sequence - at 63 s (343 groups) 8 parts measured 7 122 against 7 001
        # video-tok/s for 4 on one box (profiles/r5i_*); False: exactly `pipeline_parts`
        self.pipeline_parts_auto = True

        D, NH, Fh = self.width, self.num_heads, self.head_dim
        self.wq = nn.Linear(D, NH * Fh, bias=True)
        self.wk = nn.Linear(D, NH * Fh, bias=True)
        self.wv = nn.Linear(D, NH * Fh, bias=True)
        self.wo = nn.Linear(D, NH * Fh, bias=True)
        # per-head learning-rate gate: one Linear(D,1) per head, stacked (reference :91-106)
        self.learnable_ttt_lr_weight = nn.Parameter(torch.normal(0, 0.02, size=(NH, 1, D)))
        self.learnable_ttt_lr_bias = nn.Parameter(torch.zeros(NH, 1))
        # per-head LayerNorm affine of the inner loop (reference :108-112)
        self.ttt_norm_weight = nn.Parameter(torch.ones(NH, Fh))
        self.ttt_norm_bias = nn.Parameter(torch.zeros(NH, Fh))
        self.post_norm = nn.LayerNorm(D, eps=1e-6)
        self._perm_cache: Dict[tuple, tuple] = {}

    # -- initialisation -------------------------------------------------------------------------
    def init_weights(self):
        """Re-initialise after meta-device construction (reference :74-83)."""
        for lin in (self.wq, self.wk, self.wv, self.wo):
            nn.init.normal_(lin.weight, mean=0.0, std=0.02)
        self.post_norm.reset_parameters()
        nn.init.ones_(self.ttt_norm_weight)
        nn.init.zeros_(self.ttt_norm_bias)
        nn.init.normal_(self.learnable_ttt_lr_weight, mean=0.0, std=0.02)
        nn.init.zeros_(self.learnable_ttt_lr_bias)

    def init_device_mesh(self, tp_mesh):
        """Head-sharded tensor parallelism (reference :114-131 + ``mlp_tk.py``:297-343 ``local_map`` over heads +
        ``parallelisms.py``:106-152): every rank of ``tp_mesh`` runs the projections, RoPE, the scan and its backward for ITS
        ``NH / T`` heads over the full sequence; the head outputs are all-gathered along the feature dimension before
        ``post_norm`` / ``wo`` (the reference's one all-gather of ``[B, L, D]``, :329).  MI355X-first form: no DTensor -
        parameters stay whole on every rank (14.5 GB in bf16 is nothing next to 288 GB), each rank slices its heads' rows
        (``forward_heads``), the collective is explicit (RCCL all-gather) and differentiable, and ``tp_sync_gradients()``
        sums the per-head parameter gradients over the group after backward (a rank only produces its own heads' rows).
        ``tp_mesh``: a 1-D ``DeviceMesh`` (the reference's argument) or a process group."""
        from ttt_amd.infra.sequence_parallel import SeqParallel
        group = tp_mesh.get_group() if hasattr(tp_mesh, "get_group") else tp_mesh
        self._tp = SeqParallel(group)
        self._tp.head_range(self.num_heads)          # raises if the heads do not divide
        self.tp_mesh = tp_mesh
        TkMLP.sharded_mode = HipLinear.sharded_mode = True        # (reference class attributes; informational here)

    def tp_sync_gradients(self):
        """After backward under ``init_device_mesh``: sum the gradients of the per-head parameters over the tensor-parallel
        group (every rank holds its own heads' rows, zeros elsewhere); the replicated tail (``post_norm``, ``wo``) sees the
        same gathered activations on every rank and needs nothing.

        Call it ONCE per optimizer step, after the last micro-batch's backward (calling it per micro-batch would re-sum what
        was already summed), and only on plain-tensor gradients: under FSDP2 a gradient is a sharded DTensor after the
        reduce-scatter and this flat all-reduce would mix shards - that combination is refused here (the reference composes
        TP with FSDP through DTensor placements, parallelisms.py:106-175; this explicit-collective form does not)."""
        import torch.distributed as dist
        if getattr(self, "_tp", None) is None or self._tp.size == 1:
            return
        params = dict(self.named_parameters())
        grads = [params[n].grad for n in self._HEAD_SLICED if n in params and params[n].grad is not None]
        for g in grads:
            if type(g) is not torch.Tensor:
                raise RuntimeError(f"tp_sync_gradients: gradient of type {type(g).__name__} (FSDP-sharded?): the head-sharded tensor "
                                   "parallelism of this layer keeps whole parameters per rank and does not compose with FSDP2")
        if grads:
            flat = torch.cat([g.reshape(-1).float() for g in grads])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._tp.group)
            o = 0
            for g in grads:
                g.copy_(flat[o:o + g.numel()].view_as(g))
                o += g.numel()

    # -- pieces of process_input ------------------------------------------------------------------
    def get_qkv_projections(self, hidden_states):
        return linear3(self.wq, self.wk, self.wv, hidden_states)

    def get_eta(self, X):
        """Per-token inner-loop learning rate ``base_lr * sigmoid(x.w_h + b_h) / head_dim`` as
        ``[B, NH, NC, 1, CS]`` from mini-batched ``X [B, NC, CS, D]`` (reference :143-155)."""
        w = self.learnable_ttt_lr_weight.squeeze(1)                    # [NH, D]
        logits = F.linear(X, w.to(X.dtype), self.learnable_ttt_lr_bias.reshape(-1).to(X.dtype))   # [B,NC,CS,NH]
        lr = torch.sigmoid(logits).permute(0, 3, 1, 2).unsqueeze(3)    # [B,NH,NC,1,CS]
        return self.ttt_base_lr * lr / self.head_dim

    def _perm(self, meta: SequenceMetadata, L: int, device):
        key = (meta.num_chunks, meta.text_length, meta.init_offset, meta.base_offset, L, str(device))
        hit = self._perm_cache.get(key)
        if hit is None:
            p = scene_permutation(meta, L)
            inv = torch.empty_like(p)
            inv[p] = torch.arange(L)
            hit = (p.to(device), inv.to(device))
            self._perm_cache[key] = hit
        return hit

    def interleave(self, x: torch.Tensor, seq_metadata: SequenceMetadata):
        """[B,NH,NC,CS,*] tokens in [texts, video] order -> scene-interleaved order (reference :157-186)."""
        B, H, NC, C, HD = x.shape
        p, _ = self._perm(seq_metadata, NC * C, x.device)
        return x.reshape(B, H, NC * C, HD).index_select(2, p).reshape(B, H, NC, C, HD)

    def undo_interleave(self, x: torch.Tensor, seq_metadata: SequenceMetadata):
        """[B,L,D] scene-interleaved -> [texts, video] order (reference :188-217)."""
        _, inv = self._perm(seq_metadata, x.shape[1], x.device)
        return x.index_select(1, inv)

    def ln_reconstruction_target(self, XV, XK):
        """``gamma_h * LN(XV - XK) + beta_h + XK`` with *unbiased* std and eps added to the std
        (reference :219-235).  XV/XK are [B, L, NH, F]; computed in fp32, returned in XV's dtype."""
        d = XV.float() - XK.float()
        mean = d.mean(dim=-1, keepdim=True)
        std = d.std(dim=-1, keepdim=True)
        d = (d - mean) / (std + 1e-8)
        out = self.ttt_norm_weight.float()[None, None] * d + self.ttt_norm_bias.float()[None, None] + XK.float()
        return out.to(XV.dtype)

    def reshape_to_mini_batch(self, X, XQ, XK, XV):
        """[B,L,NH,F] -> [B,NH,NC,CS,F] and X -> [B,NC,CS,D] (reference :237-250)."""
        B, L = X.shape[:2]
        NC, CS = L // self.mini_batch_size, self.mini_batch_size
        mb = lambda t: t.transpose(1, 2).reshape(B, self.num_heads, NC, CS, self.head_dim)
        return X.reshape(B, NC, CS, self.width), mb(XQ), mb(XK), mb(XV)

    def process_input(self, hidden_states, freqs_cis, seq_metadata: SequenceMetadata):
        """Projections, L2-norm, 3-D RoPE on video tokens, LN reconstruction target, mini-batching,
        eta, scene interleave (reference :252-306).  Returns XQ/XK/XV [B,NH,NC,CS,F] and eta -
        ``[B,NH,NC,1,CS]`` on the kernel path, the full ``[B,NH,NC,CS,CS]`` tile otherwise."""
        n_text = seq_metadata.seq_text_length
        B, L = hidden_states.shape[:2]
        CS = self.mini_batch_size
        XQ, XK, XV = self.get_qkv_projections(hidden_states)
        XQ = F.normalize(XQ.view(B, L, -1, self.head_dim), p=2, dim=-1)
        XK = F.normalize(XK.view(B, L, -1, self.head_dim), p=2, dim=-1)
        XV = XV.view(B, L, -1, self.head_dim)

        rq, rk = apply_rotary_emb(XQ[:, n_text:], XK[:, n_text:], freqs_cis=freqs_cis)
        XQ = torch.cat((XQ[:, :n_text], rq), dim=1)
        XK = torch.cat((XK[:, :n_text], rk), dim=1)
        XV = self.ln_reconstruction_target(XV, XK)

        X, XQ, XK, XV = self.reshape_to_mini_batch(hidden_states, XQ, XK, XV)
        eta_row = self.get_eta(X) / CS                                   # [B,NH,NC,1,CS]  (reference :288)
        NC = eta_row.shape[2]

        if seq_metadata.is_multiscene:
            XQ = self.interleave(XQ, seq_metadata)
            XK = self.interleave(XK, seq_metadata)
            XV = self.interleave(XV, seq_metadata)
            # The reference interleaves the ROWS of the tiled eta (:294): the token landing at row i of
            # new mini-batch m brings the lr row of the mini-batch it came from.
            p, _ = self._perm(seq_metadata, L, XQ.device)
            src_mb = torch.div(p, CS, rounding_mode="floor")             # [L] original mini-batch of each slot
            if self.use_kernel:
                eta = eta_row.index_select(2, src_mb[CS - 1::CS])        # last row of every new tile
            else:
                eta = eta_row.squeeze(3).index_select(2, src_mb).reshape(B, self.num_heads, NC, CS, CS)
        else:
            eta = eta_row if self.use_kernel else eta_row.expand(-1, -1, -1, CS, -1)
        return {"XQ": XQ, "XK": XK, "XV": XV, "eta": eta}

    def ttt(self, inputs):
        raise NotImplementedError("ttt method must be implemented in TTTBase subclasses.")

    def forward(self, hidden_states: torch.Tensor, freqs_cis: torch.Tensor, seq_metadata: SequenceMetadata, reverse: bool = False,
                heads_only: bool = False):
        """``heads_only=True`` stops before ``post_norm`` / ``wo`` and returns the scan output ``[B, L, NH*F]`` in the
        original token order: the head-sharded half of the layer under sequence parallelism (see ``forward_heads``)."""
        assert hidden_states.size(1) % self.config.mini_batch_size == 0, "Sequence len must be multiple of mini batch size."
        tp = getattr(self, "_tp", None)
        if tp is not None and tp.size > 1 and not heads_only:
            from ttt_amd.infra.sequence_parallel import gather_features, replicated_input
            h0, h1 = tp.head_range(self.num_heads)
            x = replicated_input(hidden_states, tp)          # backward: the ranks' shares of d(input) are summed
            y = gather_features(self.forward_heads(x, freqs_cis, seq_metadata, reverse, h0, h1), tp)    # [B, L, NH*F]
            return self.wo(self.post_norm(y))
        if self.use_kernel and self.use_fused and fused_available(hidden_states, self.head_dim) and not freqs_cis.is_complex():
            return self._forward_fused(hidden_states, freqs_cis, seq_metadata, reverse, heads_only)
        if reverse:
            hidden_states = flip_sequence(hidden_states, seq_metadata)
        y = self.ttt(self.process_input(hidden_states, freqs_cis, seq_metadata))
        if not heads_only:
            y = self.wo(self.post_norm(y))
        if seq_metadata.is_multiscene:
            y = self.undo_interleave(y, seq_metadata)
        return flip_sequence(y, seq_metadata) if reverse else y

    _HEAD_SLICED = ("wq.weight", "wq.bias", "wk.weight", "wk.bias", "wv.weight", "wv.bias", "learnable_ttt_lr_weight",
                    "learnable_ttt_lr_bias", "ttt_norm_weight", "ttt_norm_bias", "W1", "b1", "W2", "b2")

    def forward_heads(self, hidden_states, freqs_cis, seq_metadata: SequenceMetadata, reverse: bool, h0: int, h1: int):
        """The layer restricted to heads ``[h0, h1)`` up to (not including) ``post_norm``: ``[B, L, D] -> [B, L, (h1-h0)*F]``.
        Every per-head parameter is sliced along its head dimension and the unchanged module code runs on the slice
        (inference only: sequence-parallel sampling, ``ttt_amd/infra/sequence_parallel.py``).  The reference gets the same
        head-locality from DTensor placements (``ttt_layer.py``:114-131, ``mlp_tk.py``:297-343)."""
        Fh = self.head_dim
        params = dict(self.named_parameters())
        sliced = {}
        for name in self._HEAD_SLICED:
            if name in params:
                p = params[name]
                sliced[name] = p[h0 * Fh:h1 * Fh] if name.startswith(("wq.", "wk.", "wv.")) else p[h0:h1]
        nh = self.num_heads
        self.num_heads = h1 - h0
        try:
            return torch.func.functional_call(self, sliced, (hidden_states, freqs_cis, seq_metadata, reverse), {"heads_only": True})
        finally:
            self.num_heads = nh

    # -- fused path: HIP pre/post kernels around the scan (bf16 on a HIP device) ------------------------------
    def _token_maps(self, meta: SequenceMetadata, L: int, device, reverse: bool):
        """int32 maps for the fused kernels: scan position t reads token ``src[t]`` of the input sequence and is
        rotated with ``rope[pos[t]]`` (-1: text token); ``rev`` = reversal map or None."""
        key = ("maps", meta.num_chunks, meta.text_length, meta.seq_text_length, meta.init_offset, meta.base_offset, L, str(device), reverse)
        hit = self._perm_cache.get(key)
        if hit is None:
            n_text = meta.seq_text_length
            seq = scene_permutation(meta, L) if meta.is_multiscene else torch.arange(L)   # scan pos -> index in the (reversed) sequence
            pos = torch.where(seq >= n_text, seq - n_text, torch.full_like(seq, -1))
            rev = reversal_map(meta, L) if reverse else None
            src = rev[seq] if reverse else seq
            to = lambda t: None if t is None else t.to(device=device, dtype=torch.int32).contiguous()
            hit = (to(src), to(pos), None if rev is None else rev.to(device))
            hit[1]._ttt_max_pos = int(pos.max()) + 1          # read by the binding's RoPE-table bound check (no device sync)
            self._perm_cache[key] = hit
        return hit

    def _forward_fused(self, x, freqs_cis, meta: SequenceMetadata, reverse: bool, heads_only: bool = False):
        B, L, _ = x.shape
        src, pos, rev = self._token_maps(meta, L, x.device, reverse)
        rope = freqs_cis if freqs_cis.dtype == torch.float32 and freqs_cis.is_contiguous() else freqs_cis.float().contiguous()
        eta = self._eta_rows(x, rev, meta, L)
        parts = self._pipeline_plan(x, meta, L, reverse, heads_only)
        if parts is None:
            return self._fused_nodes(x, rope, src, pos, eta, heads_only, piped=False)
        # Pipelined forward (ttt_amd/models/ssm/pipeline.py): every kernel and GEMM of the forward runs in the pre-pass, part by part,
        # the scan on a side stream beside the neighbouring parts' projections; the autograd Functions then TAKE its results.
        import test_time_training as ext
        from ttt_amd.models.ssm import pipeline
        with torch.no_grad():
            st = [self._per_batch(p, B) for p in (self.W1, self.b1, self.W2, self.b2)]
            last_eta = eta.detach().to(torch.bfloat16)[:, :, :, -1, :, None].contiguous()
            w32 = self.ttt_norm_weight.detach().to(torch.float32).contiguous()
            b32 = self.ttt_norm_bias.detach().to(torch.float32).contiguous()
            res = pipeline.prepass(ext, x.detach(), self.wq, self.wk, self.wv, self.wo, self.post_norm, w32, b32, rope, src, pos,
                                   getattr(pos, "_ttt_max_pos", None), self.num_heads, st, last_eta,
                                   self._group_size(L // self.mini_batch_size), parts)
        with pipeline.injecting(res):
            return self._fused_nodes(x, rope, src, pos, eta, heads_only, piped=True)

    def _pipeline_plan(self, x, meta, L, reverse, heads_only):
        """the parts of a pipelined forward, or None when this call runs as one piece: not TTT-MLP on the MFMA scan at mini-batches of
        64, fewer than two checkpoint groups per part, a head shard, a re-materialisation that gets its scan result handed back"""
        from ttt_amd.infra import remat_cache
        n = self.pipeline_parts
        CS = self.mini_batch_size
        if n < 2 or heads_only or not isinstance(self, TTTMLP) or CS != 64 or remat_cache.replaying("scan"):
            return None
        if not linear3_applies(self.wq, self.wk, self.wv, x) or not linear3_applies(self.wo, self.wo, self.wo, x):
            return None                    # (DTensor parameters, autocast: the pre-pass's raw GEMMs would not be the modules' arithmetic)
        NC = L // CS
        G = self._group_size(NC)
        if self.pipeline_parts_auto:
            n = max(n, min(8, -(-NC // G) // 40))
        if -(-NC // G) < 8:                    # (rounds 5 / 6: a scan of fewer than eight checkpoint groups runs as one piece)
            return None
        n = min(n, -(-NC // G) // 2)           # at least two checkpoint groups per part on average
        import test_time_training as ext
        if ext.resolved_impl(x.shape[0], self.num_heads, NC, CS, self.head_dim, G, torch.bfloat16, mlp=True, backward=False) != "mfma":
            return None
        key = ("parts", n, meta.num_chunks, meta.text_length, meta.seq_text_length, meta.init_offset, meta.base_offset, L, reverse)
        hit = self._perm_cache.get(key)
        if hit is None:
            from ttt_amd.models.ssm.pipeline import plan_parts
            seq = scene_permutation(meta, L) if meta.is_multiscene else None      # scan position -> index in the (reversed) sequence
            if reverse:
                r = reversal_map(meta, L)
                seq = r if seq is None else r[seq]                                 # -> token of the input sequence (= the maps' src)
            hit = self._perm_cache[key] = plan_parts(seq, L, CS, G, n)
        return hit

    def _eta_rows(self, x, rev, meta, L):
        """eta (tiny: [B, L, NH]): per-token learning rate in the order of the (reversed) sequence, then the reference's tile
        bookkeeping - the kernels read the row of the mini-batch the LAST token of a tile came from: [B,NH,NC,1,CS]"""
        B = x.shape[0]
        Fh, CS = self.head_dim, self.mini_batch_size
        NC = L // CS
        w = self.learnable_ttt_lr_weight.squeeze(1)
        lr = torch.sigmoid(F.linear(x, w.to(x.dtype), self.learnable_ttt_lr_bias.reshape(-1).to(x.dtype)))      # [B, L, NH]
        if rev is not None:
            lr = lr.index_select(1, rev)
        eta = (self.ttt_base_lr * lr.view(B, NC, CS, self.num_heads).permute(0, 3, 1, 2).unsqueeze(3) / Fh) / CS    # [B,NH,NC,1,CS]
        if meta.is_multiscene:
            p, _ = self._perm(meta, L, x.device)
            eta = eta.index_select(2, torch.div(p, CS, rounding_mode="floor")[CS - 1::CS])
        return eta

    def _fused_nodes(self, x, rope, src, pos, eta, heads_only, piped):
        """the autograd nodes of the fused path: projections -> pre + scan -> post-norm -> output projection"""
        B, L, _ = x.shape
        NH, Fh, CS = self.num_heads, self.head_dim, self.mini_batch_size
        NC = L // CS
        mb = lambda t: t.view(B, NH, NC, CS, Fh)
        XQr, XKr, XVr = linear3(self.wq, self.wk, self.wv, x, always=piped)
        if isinstance(self, TTTMLP):       # pre + scan as one autograd node: only the raw projections stay alive for backward
            st = [self._per_batch(p, B) for p in (self.W1, self.b1, self.W2, self.b2)]
            Y = FusedPreScanMLP.apply(XQr, XKr, XVr, self.ttt_norm_weight, self.ttt_norm_bias, rope, src, pos, NH, *st, eta,
                                      self._group_size(NC))
        else:
            XQ, XK, XV = FusedPre.apply(XQr, XKr, XVr, self.ttt_norm_weight, self.ttt_norm_bias, rope, src, pos, NH)
            Y = self.ttt_raw({"XQ": mb(XQ), "XK": mb(XK), "XV": mb(XV), "eta": eta})                              # [B,NH,NC,CS,F]
        if heads_only:     # scan position t holds token src[t]: back to the original order, [B, L, NH*F]
            out = torch.empty(B, L, NH, Fh, device=Y.device, dtype=Y.dtype)
            out.index_copy_(1, src.long(), Y.reshape(B, NH, L, Fh).transpose(1, 2))
            return out.view(B, L, NH * Fh)
        y = FusedPost.apply(Y.reshape(B, NH, L, Fh), self.post_norm.weight, self.post_norm.bias, src, self.post_norm.eps)
        if self.pipeline_parts >= 2 and isinstance(self, TTTMLP) and linear3_applies(self.wo, self.wo, self.wo, y):
            # the output projection as a node of our own whenever this layer MAY run pipelined - also in the calls that do not (a
            # re-materialisation that gets its scan result handed back): torch's checkpoint wants the recomputation to save what
            # the forward pass saved, node for node; the node takes the pre-pass's result when there is one
            from ttt_amd.models.ssm.pipeline import InjectedLinear
            return InjectedLinear.apply(y, self.wo.weight, self.wo.bias)
        return self.wo(y)

    # helpers shared by the two variants
    def _group_size(self, num_mini_batch: int) -> int:
        return min(max(self.config.scan_checkpoint_group_size, 1), num_mini_batch)   # reference :368,:439

    @staticmethod
    def _per_batch(p: torch.Tensor, B: int) -> torch.Tensor:
        return p.unsqueeze(0).expand(B, *p.shape)   # kernels read it; no need to materialise B copies


class TTTLinear(TTTBase):
    def __init__(self, config: ModelConfig, use_kernel: bool = True):
        super().__init__(config)
        self.W1 = nn.Parameter(torch.normal(0, 0.02, size=(self.num_heads, self.head_dim, self.head_dim)))
        self.b1 = nn.Parameter(torch.zeros(self.num_heads, 1, self.head_dim))
        self.use_kernel = use_kernel

    def init_weights(self):
        super().init_weights()
        nn.init.normal_(self.W1, mean=0.0, std=0.02)
        nn.init.zeros_(self.b1)

    def ttt_raw(self, inputs):
        """Scan on the HIP kernels, result left in the kernel layout [B,NH,NC,CS,F]."""
        B, _, NC, _, _ = inputs["XV"].shape
        return HipLinear.apply(self.ttt_norm_weight, self.ttt_norm_bias, self._per_batch(self.W1, B), self._per_batch(self.b1, B),
                               inputs["XQ"], inputs["XV"], inputs["XK"], inputs["eta"], self._group_size(NC))

    def ttt(self, inputs):
        B, _, NC, CS, _ = inputs["XV"].shape
        W1, b1 = self._per_batch(self.W1, B), self._per_batch(self.b1, B)
        G = self._group_size(NC)
        if self.use_kernel:
            out = HipLinear.apply(self.ttt_norm_weight, self.ttt_norm_bias, W1, b1, inputs["XQ"], inputs["XV"],
                                  inputs["XK"], inputs["eta"], G)
            out = out.permute(0, 2, 3, 1, 4)
        else:
            out = ttt_linear(inputs["XK"], inputs["XQ"], inputs["XV"], inputs["eta"], self.ttt_norm_weight,
                             self.ttt_norm_bias, W1, b1, G)
        return out.reshape(B, NC * CS, -1)          # all heads of this module (a head shard under sequence parallelism)


class TTTMLP(TTTBase):
    def __init__(self, config: ModelConfig, use_kernel: bool = True):
        super().__init__(config)
        NH, Fh = self.num_heads, self.head_dim
        self.W1 = nn.Parameter(torch.normal(0, 0.02, size=(NH, Fh, 4 * Fh)))
        self.b1 = nn.Parameter(torch.zeros(NH, 1, 4 * Fh))
        self.W2 = nn.Parameter(torch.normal(0, 0.02, size=(NH, 4 * Fh, Fh)))
        self.b2 = nn.Parameter(torch.zeros(NH, 1, Fh))
        self.use_kernel = use_kernel

    def init_weights(self):
        super().init_weights()
        nn.init.normal_(self.W1, mean=0.0, std=0.02)
        nn.init.zeros_(self.b1)
        nn.init.normal_(self.W2, mean=0.0, std=0.02)
        nn.init.zeros_(self.b2)

    def ttt_raw(self, inputs):
        """Scan on the HIP kernels, result left in the kernel layout [B,NH,NC,CS,F]."""
        B, _, NC, _, _ = inputs["XV"].shape
        st = [self._per_batch(p, B) for p in (self.W1, self.b1, self.W2, self.b2)]
        return TkMLP.apply(self.ttt_norm_weight, self.ttt_norm_bias, *st, inputs["XQ"], inputs["XV"], inputs["XK"],
                           inputs["eta"], self._group_size(NC))

    def ttt(self, inputs):
        B, _, NC, CS, _ = inputs["XV"].shape
        st = [self._per_batch(p, B) for p in (self.W1, self.b1, self.W2, self.b2)]
        G = self._group_size(NC)
        if self.use_kernel:
            out = TkMLP.apply(self.ttt_norm_weight, self.ttt_norm_bias, *st, inputs["XQ"], inputs["XV"], inputs["XK"],
                              inputs["eta"], G)
            out = out.permute(0, 2, 3, 1, 4)
        else:
            out = ttt_mlp(inputs["XK"], inputs["XQ"], inputs["XV"], inputs["eta"], self.ttt_norm_weight,
                          self.ttt_norm_bias, *st, G)
        return out.reshape(B, NC * CS, -1)          # all heads of this module (a head shard under sequence parallelism)
