#!/usr/bin/env python
"""Times THE REFERENCE's own code - ttt/models/cogvideo/dit.py:TransformerLayer with the PyTorch ops path (use_kernel=False ->
ttt/models/ssm/ops/ttt_mlp.py:9-99 under ttt/models/ssm/utils.py:scan) - at the 5B geometry on the host CPU, forward + backward of ONE
layer, on the same samples as bench.py's `cpu_baseline` leg (one scene of f latent frames + text tokens): f = 1 (L = 1 408, what the
leg's port usually times), f = 4 (L = 5 440) and, with --full, f = 13 (L = 18 048 = the whole 3-second attention segment).
Build container only (needs /root/reference); the result is recorded in BASELINE.md and bounds the extrapolation that
`cpu_baseline.sample` states (round-3 verdict, missing #4 / next #8).  fp32, eager, all cores.

    TORCHDYNAMO_DISABLE=1 python tools/ref_cpu_layer_baseline.py [--full]
"""
import os
import sys
import time
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
import torch

sys.modules["wandb"] = types.ModuleType("wandb")
import tomli

sys.modules["tomllib"] = tomli
sys.path.insert(0, "/root/reference")
from ttt.models.cogvideo.dit import TransformerLayer  # noqa: E402
from ttt.models.cogvideo.utils import SequenceMetadata  # noqa: E402
from ttt.models.configs import ModelConfig  # noqa: E402

TPF = 30 * 45
torch.set_num_threads(os.cpu_count())
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
print(f"cores {os.cpu_count()}  cpu {cpu}  torch {torch.__version__}", flush=True)


def timed(frames, n_text):
    torch.manual_seed(0)
    cfg = ModelConfig.get_preset("5B", "3sec")
    cfg.ssm_layer, cfg.adapter_method, cfg.compressed_num_frames = "ttt_mlp", "sft", frames
    layer = TransformerLayer(cfg)
    for m in layer.modules():
        if hasattr(m, "use_kernel"):
            m.use_kernel = False
        if hasattr(m, "init_freqs"):
            m.init_freqs()
    if hasattr(layer.seq_modeling_block.ssm.ttt, "init_weights"):
        layer.seq_modeling_block.ssm.ttt.init_weights()
    n_vid = frames * TPF
    meta = SequenceMetadata(text_length=n_text, seq_text_length=n_text, num_frames=frames, num_chunks=1, tokens_per_frame=TPF,
                            latent_height=60, latent_width=90, t_emb=torch.randn(1, cfg.time_embed_dim))
    vid = torch.randn(1, n_vid, cfg.model_dim, requires_grad=True)
    txt = torch.randn(1, n_text, cfg.model_dim, requires_grad=True)
    t0 = time.perf_counter()
    v, t = layer(vid, txt, meta)
    t1 = time.perf_counter()
    (v.square().mean() + t.square().mean()).backward()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, n_vid, n_vid + n_text


samples = [(1, 58), (4, 40)] + ([(13, 498)] if "--full" in sys.argv else [])
timed(1, 58)                       # warm-up
base = None
for f, nt in samples:
    fw, bw, n_vid, L = timed(f, nt)
    per_tok = (fw + bw) / n_vid
    base = base or per_tok
    print(f"reference TransformerLayer (5B, ttt_mlp ops path), {f:2d} frame(s) + {nt} text tokens, L = {L}: forward {fw:.2f} s, backward {bw:.2f} s, "
          f"{1e3 * per_tok:.3f} ms per video token and layer ({per_tok / base:.2f} x the 1-frame sample) -> {n_vid / ((fw + bw) * 42):.2f} video-tok/s for 42 layers", flush=True)
