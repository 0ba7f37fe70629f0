#!/usr/bin/env python
"""Times THE REFERENCE's own PyTorch path (ttt/models/ssm/ops through TTTWrapper, use_kernel=False) on the host CPU at BASELINE.json
configs[0] - the only reference timing that can be produced.  Runs only where /root/reference is mounted (the
build container); the result is recorded in BASELINE.md by hand.  fp32, eager (TORCHDYNAMO_DISABLE=1), all cores, 3 warm-ups + 20 timed.

    TORCHDYNAMO_DISABLE=1 python tools/ref_cpu_baseline.py
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
from ttt.models.cogvideo.utils import SequenceMetadata  # noqa: E402
from ttt.models.configs import ModelConfig  # noqa: E402
from ttt.models.ssm.ttt_layer import TTTWrapper  # noqa: E402

torch.set_num_threads(os.cpu_count())
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
print(f"cores {os.cpu_count()}  cpu {cpu}  torch {torch.__version__}")
for ssm in ("ttt_mlp", "ttt_linear"):
    torch.manual_seed(0)
    cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=64, latent_height=4, latent_width=8,
                      compressed_num_frames=8, ssm_layer=ssm)
    m = TTTWrapper(cfg)
    m.ttt.init_weights()
    m.ttt.use_kernel = False
    meta = SequenceMetadata(text_length=0, seq_text_length=0, num_frames=8, num_chunks=1, tokens_per_frame=32, latent_height=4,
                            latent_width=8, t_emb=torch.zeros(1, 512))
    x = torch.randn(1, 256, 128, requires_grad=True)
    fwd = bwd = 0.0
    for it in range(23):
        t0 = time.perf_counter()
        y = m(x, meta)
        t1 = time.perf_counter()
        y.sum().backward()
        t2 = time.perf_counter()
        if it >= 3:
            fwd += t1 - t0
            bwd += t2 - t1
    fwd, bwd = fwd / 20, bwd / 20
    print(f"{ssm:11s} reference ops path, batch 1 seq 256 d_model 128: forward {1e3 * fwd:.1f} ms  backward {1e3 * bwd:.1f} ms  "
          f"fwd+bwd {256 / (fwd + bwd):.0f} tokens/s")
