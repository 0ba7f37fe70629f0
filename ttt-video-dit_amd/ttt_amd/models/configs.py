"""Model hyper-parameters.  Mirrors the fields and presets of the reference's
``ttt/models/configs.py:9-125`` (ModelConfig) without its dependency on the job-config system
(out of scope): remat / ssm knobs are plain constructor arguments here."""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, fields

_SIZES = {
    # ttt/models/configs.py:57-69
    "debug": dict(model_dim=512, num_heads=8, num_layers=6),
    "5B": dict(model_dim=3072, num_heads=48, num_layers=42, text_dim=4096),
}
_DURATIONS = {"3sec": 13, "9sec": 37, "18sec": 73, "30sec": 121, "63sec": 253}  # configs.py:71-87 (latent frames)


@dataclass
class ModelConfig:
    model_dim: int
    num_heads: int
    num_layers: int

    ssm_layer: str = "ttt_mlp"          # "ttt_mlp" | "ttt_linear"
    layer_norm_eps: float = 1e-6

    # TTT
    mini_batch_size: int = 64
    ttt_base_lr: float = 0.1
    rope_theta: float = 10000
    scan_checkpoint_group_size: int = 16

    adapter_method: str = "none"        # none | sft | qkvo

    # network
    time_embed_dim: int = 512
    sigma_interval: int = 1000
    patch_size: int = 2
    in_channels: int = 16
    out_channels: int = 16
    scale_factor: float = 1.0

    # RoPE / latent geometry
    latent_height: int = 30
    latent_width: int = 45
    compressed_num_frames: int = 13
    theta: float = 10000

    text_dim: int = 512

    # local attention / gating
    gating_alpha_init: float = 0.1
    attn_length: int = 12
    prefix_temporal_length: int = 1

    # activation re-materialisation
    remat_transformer_layer_group_size: int = 1
    # MI355X (288 GB HBM3E): the first `remat_free_layers` transformer layers keep their activations instead of being
    # re-materialised in backward (0 = the reference's behaviour: every layer group is checkpointed, dit.py:493-499).
    remat_free_layers: int = 0
    # MI355X: kernel outputs a re-materialised layer keeps instead of recomputing them ("attn", "scan"; () = reference behaviour)
    remat_keep: tuple = ()
    remat_keep_layers: object = None      # int: only the first N re-materialised layers keep their kernel outputs (None: all)
    remat_keep_limits: object = None      # dict kind -> N: that kind only in the first N re-materialised layers
    remat_forward_ssm: bool = False
    remat_reverse_ssm: bool = False
    remat_attention: bool = False
    remat_mlp: bool = False
    remat_seq_modeling_block: bool = False
    shard_transformer_inputs: bool = False

    @classmethod
    def get_preset(cls, preset: str, video_length: str, **overrides) -> "ModelConfig":
        if preset not in _SIZES:
            raise ValueError("Pre-defined config not found.")
        if video_length not in _DURATIONS:
            raise ValueError("Pre-defined video duration config not found.")
        kw = dict(_SIZES[preset], compressed_num_frames=_DURATIONS[video_length])
        known = {f.name for f in fields(cls)}
        for k, v in overrides.items():
            if k not in known:
                raise ValueError(f"unknown ModelConfig field {k!r}")
            kw[k] = v
        return cls(**kw)

    @property
    def head_dim(self) -> int:
        return self.model_dim // self.num_heads

    def __str__(self) -> str:
        return json.dumps(asdict(self), indent=4)
