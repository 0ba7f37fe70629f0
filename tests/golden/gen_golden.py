"""Generate golden vectors by EXECUTING THE REFERENCE ITSELF (build container only).

    TORCHDYNAMO_DISABLE=1 python tests/golden/gen_golden.py

Imports /root/reference (read-only) with the two stubs of SURVEY.md section 8c, runs the
reference's PyTorch ops path (ttt/models/ssm/ops/*.py through TTTMLP/TTTLinear with
use_kernel=False, ttt/models/cogvideo/dit.py) on seeded inputs and stores inputs, outputs and
torch.autograd gradients as small .pt fixtures next to this file.  /root/reference does not
exist on the GPU box, so tests only ever read the fixtures.

Nothing from the reference is copied into the repo: only numbers are saved.
"""
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

sys.modules["wandb"] = types.ModuleType("wandb")
import tomli  # noqa: E402

sys.modules["tomllib"] = tomli
sys.path.insert(0, "/root/reference")

from ttt.models.configs import ModelConfig  # noqa: E402
from ttt.models.cogvideo.dit import DiffusionTransformer  # noqa: E402
from ttt.models.cogvideo.utils import SequenceMetadata  # noqa: E402
from ttt.models.ssm.ops import ttt_linear, ttt_mlp  # noqa: E402
from ttt.models.ssm.ttt_layer import TTTWrapper  # noqa: E402

from oracle.ttt_oracle import make_inputs  # noqa: E402  (input generator only)


def op_case(kind, B, NH, NC, CS, Fd, seed, dtype, identical_rows=True, G=2):
    """Reference ops path at op level: forward + autograd of every input."""
    d = make_inputs(kind, B, NH, NC, CS, Fd, seed=seed, dtype=dtype, identical_rows=identical_rows)
    leaf = {k: v.clone().requires_grad_(True) for k, v in d.items() if k != "dOut"}
    tile = lambda w: torch.tile(w.unsqueeze(0), dims=(B, 1, 1, 1))
    states = {k: tile(leaf[k]) for k in ("W1", "b1", "W2", "b2") if k in leaf}
    for s in states.values():
        s.retain_grad()
    if kind == "mlp":
        out = ttt_mlp(leaf["XK"], leaf["XQ"], leaf["XV"], leaf["eta"], leaf["ln_w"], leaf["ln_b"],
                      states["W1"], states["b1"], states["W2"], states["b2"], G)
    else:
        out = ttt_linear(leaf["XK"], leaf["XQ"], leaf["XV"], leaf["eta"], leaf["ln_w"], leaf["ln_b"],
                         states["W1"], states["b1"], G)
    # reference returns [B,NC,CS,NH,F]; store in kernel layout [B,NH,NC,CS,F]
    out_k = out.permute(0, 3, 1, 2, 4).contiguous()
    out_k.backward(d["dOut"])
    g = {"XQW": out_k.detach()}
    for k in ("XQ", "XK", "XV", "eta", "ln_w", "ln_b"):
        g["d" + k] = leaf[k].grad.detach()
    for k, s in states.items():
        g["d" + k + "_states"] = s.grad.detach()   # per-batch state gradient [B,NH,...]
    # inputs are NOT stored: tests regenerate them with oracle.make_inputs(**gen) and verify the checksums
    return {"kind": kind, "G": G, "identical_rows": identical_rows,
            "gen": dict(kind=kind, B=B, NH=NH, NC=NC, CS=CS, Fd=Fd, seed=seed, identical_rows=identical_rows),
            "dtype": str(dtype).split(".")[-1],
            "input_checksums": {k: float(v.double().abs().sum()) for k, v in d.items()}, "ref": g}


def module_case(ssm_layer, multiscene, seed, dtype=torch.float32):
    """TTTWrapper fwd/bwd through the reference module (use_kernel=False)."""
    torch.manual_seed(seed)
    if not multiscene:
        # BASELINE.json config 1: batch=1 seq=256 d_model=128, NH=2, CS=64
        cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=64, latent_height=4,
                          latent_width=8, compressed_num_frames=8, ssm_layer=ssm_layer,
                          scan_checkpoint_group_size=2)
        meta = dict(text_length=0, seq_text_length=0, num_frames=8, num_chunks=1, tokens_per_frame=32,
                    latent_height=4, latent_width=8)
        L = 256
    else:
        # 2 scenes, 5 frames of 4x4 latent (16 tok/frame), text 8 per scene: L = 16 + 80 = 96, CS=16.
        # Scene offsets (56, 40) are NOT multiples of CS, so interleave mixes mini-batches: eta rows of a
        # tile differ (SURVEY.md hazard C2), as in the real 9 s config (text 502, 1350 tokens/frame, CS 64).
        cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=16, latent_height=4,
                          latent_width=4, compressed_num_frames=5, ssm_layer=ssm_layer,
                          scan_checkpoint_group_size=4)
        meta = dict(text_length=8, seq_text_length=16, num_frames=5, num_chunks=2, tokens_per_frame=16,
                    latent_height=4, latent_width=4)
        L = 96
    m = TTTWrapper(cfg)  # NOT .to(dtype): that would cast the complex RoPE table to real
    m.ttt.init_weights()
    with torch.no_grad():  # make LN params / biases non-trivial so their gradients are exercised
        m.ttt.ttt_norm_weight.add_(0.1 * torch.randn_like(m.ttt.ttt_norm_weight))
        m.ttt.ttt_norm_bias.add_(0.1 * torch.randn_like(m.ttt.ttt_norm_bias))
        m.ttt.b1.add_(0.01 * torch.randn_like(m.ttt.b1))
        m.ttt.learnable_ttt_lr_bias.add_(0.1 * torch.randn_like(m.ttt.learnable_ttt_lr_bias))
    m.ttt.use_kernel = False
    x = torch.randn(1, L, 128, dtype=dtype, requires_grad=True)
    sm = SequenceMetadata(t_emb=torch.zeros(1, 512), **meta)
    if multiscene:
        sm.init_multiscene_offsets()
    y = m(x, sm)
    dy = torch.randn_like(y)
    y.backward(dy)
    return {"ssm_layer": ssm_layer, "multiscene": multiscene, "cfg": {k: getattr(cfg, k) for k in (
                "model_dim", "num_heads", "num_layers", "mini_batch_size", "latent_height", "latent_width",
                "compressed_num_frames", "ssm_layer", "scan_checkpoint_group_size", "ttt_base_lr", "rope_theta")},
            "meta": meta, "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()},
            "x": x.detach(), "dy": dy, "y": y.detach(), "dx": x.grad.detach(),
            "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}}


def dit_case(ssm_layer, scenes, seed, dtype=torch.float32):
    """Tiny DiffusionTransformer forward + grads through the reference model code."""
    torch.manual_seed(seed)
    frames = 1 + 2 * scenes  # attn_length=2, prefix 1
    cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=16, latent_height=8,
                      latent_width=8, compressed_num_frames=frames, ssm_layer=ssm_layer, text_dim=32,
                      time_embed_dim=64, attn_length=2, prefix_temporal_length=1, adapter_method="sft",
                      scan_checkpoint_group_size=2, remat_transformer_layer_group_size=1)
    m = DiffusionTransformer(cfg)  # fp32 default; never .to() (complex RoPE buffer)
    for mod in m.modules():
        if hasattr(mod, "use_kernel"):
            mod.use_kernel = False
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "ttt." in n and n.split(".")[-1] in ("W1", "W2", "learnable_ttt_lr_weight") or "wq" in n or "wk" in n or "wv" in n or "wo" in n:
                continue
            if p.ndim >= 2:
                p.normal_(0, 0.02)
            elif "bias" in n:
                p.normal_(0, 0.01)
        for layer in m.layers:
            layer.seq_modeling_block.ssm.ttt.init_weights()
    # tokens/frame = 16 ; text 16/scene -> L = 16*scenes + 16*frames, multiple of 16
    video = torch.randn(1, frames, 16, 8, 8, dtype=dtype)
    text = torch.randn(1, scenes, 16, 32, dtype=dtype)
    ts = torch.tensor([417])
    out = m(video, text, ts)
    dout = torch.randn_like(out)
    out.backward(dout)
    return {"ssm_layer": ssm_layer, "scenes": scenes,
            "cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__},
            "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()},
            "video": video, "text": text, "timesteps": ts, "out": out.detach(), "dout": dout,
            "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters()
                      if p.grad is not None and (p.numel() <= 20000 or k.endswith("ssm.ttt.wq.weight")
                                                 or k.endswith("seq_modeling_block.q.weight"))}}


def main():
    save = lambda name, obj: (torch.save(obj, os.path.join(HERE, name)), print("wrote", name))[1]
    # --- op level, fp64 pins (small) -------------------------------------------------------
    save("op_mlp_f64_cs16.pt", op_case("mlp", 1, 2, 5, 16, 64, seed=1, dtype=torch.float64, G=2))
    save("op_mlp_f64_cs64.pt", op_case("mlp", 1, 1, 3, 64, 64, seed=2, dtype=torch.float64, G=2))
    save("op_lin_f64_cs16.pt", op_case("linear", 1, 2, 5, 16, 64, seed=3, dtype=torch.float64, G=2))
    save("op_lin_f64_cs64.pt", op_case("linear", 1, 1, 3, 64, 64, seed=4, dtype=torch.float64, G=3))
    # hazard C2: non-identical eta rows (multi-scene) - dual form only
    save("op_mlp_f64_rows.pt", op_case("mlp", 1, 1, 3, 16, 64, seed=5, dtype=torch.float64, identical_rows=False))
    save("op_lin_f64_rows.pt", op_case("linear", 1, 1, 3, 16, 64, seed=6, dtype=torch.float64, identical_rows=False))
    # --- op level, fp32, kernel geometry (B=2 checks the per-batch LN-grad contract) ---------
    save("op_mlp_f32_b2.pt", op_case("mlp", 2, 2, 5, 64, 64, seed=7, dtype=torch.float32, G=4))
    save("op_lin_f32_b2.pt", op_case("linear", 2, 2, 5, 16, 64, seed=8, dtype=torch.float32, G=4))
    # --- module level ------------------------------------------------------------------------
    save("mod_mlp_cfg1.pt", module_case("ttt_mlp", False, seed=10))
    save("mod_lin_cfg1.pt", module_case("ttt_linear", False, seed=11))
    save("mod_mlp_multi.pt", module_case("ttt_mlp", True, seed=12))
    save("mod_lin_multi.pt", module_case("ttt_linear", True, seed=13))
    # --- DiT level ---------------------------------------------------------------------------
    save("dit_mlp_3scene.pt", dit_case("ttt_mlp", 3, seed=21))
    save("dit_lin_1scene.pt", dit_case("ttt_linear", 1, seed=22))


if __name__ == "__main__":
    main()
