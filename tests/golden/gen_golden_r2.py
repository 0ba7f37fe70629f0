"""Round-2 golden vectors, again produced by EXECUTING THE REFERENCE (build container only):

    TORCHDYNAMO_DISABLE=1 python tests/golden/gen_golden_r2.py

  * ``mod_{mlp,lin}_multi_lastrow.pt``, ``mod_mlp_multi64_lastrow.pt`` - the *kernel contract* in multi-scene mode.  The
    reference's module (ttt_layer.py:252-334) is run with its ops path, but every eta tile is replaced by its LAST ROW before
    the op sees it - exactly what ``TkMLP`` hands to ttt-tk (mlp_tk.py:104-105: ``last_eta = eta[:, :, :, -1, :, None]``) and
    what the Triton kernels index themselves (linear_forward.py:90-101).  With identical rows the ops path's dual form is the
    primal recurrence the kernels implement, so these fixtures are the reference-pinned target of the fused HIP module path for
    every >= 9 s configuration (SURVEY.md hazard C2).  Forward and time-reversed (cogvideo/dit.py:247-263) passes.
  * ``dit_mlp64_1scene.pt`` - a small single-scene DiffusionTransformer with TTT-MLP at mini_batch_size = 64 (the training
    geometry of the MFMA kernels), fp32 reference run: target of the bf16 HIP path on the GPU.
  * ``cogvideox_loss.pt`` - ``CogVideoX.forward`` (cogvideo/model.py:46-66) under a 1-rank gloo group (SURVEY.md hazard C8)
    with a CPU noise generator: sampled indices, loss, and a few parameter gradients.

Nothing from the reference is copied into the repo: only numbers are saved.
"""
import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden as G0  # noqa: E402  (sets up the import stubs and sys.path for /root/reference)
import torch  # noqa: E402

import ttt.models.ssm.ttt_layer as ref_layer  # noqa: E402
from ttt.models.cogvideo.utils import SequenceMetadata  # noqa: E402
from ttt.models.configs import ModelConfig  # noqa: E402
from ttt.models.ssm.ttt_layer import TTTWrapper  # noqa: E402


def _last_row(eta):
    cs = eta.shape[3]
    return eta[:, :, :, -1:, :].expand(-1, -1, -1, cs, -1)


class kernel_contract:
    """Context: the reference module's ops calls see last-row eta tiles (what its kernels consume)."""

    def __enter__(self):
        self.saved = (ref_layer.ttt_mlp, ref_layer.ttt_linear)
        mlp, lin = self.saved
        ref_layer.ttt_mlp = lambda XK, XQ, XV, eta, *a: mlp(XK, XQ, XV, _last_row(eta), *a)
        ref_layer.ttt_linear = lambda XK, XQ, XV, eta, *a: lin(XK, XQ, XV, _last_row(eta), *a)

    def __exit__(self, *exc):
        ref_layer.ttt_mlp, ref_layer.ttt_linear = self.saved


def flip_like_reference(emb, n_text, num_chunks, multiscene):
    """cogvideo/dit.py:247-263: text chunks in reverse order (multi-scene), all video tokens flipped."""
    txt, vid = emb[:, :n_text], emb[:, n_text:]
    if multiscene:
        b, n, e = txt.shape
        txt = txt.reshape(b, num_chunks, n // num_chunks, e).flip(1).reshape(b, n, e)
    return torch.cat((txt, vid.flip(1)), dim=1)


def lastrow_case(ssm_layer, geom, seed):
    torch.manual_seed(seed)
    if geom == "cs16":      # same layout as mod_*_multi.pt: 2 scenes, offsets (56, 40) not multiples of 16
        cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=16, latent_height=4, latent_width=4,
                          compressed_num_frames=5, ssm_layer=ssm_layer, scan_checkpoint_group_size=4)
        meta = dict(text_length=8, seq_text_length=16, num_frames=5, num_chunks=2, tokens_per_frame=16, latent_height=4, latent_width=4)
        L = 96
    else:                   # MFMA geometry: 3 scenes of 4 / 3 / 3 frames x 16 tokens + 32 text tokens each = 256 = 4 x 64;
        #                     scene offsets 96 / 80 are not multiples of 64
        cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=64, latent_height=4, latent_width=4,
                          compressed_num_frames=10, ssm_layer=ssm_layer, scan_checkpoint_group_size=2)
        meta = dict(text_length=32, seq_text_length=96, num_frames=10, num_chunks=3, tokens_per_frame=16, latent_height=4, latent_width=4)
        L = 256
    m = TTTWrapper(cfg)
    m.ttt.init_weights()
    with torch.no_grad():
        m.ttt.ttt_norm_weight.add_(0.1 * torch.randn_like(m.ttt.ttt_norm_weight))
        m.ttt.ttt_norm_bias.add_(0.1 * torch.randn_like(m.ttt.ttt_norm_bias))
        m.ttt.b1.add_(0.01 * torch.randn_like(m.ttt.b1))
        m.ttt.learnable_ttt_lr_bias.add_(0.1 * torch.randn_like(m.ttt.learnable_ttt_lr_bias))
        # spread the learning-rate gate so that the eta rows of a tile differ visibly between source mini-batches
        m.ttt.learnable_ttt_lr_weight.mul_(8.0)
    m.ttt.use_kernel = False
    sm = SequenceMetadata(t_emb=torch.zeros(1, 512), **meta)
    sm.init_multiscene_offsets()
    x0 = torch.randn(1, L, 128)
    dy = torch.randn(1, L, 128)
    out = {"ssm_layer": ssm_layer, "multiscene": True, "cfg": {k: getattr(cfg, k) for k in (
               "model_dim", "num_heads", "num_layers", "mini_batch_size", "latent_height", "latent_width",
               "compressed_num_frames", "ssm_layer", "scan_checkpoint_group_size", "ttt_base_lr", "rope_theta")},
           "meta": meta, "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()}, "x": x0, "dy": dy}
    flip = lambda t: flip_like_reference(t, meta["seq_text_length"], meta["num_chunks"], True)
    for reverse in (False, True):
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        with kernel_contract():
            y = m(flip(x), sm) if reverse else m(x, sm)
        y = flip(y) if reverse else y
        y.backward(dy)
        key = "rev" if reverse else "fwd"
        out[key] = {"y": y.detach(), "dx": x.grad.detach(),
                    "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}}
    # how different is the dual form on the full (non-identical) tiles?  Recorded so the test can state it.
    with torch.no_grad():
        y_full = m(x0, sm)
    out["dual_form_full_tile_y"] = y_full
    return out


def dit_mlp64_case(seed):
    torch.manual_seed(seed)
    from ttt.models.cogvideo.dit import DiffusionTransformer
    cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=2, mini_batch_size=64, latent_height=16, latent_width=16,
                      compressed_num_frames=3, ssm_layer="ttt_mlp", text_dim=32, time_embed_dim=64, attn_length=2,
                      prefix_temporal_length=1, adapter_method="sft", scan_checkpoint_group_size=2,
                      remat_transformer_layer_group_size=1)
    m = DiffusionTransformer(cfg)
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
    video = torch.randn(1, 3, 16, 16, 16)            # 3 frames x 64 tokens = 192 video tokens
    text = torch.randn(1, 1, 64, 32)                 # + 64 text tokens = 256 = 4 mini-batches of 64
    ts = torch.tensor([417])
    out = m(video, text, ts)
    dout = torch.randn_like(out)
    out.backward(dout)
    return {"ssm_layer": "ttt_mlp", "scenes": 1, "cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__},
            "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()},
            "video": video, "text": text, "timesteps": ts, "out": out.detach(), "dout": dout,
            "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters()
                      if p.grad is not None and (p.numel() <= 20000 or k.endswith("ssm.ttt.wq.weight")
                                                 or k.endswith("seq_modeling_block.q.weight") or k.endswith("mlp.layer2.weight"))}}


def dit_mlp64_multiscene_lastrow_case(seed, base_lr=1.0, lr_scale=8.0, gate=0.5, latent=(8, 16), frames=7, text_len=32):
    """A 3-scene DiffusionTransformer at mini_batch_size = 64 run by the reference's own model code with every eta tile
    replaced by its last row (``kernel_contract``): the reference-pinned target of the ASSEMBLED multi-scene HIP model - the
    case the driver benchmarks (9 s, 3 interleaved scenes).  7 frames x 32 tokens + 3 x 32 text tokens = 320 = 5 mini-batches;
    scene lengths 128 / 96 / 96, so the third scene starts at 224, not a multiple of 64: the eta rows of a tile differ and the
    dual form on the full tiles (``dit_mlp_3scene.pt``) is NOT this function (hazard C2; the difference is recorded)."""
    torch.manual_seed(seed)
    from ttt.models.cogvideo.dit import DiffusionTransformer
    cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=2, mini_batch_size=64, latent_height=latent[0], latent_width=latent[1],
                      compressed_num_frames=frames, ssm_layer="ttt_mlp", text_dim=32, time_embed_dim=64, attn_length=2,
                      prefix_temporal_length=1, adapter_method="sft", scan_checkpoint_group_size=2,
                      remat_transformer_layer_group_size=1, ttt_base_lr=base_lr, gating_alpha_init=gate)
    m = DiffusionTransformer(cfg)
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
            # a model in which the inner loop matters (at initialisation the TTT update barely moves the output and the eta
            # rows of a tile are nearly identical): larger base learning rate and gates, a spread learning-rate gate
            layer.seq_modeling_block.ssm.ttt.learnable_ttt_lr_weight.mul_(lr_scale)
    video = torch.randn(1, frames, 16, latent[0], latent[1])
    text = torch.randn(1, (frames - 1) // 2, text_len, 32)
    ts = torch.tensor([417])
    with kernel_contract():
        out = m(video, text, ts)
        dout = torch.randn_like(out)
        out.backward(dout)
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    # the dual form on the full (non-identical) tiles, for the record: at this geometry it differs visibly only in the
    # learning-rate-gate gradients (the outputs agree to 1e-7: the eta rows of a tile differ by a few per cent)
    m.zero_grad(set_to_none=True)
    out_full = m(video, text, ts)
    out_full.backward(dout)
    lrw = "layers.0.seq_modeling_block.ssm.ttt.learnable_ttt_lr_weight"
    full_lr_grad = dict(m.named_parameters())[lrw].grad.detach().clone()
    return {"ssm_layer": "ttt_mlp", "scenes": 3, "lastrow": True, "cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__},
            "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()},
            "video": video, "text": text, "timesteps": ts, "out": out.detach(), "dout": dout,
            "dual_form_full_tile_out": out_full.detach(), "dual_form_full_tile_lr_grad": (lrw, full_lr_grad),
            "grads": {k: v for k, v in grads.items()
                      if v.numel() <= 20000 or k.endswith("ssm.ttt.wq.weight")
                      or k.endswith("seq_modeling_block.q.weight") or k.endswith("mlp.layer2.weight")}}


def cogvideox_loss_case(seed):
    """Reference CogVideoX.forward on CPU: needs an initialised process group for its DiscreteSampler (hazard C8)."""
    import torch.distributed as dist
    from ttt.models.cogvideo.model import CogVideoX
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29871")
        dist.init_process_group("gloo", rank=0, world_size=1)
    torch.manual_seed(seed)
    cfg = ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=16, latent_height=8, latent_width=8,
                      compressed_num_frames=3, ssm_layer="ttt_mlp", text_dim=32, time_embed_dim=64, attn_length=2,
                      prefix_temporal_length=1, adapter_method="sft", scan_checkpoint_group_size=2,
                      remat_transformer_layer_group_size=1)
    m = CogVideoX(cfg, effective_rank=0, effective_world_size=1)
    for mod in m.modules():
        if hasattr(mod, "use_kernel"):
            mod.use_kernel = False
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim >= 2 and "ttt." not in n:
                p.normal_(0, 0.02)
        for layer in m.dit.layers:
            layer.seq_modeling_block.ssm.ttt.init_weights()
    m.noise_generator = torch.Generator(device="cpu")
    m.noise_generator.manual_seed(1234)
    vid = torch.randn(2, 3, 16, 8, 8)
    text = torch.randn(2, 1, 16, 32)
    loss = m(vid, text)
    loss.sum().backward()
    # replay the generator to record what was drawn (same call sequence as model.py:49-52)
    g = torch.Generator(device="cpu")
    g.manual_seed(1234)
    idx = torch.randint(0, cfg.sigma_interval, (2,), generator=g)
    noise = torch.randn(vid.shape, generator=g)
    return {"cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, "generator_seed": 1234,
            "state_dict": {k: v.detach().clone() for k, v in m.state_dict().items()},
            "vid": vid, "text": text, "idx": idx, "noise": noise, "sigmas_at_idx": m.sigma_sampler.sigmas[idx].clone(),
            "sigma_table_interval_250": None, "loss": loss.detach(),
            "grads": {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None and p.numel() <= 20000}}


def dit_bf16_yardstick(names, lastrow=False):
    """How far the REFERENCE's own bf16 arithmetic is from its fp32 run, per gradient: the DiT fixtures' models re-run through
    the reference code under torch.autocast(bfloat16) (GEMMs in bf16, as under FSDP mixed precision).  The GPU test bounds the
    bf16 HIP path by max(stated tolerance, 2 x this) - small, cancellation-heavy gradients (b1, the learning-rate gate) are
    ill-conditioned in bf16 for ANY implementation."""
    from ttt.models.cogvideo.dit import DiffusionTransformer
    out = {}
    for name in names:
        g = torch.load(os.path.join(HERE, name), weights_only=False)
        m = DiffusionTransformer(ModelConfig(**g["cfg"]))
        m.load_state_dict(g["state_dict"], strict=True)
        for mod in m.modules():
            if hasattr(mod, "use_kernel"):
                mod.use_kernel = False
        import contextlib
        with (kernel_contract() if lastrow else contextlib.nullcontext()):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o = m(g["video"], g["text"], g["timesteps"])
            o.float().backward(g["dout"])
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
        errs = {"out": rel(o.float(), g["out"])}
        params = dict(m.named_parameters())
        for k, r in g["grads"].items():
            if params[k].grad is not None:
                errs[k] = rel(params[k].grad, r)
        out[name] = errs
        worst = max(errs.items(), key=lambda kv: kv[1])
        print(name, "reference bf16-autocast vs its fp32 run: out", round(errs["out"], 4), "worst grad", worst[0], round(worst[1], 4))
    return out


def sigma_tables():
    """DiscreteSampler's table for sigma_interval != 1000 (sub-sampled 1000-step schedule, then rescaled)."""
    from ttt.models.cogvideo.utils import ZeroSNRDDPMDiscretization
    return {n: ZeroSNRDDPMDiscretization()(n, device="cpu", flip=True).clone() for n in (1000, 250, 50)}


def main():
    save = lambda name, obj: (torch.save(obj, os.path.join(HERE, name)), print("wrote", name))[1]
    if sys.argv[1:] == ["r3"]:     # round 3: only the new fixture (the others are unchanged, bit for bit)
        save("dit_mlp64_3scene_lastrow.pt", dit_mlp64_multiscene_lastrow_case(seed=36))
        save("dit_bf16_yardstick_r3.pt", dit_bf16_yardstick(["dit_mlp64_3scene_lastrow.pt"], lastrow=True))
        return
    save("mod_mlp_multi_lastrow.pt", lastrow_case("ttt_mlp", "cs16", seed=31))
    save("mod_lin_multi_lastrow.pt", lastrow_case("ttt_linear", "cs16", seed=32))
    save("mod_mlp_multi64_lastrow.pt", lastrow_case("ttt_mlp", "cs64", seed=33))
    save("dit_mlp64_1scene.pt", dit_mlp64_case(seed=34))
    c = cogvideox_loss_case(seed=35)
    c["sigma_tables"] = sigma_tables()
    del c["sigma_table_interval_250"]
    save("cogvideox_loss.pt", c)
    save("dit_bf16_yardstick.pt", dit_bf16_yardstick(["dit_mlp64_1scene.pt", "dit_lin_1scene.pt", "dit_mlp_3scene.pt"]))
    save("dit_mlp64_3scene_lastrow.pt", dit_mlp64_multiscene_lastrow_case(seed=36))
    save("dit_bf16_yardstick_r3.pt", dit_bf16_yardstick(["dit_mlp64_3scene_lastrow.pt"], lastrow=True))


if __name__ == "__main__":
    main()
