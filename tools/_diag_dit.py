"""DEBUG (gpurun): where do the b1 / learning-rate-gate gradient errors of the bf16 DiT come from?  Variants of the same model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_time_training as ext
from helpers import load_golden, rel_l2
from ttt_amd.models.cogvideo.dit import DiffusionTransformer
from ttt_amd.models.configs import ModelConfig

DEV = "cuda:0"
ext.load_library()
name = sys.argv[1] if len(sys.argv) > 1 else "dit_mlp64_1scene.pt"
g = load_golden(name)
keys = [k for k in g["grads"] if k.split(".")[-1] in ("b1", "b2", "learnable_ttt_lr_bias", "learnable_ttt_lr_weight", "ttt_norm_weight", "ttt_norm_bias")
        or k.endswith("ssm.ttt.wq.weight") or k.endswith("k_norm.bias")]


def run(tag, impl="auto", fused=True, kernel=True, dtype=torch.bfloat16, glue=True):
    ext.set_impl(impl)
    m = DiffusionTransformer(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    m = m.to(DEV).to(dtype)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
        if hasattr(mod, "use_fused"):
            mod.use_fused = fused
        if hasattr(mod, "use_kernel"):
            mod.use_kernel = kernel
        if hasattr(mod, "use_fused_glue"):
            mod.use_fused_glue = glue
    out = m(g["video"].to(DEV, dtype), g["text"].to(DEV, dtype), g["timesteps"].to(DEV))
    out.backward(g["dout"].to(DEV, out.dtype))
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    errs = {k.replace("seq_modeling_block.ssm.ttt.", "ttt.").replace("layers.", "L"): round(rel_l2(params[k].grad, g["grads"][k]), 4) for k in keys if params[k].grad is not None}
    print(f"{tag:34s} out {rel_l2(out, g['out']):.4f}", errs, flush=True)
    ext.set_impl("auto")


run("default (mfma, fused)")
run("generic kernels, fused", impl="generic")
run("mfma, unfused pre/post", fused=False)
run("torch dual form bf16 (no kernel)", kernel=False)
run("unfused glue", glue=False)
try:
    run("fp32 model, generic kernels", impl="generic", dtype=torch.float32)
except Exception as ex:
    print("fp32 run failed:", repr(ex)[:200])
