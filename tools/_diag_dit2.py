"""DEBUG (gpurun): capture the TTT-MLP op inputs inside the small DiT and compare MFMA / generic backward with the fp64 oracle on them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_time_training as ext
from helpers import load_golden, rel_l2
from oracle import ttt_oracle as O
from ttt_amd.models.cogvideo.dit import DiffusionTransformer
from ttt_amd.models.configs import ModelConfig

DEV = "cuda:0"
ext.load_library()
if len(sys.argv) > 1:
    ext.debug_option("bwd_cluster", int(sys.argv[1]))
g = load_golden("dit_mlp64_1scene.pt")
m = DiffusionTransformer(ModelConfig(**g["cfg"]))
m.load_state_dict(g["state_dict"], strict=True)
m = m.to(DEV).to(torch.bfloat16)
for mod in m.modules():
    if hasattr(mod, "init_freqs"):
        mod.init_freqs()
calls = []
orig_b = ext.ttt_backward
def rec_b(*a):
    calls.append([t.detach().clone() if isinstance(t, torch.Tensor) else t for t in a])
    return orig_b(*a)
ext.ttt_backward = rec_b
out = m(g["video"].to(DEV, torch.bfloat16), g["text"].to(DEV, torch.bfloat16), g["timesteps"].to(DEV))
out.backward(g["dout"].to(DEV, out.dtype))
torch.cuda.synchronize()
ext.ttt_backward = orig_b
print("captured", len(calls), "backward calls")
names = ["dln_w", "dln_b", "dW1", "db1", "dW2", "db2", "dlast_eta", "dXQ", "dXK", "dXV"]
for ci, a in enumerate(calls):
    XQ, XK, XV, le, lnw, lnb, W1c, b1c, W2c, b2c, XQW = a[:11]
    G = a[-1]
    rest = a[11:-1]
    ups, gout = rest[16:20], rest[20]
    f64 = lambda t: t.detach().double().cpu()
    ref = O.mlp_backward(f64(XQ), f64(XK), f64(XV), f64(le), f64(lnw), f64(lnb), tuple(f64(c) for c in (W1c, b1c, W2c, b2c)), G, f64(gout),
                         dst_last=tuple(f64(u) for u in ups))
    print(f"call {ci}: shapes XQ {tuple(XQ.shape)} G {G}; |dOut| {float(gout.float().norm()):.3e} |XV| {float(XV.float().norm()):.3e} "
          f"|eta| {float(le.float().abs().mean()):.3e} |W1c| {float(W1c.norm()):.3e} |b1c| {float(b1c.norm()):.3e} lnw {float(lnw.mean()):.3f}")
    for impl in ("mfma", "generic"):
        ext.set_impl(impl)
        outs = [torch.full_like(t, float("nan")) for t in rest[21:]]
        args = list(a[:11]) + [t.clone() for t in rest[:21]] + outs + [G]
        orig_b(*args)
        torch.cuda.synchronize()
        errs = {n: round(rel_l2(o, ref[n]), 4) for n, o in zip(names, outs)}
        print(f"   {impl:8s}", errs)
    ext.set_impl("auto")
    print("   ref norms", {n: f"{float(ref[n].norm()):.2e}" for n in names})
