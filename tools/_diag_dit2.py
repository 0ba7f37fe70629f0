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


# ---- which property of the captured inputs makes the MFMA backward inaccurate?  variants of call 2 -------------------------
a = calls[2]
G = a[-1]
def run_variant(tag, mod):
    b = [t.clone() if isinstance(t, torch.Tensor) else t for t in a]
    mod(b)
    XQ, XK, XV, le, lnw, lnb = b[:6]
    # checkpoints must belong to the inputs: re-run the forward (mfma) to regenerate them
    ext.set_impl("mfma")
    W1c, b1c, W2c, b2c, XQW = b[6:11]
    ext.ttt_forward(XQ, XK, XV, le, lnw, lnb, W1c[:, :, 0].contiguous(), b1c[:, :, 0].contiguous(), W2c[:, :, 0].contiguous(), b2c[:, :, 0].contiguous(),
                    W1c, b1c, W2c, b2c, XQW, G)
    torch.cuda.synchronize()
    rest = b[11:-1]
    ups, gout = rest[16:20], rest[20]
    f64 = lambda t: t.detach().double().cpu()
    ref = O.mlp_backward(f64(XQ), f64(XK), f64(XV), f64(le), f64(lnw), f64(lnb), tuple(f64(c) for c in (W1c, b1c, W2c, b2c)), G, f64(gout),
                         dst_last=tuple(f64(u) for u in ups))
    outs = [torch.full_like(t, float("nan")) for t in rest[21:]]
    orig_b(*(list(b[:11]) + [t.clone() for t in rest[:21]] + outs + [G]))
    torch.cuda.synchronize()
    errs = {n: round(rel_l2(o, ref[n]), 4) for n, o in zip(names, outs)}
    print(f"{tag:40s}", {k: errs[k] for k in ("dW1", "db1", "dW2", "dXK", "dXQ", "dlast_eta")}, flush=True)
    ext.set_impl("auto")

gen = torch.Generator(device=DEV).manual_seed(0)
rn = lambda t: torch.randn(t.shape, device=DEV, generator=gen)
run_variant("captured", lambda b: None)
def m_dout_scale(b): b[11 + 20].mul_(1024.0)
run_variant("dOut x 1024", m_dout_scale)
def m_dout_rand(b): b[11 + 20].copy_(rn(b[11 + 20]).bfloat16())
run_variant("dOut random N(0,1)", m_dout_rand)
def m_xv_rand(b): b[2].copy_(rn(b[2]).bfloat16())
run_variant("XV random N(0,1)", m_xv_rand)
def m_qk_rand(b):
    b[0].copy_(torch.nn.functional.normalize(rn(b[0]), dim=-1).bfloat16()); b[1].copy_(torch.nn.functional.normalize(rn(b[1]), dim=-1).bfloat16())
run_variant("XQ, XK random unit", m_qk_rand)
def m_eta_const(b): b[3].fill_(1.2e-5)
run_variant("eta constant 1.2e-5", m_eta_const)
def m_all(b): m_dout_rand(b); m_xv_rand(b); m_qk_rand(b)
run_variant("dOut, XV, XQ, XK random", m_all)
