"""DEBUG: one-step (local) error of both MFMA revisions from identical entering states (generic fp32 checkpoints), head 22."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
d0 = T.round_acts(O.make_inputs("mlp", seed=2, **T.FULL), torch.bfloat16)
hd, NS = 22, 49
sl = lambda t: t[:, hd:hd + 1, :NS].contiguous()
d = {k: (sl(v) if k in ("XQ", "XK", "XV", "eta", "dOut") else v[hd:hd + 1].contiguous()) for k, v in d0.items()}
o1, c1, _ = T.run_mlp(e, d, 1, torch.bfloat16, impl="generic")       # G=1: checkpoint = state entering every step
dev = "cuda:0"
def one_step(i, var):
    e.debug_variant(var); e.set_impl("mfma")
    XQ, XK, XV = (d[k][:, :, i:i + 2].to(dev, torch.bfloat16).contiguous() for k in ("XQ", "XK", "XV"))
    le = d["eta"][:, :, i:i + 2, -1, :, None].to(dev, torch.bfloat16).contiguous()
    f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
    out = torch.empty_like(XQ)
    ck = (f32(1, 1, 2, 64, 256), f32(1, 1, 2, 1, 256), f32(1, 1, 2, 256, 64), f32(1, 1, 2, 1, 64))
    e.ttt_forward(XQ, XK, XV, le, d["ln_w"].reshape(1, 1, 1, 64).to(dev), d["ln_b"].reshape(1, 1, 1, 64).to(dev),
                  *[c[:, :, i].contiguous() for c in c1], *ck, out, 1)
    torch.cuda.synchronize()
    e.set_impl("auto")
    return out, ck
names = ("W1", "b1", "W2", "b2")
for i in (0, 4, 8, 12, 16, 20, 24, 32, 40):
    row = []
    for var in (1, 2):
        out, ck = one_step(i, var)
        errs = []
        for k in range(4):
            ref_delta = (c1[k][:, :, i + 1] - c1[k][:, :, i]).double().cpu()
            got_delta = (ck[k][:, :, 1] - c1[k][:, :, i]).double().cpu()
            errs.append(float((got_delta - ref_delta).norm() / ref_delta.norm()))
        errs.append(T.rel_l2(out[:, :, 0], o1[:, :, i]))
        row.append(errs)
    print(f"step {i:2d}: one-step update error (W1,b1,W2,b2,out)  v1 " + " ".join(f"{x:.4f}" for x in row[0]) + "   v2 " + " ".join(f"{x:.4f}" for x in row[1]))
e.debug_variant(2)
