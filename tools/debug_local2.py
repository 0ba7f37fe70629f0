"""DEBUG: k-step error growth of both MFMA revisions from the same entering state (generic fp32 checkpoint), head 22."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
d0 = T.round_acts(O.make_inputs("mlp", seed=2, **T.FULL), torch.bfloat16)
NS = 49
dev = "cuda:0"
for hd in (22, 5):
    sl = lambda t: t[:, hd:hd + 1, :NS].contiguous()
    d = {k: (sl(v) if k in ("XQ", "XK", "XV", "eta", "dOut") else v[hd:hd + 1].contiguous()) for k, v in d0.items()}
    o1, c1, _ = T.run_mlp(e, d, 1, torch.bfloat16, impl="generic")
    def run(i, n, var):
        e.debug_variant(var); e.set_impl("mfma")
        XQ, XK, XV = (d[k][:, :, i:i + n + 1].to(dev, torch.bfloat16).contiguous() for k in ("XQ", "XK", "XV"))
        le = d["eta"][:, :, i:i + n + 1, -1, :, None].to(dev, torch.bfloat16).contiguous()
        f32 = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        out = torch.empty_like(XQ)
        ck = (f32(1, 1, n + 1, 64, 256), f32(1, 1, n + 1, 1, 256), f32(1, 1, n + 1, 256, 64), f32(1, 1, n + 1, 1, 64))
        e.ttt_forward(XQ, XK, XV, le, d["ln_w"].reshape(1, 1, 1, 64).to(dev), d["ln_b"].reshape(1, 1, 1, 64).to(dev),
                      *[c[:, :, i].contiguous() for c in c1], *ck, out, 1)
        torch.cuda.synchronize(); e.set_impl("auto")
        return out, ck
    i0 = 8
    for var in (1, 2):
        out, ck = run(i0, 16, var)
        for k, nm in ((1, "b1"), (3, "b2"), (2, "W2")):
            errs = [float(((ck[k][:, :, j] - c1[k][:, :, i0 + j]).double().norm() / (c1[k][:, :, i0 + j] - c1[k][:, :, i0]).double().norm().clamp_min(1e-30))) for j in range(1, 17)]
            print(f"head {hd} v{var} {nm} error after j steps / total change since start:", [round(x, 4) for x in errs])
e.debug_variant(2)
