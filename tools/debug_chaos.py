"""DEBUG: is the head-22 deviation implementation luck?  Perturb inputs by a few bf16 ulps, re-run generic + both MFMA
revisions on the SAME perturbed inputs, report the head-22 output error of each revision."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
d0 = T.round_acts(O.make_inputs("mlp", seed=2, **T.FULL), torch.bfloat16)
hd = 22
for trial in range(6):
    dp = dict(d0)
    if trial:
        xv = d0["XV"].clone().bfloat16()
        g = torch.Generator().manual_seed(trial)
        idx = torch.randint(0, 8 * 64 * 64, (64,), generator=g)
        flat = xv[0, hd, :8].reshape(-1).view(torch.int16)
        flat[idx] = flat[idx] + 1
        dp["XV"] = xv.float()
    o1, _, _ = T.run_mlp(e, dp, 16, torch.bfloat16, impl="generic")
    res = []
    for var in (1, 2):
        e.debug_variant(var)
        o2, _, _ = T.run_mlp(e, dp, 16, torch.bfloat16, impl="mfma", bwd_impl="generic")
        a, b = o2.float()[0, hd], o1.float()[0, hd]
        res.append(float((a - b).norm() / b.norm()))
    print(f"trial {trial}: head {hd} rel-L2 vs generic: v1 {res[0]:.3f}  v2 {res[1]:.3f}")
e.debug_variant(2)
