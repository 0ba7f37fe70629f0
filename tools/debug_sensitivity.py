"""DEBUG: conditioning of head 22 (seed 2): perturb the bf16 inputs by one ulp in a few places and compare fp32-arithmetic outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
d = T.round_acts(O.make_inputs("mlp", seed=2, **T.FULL), torch.bfloat16)
o1, c1, _ = T.run_mlp(e, d, 16, torch.bfloat16, impl="generic")
for nper in (1, 16, 4096):
    dp = dict(d)
    xv = d["XV"].clone().bfloat16()
    g = torch.Generator().manual_seed(nper)
    # one-ulp perturbation of `nper` random elements of XV in EVERY head, steps 0..63
    for hd in range(48):
        idx = torch.randint(0, 64 * 64 * 64, (nper,), generator=g)
        flat = xv[0, hd, :64].reshape(-1).view(torch.int16)
        flat[idx] = flat[idx] + 1
    dp["XV"] = xv.float()
    o2, c2, _ = T.run_mlp(e, dp, 16, torch.bfloat16, impl="generic")
    a, b = o2.float(), o1.float()
    ph = ((a - b).flatten(2).norm(dim=2) / b.flatten(2).norm(dim=2))[0]
    srt = sorted(enumerate(ph.tolist()), key=lambda t: -t[1])[:4]
    print(f"{nper} one-ulp perturbations per head: output rel-L2 change - top heads {[(i, round(v, 4)) for i, v in srt]}, median {float(ph.median()):.2e}")
