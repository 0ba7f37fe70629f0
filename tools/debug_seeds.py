"""DEBUG: scan seeds for unstable heads; compare forward error of both MFMA revisions (vs generic fp32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
for seed in range(10, 22):
    d = T.round_acts(O.make_inputs("mlp", seed=seed, **T.FULL), torch.bfloat16)
    o1, _, _ = T.run_mlp(e, d, 16, torch.bfloat16, impl="generic")
    res = {}
    for var in (1, 2):
        e.debug_variant(var)
        o2, _, _ = T.run_mlp(e, d, 16, torch.bfloat16, impl="mfma", bwd_impl="generic")
        a, b = o2.float(), o1.float()
        res[var] = ((a - b).flatten(2).norm(dim=2) / b.flatten(2).norm(dim=2))[0]
    bad = [(i, round(float(res[1][i]), 3), round(float(res[2][i]), 3)) for i in range(48) if max(float(res[1][i]), float(res[2][i])) > 0.01]
    print(f"seed {seed}: median v1 {float(res[1].median()):.4f} v2 {float(res[2].median()):.4f}; heads>0.01 (head, v1, v2): {bad}")
e.debug_variant(2)
