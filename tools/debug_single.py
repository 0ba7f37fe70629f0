"""DEBUG: head 22 alone (NH=1) vs inside the 48-head launch, both MFMA revisions."""
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
d = {k: (v[:, hd:hd + 1].contiguous() if k in ("XQ", "XK", "XV", "eta", "dOut") else v[hd:hd + 1].contiguous()) for k, v in d0.items()}
for G in (16, 1):
    o1, c1, _ = T.run_mlp(e, d, G, torch.bfloat16, impl="generic")
    for var in (1, 2):
        e.debug_variant(var)
        o2, c2, _ = T.run_mlp(e, d, G, torch.bfloat16, impl="mfma", bwd_impl="generic")
        ps = (o2.float() - o1.float())[0, 0].flatten(1).norm(dim=1) / o1.float()[0, 0].flatten(1).norm(dim=1)
        print(f"single head, G={G}, v{var}: overall {T.rel_l2(o2, o1):.4f}; per-step (every 8th, first 96): {[round(float(x), 4) for x in ps[:96:8].tolist()]}")
e.debug_variant(2)
