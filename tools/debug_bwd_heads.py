"""DEBUG: per-head error of the MFMA backward vs the generic kernels at full size, for several chunk sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
import test_kernels_gpu as T

e = T.ext()
d = T.round_acts(O.make_inputs("mlp", seed=2, **T.FULL), torch.bfloat16)
o1, c1, g1 = T.run_mlp(e, d, 16, torch.bfloat16, impl="generic")
for gpc in (6, 5, 3, 18, 6):
    e.debug_groups_per_chunk(gpc)
    o2, c2, g2 = T.run_mlp(e, d, 16, torch.bfloat16, impl="mfma")
    e.debug_groups_per_chunk(0)
    a, b = g2["dXK"].float(), g1["dXK"].float()
    per_head = ((a - b).flatten(2).norm(dim=2) / b.flatten(2).norm(dim=2))[0]
    per_step = ((a - b).permute(2, 0, 1, 3, 4).flatten(1).norm(dim=1) / b.permute(2, 0, 1, 3, 4).flatten(1).norm(dim=1))
    bad_heads = [i for i, v in enumerate(per_head.tolist()) if v > 0.05]
    bad_steps = [i for i, v in enumerate(per_step.tolist()) if v > 0.05]
    print(f"gpc={gpc}: bad heads {bad_heads}; bad steps {bad_steps[:6]}..{bad_steps[-3:]} ({len(bad_steps)}); "
          f"dW1 per-head err {[round(float(x), 3) for x in ((g2['dW1'] - g1['dW1']).flatten(2).norm(dim=2) / g1['dW1'].flatten(2).norm(dim=2))[0].tolist()]}")
