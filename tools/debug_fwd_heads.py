"""DEBUG: per-head / per-step forward error of the MFMA kernels vs the generic kernels at full size."""
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
for var in (2, 1, 2):
    e.debug_variant(var)
    o2, c2, _ = T.run_mlp(e, d, 16, torch.bfloat16, impl="mfma", bwd_impl="generic")
    a, b = o2.float(), o1.float()
    ph = ((a - b).flatten(2).norm(dim=2) / b.flatten(2).norm(dim=2))[0]
    ps = ((a - b).permute(2, 0, 1, 3, 4).flatten(1).norm(dim=1) / b.permute(2, 0, 1, 3, 4).flatten(1).norm(dim=1))
    print(f"variant {var}: overall {T.rel_l2(o2, o1):.4f}; heads>0.03: {[(i, round(v, 3)) for i, v in enumerate(ph.tolist()) if v > 0.03]}")
    print("   per-step (every 16th):", [round(float(x), 4) for x in ps[::16].tolist()])
    print("   ck W1 err per group:", [round(T.rel_l2(c2[0][:, :, k], c1[0][:, :, k]), 5) for k in range(0, 18, 2)], " b1:", [round(T.rel_l2(c2[1][:, :, k], c1[1][:, :, k]), 4) for k in range(1, 18, 4)])
e.debug_variant(2)
hd = 22
for var in (1, 2):
    e.debug_variant(var)
    o2, c2, _ = T.run_mlp(e, d, 16, torch.bfloat16, impl="mfma", bwd_impl="generic")
    a, b = o2.float()[0, hd], o1.float()[0, hd]
    ps = (a - b).flatten(1).norm(dim=1) / b.flatten(1).norm(dim=1)
    print(f"variant {var} head {hd} per-step err, steps 0..71:", [round(float(x), 4) for x in ps[:72].tolist()])
    print("    |out| per step:", [round(float(x), 1) for x in b.flatten(1).norm(dim=1)[:72:4].tolist()])
    for k, nm in enumerate(("W1", "b1", "W2", "b2")):
        print(f"    ck {nm} err groups 0..7:", [round(T.rel_l2(c2[k][0, hd, g], c1[k][0, hd, g]), 5) for g in range(8)], " |ref|", [round(float(c1[k][0, hd, g].norm()), 3) for g in range(8)])
e.debug_variant(2)
