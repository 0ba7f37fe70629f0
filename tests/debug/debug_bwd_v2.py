#!/usr/bin/env python
"""DEBUG: revision-2 MFMA backward vs revision 1 and vs the fp64 oracle on small cases, per gradient and per step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle import ttt_oracle as O  # noqa: E402
import test_kernels_gpu as T  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    e = T.ext()
    for (B, NH, NC, G, gpc) in [(1, 1, 1, 1, 0), (1, 1, 2, 2, 0), (1, 2, 5, 2, 0), (1, 2, 5, 2, 1), (2, 3, 7, 3, 2)]:
        d = T.round_acts(O.make_inputs("mlp", B, NH, NC, 64, 64, seed=55 + NC), torch.bfloat16)
        e.debug_groups_per_chunk(gpc)
        e.debug_variant(1)
        o1, c1, g1 = T.run_mlp(e, d, G, torch.bfloat16, impl="mfma")
        e.debug_variant(2)
        o2, c2, g2 = T.run_mlp(e, d, G, torch.bfloat16, impl="mfma")
        e.debug_groups_per_chunk(0)
        ro, rc, rg = T.oracle_on(d, G, "mlp")
        print(f"=== B{B} NH{NH} NC{NC} G{G} gpc{gpc}")
        for k in g2:
            line = f"  {k:10s} v2-vs-oracle {rel(g2[k], rg[k]):.3e}  v1-vs-oracle {rel(g1[k], rg[k]):.3e}  v2-vs-v1 {rel(g2[k], g1[k]):.3e}"
            if g2[k].ndim == 5 and g2[k].shape[2] == NC:
                per = [rel(g2[k][:, :, i], rg[k][:, :, i]) for i in range(NC)]
                line += "  per-step " + " ".join(f"{x:.1e}" for x in per)
            print(line)
        nan = {k: bool(torch.isnan(v.float()).any()) for k, v in g2.items()}
        if any(nan.values()):
            print("  NaN in:", [k for k, v in nan.items() if v])


if __name__ == "__main__":
    main()
