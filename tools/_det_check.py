"""DEBUG (gpurun): is the TTT-MLP backward run-to-run deterministic?  Repeats the same call (one-stream and two-stream schedule)
and reports which outputs differ between repetitions, and where.  (Round 3 ran it for revisions 3 and 4 side by side,
profiles/r3c_*, r3d_*; revision 3 has since been removed.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import ttt_oracle as O
from test_kernels_gpu import ext, round_acts, run_mlp

e = ext()
B, NH, NC, G = 1, 48, 96, 16
d = round_acts(O.make_inputs("mlp", B, NH, NC, 64, 64, seed=900 + NC), torch.bfloat16)
for rev in (4,):
    for overlap in (0, 1, 2):
        e.debug_option("overlap_tail", overlap)
        ref = None
        diffs = {}
        for rep in range(6):
            junk = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")
            out, cks, g = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
            torch.cuda.synchronize()
            if ref is None:
                ref = {k: v.clone() for k, v in g.items()}
                continue
            for k, v in g.items():
                if not torch.equal(v, ref[k]):
                    bad = (v != ref[k])
                    idx = bad.nonzero()
                    heads = sorted(set(idx[:, 1].tolist()))
                    steps = sorted(set(idx[:, 2].tolist())) if v.dim() == 5 else None
                    diffs.setdefault(k, []).append((rep, int(bad.sum()), heads[:12], (steps[:6], steps[-3:]) if steps else None))
        print(f"rev {rev} overlap {overlap}: sweep_error {e.sweep_error()} ->", "DETERMINISTIC" if not diffs else diffs, flush=True)
e.debug_option("overlap_tail", 2)
