#!/usr/bin/env python
"""Timing of the token-wise glue kernels (AdaLN, post-norm, gated residual) at a CogVideoX-5B geometry: forward + backward of the
autograd nodes the model uses, HIP events around each call (includes the small torch reductions of the partials), GB/s on the
algorithmic bytes.  Run under `rocprofv3 --kernel-trace --stats` for the kernels alone.

    python tools/glue_bench.py [--video-length 9sec|3sec] [--iters 10]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--video-length", default="9sec")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    import test_time_training as ext
    from ttt_amd.models.ssm.fused import FusedAdaLN, FusedPost
    ext.load_library()
    dev = torch.device("cuda:0")
    Lt, Lv = {"3sec": (498, 17550), "9sec": (1506, 49950)}[a.video_length]
    B, D, NH = 1, 3072, 48
    L = Lt + Lv
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g).bfloat16()
    vid, text = mk(B, Lv, D).requires_grad_(True), mk(B, Lt, D).requires_grad_(True)
    mods = [mk(B, D).requires_grad_(True) for _ in range(4)]
    w, b = torch.ones(D, device=dev, requires_grad=True), torch.zeros(D, device=dev, requires_grad=True)
    dout = mk(B, L, D)
    Y = mk(B, NH, L, 64).requires_grad_(True)
    src = torch.randperm(L, device=dev, generator=g).to(torch.int32)
    res = {}

    def timed(name, fn, nbytes):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for s, e in ev:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev)
        res[name] = {"avg_ms": sum(ms) / len(ms), "min_ms": ms[0], "alg_GBps": nbytes / (ms[0] * 1e-3) / 1e9}

    t2 = 2 * B * L * D           # bytes of one bf16 [B, L, D] tensor
    out = FusedAdaLN.apply(vid, text, w, b, *mods, 1e-6)
    timed("adaln_fwd", lambda: FusedAdaLN.apply(vid, text, w, b, *mods, 1e-6), 2 * t2)
    timed("adaln_bwd", lambda: torch.autograd.grad(out, (vid, text, w, b, *mods), dout, retain_graph=True), 3 * t2)
    o = FusedPost.apply(Y, w, b, src, 1e-6)
    timed("post_fwd", lambda: FusedPost.apply(Y, w, b, src, 1e-6), 2 * t2)
    timed("post_bwd", lambda: torch.autograd.grad(o, (Y, w, b), dout, retain_graph=True), 3 * t2)
    print(json.dumps({"geometry": {"B": B, "Lt": Lt, "Lv": Lv, "D": D}, **res}))


if __name__ == "__main__":
    main()
