"""Time the TTT-MLP forward scan at the evaluation geometry (mini-batches of 16 tokens): MFMA kernel vs generic kernel.

    python tools/cs16_bench.py [--nh 48] [--nc 1128] [--batch 1] [--iters 5]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttt-video-dit_amd"))
import test_time_training as e  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nh", type=int, default=48)
ap.add_argument("--nc", type=int, default=1128)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--phases", action="store_true")
ap.add_argument("--no-generic", action="store_true")
ap.add_argument("--linear", action="store_true", help="TTT-Linear instead of TTT-MLP")
a = ap.parse_args()
dev = "cuda:0"
B, NH, NC, CS, F = a.batch, a.nh, a.nc, 16, 64
torch.manual_seed(0)
n = lambda *s: torch.randn(*s, device=dev)
XQ = torch.nn.functional.normalize(n(B, NH, NC, CS, F), dim=-1).bfloat16()
XK = torch.nn.functional.normalize(n(B, NH, NC, CS, F), dim=-1).bfloat16()
XV = n(B, NH, NC, CS, F).bfloat16()
le = (0.1 * torch.sigmoid(n(B, NH, NC, CS, 1)) / (F * CS)).bfloat16()
lw, lb = torch.ones(1, NH, 1, F, device=dev), torch.zeros(1, NH, 1, F, device=dev)
W1, b1 = 0.02 * n(B, NH, F, 4 * F), torch.zeros(B, NH, 1, 4 * F, device=dev)
W2, b2 = 0.02 * n(B, NH, 4 * F, F), torch.zeros(B, NH, 1, F, device=dev)
G = NC
if a.linear:
    W1l, b1l = 0.02 * n(B, NH, F, F), torch.zeros(B, NH, 1, F, device=dev)
    lwl, lbl = torch.ones(NH, F, device=dev), torch.zeros(NH, F, device=dev)
    lel = (1.0 * torch.sigmoid(n(B, NH, NC, CS, 1)) / (F * CS)).bfloat16()
    ckl = (torch.empty(B, NH, 1, F, F, device=dev), torch.empty(B, NH, 1, 1, F, device=dev))
cks = (torch.empty(B, NH, 1, F, 4 * F, device=dev), torch.empty(B, NH, 1, 1, 4 * F, device=dev),
       torch.empty(B, NH, 1, 4 * F, F, device=dev), torch.empty(B, NH, 1, 1, F, device=dev))
outs = {}
for impl in (("mfma",) if a.no_generic else ("mfma", "generic")):
    e.set_impl(impl)
    out = torch.empty_like(XQ)
    if a.linear:
        run = lambda: e.ttt_linear_forward(XQ, XK, XV, lel, lwl, lbl, W1l, b1l, *ckl, out, G)
    else:
        run = lambda: e.ttt_forward(XQ, XK, XV, le, lw, lb, W1, b1, W2, b2, *cks, out, G)
    run()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.iters):
        run()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.iters
    outs[impl] = out.float()
    flops = B * NH * NC * (3 * 2 * CS * F * F if a.linear else 7 * 2 * CS * F * 4 * F)
    print(f"{'linear ' if a.linear else ''}{impl:8s} fwd {ms:8.3f} ms  {ms * 1e3 / NC:6.2f} us/step  {flops / ms / 1e9:7.2f} TFLOP/s ({3 if a.linear else 7} GEMMs/step)")
if not a.no_generic:
  d = (outs["mfma"] - outs["generic"]).flatten(2).norm(dim=2) / outs["generic"].flatten(2).norm(dim=2)
  print("per-head rel-L2 mfma vs generic: median %.3e max %.3e" % (d.median().item(), d.max().item()))
if a.linear:
    # backward (training geometry of TTT-Linear: G = 4, configs/train/ttt-linear/3s.toml)
    Gb = 4
    Kb = -(-NC // Gb)
    ckb = (torch.empty(B, NH, Kb, F, F, device=dev), torch.empty(B, NH, Kb, 1, F, device=dev))
    dOut = n(B, NH, NC, CS, F).bfloat16()
    z = lambda *s: torch.zeros(*s, device=dev)
    scr = (torch.empty(B, NH, Gb, F, F, device=dev), torch.empty(B, NH, Gb, 1, F, device=dev))
    grads = (torch.empty(B, NH, 1, F, device=dev), torch.empty(B, NH, 1, F, device=dev), torch.empty(B, NH, F, F, device=dev),
             torch.empty(B, NH, 1, F, device=dev), torch.empty(B, NH, NC, CS, 1, device=dev, dtype=torch.bfloat16),
             torch.empty_like(XQ), torch.empty_like(XQ), torch.empty_like(XQ))
    for impl in (("mfma",) if a.no_generic else ("mfma", "generic")):
        e.set_impl(impl)
        out = torch.empty_like(XQ)
        e.ttt_linear_forward(XQ, XK, XV, lel, lwl, lbl, W1l, b1l, *ckb, out, Gb)
        run = lambda: e.ttt_linear_backward(XQ, XK, XV, lel, lwl, lbl, *ckb, z(B, NH, F, F), z(B, NH, 1, F), dOut, *scr, *grads, Gb)
        run()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.iters):
            run()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.iters
        print(f"linear {impl:8s} bwd {ms:8.3f} ms  {ms * 1e3 / NC:6.2f} us/step (G = {Gb}, incl. the group recompute)")
if a.phases and not a.linear:
    e.set_impl("mfma")
    buf = torch.zeros(16, dtype=torch.int64, device=dev)
    e.debug_timing(buf)
    e.ttt_forward(XQ, XK, XV, le, lw, lb, W1, b1, W2, b2, *cks, out, G)
    torch.cuda.synchronize()
    e.debug_timing(None)
    c = (buf.cpu().double() / NC).tolist()
    print("cycles/step of workgroup 0, wave 0: A1+A2 %.0f | B1 wait %.0f | park+P3 %.0f | B2 wait %.0f | C %.0f | E %.0f | total %.0f"
          % (c[0], c[4], c[1], c[5], c[2], c[3], sum(c[:6])))
e.set_impl("auto")
