#!/usr/bin/env python
"""Op-level micro-benchmark of the TTT scan kernels at a CogVideoX-5B geometry (default: 3 s,
B=1, NH=48, NC=282, CS=64, F=64, G=16).  Prints one JSON line per kernel family with the average
launch time and the achieved algorithmic TFLOP/s (fwd 7g, bwd 14g, g = 2*CS*F*4F).

    python tools/op_bench.py [--impl auto|generic|mfma] [--kind mlp|linear] [--nc 282] [--iters 10]
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="auto")
    ap.add_argument("--kind", default="mlp")
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--nh", type=int, default=48)
    ap.add_argument("--nc", type=int, default=282)
    ap.add_argument("--cs", type=int, default=64)
    ap.add_argument("--g", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--overlap", type=int, default=2, help="A/B: TTT-MLP backward schedule, 0 = one stream, 1 = tail kernel of a chunk beside the next sweep (default), 2 = the next recompute too")
    ap.add_argument("--gpc", type=int, default=0, help="DEBUG A/B: checkpoint groups per backward chunk (0 = automatic)")
    ap.add_argument("--write-through-records", action="store_true", help="DEBUG A/B: the backward sweep's hand-over records always write-through (sc1), never plain")
    ap.add_argument("--ab", default=None, metavar="OPTION", help="interleaved A/B inside one process: the named debug option alternates 0 / 1 from iteration to iteration; the backward's average is reported per value (same box, same clocks)")
    ap.add_argument("--ab-values", default="0,1", help="the two values --ab alternates between (default 0,1)")
    ap.add_argument("--ab-restore", type=int, default=1, help="value the --ab option is left at for the --phases pass")
    ap.add_argument("--ab-fixed", default=None, metavar="OPTION=VALUE", help="set one more debug option for the whole run")
    ap.add_argument("--ws-skew", type=int, default=0, help="DEBUG: the kernel workspace starts this many bytes (multiple of 256) into its allocation")
    ap.add_argument("--io-skew", type=int, default=0, help="DEBUG: input tensor k (XQ, XK, XV, eta, dOut) starts k * this many bytes (multiple of 256) into its allocation")
    ap.add_argument("--disturb", type=int, default=0, metavar="MB", help="DEBUG: between forward and backward, cast MB megabytes of bf16 to fp32 in 16 tensors (what a sharded path's per-unit gradient cast does in the middle of a backward)")
    ap.add_argument("--phases", action="store_true", help="also print per-phase cycle totals of workgroup 0")
    a = ap.parse_args()
    import test_time_training as ext
    from ttt_amd.models.ssm.linear_hip import HipLinear
    from ttt_amd.models.ssm.mlp_tk import TkMLP
    ext.load_library()
    ext.set_impl(a.impl)
    ext.debug_option("fast_records", 0 if a.write_through_records else 1)
    if a.ab_fixed:
        ext.debug_option(a.ab_fixed.split("=")[0], int(a.ab_fixed.split("=")[1]))
    ext.debug_option("overlap_tail", a.overlap)
    ext.debug_option("groups_per_chunk", a.gpc)
    dev = torch.device("cuda:0")
    if a.ws_skew:
        _orig_ws = ext._workspace
        ext._workspace = lambda device, stream, nbytes: _orig_ws(device, stream, nbytes + a.ws_skew)[a.ws_skew:]
    _k = [0]

    def skewed(t):                      # the same values at an address k * io_skew bytes into a fresh allocation
        k = _k[0]; _k[0] += 1
        if not a.io_skew:
            return t
        off = k * a.io_skew // t.element_size()
        buf = torch.empty(t.numel() + off, dtype=t.dtype, device=t.device)
        v = buf[off:].view(t.shape)
        v.copy_(t)
        return v
    B, NH, NC, CS, F, G = a.b, a.nh, a.nc, a.cs, 64, a.g
    H = 4 * F if a.kind == "mlp" else F
    gen = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
    XQ = skewed(torch.nn.functional.normalize(rn(B, NH, NC, CS, F), dim=-1).bfloat16()).requires_grad_(True)
    XK = skewed(torch.nn.functional.normalize(rn(B, NH, NC, CS, F), dim=-1).bfloat16()).requires_grad_(True)
    XV = skewed(rn(B, NH, NC, CS, F).bfloat16()).requires_grad_(True)
    base_lr = 0.1 if a.kind == "mlp" else 1.0
    eta = skewed((base_lr * torch.sigmoid(rn(B, NH, NC, 1, CS)) / (F * CS)).bfloat16()).requires_grad_(True)
    ln_w = torch.ones(NH, F, device=dev, requires_grad=True)
    ln_b = torch.zeros(NH, F, device=dev, requires_grad=True)
    W1 = (0.02 * rn(NH, F, H)).requires_grad_(True)
    b1 = torch.zeros(NH, 1, H, device=dev, requires_grad=True)
    W2 = (0.02 * rn(NH, H, F)).requires_grad_(True)
    b2 = torch.zeros(NH, 1, F, device=dev, requires_grad=True)
    dOut = skewed(rn(B, NH, NC, CS, F).bfloat16())
    ex = lambda p: p.unsqueeze(0).expand(B, *p.shape)

    def fwd():
        if a.kind == "mlp":
            return TkMLP.apply(ln_w, ln_b, ex(W1), ex(b1), ex(W2), ex(b2), XQ, XV, XK, eta, G)
        return HipLinear.apply(ln_w, ln_b, ex(W1), ex(b1), XQ, XV, XK, eta, G)

    # time the raw kernel launches with events around the extension calls
    times = {"fwd": [], "bwd": []}
    for name, key in (("ttt_forward", "fwd"), ("ttt_backward", "bwd"), ("ttt_linear_forward", "fwd"), ("ttt_linear_backward", "bwd")):
        orig = getattr(ext, name)

        def wrapped(*args, _o=orig, _k=key):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); _o(*args); e.record()
            times[_k].append((s, e))
        setattr(ext, name, wrapped)

    ab_vals = [int(v) for v in a.ab_values.split(",")]
    ab = {0: [], 1: []}
    abf = {0: [], 1: []}
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize()
            times = {"fwd": [], "bwd": []}
        if a.ab:
            ext.debug_option(a.ab, ab_vals[it & 1])
        n0, f0 = len(times["bwd"]), len(times["fwd"])
        out = fwd()
        if a.disturb:
            if it == 0:
                _src = [torch.randn(a.disturb * (1 << 20) // 32, device=dev).bfloat16() for _ in range(16)]
                _dst = [torch.empty_like(t, dtype=torch.float32) for t in _src]
            torch._foreach_copy_(_dst, _src)
        if not a.fwd_only:
            out.backward(dOut)
        if a.ab and it >= 2:
            ab[it & 1] += times["bwd"][n0:]
            abf[it & 1] += times["fwd"][f0:]
    torch.cuda.synchronize()
    if a.ab:
        ext.debug_option(a.ab, a.ab_restore)
    g = 2.0 * CS * F * H
    nfl = {"fwd": (7 if a.kind == "mlp" else 3) * g, "bwd": (14 if a.kind == "mlp" else 6) * g}
    res = {"kind": a.kind, "impl_requested": a.impl, "shape": [B, NH, NC, CS, F], "G": G}
    res["ptr_mod_2MiB"] = {n: t.data_ptr() % (2 << 20) for n, t in (("XQ", XQ), ("XK", XK), ("XV", XV), ("eta", eta), ("dOut", dOut))}
    res["ptr_mod_2MiB"].update({f"ws{i}": b.data_ptr() % (2 << 20) for i, b in enumerate(ext._ws_cache.values())})
    for k, ev in times.items():
        if not ev:
            continue
        ms = sorted(s.elapsed_time(e) for s, e in ev)
        avg = sum(ms) / len(ms)
        fl = B * NH * NC * nfl[k]
        res[k] = {"impl": ext.resolved_impl(B, NH, NC, CS, F, G, torch.bfloat16, a.kind == "mlp", k == "bwd"),
                  "avg_ms": avg, "min_ms": ms[0], "us_per_step": 1e3 * avg / NC, "tflops": fl / (avg * 1e-3) / 1e12,
                  "frac_mfma_peak": fl / (avg * 1e-3) / 2.5e15, "frac_occupied_cu_peak": fl / (avg * 1e-3) / (2.5e15 * min(B * NH, 256) / 256)}
    if a.ab:
        avg = lambda ev: sum(s.elapsed_time(e) for s, e in ev) / max(1, len(ev))
        res["ab"] = {"option": a.ab, **{str(ab_vals[v]): {"fwd_avg_ms": avg(abf[v]), "bwd_avg_ms": avg(ab[v]), "n": len(abf[v])} for v in (0, 1)}}
    if a.phases:
        buf = torch.zeros(48, dtype=torch.int64, device=dev)
        ext.debug_timing(buf)
        out = fwd()
        if not a.fwd_only:
            out.backward(dOut)
        torch.cuda.synchronize()
        ext.debug_timing(None)
        res["phase_cycles_per_step"] = [round(v / NC, 1) for v in buf.tolist()]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
