#!/usr/bin/env python
"""Op-level benchmark of the local-attention kernels at the 3 s segment geometry (B=1, NH=48, S=18 048, D=64, bf16):
HIP forward / backward (csrc/attn_*.hip) next to PyTorch's SDPA (aotriton flash) on the same tensors.  Algorithmic
FLOPs: forward 4*S^2*D*NH, backward 2.5x that.  One JSON line.

    python tools/attn_bench.py [--s 18048] [--nh 48] [--iters 5] [--no-sdpa]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--s", type=int, default=18048)
    ap.add_argument("--nh", type=int, default=48)
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-sdpa", action="store_true")
    ap.add_argument("--variants", default=None, help="DEBUG A/B: comma list of attn_prio values (0 / 1: the s_setprio form of csrc/attn_body.h) - the backward "
                                                     "timed for each in interleaved rounds inside this process, outputs compared with the first")
    a = ap.parse_args()
    import test_time_training as ext
    from ttt_amd.models.cogvideo.attention import SegmentAttention
    ext.load_library()
    dev = torch.device("cuda:0")
    B, NH, S = a.b, a.nh, a.s
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda: torch.randn(B, S, NH, 64, device=dev, generator=g).bfloat16().transpose(1, 2)
    q, k, v, do = mk(), mk(), mk(), mk()
    v_ = v
    flops = 4.0 * S * S * 64 * NH * B
    res = {"shape": [B, NH, S, 64], "fwd_flops": flops}
    out = torch.empty(B, S, NH, 64, device=dev, dtype=torch.bfloat16).transpose(1, 2)
    lse = torch.empty(B, NH, S, device=dev)
    t = timeit(lambda: ext.attn_forward(q, k, v, out, lse, 0.125), a.iters)
    res["hip_fwd"] = {"ms": t, "tflops": flops / t / 1e9}
    dq, dk, dv = (torch.empty(B, S, NH, 64, device=dev, dtype=torch.bfloat16).transpose(1, 2) for _ in range(3))
    delta = torch.empty(B, NH, S, device=dev)
    t = timeit(lambda: ext.attn_backward(q, k, v, out, do, lse, delta, dq, dk, dv, 0.125), a.iters)
    res["hip_bwd"] = {"ms": t, "tflops": 2.5 * flops / t / 1e9}
    if a.variants:
        vs = [int(v) for v in a.variants.split(",")]
        times = {v: [] for v in vs}
        ref, same = None, {}
        for rnd in range(4):
            for v in vs:
                ext.debug_option("attn_prio", v)
                times[v].append(timeit(lambda: ext.attn_backward(q, k, v_, out, do, lse, delta, dq, dk, dv, 0.125), a.iters))
                if rnd == 0:
                    torch.cuda.synchronize()
                    cur = [t.clone() for t in (dq, dk, dv)]
                    if ref is None:
                        ref = cur
                    same[v] = [bool(torch.equal(x, y)) for x, y in zip(cur, ref)]
        ext.debug_option("attn_prio", 1)
        res["variants"] = {str(v): {"bwd_ms_rounds": [round(t, 3) for t in times[v]], "bwd_ms": round(sum(times[v]) / len(times[v]), 3),
                                    "tflops": round(2.5 * flops / (sum(times[v]) / len(times[v])) / 1e9, 1), "dq_dk_dv_equal_to_first": same[v]} for v in vs}
    if not a.no_sdpa:
        qq, kk, vv = (x.detach().clone().requires_grad_(True) for x in (q, k, v))
        t = timeit(lambda: F.scaled_dot_product_attention(qq, kk, vv), a.iters)
        res["sdpa_fwd"] = {"ms": t, "tflops": flops / t / 1e9}
        o = F.scaled_dot_product_attention(qq, kk, vv)
        t = timeit(lambda: torch.autograd.grad(o, (qq, kk, vv), do, retain_graph=True), a.iters)
        res["sdpa_bwd"] = {"ms": t, "tflops": 2.5 * flops / t / 1e9}
        res["max_abs_diff_vs_sdpa"] = float((o.float() - out.float()).abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
