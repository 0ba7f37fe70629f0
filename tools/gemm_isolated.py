#!/usr/bin/env python
"""The model's large library GEMMs timed ALONE (is 46 - 50 % of the MFMA roof the library or the step?).

    python tools/gemm_isolated.py [--rows 51456] [--iters 20] [--no-tuned] > profiles/r6*_gemm_isolated.json

Shapes of one CogVideoX-5B layer at the 9 s sequence (reference dit.py:43-87 MLP, :143-160 / ttt_layer.py:134-141 projections), bf16,
through the same PyTorch entry points and the same committed hipBLASLt / rocBLAS selections the training step uses
(ttt_amd/infra/gemm_tuning_gfx950.csv): forward `F.linear`, input gradient `dy @ W`, weight gradient `dy^T @ x`.  Each shape runs
back to back `iters` times between HIP events (the inputs of consecutive iterations rotate over three buffers so that the 256-MiB
Infinity Cache does not hold them), with the shader clock / socket power sampled beside it.  One JSON object: per shape ms, TFLOP/s and
fraction of the 2.5 PFLOP/s dense bf16 peak.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as Fn  # noqa: E402

PEAK = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=51456)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-tuned", action="store_true")
    a = ap.parse_args()
    from bench import ClockSampler
    from ttt_amd.infra.parallelisms import enable_tuned_gemms
    tuned = (not a.no_tuned) and enable_tuned_gemms()
    dev = torch.device("cuda:0")
    M, D, H = a.rows, 3072, 12288
    bf = torch.bfloat16
    R = 3                                                    # rotating operand sets

    def rnd(*shape):
        return [(torch.randn(*shape, device=dev, dtype=torch.float32) * 0.05).to(bf) for _ in range(R)]

    shapes = []
    # (name, flops, closure factory)
    x_d, x_h = rnd(M, D), rnd(M, H)
    w_dd, w_hd, w_dh, w_3d = rnd(D, D), rnd(H, D), rnd(D, H), rnd(3 * D, D)
    b_d, b_h = rnd(D), rnd(H)
    dy_3d = rnd(M, 3 * D)
    shapes.append(("proj fwd  F.linear [M,3072]x[3072,3072]^T+b (q/k/v/o)", 2.0 * M * D * D, lambda i: Fn.linear(x_d[i % R], w_dd[i % R], b_d[i % R])))
    shapes.append(("mlp fc1 fwd F.linear [M,3072]x[12288,3072]^T+b", 2.0 * M * D * H, lambda i: Fn.linear(x_d[i % R], w_hd[i % R], b_h[i % R])))
    shapes.append(("mlp fc2 fwd F.linear [M,12288]x[3072,12288]^T+b", 2.0 * M * D * H, lambda i: Fn.linear(x_h[i % R], w_dh[i % R], b_d[i % R])))
    shapes.append(("mlp fc2 dgrad [M,3072]@[3072,12288]", 2.0 * M * D * H, lambda i: x_d[i % R] @ w_dh[i % R]))
    shapes.append(("mlp fc1 dgrad [M,12288]@[12288,3072]", 2.0 * M * D * H, lambda i: x_h[i % R] @ w_hd[i % R]))
    shapes.append(("proj dgrad [M,3072]@[3072,3072] (o)", 2.0 * M * D * D, lambda i: x_d[i % R] @ w_dd[i % R]))
    shapes.append(("qkv dgrad fused [M,9216]@[9216,3072] (Linear3)", 2.0 * M * 3 * D * D, lambda i: dy_3d[i % R] @ w_3d[i % R]))
    shapes.append(("proj wgrad [3072,M]@[M,3072] (dy^T x)", 2.0 * M * D * D, lambda i: x_d[i % R].t() @ x_d[(i + 1) % R]))

    out = {"rows": M, "iters": a.iters, "tuned_gemm_selections": bool(tuned), "peak_tflops": PEAK, "device": torch.cuda.get_device_name(0), "shapes": []}
    clocks = ClockSampler(0, period_s=0.05)
    for name, flops, fn in shapes:
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        clocks.samples = []
        clocks.start()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(a.iters):
            fn(i)
        e.record()
        torch.cuda.synchronize()
        clocks.stop()
        ms = s.elapsed_time(e) / a.iters
        tf = flops / (ms * 1e-3) / 1e12
        cs = clocks.summary()
        out["shapes"].append({"name": name, "ms": round(ms, 4), "tflops": round(tf, 1), "frac_of_peak": round(tf / PEAK, 4),
                              "clock_mhz_avg": cs.get("clock_mhz_avg"), "power_w_avg": cs.get("power_w_avg")})
        print(f"{name:70s} {ms:8.3f} ms {tf:8.1f} TFLOP/s {tf / PEAK:6.3f}  clk {cs.get('clock_mhz_avg')} MHz {cs.get('power_w_avg')} W", file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
