#!/usr/bin/env python
"""Round 5: the q / k / v projections' backward as ONE weight-gradient GEMM and ONE input-gradient GEMM over the concatenated output
gradient (ttt_amd/infra/fused_linear.py: Linear3.backward) against three GEMMs each.

    python tools/qkv_backward_bench.py --tune gpurun_out/x/tunableop_qkv.csv     # TunableOp search for the new shapes, then the A/B
    python tools/qkv_backward_bench.py                                            # A/B with the committed selections only

Shapes: D = 3072, rows = 51 456 (the TTT layer's wq / wk / wv at 9 s), 18 052 / 18 048 (an attention segment at 9 s / 3 s).  Interleaved
rounds in one process, medians; the results of both paths are compared (same mathematics, another summation order)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", default=None, metavar="CSV", help="run PyTorch TunableOp's solution search for the fused shapes first and write its file here")
    ap.add_argument("--rows", default="51456,18052,18048")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    from torch.cuda import tunable
    from ttt_amd.infra import fused_linear as FL
    from ttt_amd.infra.parallelisms import enable_tuned_gemms
    dev = torch.device("cuda:0")
    D = 3072
    res = {"tuned_selections_loaded": bool(enable_tuned_gemms())}
    g = torch.Generator(device=dev).manual_seed(0)
    if a.tune:
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(30)
        tunable.set_max_warmup_duration(5) if hasattr(tunable, "set_max_warmup_duration") else None
        tunable.set_filename(os.path.abspath(a.tune))
        for rows in (int(r) for r in a.rows.split(",")):
            cat = torch.randn(rows, 3 * D, device=dev, generator=g).bfloat16()
            x = torch.randn(rows, D, device=dev, generator=g).bfloat16()
            w = torch.randn(3 * D, D, device=dev, generator=g).bfloat16()
            cat.t().mm(x); cat.mm(w)
            for i in range(3):                  # the per-projection weight gradients over the strided column blocks (ld 3 D)
                cat[:, i * D:(i + 1) * D].t().mm(x)
            torch.cuda.synchronize()
        tunable.tuning_enable(False)          # (the file is written when the process exits)
        res["tuned_file"] = a.tune
    for rows in (int(r) for r in a.rows.split(",")):
        x = (torch.randn(1, rows, D, device=dev, generator=g) * 0.5).bfloat16().requires_grad_(True)
        ws = [(torch.randn(D, D, device=dev, generator=g) * 0.02).bfloat16().requires_grad_(True) for _ in range(3)]
        bs = [torch.zeros(D, device=dev, dtype=torch.bfloat16).requires_grad_(True) for _ in range(3)]
        buf = (torch.randn(1, rows, 3 * D, device=dev, generator=g) * 0.1).bfloat16()
        blocks = [buf[..., :D], buf[..., D:2 * D], buf[..., 2 * D:]]
        sep = [b.contiguous() for b in blocks]
        ys = FL.Linear3.apply(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])

        def run(grads, fuse):
            FL.FUSE_QKV_BACKWARD = fuse
            return torch.autograd.grad(ys, [x] + ws + bs, grads, retain_graph=True)

        t = {"fused_dgrad": [], "fused_both": [], "three_gemms": []}
        for _ in range(a.rounds):
            t["fused_dgrad"].append(timeit(lambda: run(blocks, "dgrad")))
            t["fused_both"].append(timeit(lambda: run(blocks, "both")))
            t["three_gemms"].append(timeit(lambda: run(sep, "")))
        gf, gs = run(blocks, "dgrad"), run(sep, "")
        FL.FUSE_QKV_BACKWARD = "dgrad"
        rl2 = lambda p, q: float((p.double() - q.double()).norm() / q.double().norm().clamp_min(1e-30))
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        flops = 2 * 2.0 * rows * D * 3 * D
        res[f"rows={rows}"] = {"median_ms": med, "tflops": {k: flops / v / 1e9 for k, v in med.items()},
                               "rel_l2_fused_vs_three": {"dx": rl2(gf[0], gs[0]), "dw": max(rl2(p, q) for p, q in zip(gf[1:4], gs[1:4])),
                                                         "db": max(rl2(p, q) for p, q in zip(gf[4:], gs[4:]))}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
