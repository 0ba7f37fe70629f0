#!/usr/bin/env python
"""How hipExtStreamCreateWithCUMask's bits map to (XCD, shader engine, CU) on this GPU, and what a confined stream costs / protects.

    python tools/cu_mask_probe.py > profiles/r6*_cu_mask_probe.json

1. 256 one-CU workgroups on the default stream: the (xcd, se, sa, cu) set of the chip.
2. For a few masks (first 32 bits, every 4th bit, bits 0 - 63, ...): which CUs workgroups of the masked stream land on.
3. The TTT-MLP backward (NC = 804, 48 heads) alone, beside a saturating bf16 GEMM loop on an UNMASKED side stream, and beside the same
   loop on a stream masked to a quarter of the chip (the complement of what the sweep needs): ms per backward, gradients compared
   bit for bit with the run alone, sweep_error.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import test_time_training as ext
    from ttt_amd.models.ssm.mlp_tk import TkMLP
    ext.load_library()
    dev = torch.device("cuda:0")
    res = {"device": torch.cuda.get_device_name(0), "cus": torch.cuda.get_device_properties(0).multi_processor_count}
    allp = ext.placement_probe(256)
    res["unmasked_distinct_cus"] = len(set(allp))
    res["unmasked_per_xcd"] = {str(x): sum(1 for p in set(allp) if p[0] == x) for x in sorted({p[0] for p in allp})}
    masks = {"bits_0_31": [0xFFFFFFFF] + [0] * 7, "bits_0_63": [0xFFFFFFFF] * 2 + [0] * 6, "every_4th_bit": [0x11111111] * 8,
             "bits_0_7": [0xFF] + [0] * 7, "last_64": [0] * 6 + [0xFFFFFFFF] * 2, "low_byte_of_every_word": [0xFF] * 8}
    res["masks"] = {}
    for name, words in masks.items():
        try:
            st = ext.masked_stream(words)
            pl = ext.placement_probe(256, stream=st)
            d = sorted(set(pl))
            res["masks"][name] = {"words": [hex(w) for w in words], "distinct_cus": len(d),
                                  "per_xcd": {str(x): sum(1 for p in d if p[0] == x) for x in sorted({p[0] for p in d})}, "first": d[:12]}
        except Exception as ex:      # noqa: BLE001
            res["masks"][name] = {"error": repr(ex)[:200]}
    # ---- the sweep beside a GEMM loop
    B, NH, NC, G = 1, 48, 804, 16
    gen = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=gen)
    nz = torch.nn.functional.normalize
    XQ, XK = (nz(rn(B, NH, NC, 64, 64), dim=-1).bfloat16().requires_grad_(True) for _ in range(2))
    XV = rn(B, NH, NC, 64, 64).bfloat16().requires_grad_(True)
    eta = (0.1 * torch.sigmoid(rn(B, NH, NC, 1, 64)) / 4096).bfloat16().requires_grad_(True)
    ln_w, ln_b = torch.ones(NH, 64, device=dev, requires_grad=True), torch.zeros(NH, 64, device=dev, requires_grad=True)
    W1, W2 = (0.02 * rn(NH, 64, 256)).requires_grad_(True), (0.02 * rn(NH, 256, 64)).requires_grad_(True)
    b1, b2 = torch.zeros(NH, 1, 256, device=dev, requires_grad=True), torch.zeros(NH, 1, 64, device=dev, requires_grad=True)
    dOut = rn(B, NH, NC, 64, 64).bfloat16()
    ex = lambda p: p.unsqueeze(0).expand(B, *p.shape)
    ins = [XQ, XK, XV, eta, W1, W2, ln_w]
    a, b = rn(8192, 8192).bfloat16(), rn(8192, 8192).bfloat16()
    times = []
    orig = ext.ttt_backward

    def timed(*args):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); orig(*args); e.record()
        times.append((s, e))
    ext.ttt_backward = timed

    def run(side, iters=6):
        times.clear()
        grads = None
        for _ in range(iters):
            out = TkMLP.apply(ln_w, ln_b, ex(W1), ex(b1), ex(W2), ex(b2), XQ, XV, XK, eta, G)
            torch.cuda.synchronize()
            if side is not None:
                with torch.cuda.stream(side):
                    for _ in range(40):                      # ~40 x 0.9 ms of GEMM: longer than one backward
                        torch.mm(a, b)
            grads = torch.autograd.grad(out, ins, dOut)
            torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in times)
        return ms[len(ms) // 2], [g.clone() for g in grads]

    # hipExtStreamCreateWithCUMask gives a BLOCKING stream: it synchronises implicitly with the legacy default stream (torch's default),
    # so everything here runs on a non-default stream - as a training step that wants the overlap has to
    work = torch.cuda.Stream()
    work.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(work)
    t0, g0 = run(None)
    res["sweep"] = {"alone_ms": round(t0, 3)}
    plain = torch.cuda.Stream()
    t1, g1 = run(plain)
    res["sweep"]["beside_unmasked_gemm_ms"] = round(t1, 3)
    res["sweep"]["beside_unmasked_same_bits"] = all(torch.equal(x, y) for x, y in zip(g0, g1))
    # a quarter of the chip for the side stream: the choice of bits follows from part 2 (set by --side-mask, default: the low byte of every 32-bit word = 8 CUs of every XCD)
    words = [int(w, 16) for w in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0xff"] * 8)]
    try:
        side = ext.masked_stream(words)
        t2, g2 = run(side)
        res["sweep"]["beside_masked_gemm_ms"] = round(t2, 3)
        res["sweep"]["beside_masked_same_bits"] = all(torch.equal(x, y) for x, y in zip(g0, g2))
        res["sweep"]["side_mask"] = [hex(w) for w in words]
        # how fast is the GEMM itself on the masked stream, alone?
        for name, st in (("gemm_8192_unmasked_ms", plain), ("gemm_8192_masked_ms", side)):
            with torch.cuda.stream(st):
                torch.mm(a, b); st.synchronize()
                t = time.perf_counter()
                for _ in range(10):
                    torch.mm(a, b)
                st.synchronize()
            res["sweep"][name] = round((time.perf_counter() - t) * 100, 3)
        # what the masked stream gets done BESIDE the scans: 24 projection-sized weight-gradient GEMMs ([3072, 51456] @ [51456, 3072]) on
        # the side stream while the main stream runs 6 backward (or 6 forward) scans; both clocks
        xw, dyw = rn(51456, 3072).bfloat16(), rn(51456, 3072).bfloat16()
        wg = lambda: torch.mm(dyw.t(), xw)

        def both(main_fn, n_main, n_side):
            torch.cuda.synchronize()
            ms0, ms1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ss0, ss1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side.wait_stream(torch.cuda.current_stream())
            ms0.record()
            with torch.cuda.stream(side):
                ss0.record()
                for _ in range(n_side):
                    wg()
                ss1.record()
            for _ in range(n_main):
                main_fn()
            ms1.record()
            torch.cuda.synchronize()
            return round(ms0.elapsed_time(ms1), 3), round(ss0.elapsed_time(ss1), 3)

        outs = [TkMLP.apply(ln_w, ln_b, ex(W1), ex(b1), ex(W2), ex(b2), XQ, XV, XK, eta, G) for _ in range(1)]
        bwd = lambda: torch.autograd.grad(outs[0], ins, dOut, retain_graph=True)
        fwd = lambda: TkMLP.apply(ln_w, ln_b, ex(W1), ex(b1), ex(W2), ex(b2), XQ, XV, XK, eta, G)
        with torch.no_grad():
            fwd_ng = lambda: TkMLP.apply(ln_w, ln_b, ex(W1), ex(b1), ex(W2), ex(b2), XQ, XV, XK, eta, G)
        bwd(); wg(); torch.cuda.synchronize()
        res["beside"] = {"wgrad_gemm_alone_unmasked_ms_each": both(lambda: None, 0, 24)[1] / 24 if False else None}
        with torch.cuda.stream(plain):
            pass
        t_main_only = both(bwd, 6, 0)[0]
        t_side_only = both(lambda: None, 0, 24)[1]
        t_both = both(bwd, 6, 24)
        res["beside"] = {"six_backwards_alone_ms": t_main_only, "24_wgrad_gemms_on_masked_stream_alone_ms": t_side_only,
                         "together_main_ms_side_ms": t_both}
        tf_only = both(fwd, 6, 0)[0]
        tf_both = both(fwd, 6, 24)
        res["beside"].update({"six_forward_scans_alone_ms": tf_only, "forward_together_main_ms_side_ms": tf_both})
        # the same 24 GEMMs on the main stream (full chip), for the price list
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(24):
            wg()
        s1.record(); torch.cuda.synchronize()
        res["beside"]["24_wgrad_gemms_full_chip_ms"] = round(s0.elapsed_time(s1), 3)
    except Exception as exn:      # noqa: BLE001
        res["sweep"]["masked_error"] = repr(exn)[:300]
    res["sweep"]["sweep_error"] = ext.sweep_error()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
