"""CPU diagnosis (round 4): the WHOLE TTT-MLP backward step with the bf16 roundings of the MFMA sweep (csrc/ttt_mfma_bwd4.hip)
injected point by point into the fp64 oracle, on the 3-scene kernel-contract DiT fixture; reports the learning-rate-gate
gradient errors with all roundings on and with each one switched off / alone.
Usage: python tools/diag/lr_gate_full_emul_cpu.py [quick]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import load_golden, rel_l2
from oracle import cpu_ext, ttt_oracle as O

bf = lambda t: t.to(torch.bfloat16).to(t.dtype)
T = lambda x: x.transpose(-1, -2)
POINTS = ["rec", "W", "dWp", "gZ1", "D1", "X2", "u", "M", "dZ2", "dZ1", "dZ2b", "dZ2b_colsum", "dZ1b", "X2b", "D1b", "deta_out"]


def step_bwd(st_in, Q, K, V, eta, gam, bet, dOut, dst, on):
    r = lambda name, x: bf(x) if name in on else x
    W1, b1, W2, b2 = st_in
    Fd = Q.shape[-1]
    (W1n, b1n, W2n, b2n), _, s = O._mlp_step_primal(W1, b1, W2, b2, Q, K, V, eta, gam, bet, O.LN_EPS)
    dW1n, db1n, dW2n, db2n = dst
    # what the record holds: Z1, Z1b, gZ2 as bf16; the owner rows in fp32
    Z1, Z1b, gZ2r = r("rec", s["Z1"]), r("rec", s["Z1b"]), r("rec", s["gZ2"])
    X2, D1 = r("X2", O.gelu_tanh(Z1)), r("D1", O.gelu_bwd(Z1))
    X2b, D1b = r("X2b", O.gelu_tanh(Z1b)), r("D1b", O.gelu_bwd(Z1b))
    W1m, W2m, W1nm, W2nm = r("W", W1), r("W", W2), r("W", W1n), r("W", W2n)
    gX2 = gZ2r @ T(W2m)
    gZ1 = r("gZ1", gX2 * D1)
    M = r("M", gX2 * O.gelu_bwd2(Z1))
    # output path
    xhl_r = r("own_xl", s["xhl"])
    dgam = (dOut * xhl_r).sum(-2, keepdim=True)
    dbet = dOut.sum(-2, keepdim=True)
    dZ2b = O._ln_bwd(dOut, xhl_r, s["stdl"], gam, Fd)
    dW2n = dW2n + T(X2b) @ r("dZ2b", dZ2b)
    # "dZ2b_colsum": db2 += column sums of the bf16 tile (a ones-MFMA; round 3) instead of the owners' fp32 values (round 4)
    db2n = db2n + r("dZ2b_colsum", dZ2b).sum(-2, keepdim=True)
    dZ1b = (r("dZ2b", dZ2b) @ T(W2nm)) * D1b
    db1n = db1n + dZ1b.sum(-2, keepdim=True)
    dZ1b = r("dZ1b", dZ1b)
    dW1n = dW1n + T(Q) @ dZ1b
    dQ = dOut + dZ1b @ T(W1nm)
    # S1
    d1p, d2p = r("dWp", dW1n), r("dWp", dW2n)
    e1 = K @ d1p + db1n
    a2 = gZ2r @ T(d2p)
    deta = -((gZ1 * e1).sum(-1, keepdim=True) + (X2 * a2).sum(-1, keepdim=True) + (s["gZ2"] * db2n).sum(-1, keepdim=True))
    deta = r("deta_out", deta)
    u = r("u", -eta * e1 * D1)
    if "P_rec" in on:       # option "sweep_records_bf16": the four hidden slices' partial d(gZ2) tiles rounded to bf16 before they are summed
        parts = [bf(-eta * (X2[..., :, 64 * q:64 * q + 64] @ d2p[..., 64 * q:64 * q + 64, :]) + u[..., :, 64 * q:64 * q + 64] @ W2m[..., 64 * q:64 * q + 64, :])
                 for q in range(4)]
        dgZ2 = ((parts[0] + parts[1]) + parts[2]) + parts[3] - eta * db2n
    else:
        dgZ2 = -eta * (X2 @ d2p) + u @ W2m - eta * db2n
    # option "own_bf16": the inner LayerNorm's owner rows of the step record (x_hat, y - target) as bf16; "own_xl" = the OUTPUT
    # LayerNorm's x_hat too - which is NOT affordable (it feeds dZ2b and through it db2: lr-gate gradients off by 0.3)
    xh_r, go_r = r("own_xh", s["xh"]), r("own_go", s["go"])
    gxh_r = go_r * gam
    gZ2_own = (Fd * gxh_r - gxh_r.sum(-1, keepdim=True) - xh_r * (gxh_r * xh_r).sum(-1, keepdim=True)) / (Fd * s["std"])
    dZ2, dgam2, dbet2, dt = O._ln_l2_bwd_bwd(dgZ2, xh_r, s["std"], go_r, gxh_r, gZ2_own, gam, Fd)
    dgam = dgam + dgam2.sum(-2, keepdim=True)
    dbet = dbet + dbet2.sum(-2, keepdim=True)
    dV = dt
    dZ2 = r("dZ2", dZ2)
    # S4a
    dg = -eta * e1
    dx = -eta * a2 + dZ2 @ T(W2m)
    dZ1 = dg * M + dx * D1
    db1 = db1n + dZ1.sum(-2, keepdim=True)
    dZ1 = r("dZ1", dZ1)
    dW1 = dW1n + T(K) @ dZ1
    dW2 = dW2n + T(u) @ gZ2r + T(X2) @ dZ2
    db2 = db2n + dZ2.sum(-2, keepdim=True)
    dK = -eta * (gZ1 @ T(d1p)) + dZ1 @ T(W1m) - dt
    return (dW1, db1, dW2, db2), dQ, dK, dV, deta, dgam, dbet


def backward(XQ, XK, XV, le, lnw, lnb, cks, G, dOut, dst_last, on):
    B, NH, NC, CS, Fd = XQ.shape
    gam, bet = lnw.reshape(1, NH, 1, Fd), lnb.reshape(1, NH, 1, Fd)
    W1c, b1c, W2c, b2c = cks
    dst = dst_last
    res = {k: torch.zeros_like(XQ) for k in ("dXQ", "dXK", "dXV")}
    res["dlast_eta"] = torch.zeros_like(le)
    dgam = torch.zeros(B, NH, 1, Fd, dtype=XQ.dtype); dbet = torch.zeros_like(dgam)
    for k in reversed(range(W1c.shape[2])):
        lo, hi = k * G, min((k + 1) * G, NC)
        st = (W1c[:, :, k], b1c[:, :, k], W2c[:, :, k], b2c[:, :, k])
        states = []
        for i in range(lo, hi):
            states.append(st)
            st, _, _ = O._mlp_step_primal(*st, XQ[:, :, i], XK[:, :, i], XV[:, :, i], le[:, :, i], gam, bet, O.LN_EPS)
        for i in reversed(range(lo, hi)):
            dst, q, kk, v, e, g_, b_ = step_bwd(states[i - lo], XQ[:, :, i], XK[:, :, i], XV[:, :, i], le[:, :, i], gam, bet,
                                                dOut[:, :, i], dst, on)
            res["dXQ"][:, :, i], res["dXK"][:, :, i], res["dXV"][:, :, i], res["dlast_eta"][:, :, i] = q, kk, v, e
            dgam += g_; dbet += b_
    res.update(dW1=dst[0], db1=dst[1], dW2=dst[2], db2=dst[3], dln_w=dgam, dln_b=dbet)
    return res


def run(on, g):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    fake = cpu_ext.install()

    def patched(XQ, XK, XV, le, lnw, lnb, W1c, b1c, W2c, b2c, XQW, *rest):
        f = cpu_ext._f
        uW1, ub1, uW2, ub2, gout = rest[16:21]
        res = backward(f(XQ), f(XK), f(XV), f(le), f(lnw), f(lnb), tuple(f(c) for c in (W1c, b1c, W2c, b2c)), rest[-1], f(gout),
                       tuple(f(u) for u in (uW1, ub1, uW2, ub2)), on)
        for dst, k in zip(rest[21:-1], ("dln_w", "dln_b", "dW1", "db1", "dW2", "db2", "dlast_eta", "dXQ", "dXK", "dXV")):
            dst.copy_(res[k])
    fake.ttt_backward = patched
    m = DiffusionTransformer(ModelConfig(**g["cfg"]))
    m.load_state_dict(g["state_dict"], strict=True)
    for mod in m.modules():
        if hasattr(mod, "use_kernel"):
            mod.use_kernel = True
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = m(g["video"], g["text"], g["timesteps"])
    out.float().backward(g["dout"])
    params = dict(m.named_parameters())
    short = lambda k: k.split("layers.")[1].replace("seq_modeling_block.ssm.ttt.learnable_ttt_", "")
    lr = {short(k): round(rel_l2(params[k].grad, r), 4) for k, r in g["grads"].items() if "ttt_lr" in k and params[k].grad is not None}
    worst = max(rel_l2(params[k].grad, r) for k, r in g["grads"].items() if "ttt_lr" not in k and params[k].grad is not None)
    return lr, round(worst, 4)


if __name__ == "__main__":
    g = load_golden(sys.argv[2] if len(sys.argv) > 2 else "dit_mlp64_3scene_lastrow.pt")
    print("none                         ", *run(set(), g), flush=True)
    print("ALL (round 3 sweep)          ", *run(set(POINTS), g), flush=True)
    print("all but dZ2b_colsum (round 4)", *run(set(POINTS) - {"dZ2b_colsum"}, g), flush=True)
    print("round 4 + bf16 records       ", *run(set(POINTS) - {"dZ2b_colsum"} | {"P_rec"}, g), flush=True)
    print("  + bf16 inner owner rows    ", *run(set(POINTS) - {"dZ2b_colsum"} | {"P_rec", "own_xh", "own_go"}, g), flush=True)
    print("  + bf16 OUTPUT-LN x_hat too ", *run(set(POINTS) - {"dZ2b_colsum"} | {"P_rec", "own_xh", "own_go", "own_xl"}, g), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        sys.exit(0)
    for pnt in POINTS:
        print(f"only {pnt:10s}", *run({pnt}, g), flush=True)
    for pnt in POINTS:
        print(f"all but {pnt:7s}", *run(set(POINTS) - {pnt}, g), flush=True)
