"""CPU diagnosis (round 4): which bf16 rounding inside the TTT-MLP backward's d(eta) carries the error of the learning-rate-gate
gradient on the 3-scene kernel-contract DiT fixture?  Runs the assembled DiT on the CPU with the oracle-backed stand-in of the
extension (oracle/cpu_ext.py) and replaces d(last_eta) of every backward call by a variant of the fp64 oracle's with bf16
rounding injected at chosen operands - the roundings the MFMA sweep (csrc/ttt_mfma_bwd4.hip, stage S1) applies.
Usage: python tools/diag/lr_gate_rounding_cpu.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import load_golden, rel_l2
from oracle import cpu_ext, ttt_oracle as O

bf = lambda t: t.to(torch.bfloat16).to(t.dtype)
T = lambda x: x.transpose(-1, -2)


def split(t):            # hi + lo bf16 pair
    hi = bf(t)
    return hi, bf(t - hi)


def deta_variant(XQ, XK, XV, le, lnw, lnb, cks, G, dOut, dst_last, inj):
    """d(last_eta) [B,NH,NC,CS,1] of the oracle's backward with roundings `inj` (a set of names) in the d(eta) statement only;
    everything that is carried (state gradients) stays fp64."""
    B, NH, NC, CS, Fd = XQ.shape
    gam, bet = lnw.reshape(1, NH, 1, Fd), lnb.reshape(1, NH, 1, Fd)
    W1c, b1c, W2c, b2c = cks
    dst = dst_last
    deta = torch.zeros_like(le)
    for k in reversed(range(W1c.shape[2])):
        lo, hi = k * G, min((k + 1) * G, NC)
        st = (W1c[:, :, k], b1c[:, :, k], W2c[:, :, k], b2c[:, :, k])
        states = []
        for i in range(lo, hi):
            states.append(st)
            st, _, _ = O._mlp_step_primal(*st, XQ[:, :, i], XK[:, :, i], XV[:, :, i], le[:, :, i], gam, bet, O.LN_EPS)
        for i in reversed(range(lo, hi)):
            Q, K, V, eta, dO = XQ[:, :, i], XK[:, :, i], XV[:, :, i], le[:, :, i], dOut[:, :, i]
            (W1n, b1n, W2n, b2n), _, s = O._mlp_step_primal(*states[i - lo], Q, K, V, eta, gam, bet, O.LN_EPS)
            dW1n, db1n, dW2n, db2n = dst
            # the output path adds to the carried state first (as in _mlp_step_bwd)
            dZ2b = O._ln_bwd(dO, s["xhl"], s["stdl"], gam, Fd)
            dW2n_ = dW2n + T(s["X2b"]) @ dZ2b
            db2n_ = db2n + dZ2b.sum(-2, keepdim=True)
            dZ1b = (dZ2b @ T(W2n)) * O.gelu_bwd(s["Z1b"])
            dW1n_ = dW1n + T(Q) @ dZ1b
            db1n_ = db1n + dZ1b.sum(-2, keepdim=True)
            gZ1, X2, gZ2 = s["gZ1"], s["X2"], s["gZ2"]
            if "act" in inj:
                gZ1, X2 = bf(gZ1), bf(X2)
            gZ2m = bf(gZ2) if "act" in inj else gZ2
            d1, d2 = dW1n_, dW2n_
            if "dW1" in inj: d1 = bf(d1)
            if "dW2" in inj: d2 = bf(d2)
            if "dW1hl" in inj: d1 = sum(split(d1))
            if "dW2hl" in inj: d2 = sum(split(d2))
            A2 = gZ2m @ T(d2)
            A1 = gZ1 @ T(d1) if "a1form" not in inj else None
            # kernel form of the first-layer terms: e1 = db1 + K dW1 (rows n), se = sum_n gZ1 * e1
            e1 = K @ d1 + db1n_
            t1 = (gZ1 * e1).sum(-1, keepdim=True)                 # = rowsum(K * A1) + rowsum(gZ1 * db1)
            t2 = (X2 * A2).sum(-1, keepdim=True)
            t3 = (gZ2 * db2n_).sum(-1, keepdim=True)              # owner: fp32 gZ2, fp32 db2
            de = -(t1 + t2 + t3)
            deta[:, :, i] = de
            dst, *_ = O._mlp_step_bwd(states[i - lo], Q, K, V, eta, gam, bet, O.LN_EPS, dO, dst)
    return deta


def main():
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    g = load_golden("dit_mlp64_3scene_lastrow.pt")
    yard = load_golden("dit_bf16_yardstick_r3.pt")["dit_mlp64_3scene_lastrow.pt"]
    variants = [("fp64 d(eta), bf16 output", set()), ("act", {"act"}), ("dW1", {"dW1"}), ("dW2", {"dW2"}),
                ("dW1+dW2", {"dW1", "dW2"}), ("all (kernel)", {"act", "dW1", "dW2"}), ("act + hi/lo dW", {"act", "dW1hl", "dW2hl"}),
                ("fp32 output, all", {"act", "dW1", "dW2", "f32out"}), ("fp32 output, act + hi/lo", {"act", "dW1hl", "dW2hl", "f32out"})]
    fake = cpu_ext.install()
    orig = fake.ttt_backward
    for name, inj in variants:
        def patched(XQ, XK, XV, le, lnw, lnb, W1c, b1c, W2c, b2c, XQW, *rest):
            orig(XQ, XK, XV, le, lnw, lnb, W1c, b1c, W2c, b2c, XQW, *rest)
            f = cpu_ext._f
            uW1, ub1, uW2, ub2, gout = rest[16:21]
            de = deta_variant(f(XQ), f(XK), f(XV), f(le), f(lnw), f(lnb), tuple(f(c) for c in (W1c, b1c, W2c, b2c)), rest[-1], f(gout),
                              tuple(f(u) for u in (uW1, ub1, uW2, ub2)), inj)
            out = rest[21:-1][6]
            if "f32out" in inj:
                patched.side.append(de)          # fp32 side channel: the autograd wrapper reads bf16, so scale trick below
            out.copy_(de)
        patched.side = []
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
        res = {k.split("layers.")[1].replace("seq_modeling_block.ssm.ttt.learnable_ttt_", ""): round(rel_l2(params[k].grad, r), 4)
               for k, r in g["grads"].items() if "ttt_lr" in k and params[k].grad is not None}
        print(f"{name:32s}", res, flush=True)
    print("reference's own bf16-autocast run:", {k.split("layers.")[1].replace("seq_modeling_block.ssm.ttt.learnable_ttt_", ""): round(v, 4)
                                                 for k, v in yard.items() if "ttt_lr" in k})


if __name__ == "__main__":
    main()
