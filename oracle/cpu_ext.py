"""TEST / CPU-BASELINE-ONLY stand-in for the ``test_time_training`` HIP extension, backed by the CPU oracle.

Lets the host-side plumbing (TkMLP / HipLinear autograd wrappers, TTT modules, FSDP wiring) be
exercised on a GPU-less machine through exactly the positional-buffer contract of the real
extension.  Installed only by tests and by the ``cpu_baseline`` leg of bench.py (``install()``); the product never imports it."""
import sys
import types

import torch

from oracle import ttt_oracle as O


def _f(t):
    return t.detach().to(torch.float64)


def ttt_forward(XQ, XK, XV, last_eta, ln_w, ln_b, W1, b1, W2, b2, W1c, b1c, W2c, b2c, XQW, G):
    out, cks, _ = O.mlp_forward(_f(XQ), _f(XK), _f(XV), _f(last_eta), _f(ln_w), _f(ln_b), _f(W1), _f(b1), _f(W2), _f(b2), G)
    XQW.copy_(out)
    for dst, src in zip((W1c, b1c, W2c, b2c), cks):
        dst.copy_(src)


def ttt_backward(XQ, XK, XV, last_eta, ln_w, ln_b, W1c, b1c, W2c, b2c, XQW, *rest):
    G = rest[-1]
    remat, (uW1, ub1, uW2, ub2, gout), outs = rest[:16], rest[16:21], rest[21:-1]
    dlnw, dlnb, dW1, db1, dW2, db2, deta, dQ, dK, dV = outs
    g = O.mlp_backward(_f(XQ), _f(XK), _f(XV), _f(last_eta), _f(ln_w), _f(ln_b), tuple(_f(c) for c in (W1c, b1c, W2c, b2c)),
                       G, _f(gout), dst_last=tuple(_f(u) for u in (uW1, ub1, uW2, ub2)))
    for dst, k in ((dlnw, "dln_w"), (dlnb, "dln_b"), (dW1, "dW1"), (db1, "db1"), (dW2, "dW2"), (db2, "db2"),
                   (deta, "dlast_eta"), (dQ, "dXQ"), (dK, "dXK"), (dV, "dXV")):
        dst.copy_(g[k])


def ttt_linear_forward(XQ, XK, XV, last_eta, ln_w, ln_b, W1, b1, W1c, b1c, XQW, G):
    out, cks, _ = O.linear_forward(_f(XQ), _f(XK), _f(XV), _f(last_eta), _f(ln_w), _f(ln_b), _f(W1), _f(b1), G)
    XQW.copy_(out)
    W1c.copy_(cks[0]); b1c.copy_(cks[1])


def ttt_linear_backward(XQ, XK, XV, last_eta, ln_w, ln_b, W1c, b1c, uW1, ub1, gout, W1g, b1g, dlnw, dlnb, dW1, db1,
                        deta, dQ, dK, dV, G):
    g = O.linear_backward(_f(XQ), _f(XK), _f(XV), _f(last_eta), _f(ln_w), _f(ln_b), (_f(W1c), _f(b1c)), G, _f(gout))
    for dst, k in ((dlnw, "dln_w"), (dlnb, "dln_b"), (dW1, "dW1"), (db1, "db1"), (deta, "dlast_eta"), (dQ, "dXQ"),
                   (dK, "dXK"), (dV, "dXV")):
        dst.copy_(g[k])


def install():
    m = types.ModuleType("test_time_training")
    m.ttt_forward, m.ttt_backward = ttt_forward, ttt_backward
    m.ttt_linear_forward, m.ttt_linear_backward = ttt_linear_forward, ttt_linear_backward
    m.__fake__ = True
    sys.modules["test_time_training"] = m
    return m


def uninstall():
    if getattr(sys.modules.get("test_time_training"), "__fake__", False):
        del sys.modules["test_time_training"]
