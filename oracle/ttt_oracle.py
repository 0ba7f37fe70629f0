"""CPU oracle for the TTT-MLP / TTT-Linear scan (forward + analytic backward).

TEST INFRASTRUCTURE ONLY.  Nothing under ``ttt-video-dit_amd/`` may import this
file; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it, and only as the checker / the reported CPU baseline.

It is a restatement, in plain torch-on-CPU tensor arithmetic (any float dtype,
fp64 for pinning), of the reference's PyTorch "ops" path:

  * ``ln_fwd`` / ``ln_fused_l2_bwd`` / ``gelu_bwd``  <- reference
    ``ttt/models/ssm/ops/utils.py:4-54``
  * ``mlp_step_dual``      <- ``ttt/models/ssm/ops/ttt_mlp.py:9-67``   (dual form, full eta tile)
  * ``linear_step_dual``   <- ``ttt/models/ssm/ops/ttt_linear.py:8-54``
  * ``scan_dual``          <- ``ttt/models/ssm/utils.py:111-146`` + ``ops/ttt_mlp.py:70-99``
  * ``mlp_forward`` / ``linear_forward``  primal form with the *kernel* contract
    (last-row eta, fp32 checkpoints every G steps) <- call site
    ``ttt/models/ssm/mlp_tk.py:92-133`` and ``kernels/linear_forward.py:54-145``
  * ``mlp_backward`` / ``linear_backward`` hand-derived reverse sweep with the
    kernel contract of ``mlp_tk.py:179-275`` / ``kernels/linear_backward.py:73-197``
    (group recompute from checkpoints, state-gradient carried backwards).

Parity pinning: tests/test_oracle_golden.py checks every function here against
golden vectors produced by *executing the reference itself* (its ops path and
torch.autograd through it) in the build container - see tests/golden/gen_golden.py.
The reference ships no tests / KATs of its own.

The third-party ttt-tk CUDA kernel (module ``test-time-training/ttt-tk``, commit
unpinned - .gitmodules:1-3 carries no SHA and the directory is empty) is NOT
available; for its arithmetic parity is anchored on the reference's ops path, which
is what north_star names as the numerical reference.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

LN_EPS = 1e-8  # ops/utils.py:4,21  (the Triton kernels use 1e-6: hazard C1)

GELU_A = 0.79788456
GELU_C = 0.044715
GELU_3AC = 0.1070322243  # ops/utils.py:53


# --------------------------------------------------------------------------- helpers
def ln_fwd(x, gamma, beta, eps=LN_EPS):
    """ops/utils.py:4-18 - LayerNorm over the last dim, biased variance."""
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, keepdim=True, unbiased=False)
    std = torch.sqrt(var + eps)
    x_hat = (x - mu) / std
    return gamma * x_hat + beta


def ln_fused_l2_bwd(x, l2_target, gamma, beta, eps=LN_EPS):
    """ops/utils.py:21-48 - d/dx of 0.5*||LN(x)-target||^2."""
    D = x.shape[-1]
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, keepdim=True, unbiased=False)
    std = torch.sqrt(var + eps)
    x_hat = (x - mu) / std
    y = gamma * x_hat + beta
    grad_output = y - l2_target
    grad_x_hat = grad_output * gamma
    z = (
        (1.0 / D)
        * (
            D * grad_x_hat
            - grad_x_hat.sum(dim=-1, keepdim=True)
            - x_hat * (grad_x_hat * x_hat).sum(dim=-1, keepdim=True)
        )
        / std
    )
    return z


def gelu_tanh(x):
    """F.gelu(x, approximate='tanh') (ops/ttt_mlp.py:29,42)."""
    return 0.5 * x * (1.0 + torch.tanh(GELU_A * x * (1.0 + GELU_C * x * x)))


def gelu_bwd(x):
    """ops/utils.py:51-54 - derivative of tanh-GELU."""
    tanh_out = torch.tanh(GELU_A * x * (1 + GELU_C * x * x))
    return 0.5 * x * ((1 - tanh_out * tanh_out) * (GELU_A + GELU_3AC * x * x)) + 0.5 * (1 + tanh_out)


def gelu_bwd2(x):
    """Second derivative of tanh-GELU (not in the reference; needed by the backward of
    ``gelu_bwd`` - SURVEY.md Appendix A)."""
    u = GELU_A * x * (1 + GELU_C * x * x)
    t = torch.tanh(u)
    du = GELU_A + GELU_3AC * x * x
    d2u = 2.0 * GELU_3AC * x
    s = 1 - t * t
    return s * du + 0.5 * x * (s * d2u - 2.0 * t * s * du * du)


# --------------------------------------------------------------------------- dual form (reference ops path)
def mlp_step_dual(p: Dict[str, torch.Tensor], XQ, XK, XV, eta):
    """One mini-batch of ops/ttt_mlp.py:9-67.  XQ/XK/XV [B,NH,CS,F], eta [B,NH,CS,CS]."""
    W1, b1, W2, b2 = p["W1"], p["b1"], p["W2"], p["b2"]
    NH, F = XQ.shape[1], XQ.shape[-1]
    ln_w = p["ln_w"].reshape(NH, 1, F)
    ln_b = p["ln_b"].reshape(NH, 1, F)

    Z1 = XK @ W1 + b1
    X2 = gelu_tanh(Z1)
    Z2 = X2 @ W2 + b2
    target = XV - XK
    gZ2 = ln_fused_l2_bwd(Z2, target, ln_w, ln_b)
    gZ1 = gZ2 @ W2.transpose(-2, -1) * gelu_bwd(Z1)

    Attn1 = XQ @ XK.transpose(-2, -1)
    b1_bar = b1 - eta @ gZ1
    Z1_bar = XQ @ W1 - (eta * Attn1) @ gZ1 + b1_bar
    X2_bar = gelu_tanh(Z1_bar)
    Attn2 = X2_bar @ X2.transpose(-2, -1)
    b2_bar = b2 - eta @ gZ2
    Z2_bar = X2_bar @ W2 - (eta * Attn2) @ gZ2 + b2_bar

    last_eta = eta[:, :, -1, :, None]
    W1n = W1 - (last_eta * XK).transpose(-1, -2) @ gZ1
    b1n = b1 - torch.sum(last_eta * gZ1, dim=-2, keepdim=True)
    W2n = W2 - (last_eta * X2).transpose(-1, -2) @ gZ2
    b2n = b2 - torch.sum(last_eta * gZ2, dim=-2, keepdim=True)

    out = XQ + ln_fwd(Z2_bar, ln_w, ln_b)
    return dict(p, W1=W1n, b1=b1n, W2=W2n, b2=b2n), out


def linear_step_dual(p: Dict[str, torch.Tensor], XQ, XK, XV, eta):
    """One mini-batch of ops/ttt_linear.py:8-54."""
    W1, b1 = p["W1"], p["b1"]
    NH, F = XQ.shape[1], XQ.shape[-1]
    ln_w = p["ln_w"].reshape(NH, 1, F)
    ln_b = p["ln_b"].reshape(NH, 1, F)

    Z1 = XK @ W1 + b1
    target = XV - XK
    gZ1 = ln_fused_l2_bwd(Z1, target, ln_w, ln_b)
    Attn1 = XQ @ XK.transpose(-2, -1)
    b1_bar = b1 - eta @ gZ1
    Z1_bar = XQ @ W1 - (eta * Attn1) @ gZ1 + b1_bar

    last_eta = eta[:, :, -1, :, None]
    W1n = W1 - (last_eta * XK).transpose(-1, -2) @ gZ1
    b1n = b1 - torch.sum(last_eta * gZ1, dim=-2, keepdim=True)
    out = XQ + ln_fwd(Z1_bar, ln_w, ln_b)
    return dict(p, W1=W1n, b1=b1n), out


def scan_dual(kind: str, XQ, XK, XV, eta, ln_w, ln_b, W1, b1, W2=None, b2=None, checkpoint_group_size: int = 0):
    """Whole-sequence dual-form scan: ops/ttt_mlp.py:70-99 (resp. ttt_linear.py:57-84) with
    ssm/utils.py:111-146's loop.  ``checkpoint_group_size`` > 0 wraps groups of that many steps in
    ``torch.utils.checkpoint(use_reentrant=False)`` exactly as the reference's scan does (:131-142): a memory
    device, not arithmetic - used only by bench.py's timed CPU baseline, where autograd runs through the scan.

    Inputs [B,NH,NC,CS,F], eta [B,NH,NC,CS,CS]; returns [B,NH,NC,CS,F] (the kernel layout;
    the reference permutes to [B,NC,CS,NH,F] at ops/ttt_mlp.py:99)."""
    keys = ("W1", "b1", "W2", "b2") if kind == "mlp" else ("W1", "b1")
    consts = {"ln_w": ln_w, "ln_b": ln_b}
    step = mlp_step_dual if kind == "mlp" else linear_step_dual
    NC = XQ.shape[2]

    def run_group(lo, hi, *state):
        p = dict(consts, **dict(zip(keys, state)))
        outs = []
        for i in range(lo, hi):
            p, o = step(p, XQ[:, :, i], XK[:, :, i], XV[:, :, i], eta[:, :, i])
            outs.append(o)
        return (torch.stack(outs, dim=2),) + tuple(p[k] for k in keys)

    state = (W1, b1, W2, b2) if kind == "mlp" else (W1, b1)
    G = checkpoint_group_size if checkpoint_group_size > 0 else NC
    outs = []
    for lo in range(0, NC, G):
        hi = min(lo + G, NC)
        if checkpoint_group_size > 0:
            from torch.utils.checkpoint import checkpoint
            res = checkpoint(run_group, lo, hi, *state, use_reentrant=False)
        else:
            res = run_group(lo, hi, *state)
        outs.append(res[0])
        state = res[1:]
    return torch.cat(outs, dim=2), dict(consts, **dict(zip(keys, state)))


# --------------------------------------------------------------------------- primal form, kernel contract
def _ln_stats(x, eps):
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, keepdim=True, unbiased=False)
    std = torch.sqrt(var + eps)
    return (x - mu) / std, std


def _mlp_step_primal(W1, b1, W2, b2, Q, K, V, eta, gam, bet, eps):
    """SURVEY.md Appendix A forward.  eta [B,NH,CS,1] (last row of the tile, as a column)."""
    Fd = Q.shape[-1]
    Z1 = K @ W1 + b1
    X2 = gelu_tanh(Z1)
    Z2 = X2 @ W2 + b2
    xh, std = _ln_stats(Z2, eps)
    go = gam * xh + bet - (V - K)
    gxh = go * gam
    gZ2 = (Fd * gxh - gxh.sum(-1, keepdim=True) - xh * (gxh * xh).sum(-1, keepdim=True)) / (Fd * std)
    D1 = gelu_bwd(Z1)
    gX2 = gZ2 @ W2.transpose(-1, -2)
    gZ1 = gX2 * D1
    W1n = W1 - (eta * K).transpose(-1, -2) @ gZ1
    b1n = b1 - (eta * gZ1).sum(-2, keepdim=True)
    W2n = W2 - (eta * X2).transpose(-1, -2) @ gZ2
    b2n = b2 - (eta * gZ2).sum(-2, keepdim=True)
    Z1b = Q @ W1n + b1n
    X2b = gelu_tanh(Z1b)
    Z2b = X2b @ W2n + b2n
    xhl, stdl = _ln_stats(Z2b, eps)
    out = Q + gam * xhl + bet
    saved = dict(Z1=Z1, X2=X2, xh=xh, std=std, go=go, gxh=gxh, gZ2=gZ2, D1=D1, gX2=gX2, gZ1=gZ1,
                 Z1b=Z1b, X2b=X2b, xhl=xhl, stdl=stdl, W1n=W1n, W2n=W2n)
    return (W1n, b1n, W2n, b2n), out, saved


def mlp_forward(XQ, XK, XV, last_eta, ln_w, ln_b, W1, b1, W2, b2, G: int, eps: float = LN_EPS):
    """TTT-MLP forward with the ``test_time_training.ttt_forward`` contract (mlp_tk.py:116-133).

    XQ/XK/XV [B,NH,NC,CS,F]; last_eta [B,NH,NC,CS,1]; ln_w/ln_b [1,NH,1,F] (or [NH,F]);
    W1 [B,NH,F,H] b1 [B,NH,1,H] W2 [B,NH,H,F] b2 [B,NH,1,F].
    Returns XQW [B,NH,NC,CS,F] and the four checkpoint tensors [B,NH,K,...] holding the state
    *entering* steps 0, G, 2G, ...  (K = ceil(NC/G))."""
    B, NH, NC, CS, Fd = XQ.shape
    gam = ln_w.reshape(1, NH, 1, Fd)
    bet = ln_b.reshape(1, NH, 1, Fd)
    K = math.ceil(NC / G)
    st = (W1, b1, W2, b2)
    ck = [[], [], [], []]
    outs = []
    for i in range(NC):
        if i % G == 0:
            for c, s in zip(ck, st):
                c.append(s)
        st, o, _ = _mlp_step_primal(*st, XQ[:, :, i], XK[:, :, i], XV[:, :, i], last_eta[:, :, i], gam, bet, eps)
        outs.append(o)
    cks = tuple(torch.stack(c, dim=2) for c in ck)
    assert cks[0].shape[2] == K
    return torch.stack(outs, dim=2), cks, st


def _ln_l2_bwd_bwd(G_, xh, std, go, gxh, gZ2, gam, Fd):
    """Backward of ln_fused_l2_bwd w.r.t. (x, gamma, beta, target) given G_ = dL/d(gZ2).
    SURVEY.md Appendix A; structure follows kernels/linear_backward.py:137-169."""
    r = 1.0 / std
    mGr = -G_ * r
    s1 = mGr.sum(-1, keepdim=True)
    s2 = (mGr * xh).sum(-1, keepdim=True)
    dgxh = r * G_ + s1 / Fd + xh * s2 / Fd
    dy = gam * dgxh
    dgam = (go * dgxh + dy * xh)
    dbet = dy
    dt = -dy
    dxh = dy * gam + gxh * s2 / Fd + (gxh * xh).sum(-1, keepdim=True) * mGr / Fd
    dstd = -dxh * xh * r - G_ * gZ2 * r
    dZ = dxh * r - dxh.sum(-1, keepdim=True) * r / Fd + dstd.sum(-1, keepdim=True) * xh / Fd
    return dZ, dgam, dbet, dt


def _ln_bwd(dy, xh, std, gam, Fd):
    """LayerNorm input-gradient (kernels/linear_backward.py:103-115)."""
    dxh = dy * gam
    return (Fd * dxh - dxh.sum(-1, keepdim=True) - xh * (dxh * xh).sum(-1, keepdim=True)) / (Fd * std)


def _mlp_step_bwd(st_in, Q, K, V, eta, gam, bet, eps, dOut, dst):
    """Reverse of one primal step.  dst = (dW1', db1', dW2', db2') flowing from later steps."""
    W1, b1, W2, b2 = st_in
    Fd = Q.shape[-1]
    (W1n, b1n, W2n, b2n), _, s = _mlp_step_primal(W1, b1, W2, b2, Q, K, V, eta, gam, bet, eps)
    dW1n, db1n, dW2n, db2n = dst
    T = lambda x: x.transpose(-1, -2)

    # out = Q + LN(Z2b)
    dgam = (dOut * s["xhl"]).sum(-2, keepdim=True)
    dbet = dOut.sum(-2, keepdim=True)
    dZ2b = _ln_bwd(dOut, s["xhl"], s["stdl"], gam, Fd)
    dW2n = dW2n + T(s["X2b"]) @ dZ2b
    db2n = db2n + dZ2b.sum(-2, keepdim=True)
    dX2b = dZ2b @ T(W2n)
    dZ1b = dX2b * gelu_bwd(s["Z1b"])
    dW1n = dW1n + T(Q) @ dZ1b
    db1n = db1n + dZ1b.sum(-2, keepdim=True)
    dQ = dOut + dZ1b @ T(W1n)

    # state updates  W' = W - (eta*X)^T g ; b' = b - sum(eta*g)
    A2 = s["gZ2"] @ T(dW2n)            # [CS,H]
    dgZ2 = -(eta * s["X2"]) @ dW2n - eta * db2n
    dX2 = -eta * A2
    A1 = s["gZ1"] @ T(dW1n)            # [CS,F]
    dgZ1 = -(eta * K) @ dW1n - eta * db1n
    dK = -eta * A1
    deta = (-(s["X2"] * A2).sum(-1, keepdim=True) - (s["gZ2"] * db2n).sum(-1, keepdim=True)
            - (K * A1).sum(-1, keepdim=True) - (s["gZ1"] * db1n).sum(-1, keepdim=True))

    # gZ1 = (gZ2 @ W2^T) * gelu'(Z1)
    u = dgZ1 * s["D1"]
    dgZ2 = dgZ2 + u @ W2
    dW2 = dW2n + T(T(s["gZ2"]) @ u)
    dZ1 = dgZ1 * s["gX2"] * gelu_bwd2(s["Z1"])

    # gZ2 = ln_fused_l2_bwd(Z2, V-K)
    dZ2, dgam2, dbet2, dt = _ln_l2_bwd_bwd(dgZ2, s["xh"], s["std"], s["go"], s["gxh"], s["gZ2"], gam, Fd)
    dgam = dgam + dgam2.sum(-2, keepdim=True)
    dbet = dbet + dbet2.sum(-2, keepdim=True)
    dV = dt
    dK = dK - dt

    # Z2 = X2 W2 + b2 ; X2 = gelu(Z1) ; Z1 = K W1 + b1
    dX2 = dX2 + dZ2 @ T(W2)
    dW2 = dW2 + T(s["X2"]) @ dZ2
    db2 = db2n + dZ2.sum(-2, keepdim=True)
    dZ1 = dZ1 + dX2 * s["D1"]
    dK = dK + dZ1 @ T(W1)
    dW1 = dW1n + T(K) @ dZ1
    db1 = db1n + dZ1.sum(-2, keepdim=True)
    return (dW1, db1, dW2, db2), dQ, dK, dV, deta, dgam, dbet


def mlp_backward(XQ, XK, XV, last_eta, ln_w, ln_b, cks, G: int, dXQW, dst_last=None, eps: float = LN_EPS):
    """TTT-MLP backward with the ``test_time_training.ttt_backward`` contract (mlp_tk.py:227-275):
    recompute each checkpoint group forward, sweep it in reverse carrying the state gradient.

    Returns dict with dXQ,dXK,dXV [B,NH,NC,CS,F], dlast_eta [B,NH,NC,CS,1],
    dW1,db1,dW2,db2 (gradient w.r.t. the initial state, [B,NH,...]),
    dln_w,dln_b [B,NH,1,F] (per batch element; the caller sums over B, mlp_tk.py:277-278)."""
    B, NH, NC, CS, Fd = XQ.shape
    gam = ln_w.reshape(1, NH, 1, Fd)
    bet = ln_b.reshape(1, NH, 1, Fd)
    W1c, b1c, W2c, b2c = cks
    if dst_last is None:
        dst = tuple(torch.zeros_like(c[:, :, 0]) for c in cks)
    else:
        dst = dst_last
    dQ = torch.zeros_like(XQ)
    dK = torch.zeros_like(XQ)
    dV = torch.zeros_like(XQ)
    deta = torch.zeros_like(last_eta)
    dgam = torch.zeros(B, NH, 1, Fd, dtype=XQ.dtype)
    dbet = torch.zeros(B, NH, 1, Fd, dtype=XQ.dtype)
    Kc = W1c.shape[2]
    for k in reversed(range(Kc)):
        lo, hi = k * G, min((k + 1) * G, NC)
        st = (W1c[:, :, k], b1c[:, :, k], W2c[:, :, k], b2c[:, :, k])
        states = []
        for i in range(lo, hi):
            states.append(st)
            st, _, _ = _mlp_step_primal(*st, XQ[:, :, i], XK[:, :, i], XV[:, :, i], last_eta[:, :, i], gam, bet, eps)
        for i in reversed(range(lo, hi)):
            dst, q, kk, v, e, g_, b_ = _mlp_step_bwd(states[i - lo], XQ[:, :, i], XK[:, :, i], XV[:, :, i],
                                                     last_eta[:, :, i], gam, bet, eps, dXQW[:, :, i], dst)
            dQ[:, :, i], dK[:, :, i], dV[:, :, i], deta[:, :, i] = q, kk, v, e
            dgam += g_
            dbet += b_
    return dict(dXQ=dQ, dXK=dK, dXV=dV, dlast_eta=deta, dW1=dst[0], db1=dst[1], dW2=dst[2], db2=dst[3],
                dln_w=dgam, dln_b=dbet)


# ---- TTT-Linear ------------------------------------------------------------------------------
def _lin_step_primal(W1, b1, Q, K, V, eta, gam, bet, eps):
    Fd = Q.shape[-1]
    Z1 = K @ W1 + b1
    xh, std = _ln_stats(Z1, eps)
    go = gam * xh + bet - (V - K)
    gxh = go * gam
    gZ1 = (Fd * gxh - gxh.sum(-1, keepdim=True) - xh * (gxh * xh).sum(-1, keepdim=True)) / (Fd * std)
    W1n = W1 - (eta * K).transpose(-1, -2) @ gZ1
    b1n = b1 - (eta * gZ1).sum(-2, keepdim=True)
    Z1b = Q @ W1n + b1n
    xhl, stdl = _ln_stats(Z1b, eps)
    out = Q + gam * xhl + bet
    return (W1n, b1n), out, dict(xh=xh, std=std, go=go, gxh=gxh, gZ1=gZ1, xhl=xhl, stdl=stdl)


def linear_forward(XQ, XK, XV, last_eta, ln_w, ln_b, W1, b1, G: int, eps: float = LN_EPS):
    """TTT-Linear forward, primal form (kernels/linear_forward.py:54-145 structure, eps as the ops path)."""
    B, NH, NC, CS, Fd = XQ.shape
    gam = ln_w.reshape(1, NH, 1, Fd)
    bet = ln_b.reshape(1, NH, 1, Fd)
    st = (W1, b1)
    ck = [[], []]
    outs = []
    for i in range(NC):
        if i % G == 0:
            ck[0].append(st[0]); ck[1].append(st[1])
        st, o, _ = _lin_step_primal(*st, XQ[:, :, i], XK[:, :, i], XV[:, :, i], last_eta[:, :, i], gam, bet, eps)
        outs.append(o)
    return torch.stack(outs, dim=2), (torch.stack(ck[0], 2), torch.stack(ck[1], 2)), st


def _lin_step_bwd(st_in, Q, K, V, eta, gam, bet, eps, dOut, dst):
    W1, b1 = st_in
    Fd = Q.shape[-1]
    (W1n, b1n), _, s = _lin_step_primal(W1, b1, Q, K, V, eta, gam, bet, eps)
    dW1n, db1n = dst
    T = lambda x: x.transpose(-1, -2)
    dgam = (dOut * s["xhl"]).sum(-2, keepdim=True)
    dbet = dOut.sum(-2, keepdim=True)
    dZ1b = _ln_bwd(dOut, s["xhl"], s["stdl"], gam, Fd)
    dW1n = dW1n + T(Q) @ dZ1b
    db1n = db1n + dZ1b.sum(-2, keepdim=True)
    dQ = dOut + dZ1b @ T(W1n)
    A1 = s["gZ1"] @ T(dW1n)
    dgZ1 = -(eta * K) @ dW1n - eta * db1n
    dK = -eta * A1
    deta = -(K * A1).sum(-1, keepdim=True) - (s["gZ1"] * db1n).sum(-1, keepdim=True)
    dZ1, dgam2, dbet2, dt = _ln_l2_bwd_bwd(dgZ1, s["xh"], s["std"], s["go"], s["gxh"], s["gZ1"], gam, Fd)
    dgam = dgam + dgam2.sum(-2, keepdim=True)
    dbet = dbet + dbet2.sum(-2, keepdim=True)
    dV = dt
    dK = dK - dt + dZ1 @ T(W1)
    dW1 = dW1n + T(K) @ dZ1
    db1 = db1n + dZ1.sum(-2, keepdim=True)
    return (dW1, db1), dQ, dK, dV, deta, dgam, dbet


def linear_backward(XQ, XK, XV, last_eta, ln_w, ln_b, cks, G: int, dXQW, eps: float = LN_EPS):
    B, NH, NC, CS, Fd = XQ.shape
    gam = ln_w.reshape(1, NH, 1, Fd)
    bet = ln_b.reshape(1, NH, 1, Fd)
    W1c, b1c = cks
    dst = (torch.zeros_like(W1c[:, :, 0]), torch.zeros_like(b1c[:, :, 0]))
    dQ = torch.zeros_like(XQ); dK = torch.zeros_like(XQ); dV = torch.zeros_like(XQ)
    deta = torch.zeros_like(last_eta)
    dgam = torch.zeros(B, NH, 1, Fd, dtype=XQ.dtype)
    dbet = torch.zeros(B, NH, 1, Fd, dtype=XQ.dtype)
    for k in reversed(range(W1c.shape[2])):
        lo, hi = k * G, min((k + 1) * G, NC)
        st = (W1c[:, :, k], b1c[:, :, k])
        states = []
        for i in range(lo, hi):
            states.append(st)
            st, _, _ = _lin_step_primal(*st, XQ[:, :, i], XK[:, :, i], XV[:, :, i], last_eta[:, :, i], gam, bet, eps)
        for i in reversed(range(lo, hi)):
            dst, q, kk, v, e, g_, b_ = _lin_step_bwd(states[i - lo], XQ[:, :, i], XK[:, :, i], XV[:, :, i],
                                                     last_eta[:, :, i], gam, bet, eps, dXQW[:, :, i], dst)
            dQ[:, :, i], dK[:, :, i], dV[:, :, i], deta[:, :, i] = q, kk, v, e
            dgam += g_
            dbet += b_
    return dict(dXQ=dQ, dXK=dK, dXV=dV, dlast_eta=deta, dW1=dst[0], db1=dst[1], dln_w=dgam, dln_b=dbet)


# --------------------------------------------------------------------------- synthetic inputs
def make_inputs(kind: str, B, NH, NC, CS, Fd, seed=0, dtype=torch.float32, base_lr=None, identical_rows=True):
    """Seeded op-level inputs of SURVEY.md section 8(d): L2-normalised Q/K, randn V, eta =
    base_lr*sigmoid(randn)/(F*CS) tiled over rows, W ~ N(0,0.02^2), b = 0, ln_w = 1 (+noise), ln_b = 0 (+noise)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    H = 4 * Fd
    if base_lr is None:
        base_lr = 0.1 if kind == "mlp" else 1.0
    XQ = torch.nn.functional.normalize(rn(B, NH, NC, CS, Fd), dim=-1)
    XK = torch.nn.functional.normalize(rn(B, NH, NC, CS, Fd), dim=-1)
    XV = rn(B, NH, NC, CS, Fd)
    eta_row = base_lr * torch.sigmoid(rn(B, NH, NC, 1, CS)) / (Fd * CS)
    eta = eta_row.repeat(1, 1, 1, CS, 1)
    if not identical_rows:
        eta = eta * (1.0 + 0.5 * torch.rand(B, NH, NC, CS, CS, generator=g, dtype=torch.float64))
    d = dict(XQ=XQ, XK=XK, XV=XV, eta=eta,
             ln_w=1.0 + 0.1 * rn(NH, Fd), ln_b=0.1 * rn(NH, Fd),
             dOut=rn(B, NH, NC, CS, Fd))
    if kind == "mlp":
        d.update(W1=0.02 * rn(NH, Fd, H), b1=0.01 * rn(NH, 1, H), W2=0.02 * rn(NH, H, Fd), b2=0.01 * rn(NH, 1, Fd))
    else:
        d.update(W1=0.02 * rn(NH, Fd, Fd), b1=0.01 * rn(NH, 1, Fd))
    return {k: v.to(dtype) for k, v in d.items()}
