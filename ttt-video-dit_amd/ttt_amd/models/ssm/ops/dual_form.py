"""Device-agnostic PyTorch implementation of the TTT scan in *dual form* with the full eta tile:
the ``use_kernel=False`` branch of the reference layer (``ttt/models/ssm/ttt_layer.py:383-395,
457-470`` -> ``ttt/models/ssm/ops/ttt_mlp.py`` / ``ttt_linear.py``).

It is an explicit opt-in mode of the module API, not a fallback: ``use_kernel`` defaults to True
and the HIP path raises when its library is missing.  It is kept because it is the only mode that
honours a general (non row-identical) eta tile, which the multi-scene ``interleave`` produces
(SURVEY.md hazard C2); the kernels implement the reference *kernel* contract (last row only).
Gradients come from autograd; groups of ``checkpoint_group_size`` steps are re-materialised with
``torch.utils.checkpoint`` like the reference's scan (ssm/utils.py:131-142).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

_EPS = 1e-8
_A, _C3 = 0.79788456, 0.1070322243


def _norm_stats(z):
    mu = z.mean(-1, keepdim=True)
    sd = torch.sqrt(z.var(-1, keepdim=True, unbiased=False) + _EPS)
    return (z - mu) / sd, sd


def _l2_target_grad(z, target, gamma, beta):
    """gradient of 0.5*||LN(z) - target||^2 w.r.t. z, closed form (ops/utils.py:21-48)."""
    n = z.shape[-1]
    zh, sd = _norm_stats(z)
    g = (gamma * zh + beta - target) * gamma
    return (n * g - g.sum(-1, keepdim=True) - zh * (g * zh).sum(-1, keepdim=True)) / (n * sd)


def _dgelu(x):
    t = torch.tanh(_A * x * (1 + 0.044715 * x * x))
    return 0.5 * x * ((1 - t * t) * (_A + _C3 * x * x)) + 0.5 * (1 + t)


def _mlp_group(W1, b1, W2, b2, gamma, beta, XQ, XK, XV, eta):
    outs = []
    for i in range(XQ.shape[0]):
        q, k, v, e = XQ[i], XK[i], XV[i], eta[i]
        z1 = k @ W1 + b1
        x2 = F.gelu(z1, approximate="tanh")
        z2 = x2 @ W2 + b2
        g2 = _l2_target_grad(z2, v - k, gamma, beta)
        g1 = (g2 @ W2.mT) * _dgelu(z1)
        z1b = q @ W1 - (e * (q @ k.mT)) @ g1 + (b1 - e @ g1)
        x2b = F.gelu(z1b, approximate="tanh")
        z2b = x2b @ W2 - (e * (x2b @ x2.mT)) @ g2 + (b2 - e @ g2)
        el = e[..., -1, :, None]
        W1 = W1 - (el * k).mT @ g1
        b1 = b1 - (el * g1).sum(-2, keepdim=True)
        W2 = W2 - (el * x2).mT @ g2
        b2 = b2 - (el * g2).sum(-2, keepdim=True)
        zh, _ = _norm_stats(z2b)
        outs.append(q + gamma * zh + beta)
    return W1, b1, W2, b2, torch.stack(outs)


def _lin_group(W1, b1, gamma, beta, XQ, XK, XV, eta):
    outs = []
    for i in range(XQ.shape[0]):
        q, k, v, e = XQ[i], XK[i], XV[i], eta[i]
        g1 = _l2_target_grad(k @ W1 + b1, v - k, gamma, beta)
        z1b = q @ W1 - (e * (q @ k.mT)) @ g1 + (b1 - e @ g1)
        el = e[..., -1, :, None]
        W1 = W1 - (el * k).mT @ g1
        b1 = b1 - (el * g1).sum(-2, keepdim=True)
        zh, _ = _norm_stats(z1b)
        outs.append(q + gamma * zh + beta)
    return W1, b1, torch.stack(outs)


def _scan(group_fn, state, gamma, beta, XQ, XK, XV, eta, G):
    # iterate mini-batch-major
    xs = [t.permute(2, 0, 1, 3, 4) for t in (XQ, XK, XV, eta)]
    NC = xs[0].shape[0]
    chunks = []
    for lo in range(0, NC, G):
        sl = [t[lo:lo + G] for t in xs]
        if torch.is_grad_enabled():
            *state, out = checkpoint(group_fn, *state, gamma, beta, *sl, use_reentrant=False)
        else:
            *state, out = group_fn(*state, gamma, beta, *sl)
        chunks.append(out)
    out = torch.cat(chunks, dim=0)            # [NC, B, NH, CS, F]
    return out.permute(1, 0, 3, 2, 4)         # [B, NC, CS, NH, F] like ops/ttt_mlp.py:99


def ttt_mlp(XK, XQ, XV, eta, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, W2_init, b2_init, checkpoint_group_size):
    NH, Fd = XQ.shape[1], XQ.shape[-1]
    gamma = ttt_norm_weight.reshape(NH, 1, Fd)
    beta = ttt_norm_bias.reshape(NH, 1, Fd)
    return _scan(_mlp_group, [W1_init, b1_init, W2_init, b2_init], gamma, beta, XQ, XK, XV, eta, max(int(checkpoint_group_size), 1))


def ttt_linear(XK, XQ, XV, eta, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, checkpoint_group_size):
    NH, Fd = XQ.shape[1], XQ.shape[-1]
    gamma = ttt_norm_weight.reshape(NH, 1, Fd)
    beta = ttt_norm_bias.reshape(NH, 1, Fd)
    return _scan(_lin_group, [W1_init, b1_init], gamma, beta, XQ, XK, XV, eta, max(int(checkpoint_group_size), 1))
