"""The TTT-MLP layer's forward as a pipeline over parts of the sequence (round 5).

The forward scan is sequential and occupies ``B * NH`` = 48 of the 256 CUs for 6.2 ms per call at the 9 s geometry (84 calls per
training step), with 208 CUs idle; the projections in front of it (``wq / wk / wv``: 2.2 ms, pre-processing 0.4 ms) and behind it
(``post_norm`` 0.2 ms, ``wo`` 0.75 ms) wait for it or make it wait.  Here the sequence is cut into parts at checkpoint-group
boundaries of the SCAN order; the scan walks part c on a side stream (``ttt_hip_mlp_forward_chunk``: it starts from the state the
previous part left and hands its own on in fp32 - the bits of the one-call scan) while the compute stream runs the projections +
pre-processing of part c + 1 and the post-norm + output projection of part c - 1.  The pre- / post-processing kernels take a range
of scan positions (``ttt_hip_pre_forward_range`` / ``ttt_hip_post_forward_range``); the GEMMs work on the token runs a part covers
(a part is a few contiguous runs of the [texts | video] sequence: the scan order interleaves scenes and may be time-reversed).

This is FORWARD work only and changes no autograd node: the pre-pass runs under ``no_grad`` and fills the very tensors the layer's
autograd Functions would produce (raw projections, scan output + state checkpoints, post-norm output, layer output); the Functions
are then called as always and TAKE those tensors instead of launching their kernels (``injected`` below), so the saved tensors, the
backward graph and the re-materialisation cache see what they always saw.

Reference order of operations: ttt/models/ssm/ttt_layer.py:314-334 (projections -> process_input -> ttt -> post_norm -> wo),
cogvideo/dit.py:224-266."""
from __future__ import annotations

import math
import threading

import torch

_BF16, _F32 = torch.bfloat16, torch.float32
_tls = threading.local()
_side = {}


def side_stream(device) -> torch.cuda.Stream:
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    st = _side.get(idx)
    if st is None:
        # TTT_SCAN_STREAM_PRIORITY (A/B knob, round 6): -1 = a high-priority queue for the scan - its workgroups take the CUs that come
        # free before the workgroups of the GEMMs queued beside it do; default 0
        import os
        st = _side[idx] = torch.cuda.Stream(device=idx, priority=int(os.environ.get("TTT_SCAN_STREAM_PRIORITY", "0")))
    return st


# ---- injection: results of the pre-pass that the autograd Functions take instead of computing -----------------------------------
class injecting:
    """``with injecting({"linear3": (q, k, v), "scan": (out, *cks), "post": y, "wo": o}): ...`` - every entry must be taken once."""

    def __init__(self, results):
        self.results = dict(results)

    def __enter__(self):
        assert getattr(_tls, "inj", None) is None, "nested injection"
        _tls.inj = self.results
        return self

    def __exit__(self, et, ev, tb):
        left = list(_tls.inj)
        _tls.inj = None
        if et is None and left:
            raise RuntimeError(f"ttt pipeline: injected results {left} were not consumed (the layer's call sequence changed?)")
        return False


def injected(kind):
    """the pre-pass's result of this kind (removed from the set), or None outside a pipelined forward"""
    inj = getattr(_tls, "inj", None)
    return None if inj is None else inj.pop(kind, None)


class InjectedLinear(torch.autograd.Function):
    """``F.linear(x, w, b)`` whose forward takes the pre-pass's output; the backward is the linear layer's
    (dx = dy w, dw = dy^T x, db = sum dy - the products autograd forms for ``F.linear``)."""

    @staticmethod
    def forward(ctx, x, w, b):
        out = injected("wo")
        if out is None:
            out = torch.nn.functional.linear(x, w, b)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dy.matmul(w) if need_x else None
        dw = dy2.t().mm(x.reshape(-1, x.shape[-1])) if need_w else None
        db = dy2.sum(0) if (need_b and ctx.has_bias) else None
        return dx, dw, db


# ---- the plan: parts of the scan order and the token runs they cover ---------------------------------------------------------------
# Shares of the checkpoint groups per part.  Round 6: with the pair scan (4.9 ms on 96 CUs beside the GEMMs) the compute stream is the
# pipeline's co-bottleneck - its GEMMs run 1.7 - 1.85x slower on the 160 CUs the scan leaves (profiles/r6x_layer_fwd_timeline_parts4.txt) -
# and the forward ends with the LAST part's scan running alone (0.6 ms at four equal parts) + its output projection.  Parts that TAPER
# towards the end cut that tail; a two-resource model of the timeline fitted to the trace puts the optimum for four parts near these
# shares (6.26 against 6.50 ms).  TTT_PIPELINE_WEIGHTS="w0,w1,..." overrides (A/B knob; "equal" = the round-5 plan).
TAPER = {4: (0.31, 0.33, 0.24, 0.12), 5: (0.31, 0.31, 0.22, 0.12, 0.04), 6: (0.26, 0.27, 0.21, 0.14, 0.08, 0.04)}


def part_group_counts(K: int, n_parts: int):
    import os
    env = os.environ.get("TTT_PIPELINE_WEIGHTS", "")
    w = None
    if env and env != "equal":
        w = [float(v) for v in env.split(",")]
        if len(w) != n_parts:
            w = None
    elif env != "equal":
        w = TAPER.get(n_parts)
        if w is None and n_parts >= 7:              # long scans (63 s: 8 parts): the same shape - the last three parts at 0.8 / 0.5 / 0.2 of a full one
            w = [1.1] * (n_parts - 4) + [1.0, 0.8, 0.5, 0.2]
    if w is None:
        per, extra = divmod(K, n_parts)
        return [per + (1 if c < extra else 0) for c in range(n_parts)]
    tot = sum(w)
    counts = [max(1, int(round(K * v / tot))) for v in w]
    while sum(counts) > K:                      # (rounding: take from / give to the largest part)
        counts[counts.index(max(counts))] -= 1
    while sum(counts) < K:
        counts[counts.index(max(counts))] += 1
    return counts


def plan_parts(src_cpu, L: int, CS: int, G: int, n_parts: int):
    """[(step0, nsteps, [(r0, r1), ...])]: parts of whole checkpoint groups, as equal as they come; ``src_cpu`` maps scan position ->
    token (None: identity).  The runs of a part are the maximal contiguous token ranges it covers, ascending."""
    NC = L // CS
    K = math.ceil(NC / G)
    n_parts = max(1, min(n_parts, K))
    counts = part_group_counts(K, n_parts)
    parts, g0 = [], 0
    for c in range(n_parts):
        g1 = g0 + counts[c]
        s0, s1 = g0 * G, min(g1 * G, NC)
        t0, t1 = s0 * CS, s1 * CS
        if src_cpu is None:
            runs = [(t0, t1)]
        else:
            toks = torch.sort(src_cpu[t0:t1].to(torch.int64)).values
            cut = torch.nonzero(toks[1:] != toks[:-1] + 1).flatten() + 1
            edges = [0] + cut.tolist() + [toks.numel()]
            runs = [(int(toks[a]), int(toks[b - 1]) + 1) for a, b in zip(edges[:-1], edges[1:])]
        parts.append((s0, s1 - s0, runs))
        g0 = g1
    return parts


def prepass(ext, x, wq, wk, wv, wo, post_norm, ln_w32, ln_b32, rope, src, pos, n_pos, NH, state, last_eta, G, parts):
    """Fills and returns {"linear3": (XQr, XKr, XVr), "scan": (out, W1c, b1c, W2c, b2c), "post": y, "wo": o} (no autograd)."""
    B, L, D = x.shape
    Fh = D // NH
    CS = last_eta.shape[-2]
    NC = L // CS
    K = math.ceil(NC / G)
    dev = x.device
    main, side = torch.cuda.current_stream(dev), side_stream(dev)
    with torch.no_grad():
        e16 = lambda *s: torch.empty(*s, device=dev, dtype=_BF16)
        e32 = lambda *s: torch.empty(*s, device=dev, dtype=_F32)
        XQr, XKr, XVr, y, o = (e16(B, L, D) for _ in range(5))
        XQ, XK, XV = (e16(B, NH, L, Fh) for _ in range(3))
        out = e16(B, NH, NC, CS, Fh)
        cks = (e32(B, NH, K, Fh, 4 * Fh), e32(B, NH, K, 1, 4 * Fh), e32(B, NH, K, 4 * Fh, Fh), e32(B, NH, K, 1, Fh))
        carry = [t.to(_F32).contiguous().clone() for t in state]          # the state entering the next part (the scan replaces it)
        lw, lb = ln_w32.reshape(1, NH, 1, Fh), ln_b32.reshape(1, NH, 1, Fh)
        pw32, pb32 = post_norm.weight.detach().to(_F32).contiguous(), post_norm.bias.detach().to(_F32).contiguous()
        mb = lambda t: t.view(B, NH, NC, CS, Fh)
        wts = [(m.weight.t(), m.bias, dst) for m, dst in ((wq, XQr), (wk, XKr), (wv, XVr))]
        wo_t, wo_b = wo.weight.t(), wo.bias

        def stage_in(c):           # projections + pre-processing of part c (compute stream)
            s0, ns, runs = parts[c]
            for r0, r1 in runs:
                for b in range(B):
                    xs = x[b, r0:r1]
                    for wt, bias, dst in wts:
                        if bias is None:
                            torch.mm(xs, wt, out=dst[b, r0:r1])
                        else:
                            torch.addmm(bias, xs, wt, out=dst[b, r0:r1])
            ext.pre_forward(XQr, XKr, XVr, rope, src, pos, ln_w32, ln_b32, XQ, XK, XV, NH, n_pos=n_pos, t0=s0 * CS, tn=ns * CS)

        def stage_out(c):          # post-norm + output projection of part c (compute stream)
            s0, ns, runs = parts[c]
            ext.post_forward(out.view(B, NH, L, Fh), src, pw32, pb32, y, float(post_norm.eps), t0=s0 * CS, tn=ns * CS)
            for r0, r1 in runs:
                for b in range(B):
                    if wo_b is None:
                        torch.mm(y[b, r0:r1], wo_t, out=o[b, r0:r1])
                    else:
                        torch.addmm(wo_b, y[b, r0:r1], wo_t, out=o[b, r0:r1])

        side.wait_stream(main)     # (the buffers above were allocated on the compute stream; nothing of them is in use there yet)
        done = []
        for c in range(len(parts)):
            stage_in(c)
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                s0, ns, _ = parts[c]
                ext.ttt_forward_chunk(mb(XQ), mb(XK), mb(XV), last_eta, lw, lb, *carry, *cks, out, G, s0, ns)
                ev = torch.cuda.Event()
                ev.record(side)
            done.append(ev)
            if c >= 1:             # (issued behind part c's projections: they are what the scan of part c waits for)
                main.wait_event(done[c - 1])
                stage_out(c - 1)
        main.wait_event(done[-1])
        stage_out(len(parts) - 1)
    return {"linear3": (XQr, XKr, XVr), "scan": (out, *cks), "post": y, "wo": o}
