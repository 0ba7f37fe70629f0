"""Round-4 parity tests on the device (all through the C ABI of libttt_hip.so)."""
import os

import pytest
import torch

from helpers import load_golden, rel_l2
from test_kernels_gpu import DEV, ext

pytestmark = pytest.mark.gpu


def test_reference_tkmlp_argument_lists_replayed_on_the_device():
    """What the REFERENCE's ``TkMLP`` passes its extension, replayed on the real binding: tests/golden/tkmlp_replay.pt holds
    the positional argument lists of ``ttt_forward`` (mlp_tk.py:116-133) and ``ttt_backward`` (mlp_tk.py:227-275) recorded
    while the reference's wrapper ran in the build container (gen_golden_r4.py), and the result of the reference's ops path
    (ops/ttt_mlp.py) for the same call.  Here every recorded input goes to the device as it was passed, every output / scratch
    buffer is allocated with the recorded dtype and shape (as TkMLP allocates them), the two entry points are called
    positionally, and what comes back is post-processed as ``TkMLP._backward_core`` does (sum over the batch for the LayerNorm
    gradients, mlp_tk.py:277-278; d(eta) in the last row, :280) and compared with the ops path: output 1e-2, gradients 3e-2
    (bf16 activations at the op boundary)."""
    e = ext()
    g = load_golden("tkmlp_replay.pt")

    def materialise(spec, fwd_args=None):
        args = []
        for a in spec:
            if a["role"] == "int":
                args.append(a["value"])
            elif a["role"] == "in":
                args.append(a["value"].to(DEV).contiguous())
            elif a["role"] == "fwd_out":
                args.append(fwd_args[a["index"]])
            else:                                   # TkMLP zero-fills its outputs; NaN here: the kernel must write every element
                t = torch.empty(a["shape"], device=DEV, dtype=a["dtype"])
                args.append(t.fill_(float("nan")))
        return args

    fa = materialise(g["forward_args"])
    assert len(fa) == 16
    e.ttt_forward(*fa)
    ba = materialise(g["backward_args"], fa)
    assert len(ba) == 43
    e.ttt_backward(*ba)
    torch.cuda.synchronize()
    assert e.sweep_error() == 0
    out = fa[14]
    dlnw, dlnb, dW1, db1, dW2, db2, dle, dXQ, dXK, dXV = ba[32:42]
    got = {"ln_w": dlnw.sum(0).squeeze(1), "ln_b": dlnb.sum(0).squeeze(1),
           # TkMLP receives the state tiled over the batch (ttt_layer.py:431-436); autograd sums the tiles' gradients
           "W1": dW1.sum(0), "b1": db1.sum(0), "W2": dW2.sum(0), "b2": db2.sum(0),
           "XQ": dXQ, "XK": dXK, "XV": dXV, "eta_last_row_sum": dle.squeeze(-1)}
    errs = {"out": rel_l2(out, g["ops_out"])}
    for k, r in g["ops_grads"].items():
        assert got[k].shape == r.shape, (k, got[k].shape, r.shape)
        errs[k] = rel_l2(got[k], r)
    print("reference TkMLP argument lists on the device vs the reference ops path:", {k: round(v, 5) for k, v in errs.items()})
    assert errs["out"] < 1e-2, errs
    bad = {k: v for k, v in errs.items() if k != "out" and not v < 3e-2}
    assert not bad, (bad, errs)


def test_sweep_at_chunk_edges_vs_oracle_and_deterministic():
    """The shipped TTT-MLP backward sweep (csrc/ttt_mfma_bwd4.hip: bf16 hand-over records with the owners' partner-independent
    arithmetic under the record loads, derivers on waves 2, 3, bf16 inner-LayerNorm owner rows - each the winner of a round-4 A/B,
    the losing instantiations were removed in round 5) against the fp64 oracle head by head at the usual tolerances, and run-to-run
    deterministic.  NC = 70 with checkpoint groups of 16 and two groups per chunk: three chunks, the last of them short."""
    from oracle import ttt_oracle as O
    from test_kernels_gpu import oracle_on, round_acts, run_mlp
    from test_parity_r2_gpu import check_per_head
    e = ext()
    NH, NC, G = 8, 70, 16
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 64, 64, seed=8100), torch.bfloat16)
    ro, rc, rg = oracle_on(d, G, "mlp")
    e.debug_groups_per_chunk(2)
    try:
        out1, cks1, g1 = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
        out2, cks2, g2 = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
    finally:
        e.debug_groups_per_chunk(0)
    assert e.sweep_error() == 0
    check_per_head("TTT-MLP MFMA backward, three chunks", out1, cks1, g1, ro, rc, rg, 1e-2, 3e-2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), f"{k} differs between two identical calls"
