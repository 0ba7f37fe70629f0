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


def _run_sweep_variant(e, d, G, **opts):
    from test_kernels_gpu import run_mlp
    for k, v in opts.items():
        e.debug_option(k, v)
    e.debug_groups_per_chunk(2)
    try:
        return run_mlp(e, d, G, torch.bfloat16, impl="mfma")
    finally:
        e.debug_groups_per_chunk(0)
        sweep_defaults(e)


SWEEP_DEFAULTS = dict(sweep_records_bf16=1, sweep_deriver_wave0=2, own_bf16=1)      # the library's (csrc/ttt_mfma_bwd4.hip, ttt_mfma_bwd2.hip)


def sweep_defaults(e):
    for k, v in SWEEP_DEFAULTS.items():
        e.debug_option(k, v)


@pytest.mark.parametrize("bf16_records,deriver_wave0,own_bf16", [(1, 2, 0), (0, 4, 0), (1, 4, 0), (1, 2, 1)])
def test_sweep_schedule_and_record_variants(bf16_records, deriver_wave0, own_bf16):
    """Round-4 variants of the TTT-MLP backward sweep (csrc/ttt_mfma_bwd4.hip), each against the fp64 oracle head by head at the
    usual tolerances and required to be run-to-run deterministic: hand-over records that carry the partial d(gZ2) tiles as
    bf16 with the owners' partner-independent arithmetic under the record loads (debug option "sweep_records_bf16", the default
    since the round-4 A/B: 11.8 against 13.4 ms per backward at NC = 804) or the round-3 sweep (0); the deriver role on waves
    2, 3 (default: -2.5 % in three A/Bs) or 4, 5 ("sweep_deriver_wave0"); the inner LayerNorm's owner rows of the step record
    as bf16 ("own_bf16").  NC = 70 with checkpoint groups of 16 and two groups per chunk: three chunks, the last of them short.
    (No bit-equality ACROSS variants: separate instantiations, and the compiler contracts their multiply-adds differently -
    measured on dln_w.)"""
    from oracle import ttt_oracle as O
    from test_kernels_gpu import oracle_on, round_acts
    from test_parity_r2_gpu import check_per_head
    e = ext()
    NH, NC, G = 8, 70, 16
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 64, 64, seed=8100), torch.bfloat16)
    ro, rc, rg = oracle_on(d, G, "mlp")
    opts = dict(sweep_records_bf16=bf16_records, sweep_deriver_wave0=deriver_wave0, own_bf16=own_bf16)
    out1, cks1, g1 = _run_sweep_variant(e, d, G, **opts)
    out2, cks2, g2 = _run_sweep_variant(e, d, G, **opts)
    assert e.sweep_error() == 0
    check_per_head(f"TTT-MLP MFMA backward, bf16 records={bf16_records} derivers on waves {deriver_wave0}, {deriver_wave0 + 1} bf16 owner rows={own_bf16}",
                   out1, cks1, g1, ro, rc, rg, 1e-2, 3e-2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), f"{k} differs between two identical calls"


def test_forward_scan_half_chunk_swap_variant():
    """Debug option "scan_swap" of the TTT-MLP forward scan (csrc/ttt_mfma2.hip, template parameter SW): the two 8-byte units of
    every 16-byte chunk of an LDS tile row are stored swapped in rows with bit 3 ^ bit 4 set, which removes the 2-way bank
    conflicts of the accesses that walk rows at a fixed column (tools/lds_bank_model.py --half-swap).  A pure change of
    addresses: the variant is held to the fp64 oracle at the usual tolerances (outputs AND the checkpointed states, i.e. every
    LDS tile of every phase was read back as it was written) and to the other variant (same data, so the same result up to
    the contraction choices of a separate instantiation: reported, bounded far below a layout error)."""
    from oracle import ttt_oracle as O
    from test_kernels_gpu import oracle_on, round_acts, run_mlp
    from test_parity_r2_gpu import check_per_head
    e = ext()
    NH, NC, G = 8, 70, 16
    d = round_acts(O.make_inputs("mlp", 1, NH, NC, 64, 64, seed=8200), torch.bfloat16)
    ro, rc, rg = oracle_on(d, G, "mlp")
    res = {}
    try:
        for v in (0, 1):
            e.debug_option("scan_swap", v)
            res[v] = run_mlp(e, d, G, torch.bfloat16, impl="mfma")
            check_per_head(f"TTT-MLP MFMA forward scan, scan_swap={v}", *res[v], ro, rc, rg, 1e-2, 3e-2)
    finally:
        e.debug_option("scan_swap", SCAN_SWAP_DEFAULT)
    same = torch.equal(res[0][0], res[1][0]) and all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    worst = max([rel_l2(res[1][0], res[0][0])] + [rel_l2(a, b) for a, b in zip(res[1][1], res[0][1])])
    print(f"scan_swap 1 vs 0: outputs and checkpoints bit-identical = {same}, worst relative L2 difference {worst:.2e}")
    assert worst < 2e-3, worst


SCAN_SWAP_DEFAULT = 1          # the library's (csrc/ttt_mfma2.hip g_scan_swap)


def test_backward_schedule_options_same_bits():
    """Scheduling options of the TTT-MLP backward (csrc/ttt_mfma_bwd2.hip), both aimed at the race between a chunk's tail kernel and
    the next chunk's sweep that the rocprofv3 trace of round 4 shows (profiles/r4y_sweep_launches.txt): "tail_delay_us" (a one-wave
    gate kernel in front of each tail kernel on its side stream), "flags_memset_early" (the next sweep's hand-over flags cleared
    behind the current sweep instead of in front of the next) and "tail_gate_resident" (the gate waits until the next sweep's
    workgroups have counted themselves in).  Pure scheduling: every gradient must be bit-identical."""
    from oracle import ttt_oracle as O
    from test_kernels_gpu import round_acts, run_mlp
    e = ext()
    d = round_acts(O.make_inputs("mlp", 1, 8, 70, 64, 64, seed=8400), torch.bfloat16)
    res = []
    e.debug_groups_per_chunk(2)
    try:
        # the last combination - the gate that waits for the sweep's workgroups - was written after round 4's GPU budget had ended
        # and has never run on a device: opt-in (TTT_TEST_VARIANTS=1, set by tools/_run_next_round_first_call.sh) until it has
        combos = [(0, 0, 0), (25, 0, 0), (0, 1, 0), (25, 1, 0)] + ([(0, 1, 1)] if os.environ.get("TTT_TEST_VARIANTS", "0") == "1" else [])
        for delay, early, resident in combos:
            e.debug_option("tail_delay_us", delay)
            e.debug_option("flags_memset_early", early)
            e.debug_option("tail_gate_resident", resident)   # the tail's gate waits for the next sweep's workgroups (counted in a flag word)
            res.append(run_mlp(e, d, 16, torch.bfloat16, impl="mfma"))
            assert e.sweep_error() == 0, (delay, early, resident)
    finally:
        e.debug_option("tail_delay_us", 0)
        e.debug_option("tail_gate_resident", 0)
        e.debug_option("flags_memset_early", 1)              # the library's default (csrc/ttt_mfma_bwd2.hip g_memset_early)
        e.debug_groups_per_chunk(0)
    for other in res[1:]:
        for k in res[0][2]:
            assert torch.equal(res[0][2][k], other[2][k]), k
