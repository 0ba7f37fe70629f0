"""Round-6 tests on the device (through the C ABI of libttt_hip.so): the TTT-MLP forward scan as a PAIR of workgroups per (b, h).

Role A (the state chain A1 .. C of csrc/ttt_mfma2.hip) publishes the updated state per step as the operand fragments of the output
path; role B (another CU) runs f6 / E / P6 from those records with the same instructions in the same order.  So the pair form must
return the BITS of the one-workgroup kernel (debug option ``scan_pair`` = 0) - outputs, state checkpoints, the state a part of the
sequence hands on - which the other GPU tests tie to the fp64 oracle and the reference-executed fixtures (what the reference
computes: ttt/models/ssm/ops/ttt_mlp.py:28-56)."""
import math

import pytest
import torch

from helpers import rel_l2
from oracle import ttt_oracle as O
from test_kernels_gpu import DEV, ext, oracle_on, round_acts, run_mlp
from test_parity_r5_gpu import _scan_inputs

pytestmark = pytest.mark.gpu
F, H = 64, 256


def _bufs(B, NH, NC, G):
    K = math.ceil(NC / G)
    cks = (torch.empty(B, NH, K, F, H, device=DEV), torch.empty(B, NH, K, 1, H, device=DEV),
           torch.empty(B, NH, K, H, F, device=DEV), torch.empty(B, NH, K, 1, F, device=DEV))
    for t in cks:
        t.fill_(float("nan"))
    return cks, torch.full((B, NH, NC, 64, F), float("nan"), device=DEV, dtype=torch.bfloat16)


def _forward(e, d, G, pair, cuts=None):
    B, NH, NC = d["XQ"].shape[:3]
    cks, out = _bufs(B, NH, NC, G)
    e.debug_option("scan_pair", pair)
    try:
        if cuts is None:
            e.ttt_forward(d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"], d["W1"], d["b1"], d["W2"], d["b2"], *cks, out, G)
            state = None
        else:
            state = [d[k].clone() for k in ("W1", "b1", "W2", "b2")]
            for s0, s1 in zip(cuts[:-1], cuts[1:]):
                e.ttt_forward_chunk(d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"], *state, *cks, out, G, s0, s1 - s0)
        torch.cuda.synchronize()
    finally:
        e.debug_option("scan_pair", 1)
    return out, cks, state


# 48 x 804 = the 9 s training geometry (one launch, 96 workgroups); 6 (b,h) = a launch whose role-B workgroups start at block 8;
# NC < RING, NC < G, ragged last group; 2 x 48 = the batched pair of a sampler step at mini-batches of 64 (192 workgroups)
@pytest.mark.parametrize("B,NH,NC,G", [(1, 48, 804, 16), (2, 3, 37, 4), (1, 8, 5, 16), (1, 2, 3, 16), (1, 5, 1, 1), (2, 48, 64, 16)])
def test_pair_scan_has_the_bits_of_the_one_workgroup_scan(B, NH, NC, G):
    e = ext()
    d = _scan_inputs(B, NH, NC, 11 + NC)
    e.set_impl("mfma")
    try:
        out0, cks0, _ = _forward(e, d, G, 0)
        for rep in range(2):                    # twice: the flag lines are re-zeroed in front of every launch
            out1, cks1, _ = _forward(e, d, G, 1)
            assert e.sweep_error() == 0
            assert not torch.isnan(out1.float()).any() and not any(torch.isnan(t).any() for t in cks1)
            assert torch.equal(out0, out1), rep
            for a, b in zip(cks0, cks1):
                assert torch.equal(a, b), rep
    finally:
        e.set_impl("auto")


def test_pair_scan_in_parts_hands_on_the_same_state():
    e = ext()
    B, NH, NC, G = 1, 48, 160, 16
    d = _scan_inputs(B, NH, NC, 3)
    e.set_impl("mfma")
    try:
        out0, cks0, _ = _forward(e, d, G, 0)
        _, _, st0 = _forward(e, d, G, 0, cuts=(0, 48, 112, 160))
        out1, cks1, st1 = _forward(e, d, G, 1, cuts=(0, 48, 112, 160))
    finally:
        e.set_impl("auto")
    assert torch.equal(out0, out1) and all(torch.equal(a, b) for a, b in zip(cks0, cks1))
    assert all(torch.equal(a, b) for a, b in zip(st0, st1))


def test_pair_scan_on_a_side_stream_beside_a_busy_chip():
    """What the pipelined layer forward does: the scan on a side stream while GEMMs fill the other CUs (role-B workgroups are
    dispatched when CUs come free: role A runs up to RING steps ahead and waits) - same bits, no hand-over error."""
    e = ext()
    B, NH, NC, G = 1, 48, 96, 16
    d = _scan_inputs(B, NH, NC, 17)
    e.set_impl("mfma")
    try:
        out0, cks0, _ = _forward(e, d, G, 0)
        a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
        side = torch.cuda.Stream()
        cks1, out1 = _bufs(B, NH, NC, G)
        torch.cuda.synchronize()
        for _ in range(6):
            a @ a                                # ~ 0.9 ms each on the whole chip
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            e.ttt_forward(d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"], d["W1"], d["b1"], d["W2"], d["b2"], *cks1, out1, G)
        for _ in range(12):
            a @ a
        torch.cuda.synchronize()
    finally:
        e.set_impl("auto")
    assert e.sweep_error() == 0
    assert torch.equal(out0, out1) and all(torch.equal(x, y) for x, y in zip(cks0, cks1))


def test_pair_scan_vs_oracle():
    e = ext()
    d = round_acts(O.make_inputs("mlp", 1, 4, 21, 64, 64, seed=123), torch.bfloat16)
    out, cks, g = run_mlp(e, d, 4, torch.bfloat16, impl="mfma")
    ro, rc, rg = oracle_on(d, 4, "mlp")
    assert rel_l2(out, ro) < 1e-2 and rel_l2(g["dW1"], rg["dW1"]) < 3e-2


def test_pair_scan_whose_second_workgroup_never_runs_is_loud():
    """Role B of every pair leaves at once (debug option): role A finds its ring full after RING steps, gives up after its bounded
    wait (2 ms under fault injection), stores the error word - the next extension call raises until the error is acknowledged."""
    e = ext()
    d = _scan_inputs(1, 2, 12, 5)
    e.set_impl("mfma")
    e.debug_option("scan_fault", 1)
    try:
        _forward(e, d, 4, 1)
    finally:
        e.debug_option("scan_fault", 0)
        e.set_impl("auto")
    assert e.sweep_error() != 0
    with pytest.raises(RuntimeError, match="hand-over"):
        _forward(e, d, 4, 1)
    e.sweep_error_clear()
    e.set_impl("mfma")
    try:
        out0, cks0, _ = _forward(e, d, 4, 0)
        out1, cks1, _ = _forward(e, d, 4, 1)
    finally:
        e.set_impl("auto")
    assert e.sweep_error() == 0 and torch.equal(out0, out1)
