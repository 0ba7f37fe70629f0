"""Pins the CPU oracle against golden vectors produced by executing the reference's own
PyTorch ops path + torch.autograd (tests/golden/gen_golden.py).  CPU only."""
import pytest
import torch

from oracle import ttt_oracle as O
from helpers import load_golden, op_inputs, rel_l2, tile_states

F64 = ["op_mlp_f64_cs16.pt", "op_mlp_f64_cs64.pt", "op_lin_f64_cs16.pt", "op_lin_f64_cs64.pt"]
F32 = ["op_mlp_f32_b2.pt", "op_lin_f32_b2.pt"]
ROWS = ["op_mlp_f64_rows.pt", "op_lin_f64_rows.pt"]


def _run_primal(g, d):
    kind, G = g["kind"], g["G"]
    B = d["XQ"].shape[0]
    st = tile_states(d, B)
    last_eta = d["eta"][:, :, :, -1, :, None].contiguous()
    if kind == "mlp":
        out, cks, _ = O.mlp_forward(d["XQ"], d["XK"], d["XV"], last_eta, d["ln_w"], d["ln_b"],
                                    st["W1"], st["b1"], st["W2"], st["b2"], G)
        grads = O.mlp_backward(d["XQ"], d["XK"], d["XV"], last_eta, d["ln_w"], d["ln_b"], cks, G, d["dOut"])
    else:
        out, cks, _ = O.linear_forward(d["XQ"], d["XK"], d["XV"], last_eta, d["ln_w"], d["ln_b"],
                                       st["W1"], st["b1"], G)
        grads = O.linear_backward(d["XQ"], d["XK"], d["XV"], last_eta, d["ln_w"], d["ln_b"], cks, G, d["dOut"])
    return out, grads


def _check(g, out, grads, tol):
    ref = g["ref"]
    assert rel_l2(out, ref["XQW"]) < tol
    assert rel_l2(grads["dXQ"], ref["dXQ"]) < tol
    assert rel_l2(grads["dXK"], ref["dXK"]) < tol
    assert rel_l2(grads["dXV"], ref["dXV"]) < tol
    # kernels put the whole eta gradient in the last row (mlp_tk.py:280); autograd spreads it over rows
    assert rel_l2(grads["dlast_eta"].squeeze(-1), ref["deta"].sum(dim=-2)) < tol
    assert rel_l2(grads["dW1"], ref["dW1_states"]) < tol
    assert rel_l2(grads["db1"], ref["db1_states"]) < tol
    if g["kind"] == "mlp":
        assert rel_l2(grads["dW2"], ref["dW2_states"]) < tol
        assert rel_l2(grads["db2"], ref["db2_states"]) < tol
    # per-batch LN grads are summed by the caller (mlp_tk.py:277-278)
    assert rel_l2(grads["dln_w"].sum(0).squeeze(1), ref["dln_w"]) < tol
    assert rel_l2(grads["dln_b"].sum(0).squeeze(1), ref["dln_b"]) < tol


@pytest.mark.parametrize("name", F64)
def test_primal_matches_reference_fp64(name):
    g = load_golden(name)
    d = op_inputs(g)
    out, grads = _run_primal(g, d)
    _check(g, out, grads, 1e-9)


@pytest.mark.parametrize("name", F32)
def test_primal_matches_reference_fp32(name):
    g = load_golden(name)
    d = op_inputs(g)
    out, grads = _run_primal(g, d)
    _check(g, out, grads, 2e-4)
    # and in fp64 arithmetic on the same fp32 inputs (what GPU parity tests use as the target)
    d64 = {k: v.double() for k, v in d.items()}
    out, grads = _run_primal(g, d64)
    _check(g, out, grads, 2e-4)


@pytest.mark.parametrize("name", F64 + ROWS)
def test_dual_form_matches_reference(name):
    """The dual-form restatement must equal the reference for ANY eta tile (hazard C2)."""
    g = load_golden(name)
    d = op_inputs(g)
    st = tile_states(d, d["XQ"].shape[0])
    out, _ = O.scan_dual(g["kind"], d["XQ"], d["XK"], d["XV"], d["eta"], d["ln_w"], d["ln_b"],
                         st["W1"], st["b1"], st.get("W2"), st.get("b2"))
    assert rel_l2(out, g["ref"]["XQW"]) < 1e-10


@pytest.mark.parametrize("name", ROWS)
def test_last_row_contract_differs_when_rows_differ(name):
    """Documents hazard C2: with non-identical eta rows the kernel (last-row) contract is NOT the
    dual form - the primal oracle must differ from the reference there."""
    g = load_golden(name)
    d = op_inputs(g)
    out, _ = _run_primal(g, d)
    assert rel_l2(out, g["ref"]["XQW"]) > 1e-4


def test_gelu_second_derivative_matches_autograd():
    x = torch.linspace(-4, 4, 401, dtype=torch.float64, requires_grad=True)
    (g1,) = torch.autograd.grad(O.gelu_bwd(x).sum(), x)
    assert torch.allclose(g1, O.gelu_bwd2(x.detach()), atol=1e-7)
    y = torch.nn.functional.gelu(x, approximate="tanh")
    (d1,) = torch.autograd.grad(y.sum(), x)
    assert torch.allclose(d1, O.gelu_bwd(x.detach()), atol=1e-7)
    assert torch.allclose(y, O.gelu_tanh(x), atol=1e-12)


# ---- attention oracle (oracle/attn_oracle.py) -------------------------------------------------------------------------
def test_attention_oracle_matches_sdpa_fp64():
    import torch.nn.functional as F
    from oracle import attn_oracle as AO
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 3, 70, 64, generator=g, dtype=torch.float64) for _ in range(3))
    out, lse = AO.attention(q, k, v)
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    assert (out - ref).abs().max() < 1e-12
    assert (lse - torch.logsumexp(q @ k.transpose(-1, -2) / 8.0, -1)).abs().max() < 1e-12


def test_qk_pre_oracle_matches_module_path_fp64():
    """oracle qk_pre == q_norm / k_norm (nn.LayerNorm(64, eps=1e-6)) + Rotary3DPositionEmbedding.forward on the video
    tokens, i.e. the statements of SeqModelingBlock._segment (reference cogvideo/dit.py:184-195), in fp64."""
    from oracle import attn_oracle as AO
    from ttt_amd.models.cogvideo.utils import Rotary3DPositionEmbedding
    g = torch.Generator().manual_seed(1)
    B, S, NH, n_text = 2, 90, 3, 20
    rot = Rotary3DPositionEmbedding(4, 6, 5, 64).double()
    rot.freqs_cos, rot.freqs_sin = rot.freqs_cos.double(), rot.freqs_sin.double()
    qn, kn = torch.nn.LayerNorm(64, eps=1e-6).double(), torch.nn.LayerNorm(64, eps=1e-6).double()
    for m in (qn, kn):
        m.weight.data = 1 + 0.3 * torch.randn(64, generator=g, dtype=torch.float64)
        m.bias.data = 0.2 * torch.randn(64, generator=g, dtype=torch.float64)
    q_raw, k_raw = (torch.randn(B, S, NH * 64, generator=g, dtype=torch.float64) for _ in range(2))
    heads = lambda t: t.view(B, S, NH, 64).transpose(1, 2)
    q, k = qn(heads(q_raw)), kn(heads(k_raw))
    q = torch.cat((q[:, :, :n_text], rot(q[:, :, n_text:])), dim=2)
    k = torch.cat((k[:, :, :n_text], rot(k[:, :, n_text:])), dim=2)
    oq, ok = AO.qk_pre(q_raw, k_raw, qn.weight, qn.bias, kn.weight, kn.bias, rot.freqs_cos, rot.freqs_sin, NH, n_text)
    assert (oq - q).abs().max() < 1e-12 and (ok - k).abs().max() < 1e-12
