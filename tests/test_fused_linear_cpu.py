"""``Linear3`` (ttt_amd/infra/fused_linear.py): the three projections of one input as a single autograd node must equal the three
``nn.Linear`` calls - outputs and every gradient, with unused outputs and frozen weights."""
import torch

from ttt_amd.infra import fused_linear as wgrad


def test_linear3_equals_three_linears():
    """The fused q/k/v node: outputs bit-equal to three F.linear calls, input gradient equal to the sum of the three
    (accumulated in the GEMM instead of by two additions), weight / bias gradients equal; unused outputs are handled."""
    torch.manual_seed(0)
    mods = [torch.nn.Linear(48, 32) for _ in range(3)]
    x = torch.randn(2, 7, 48, requires_grad=True)
    ys = wgrad.linear3(*mods, x)
    ref = [m(x) for m in mods]
    for a, b in zip(ys, ref):
        assert torch.equal(a, b)
    gs = [torch.randn_like(r) for r in ref]
    (ys[0] * gs[0]).sum().add((ys[2] * gs[2]).sum()).backward()          # the k output stays unused
    got = [x.grad.clone()] + [m.weight.grad.clone() if m.weight.grad is not None else None for m in mods]
    x.grad = None
    for m in mods:
        m.zero_grad()
    (ref[0] * gs[0]).sum().add((ref[2] * gs[2]).sum()).backward()
    want = [x.grad.clone()] + [m.weight.grad.clone() if m.weight.grad is not None else None for m in mods]
    assert torch.allclose(got[0], want[0], rtol=1e-5, atol=1e-6)
    assert got[2] is None and want[2] is None
    for g, w in ((got[1], want[1]), (got[3], want[3])):
        assert torch.allclose(g, w, rtol=1e-5, atol=1e-6)
