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


def test_linear3_backward_over_column_blocks_of_one_buffer():
    """Round 5: when the three output gradients are the column blocks of ONE [.., n0 + n1 + n2] buffer (what the consumers' backward
    kernels write, ttt_amd/models/ssm/fused.py: qkv_grad_blocks), Linear3.backward forms the weight gradients as one GEMM over the
    concatenation and the input gradient as one GEMM - the same numbers as the three-GEMM path up to the summation order; any other
    layout (separate tensors, blocks out of order, a missing gradient) takes the three-GEMM path."""
    from ttt_amd.infra import fused_linear as FL
    torch.manual_seed(0)
    B, L, D = 2, 37, 16
    x = torch.randn(B, L, D, requires_grad=True)
    ws = [torch.randn(n, D, requires_grad=(i != 1)) for i, n in enumerate((16, 24, 8))]          # (one frozen weight)
    bs = [torch.randn(n, requires_grad=True) for n in (16, 24, 8)]
    buf = torch.randn(B, L, 48)
    blocks = [buf[..., :16], buf[..., 16:40], buf[..., 40:]]
    assert FL._column_blocks(blocks) is not None and FL._column_blocks(blocks).shape == (B * L, 48)
    assert FL._column_blocks([b.contiguous() for b in blocks]) is None
    assert FL._column_blocks([blocks[0], None, blocks[2]]) is None
    assert FL._column_blocks([buf[..., 16:32], buf[..., :16], buf[..., 32:]]) is None
    assert FL._column_blocks([buf[:, 1:, :16], buf[:, 1:, 16:40], buf[:, 1:, 40:]]) is None       # (rows of a larger buffer: not one matrix)
    ys = FL.Linear3.apply(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])
    ins = [x, ws[0], ws[2]] + bs
    three = torch.autograd.grad(ys, ins, [b.contiguous() for b in blocks], retain_graph=True)
    for mode in ("dgrad", "both"):
        FL.FUSE_QKV_BACKWARD = mode
        try:
            fused = torch.autograd.grad(ys, ins, blocks, retain_graph=True)
        finally:
            FL.FUSE_QKV_BACKWARD = "dgrad"
        for a, b in zip(fused, three):
            assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-5, atol=1e-5), mode
    ref = torch.autograd.grad([torch.nn.functional.linear(x, w, b) for w, b in zip(ws, bs)], ins, blocks)
    for a, b in zip(fused, ref):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)


def test_fuse_mode_values_and_the_stacked_weight_cache():
    """TTT_FUSE_QKV_BACKWARD accepts only its five spellings (a typo used to select the fusion silently); the
    [n0 + n1 + n2, K] weight stack of the one-GEMM input gradient is formed once per (weights, version) - the two scan directions of a TTT
    layer share it - and again after an in-place update of a weight."""
    import pytest
    from ttt_amd.infra import fused_linear as FL
    assert [FL._fuse_mode(v) for v in ("", "0", "1", "dgrad", "both")] == ["", "", "dgrad", "dgrad", "both"]
    with pytest.raises(ValueError):
        FL._fuse_mode("off")
    ws = [torch.randn(4, 8), torch.randn(6, 8), torch.randn(2, 8)]
    a = FL._stacked_weights(ws)
    assert a.shape == (12, 8) and torch.equal(a, torch.cat(ws, 0)) and FL._stacked_weights(ws) is a
    ws[1].mul_(2.0)                                    # (what publishing new parameters does: the version moves)
    b = FL._stacked_weights(ws)
    assert b is not a and torch.equal(b, torch.cat(ws, 0))
    # other tensors at the same address with the same version (a new model after the old one was freed) are NOT the cached ones
    ws2 = [w.clone() for w in ws]
    c = FL._stacked_weights(ws2)
    assert c is not b
    FL._stack_cache.update(ws=None, key=None, value=None)
