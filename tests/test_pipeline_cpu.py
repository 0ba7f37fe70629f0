"""Host logic of the pipelined TTT-MLP forward (ttt_amd/models/ssm/pipeline.py): the plan of parts and token runs, the injection of
pre-computed results into the layer's autograd Functions."""
import pytest
import torch


def test_pipeline_plan_covers_the_sequence_once():
    """the parts of the plan: whole checkpoint groups, every token of the sequence in exactly one run of exactly one part"""
    from ttt_amd.models.ssm.pipeline import plan_parts
    L, CS, G = 64 * 23, 64, 4
    perm = torch.randperm(L // 32).repeat_interleave(32) * 32 + torch.arange(32).repeat(L // 32)      # blocks of 32 tokens shuffled
    for src in (None, perm, torch.arange(L - 1, -1, -1)):
        for n in (1, 2, 3, 5):
            parts = plan_parts(src, L, CS, G, n)
            assert sum(ns for _, ns, _ in parts) == L // CS and all(s0 % G == 0 for s0, _, _ in parts)
            seen = torch.zeros(L, dtype=torch.int32)
            for s0, ns, runs in parts:
                toks = torch.arange(s0 * CS, (s0 + ns) * CS) if src is None else src[s0 * CS:(s0 + ns) * CS]
                cover = torch.zeros(L, dtype=torch.int32)
                for r0, r1 in runs:
                    cover[r0:r1] += 1
                assert int(cover.sum()) == ns * CS and bool((cover[toks] == 1).all())
                seen += cover
            assert bool((seen == 1).all())


def test_injected_results_are_taken_once_and_must_all_be_consumed():
    from ttt_amd.infra.fused_linear import Linear3
    from ttt_amd.models.ssm import pipeline
    x = torch.randn(2, 5, 8, requires_grad=True)
    ws = [torch.randn(8, 8, requires_grad=True) for _ in range(3)]
    bs = [torch.randn(8, requires_grad=True) for _ in range(3)]
    ref = Linear3.apply(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])
    pre = tuple(t.detach().clone() for t in ref)
    with pipeline.injecting({"linear3": pre}):
        got = Linear3.apply(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])
        assert pipeline.injected("linear3") is None            # taken
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(got, pre))
    # the backward is the node's own: gradients as without injection
    g = [torch.randn_like(t) for t in ref]
    a = torch.autograd.grad(got, [x] + ws + bs, g)
    b = torch.autograd.grad(ref, [x] + ws + bs, g)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    with pytest.raises(RuntimeError, match="not consumed"):
        with pipeline.injecting({"post": torch.zeros(1)}):
            pass
    y = torch.randn(3, 8, requires_grad=True)
    w, bias = torch.randn(4, 8, requires_grad=True), torch.randn(4, requires_grad=True)
    o_ref = torch.nn.functional.linear(y, w, bias)
    with pipeline.injecting({"wo": o_ref.detach().clone()}):
        o = pipeline.InjectedLinear.apply(y, w, bias)
    go = torch.randn_like(o_ref)
    assert all(torch.allclose(p, q) for p, q in zip(torch.autograd.grad(o, [y, w, bias], go), torch.autograd.grad(o_ref, [y, w, bias], go)))
