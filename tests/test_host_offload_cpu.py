"""ttt_amd/infra/host_offload.py on CPU tensors: the bookkeeping of the saved-activation offload - which saved tensors travel (size
threshold, parameters stay, views of one storage travel once, a slice of a big buffer stays, the layer's byte budget), the slot pool that
is built in the first step and re-used by the next, the prefetch / release walk of the backward - with outputs and gradients equal, bit for
bit, to the run without it.  The device side (streams, events, pinned memory) is exercised by tests/test_host_offload_gpu.py."""
import torch

from ttt_amd.infra.host_offload import HostOffload


class Triple(torch.autograd.Function):
    """saves three column blocks of ONE buffer (like the q / k / v projections) and its input"""
    @staticmethod
    def forward(ctx, x, w):
        y = x @ w                                  # [n, 3d]
        d = y.shape[1] // 3
        a, b, c = y[:, :d], y[:, d:2 * d], y[:, 2 * d:]
        ctx.save_for_backward(x, w, a, b, c)
        return a * b + c

    @staticmethod
    def backward(ctx, g):
        x, w, a, b, c = ctx.saved_tensors
        gy = torch.cat([g * b, g * a, g], dim=1)
        return gy @ w.t(), x.t() @ gy


def layer(x, p):
    h = Triple.apply(x, p["w3"])
    h = torch.tanh(h) + x                          # tanh saves its output: a storage of its own
    big = torch.cat([h, h * 2, h * 3, h * 4, h * 5, h * 6], dim=0)
    s = big[: h.shape[0]]                          # a sixth of a big buffer, saved by the product below: stays on the device
    return h + (s * p["v"]) * 0.5


def run(off, steps=1, n_layers=4, n=64, d=32):
    g = torch.Generator().manual_seed(0)
    ps = [{"w3": torch.randn(d, 3 * d, generator=g).mul_(0.1).requires_grad_(), "v": torch.nn.Parameter(torch.randn(n, d, generator=g))}
          for _ in range(n_layers)]
    x0 = torch.randn(n, d, generator=g, requires_grad=True)
    out = None
    for _ in range(steps):
        for p in ps:
            p["w3"].grad = p["v"].grad = None
        x0.grad = None
        if off is not None:
            off.begin_step()
        x = x0
        for i, p in enumerate(ps):
            if off is not None:
                with off.layer(i):
                    x = layer(x, p)
                x.register_hook(lambda gr, i=i: off.backward_reaches(i))
            else:
                x = layer(x, p)
        if off is not None:
            off.end_forward()
        x.square().sum().backward()
        out = [x.detach().clone(), x0.grad.clone()] + [p[k].grad.clone() for p in ps for k in ("w3", "v")]
    return out


def test_offloaded_run_is_bit_identical_and_only_the_right_tensors_travel():
    ref = run(None)
    unit = 64 * 32 * 4                              # one [n, d] fp32 tensor
    off = HostOffload(bytes_per_layer=1 << 30, min_bytes=unit, pin=False)
    off.chunk_bytes = 1 << 20
    got = run(off)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    st = off.stats
    # per layer: the input x (shared by Triple and the residual add's neighbours), the [n, 3d] buffer ONCE for its three views, tanh's output;
    # the parameter v and the slice of the 6x buffer stay.  (s * v saves s - a slice - and v - a parameter.)
    # The first layer's input is a trainable leaf: it stays as well.
    assert st["offloaded_storages"] == 4 * 3 - 1, st
    assert st["views_shared"] >= 4 * 2, st
    assert st["offloaded_bytes"] == 4 * (unit + 3 * unit + unit) - unit, st
    assert st["fetched_bytes"] == st["offloaded_bytes"] and st["late_fetches"] == 0, st


def test_budget_threshold_layer_limit_and_pool_reuse():
    ref = run(None, steps=2)
    unit = 64 * 32 * 4
    # budget of two units per layer: the first-saved storages go (x, then - the 3-unit buffer does not fit - tanh's output)
    off = HostOffload(bytes_per_layer=2 * unit, layers=2, min_bytes=unit, pin=False)
    off.chunk_bytes = 1 << 16
    got = run(off, steps=2)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    # per step: layer 0 sends tanh's output (its input is the trainable leaf), layer 1 its input and tanh's output; layers 2, 3 nothing
    assert off.stats["offloaded_storages"] == 2 * 3
    assert len(off._slots) == 3                                    # the second step re-used the first step's slots
    pinned = off.stats["pinned_bytes"]
    run_again = run(off, steps=1)
    assert off.stats["pinned_bytes"] == pinned
    for a, r in zip(run_again, run(None, steps=1)):
        assert torch.equal(a, r)
    # threshold above every tensor: nothing travels, nothing breaks
    none = HostOffload(bytes_per_layer=1 << 30, min_bytes=1 << 30, pin=False)
    for a, r in zip(run(none), run(None)):
        assert torch.equal(a, r)
    assert none.stats["offloaded_storages"] == 0


def test_late_fetch_when_no_hook_announces_the_backward():
    """without backward_reaches() the unpack hook fetches on demand (slow on a device, never wrong)"""
    unit = 64 * 32 * 4
    off = HostOffload(bytes_per_layer=1 << 30, min_bytes=unit, pin=False)
    g = torch.Generator().manual_seed(3)
    p = {"w3": torch.randn(32, 96, generator=g).mul_(0.1).requires_grad_(), "v": torch.nn.Parameter(torch.randn(64, 32, generator=g))}
    x0 = torch.randn(64, 32, generator=g, requires_grad=True)
    off.begin_step()
    with off.layer(0):
        y = layer(x0, p)
    off.end_forward()
    y.sum().backward()
    g1 = x0.grad.clone()
    x0.grad = None
    layer(x0, p).sum().backward()
    assert torch.equal(g1, x0.grad)
    assert off.stats["late_fetches"] >= 1


def test_kept_kernel_outputs_of_rematerialised_regions_wait_in_host_memory():
    """remat_cache.context_fn(kinds, park=(offload, layer)): the outputs a checkpointed region keeps are parked and handed back to the
    recomputation from the host copy - one 'launch' per kernel, the bits of the plain run."""
    from torch.utils.checkpoint import checkpoint

    from test_remat_cache_cpu import LAUNCHES, block
    from ttt_amd.infra import remat_cache

    def go(off):
        for k in LAUNCHES:
            LAUNCHES[k] = 0
        g = torch.Generator().manual_seed(0)
        x = torch.randn(64, 8, generator=g, requires_grad=True)
        w1 = torch.randn(8, 8, generator=g, requires_grad=True)
        w2 = torch.randn(8, 8, generator=g, requires_grad=True)
        y = x
        if off is not None:
            off.begin_step()
        for i in range(3):
            park = (off, i) if off is not None else None
            y = checkpoint(block, y, w1, w2, use_reentrant=False, context_fn=remat_cache.context_fn(("attn", "scan"), park))
            if off is not None:
                y.register_hook(lambda gr, i=i: off.backward_reaches(i))
        if off is not None:
            off.end_forward()
        y.sum().backward()
        return y.detach(), x.grad, w1.grad, w2.grad, dict(LAUNCHES)

    ref = go(None)
    off = HostOffload(bytes_per_layer=0, min_bytes=64 * 8 * 4, pin=False, park_kept=True)
    off.chunk_bytes = 1 << 16
    got = go(off)
    assert got[4] == ref[4] == {"attn": 3, "scan": 3}
    for a, r in zip(got[:4], ref[:4]):
        assert torch.equal(a, r)
    assert off.stats["offloaded_storages"] == 6 and off.stats["fetched_bytes"] == off.stats["offloaded_bytes"] == 6 * 64 * 8 * 4
    assert off.stats["late_fetches"] == 0


def test_pinned_pool_cap_keeps_the_rest_on_the_device():
    """``max_pinned_bytes``: the pool never grows beyond the cap; tensors that find no room stay where they are - same bits, fewer travel."""
    unit = 64 * 32 * 4
    off = HostOffload(bytes_per_layer=1 << 30, min_bytes=unit, pin=False, max_pinned_bytes=1 << 16)
    off.chunk_bytes = 1 << 15                      # two chunks of 32 KiB fit under the cap: 8 units, then refusals
    ref, got = run(None), run(off)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    assert off.stats["pinned_bytes"] <= 1 << 16 and off.stats["pool_full_refusals"] > 0
    assert 0 < off.stats["offloaded_storages"] < 11
    again = run(off)                               # the next step re-uses the slots it has, refuses the same tensors
    for a, r in zip(again, ref):
        assert torch.equal(a, r)
    assert off.stats["pinned_bytes"] <= 1 << 16


def test_diffusion_transformer_wiring_on_cpu():
    """``DiffusionTransformer.host_offload`` end to end on CPU tensors (the oracle stands in for the extension, as in tests/test_remat_cache_cpu.py):
    a remat-free layer under the saved-tensor hooks, re-materialised layers whose kept kernel outputs are parked, the per-kind layer limits
    (``remat_keep_limits``), the backward's layer hooks - outputs and every gradient equal to the plain run, bit for bit."""
    from oracle import cpu_ext
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    cpu_ext.install()
    try:
        torch.manual_seed(0)
        cfg = ModelConfig(model_dim=64, num_heads=2, num_layers=4, mini_batch_size=16, latent_height=4, latent_width=8, compressed_num_frames=2,
                          ssm_layer="ttt_linear", text_dim=16, time_embed_dim=32, attn_length=2, prefix_temporal_length=1, scan_checkpoint_group_size=2)
        m = DiffusionTransformer(cfg)
        vid, text, ts = torch.randn(1, 2, 16, 8, 16), torch.randn(1, 1, 16, 16), torch.tensor([100])

        def step():
            m.zero_grad(set_to_none=True)
            out = m(vid, text, ts)
            out.square().mean().backward()
            return out.detach().clone(), [p.grad.clone() for p in m.parameters() if p.grad is not None]

        m.remat_free_layers, m.remat_keep = 1, ("attn", "scan", "fc2")
        ref = step()
        for limits, park, per_layer in (({}, True, 1 << 30), ({"scan": 1, "fc2": 2}, True, 0), ({"scan": 1}, False, 1 << 30)):
            off = HostOffload(bytes_per_layer=per_layer, min_bytes=1 << 10, pin=False, park_kept=park)
            off.chunk_bytes = 1 << 20
            m.host_offload, m.remat_keep_limits = off, limits
            for _ in range(2):                                           # the second step re-uses the first step's slots
                out, grads = step()
                assert torch.equal(out, ref[0]) and all(torch.equal(a, b) for a, b in zip(grads, ref[1]))
            st = off.stats
            assert st["late_fetches"] == 0 and st["fetched_bytes"] == st["offloaded_bytes"], dict(st)
            if per_layer:
                assert st["offloaded_storages"] >= 2 * 20, dict(st)       # the free layer's saved tensors travelled, in both steps
            else:                                                         # (the kernels whose outputs are kept do not run on CPU tensors: the parked
                assert st["views_shared"] == 0                            # path itself is the toy test above; here nothing may come through the hooks)
        m.host_offload, m.remat_keep_limits = None, {}
        out, grads = step()
        assert torch.equal(out, ref[0])
    finally:
        cpu_ext.uninstall()
