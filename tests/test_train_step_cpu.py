"""checked_optimizer_step (ttt_amd/infra/train_step.py): a backward whose cluster hand-over timed out must never reach
optimizer.step() - the round-3 advisor finding (poisoned gradients were only caught by the NEXT extension call)."""
import torch

from ttt_amd.infra.train_step import checked_optimizer_step


class FakeExt:
    def __init__(self, err=0):
        self.err, self.cleared = err, 0

    def sweep_error(self):
        return self.err

    def sweep_error_clear(self):
        self.err, self.cleared = 0, self.cleared + 1


def _model():
    torch.manual_seed(0)
    m = torch.nn.Linear(4, 3)
    opt = torch.optim.AdamW(m.parameters(), lr=0.1)
    return m, opt


def test_clean_step_updates_and_returns_the_norm():
    m, opt = _model()
    w0 = m.weight.detach().clone()
    m(torch.ones(2, 4)).sum().backward()
    n = checked_optimizer_step(opt, m.parameters(), 1.0, extension=FakeExt(0))
    assert n is not None and float(n) > 0 and not torch.equal(m.weight, w0)


def test_timed_out_handover_skips_the_step_and_acknowledges():
    m, opt = _model()
    w0 = m.weight.detach().clone()
    m(torch.ones(2, 4)).sum().backward()
    m.weight.grad.fill_(float("nan"))                     # what the poisoned sweep leaves behind
    e = FakeExt(1 + 17)
    assert checked_optimizer_step(opt, m.parameters(), 1.0, extension=e) is None
    assert torch.equal(m.weight, w0) and m.weight.grad is None and e.cleared == 1 and len(opt.state) == 0
    # the batch can be run again
    m(torch.ones(2, 4)).sum().backward()
    assert checked_optimizer_step(opt, m.parameters(), 1.0, extension=e) is not None


def test_non_finite_norm_without_an_error_word_also_skips():
    m, opt = _model()
    w0 = m.weight.detach().clone()
    m(torch.ones(2, 4)).sum().backward()
    m.bias.grad[0] = float("inf")
    e = FakeExt(0)
    assert checked_optimizer_step(opt, m.parameters(), 1.0, extension=e) is None
    assert torch.equal(m.weight, w0) and e.cleared == 0


def test_skipped_step_with_the_flat_sharded_path_leaves_it_usable():
    """FlatFSDP (one process, no group) under checked_optimizer_step with its own clip function: a poisoned backward is dropped -
    masters, compute parameters and AdamW state untouched -, and the same batch run again steps normally (the persistent gradient
    storage of the wrapper is re-used, not left half-accumulated)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import cpu_ext
    from test_flat_fsdp_gloo import _build, _loss
    from ttt_amd.infra.flat_fsdp import FlatFSDP
    cpu_ext.install()
    try:
        m = _build("qkvo")
        m.remat_free_layers = 0
        fs = FlatFSDP(m)
        opt = torch.optim.AdamW(fs.master_parameters(), lr=1e-3)
        before = [u.master.detach().clone() for u in fs.units]
        bf16_before = [u.gathered.clone() for u in fs.units]

        def backward():
            fs.zero_grad()
            _loss(m, 0).backward()
            fs.finish_backward()

        backward()
        fs.units[0].grad_shard[3] = float("nan")
        e = FakeExt(5)
        assert checked_optimizer_step(opt, fs.master_parameters(), 1.0, extension=e, clip_fn=fs.clip_grad_norm_, on_skip=fs.zero_grad) is None
        assert e.cleared == 1 and len(opt.state) == 0
        assert all(m.grad is None for m in fs.master_parameters()) and not any(u.has_grad for u in fs.units)
        assert all(torch.equal(a, u.master) for a, u in zip(before, fs.units))
        assert all(torch.equal(a, u.gathered) for a, u in zip(bf16_before, fs.units))
        backward()                                            # the batch again
        n = checked_optimizer_step(opt, fs.master_parameters(), 1.0, extension=e, clip_fn=fs.clip_grad_norm_)
        assert n is not None and bool(torch.isfinite(n)) and float(n) > 0
        fs.publish()
        assert any(not torch.equal(a, u.master) for a, u in zip(before, fs.units))
        assert any(not torch.equal(a, u.gathered) for a, u in zip(bf16_before, fs.units))
    finally:
        cpu_ext.uninstall()


def test_the_optimizer_hooks_of_the_flat_sharded_path_gate_and_publish():
    """``FlatFSDP.attach_optimizer``: under the reference's unchanged loop (optimizer.zero_grad, backward, clip, optimizer.step) the
    step PRE-hook finishes the backward and looks at the hand-over error word - a poisoned backward is dropped, AdamW changes
    nothing, ``last_step_skipped`` says so, the word is acknowledged - and the POST-hook publishes the new masters into the bf16
    compute parameters; ``optimizer.zero_grad()`` also resets the holder."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import cpu_ext
    from test_flat_fsdp_gloo import _build, _loss
    from ttt_amd.infra.flat_fsdp import FlatFSDP
    from ttt_amd.infra.optimizers import ScheduleType, create_specialized_optimizer
    cpu_ext.install()
    try:
        m = _build("qkvo")
        m.remat_free_layers = 0
        fs = FlatFSDP(m)
        opt, _ = create_specialized_optimizer(m, 1e-3, 1e-3, 1e-4, 1, 4, ScheduleType.LINEAR, ScheduleType.COSINE, "qkvo")
        e = FakeExt(0)
        fs.attach_optimizer(opt, extension=e)
        before = [u.master.detach().clone() for u in fs.units]
        bf16_before = [u.gathered.clone() for u in fs.units]
        opt.zero_grad()
        _loss(m, 0).backward()
        fs.clip_grad_norm_(0.1)
        fs.units[0].grad_shard[5] = float("inf")             # a poisoned backward
        opt.step()
        assert fs.last_step_skipped and len(opt.state) == 0
        assert all(torch.equal(a, u.master) for a, u in zip(before, fs.units))
        assert all(torch.equal(a, u.gathered) for a, u in zip(bf16_before, fs.units))
        opt.zero_grad()                                       # the batch again
        assert not any(u.has_grad for u in fs.units)
        _loss(m, 0).backward()
        n = fs.clip_grad_norm_(0.1)
        opt.step()
        assert not fs.last_step_skipped and bool(torch.isfinite(n))
        assert any(not torch.equal(a, u.master) for a, u in zip(before, fs.units))
        assert any(not torch.equal(a, u.gathered) for a, u in zip(bf16_before, fs.units))     # published by the post-hook
    finally:
        cpu_ext.uninstall()


def test_skipped_step_does_not_advance_an_attached_scheduler_and_remove_restores_the_instances():
    """the reference's loop calls ``lr_scheduler.step()`` unconditionally - a scheduler handed to ``attach_optimizer``
    stands still on a skipped step; ``checked_optimizer_step`` takes the gate decision itself and the pre-hook does not take it a second
    time; ``remove()`` gives ``optimizer.zero_grad`` / ``lr_scheduler.step`` back."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import cpu_ext
    from test_flat_fsdp_gloo import _build, _loss
    from ttt_amd.infra.flat_fsdp import FlatFSDP
    from ttt_amd.infra.optimizers import ScheduleType, create_grouped_lr_scheduler, create_specialized_optimizer
    cpu_ext.install()
    try:
        m = _build("qkvo")
        m.remat_free_layers = 0
        fs = FlatFSDP(m)
        opt, cfgs = create_specialized_optimizer(m, 1e-3, 1e-3, 1e-4, 2, 8, ScheduleType.LINEAR, ScheduleType.COSINE, "qkvo")
        sched = create_grouped_lr_scheduler(opt, cfgs)
        zg0, st0 = opt.zero_grad, sched.step

        class CountingExt(FakeExt):
            calls = 0

            def sweep_error(self):
                CountingExt.calls += 1
                return self.err
        e = CountingExt(0)
        fs.attach_optimizer(opt, extension=e, lr_scheduler=sched)
        assert opt.zero_grad is not zg0 and sched.step is not st0
        lr0 = [g["lr"] for g in opt.param_groups]
        opt.zero_grad()
        _loss(m, 0).backward()
        fs.clip_grad_norm_(0.1)
        fs.units[0].grad_shard[3] = float("nan")
        opt.step()
        sched.step()
        assert fs.last_step_skipped and [g["lr"] for g in opt.param_groups] == lr0          # the schedule stood still
        opt.zero_grad()
        _loss(m, 0).backward()
        fs.clip_grad_norm_(0.1)
        opt.step()
        sched.step()
        assert not fs.last_step_skipped and [g["lr"] for g in opt.param_groups] != lr0
        # checked_optimizer_step drives the step: ONE look at the error word, not two
        opt.zero_grad()
        _loss(m, 0).backward()
        CountingExt.calls = 0
        train = [p for g in opt.param_groups for p in g["params"]]
        assert checked_optimizer_step(opt, train, 0.1, extension=e, clip_fn=fs.clip_grad_norm_) is not None
        assert CountingExt.calls == 1 and not fs.last_step_skipped
        fs.remove()
        assert opt.zero_grad == zg0 and sched.step == st0
    finally:
        cpu_ext.uninstall()


def test_named_trainable_keeps_trainable_parameters_outside_the_wrapped_module():
    """with a master holder on ``model.dit`` the optimizer groups used to see the holder's masters ONLY; a trainable
    parameter of the model outside ``dit`` must still reach its group (the reference's model.named_parameters() includes it)."""
    from ttt_amd.infra.optimizers import named_trainable

    class Holder:
        def __init__(self, mod):
            self.m = [("layers.0.<other_wd>", torch.nn.Parameter(torch.zeros(3)))]

        def named_master_parameters(self):
            return self.m

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.dit = torch.nn.Linear(2, 2)
            self.extra = torch.nn.Parameter(torch.zeros(4))
            self.frozen = torch.nn.Parameter(torch.zeros(4), requires_grad=False)
    mod = Model()
    mod.dit._master_holder = Holder(mod.dit)
    names = [n for n, _ in named_trainable(mod)]
    assert names == ["layers.0.<other_wd>", "extra"]
