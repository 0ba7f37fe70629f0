"""Flat-buffer FSDP for the DiT: the MI355X-first form of ``apply_fsdp`` (reference ``ttt/infra/parallelisms.py``:155-175).

Same partitioning and arithmetic as the reference's FSDP2 wrapping - one unit per ``TransformerLayer`` plus the DiT root, fp32
master parameters and optimizer state sharded over the data-parallel ranks, bf16 compute parameters, gradients reduced in fp32 and
averaged over the ranks - with the data movement laid out for few, large RCCL collectives over xGMI instead of FSDP2's
per-parameter bookkeeping:

  * a unit's TRAINABLE parameters live in ONE flat bf16 buffer (the parameters are views of it).  Per step and unit: one cast of
    the rank's fp32 master shard into its slice of that buffer, ONE in-place ``all_gather_into_tensor`` (no copy-in, no copy-out:
    the gathered buffer IS the parameters), after the unit's backward one multi-tensor cast of its bf16 gradients into a flat fp32
    buffer and ONE ``reduce_scatter_tensor`` into the rank's fp32 gradient shard.  The gathered parameters stay resident between
    forward and backward (288 GB: ``reshard_after_forward=False`` of ``apply_fsdp``), so there is one all-gather per unit and step;
  * FROZEN parameters (adapter "qkvo" freezes 4.0 of the 7.2 B parameters) are replicated in bf16 and never communicated (FSDP2
    shards, casts and re-gathers them every step: 8 GB of bf16 weights are no constraint on 288 GB);
  * collectives run on a side stream: the all-gathers of all units are queued after the optimizer step and a unit's forward waits
    for its own; a unit's reduce-scatter is queued when the last of its gradients has been accumulated and runs beside the
    backward of the units in front of it (the TTT backward never blocks its stream);
  * nothing is allocated inside the backward: the reduce-scatter inputs are two persistent full-size fp32 buffers used in turn,
    the fp32 gradient shards are kept across steps.  (Measured: a 0.3-GB allocation per unit in the middle of the backward moves
    the TTT backward's buffers, and its cluster sweep then runs 16 % slower - ``profiles/r4q_*``, ``r4r_*``.)  Without
    collectives (one rank, no group) there is nothing to overlap and every unit is cast in ``finish_backward``.

Why (round-4 measurement, ``profiles/r4i_*``): over a one-rank mesh FSDP2 costs 7 % of the 9 s step against the same arithmetic
without it - 195 ms of per-parameter ``copy_`` kernels (8 914 launches) and a cluster sweep that runs 0.95 instead of 0.82 ms beside
them.  With one rank this class does what ``ReplicaMixedPrecision`` does (the collectives are skipped), so the N = 1 line and
the N > 1 path are the same code.

    fs = FlatFSDP(model.dit)                                  # after init; parameters become bf16 views
    opt, schedules = create_specialized_optimizer(model, ...)  # ttt_amd/infra/optimizers.py: the reference's four AdamW groups by NAME
    fs.attach_optimizer(opt)                                   # step pre-hook: finish_backward + the sweep-error gate; post-hook: publish
    loss.backward(); norm = fs.clip_grad_norm_(1.0); opt.step(); lr_scheduler.step()          # the reference's loop, train.py:131-166

Optimizer groups.  The reference builds four AdamW groups by parameter name (``ttt/infra/optimizers.py``:31-89: "ttt" / "ssm" names at
``ssm_lr``, the others at ``base_lr``; "bias" / "norm" / "b1" / "b2" names without weight decay).  A unit mixes all four classes, so
its flat buffer is laid out GROUP-MAJOR (the parameters of one class are contiguous) and the rank's fp32 master shard is handed to
the optimizer as one ``Parameter`` per (unit, class) - a VIEW of the shard, with a gradient that is a view of the reduce-scattered
gradient shard: ``named_master_parameters()`` yields them under names that carry the class (``"layers.3.<ttt_no_wd>"`` contains
"ttt" and "bias"), so the reference's name rules classify them unchanged.  A rank owns the elements of a class that fall into
its shard - possibly none.

Export / resume: ``full_parameters()`` / ``optimizer_state_full(opt)`` and their ``load_*`` inverses work by the reference's
parameter names and are independent of the world size (the layout of an unsharded run).

Covered on the CPU by ``tests/test_flat_fsdp_gloo.py`` (world 2 and 3, gloo) against a hand-written data-parallel reference, and
an interrupted-and-resumed run against an uninterrupted one (bit-identical).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch
import torch.distributed as dist

_ALIGN = 64          # elements: every parameter starts on a 128-byte boundary of the flat bf16 buffer


class _Unit:
    __slots__ = ("module", "params", "names", "offsets", "numel", "padded", "shard", "gathered", "shard_view", "master", "pending",
                 "ready", "grad_shard", "held", "prefix", "regions", "masters", "has_grad")


class FlatFSDP:
    def __init__(self, dit: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None, param_dtype=torch.bfloat16,
                 reduce_dtype=torch.float32, gradient_divide_factor: Optional[float] = None, always_communicate: bool = False,
                 group_of=None):
        """``group_of(name) -> class key``: which optimizer class a parameter belongs to (default: the reference's four, by name -
        ``ttt_amd.infra.optimizers.ParameterGroupManager.group_of``); ``group_of=lambda n: ""`` gives one master per unit."""
        self.dit, self.param_dtype, self.reduce_dtype = dit, param_dtype, reduce_dtype
        if group_of is None:
            from ttt_amd.infra.optimizers import ParameterGroupManager
            group_of = ParameterGroupManager.group_of
        self._group_of = group_of
        self._optimizers = []
        self.group = process_group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if on else 1
        self.rank = dist.get_rank(process_group) if on else 0
        self.divide = float(gradient_divide_factor if gradient_divide_factor is not None else self.world)
        # with one rank the collectives are skipped - unless asked for (bench.py's `fsdp1` point: the code path of N > 1 on one GPU)
        self._communicate = self.world > 1 or (always_communicate and on)
        self.units: List[_Unit] = []
        self._cuda = next(dit.parameters()).is_cuda
        self._comm = torch.cuda.Stream() if (self._cuda and self._communicate) else None      # no collectives: no side stream
        # Without collectives there is nothing to overlap: every unit is reduced (= cast to fp32) in finish_backward, as the replica
        # path does.  With them, a unit is reduced by the hook of its last gradient through PERSISTENT buffers (two rotating
        # full-size fp32 buffers, one fp32 shard per unit): no allocation happens inside the backward.  Measured on one rank
        # (profiles/r4q_*, r4r_*): a 0.3-GB allocation per unit in the middle of the backward moves the TTT backward's own
        # buffers and its cluster sweep then runs 0.95 instead of 0.82 ms per launch (+120 ms per step; every other kernel
        # unchanged, nothing concurrent) - the 2 % that separated this class from the replica path, and FSDP2's 7 % too.
        self._defer = not self._communicate or os.environ.get("FLAT_FSDP_DEFER_REDUCE", "0") == "1"
        self._full: List[torch.Tensor] = []                   # rotating reduce-scatter inputs (communicate mode)
        self._full_free: List[Optional[torch.cuda.Event]] = []
        self._turn = 0
        self._hooks = []
        names = {id(p): n for n, p in dit.named_parameters()}
        seen = set()
        for mod in list(getattr(dit, "layers", [])) + [dit]:
            train = []
            for p in mod.parameters():
                if id(p) in seen or not p.is_floating_point():
                    continue
                seen.add(id(p))
                if p.requires_grad:
                    train.append(p)
                else:
                    p.data = p.data.to(param_dtype)            # frozen: replicated compute copy, nothing to communicate
            if train:
                self.units.append(self._build_unit(mod, train, names))
        self._root_hook = dit.register_forward_pre_hook(self._cast_inputs, with_kwargs=True)
        dit._master_holder = self                             # (ttt_amd.infra.optimizers.named_trainable finds the masters here)
        self.last_step_skipped = False
        self._published = True                                # the gathered buffers were filled from the full initial values

    # ------------------------------------------------------------------------------------------------------------ construction
    def _build_unit(self, mod, train, names) -> _Unit:
        u = _Unit()
        # group-major layout: the parameters of one optimizer class are contiguous (stable within a class)
        keys = [self._group_of(names[id(p)]) for p in train]
        order = sorted(range(len(train)), key=lambda i: (self._class_rank(keys[i]), i))
        train, keys = [train[i] for i in order], [keys[i] for i in order]
        u.module, u.params, u.names = mod, train, [names[id(p)] for p in train]
        u.prefix = next((f"layers.{i}" for i, m in enumerate(getattr(self.dit, "layers", [])) if m is mod), "root")
        u.offsets, off, regions = [], 0, {}
        for p, k in zip(train, keys):
            u.offsets.append(off)
            lo_hi = regions.setdefault(k, [off, off])
            off += -(-p.numel() // _ALIGN) * _ALIGN
            lo_hi[1] = off
        u.numel = off
        q = self.world * _ALIGN
        u.padded = -(-off // q) * q
        u.shard = u.padded // self.world
        dev = train[0].device
        full32 = torch.zeros(u.padded, dtype=torch.float32, device=dev)
        for p, o in zip(train, u.offsets):
            full32[o:o + p.numel()].copy_(p.detach().reshape(-1))
        lo = self.rank * u.shard
        u.master = full32[lo:lo + u.shard].clone()           # the rank's fp32 shard (a plain tensor; the optimizer steps views of it)
        u.regions, u.masters = {}, {}
        for k, (a, b) in regions.items():                     # this rank's part of every class: [a, b) in shard coordinates
            a, b = max(a, lo) - lo, min(b, lo + u.shard) - lo
            if b > a:
                u.regions[k] = (a, b)
                m = torch.nn.Parameter(u.master[a:b], requires_grad=True)
                assert m.data_ptr() == u.master[a:b].data_ptr()
                u.masters[k] = m
        u.gathered = full32.to(self.param_dtype)
        u.shard_view = u.gathered[lo:lo + u.shard]
        for p, o in zip(train, u.offsets):
            p.data = u.gathered[o:o + p.numel()].view(p.shape)
        u.pending, u.ready, u.grad_shard, u.held, u.has_grad = len(train), None, None, [], False
        for p in train:
            self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, _u=u: self._on_grad(_u)))
        if self._cuda:
            self._hooks.append(mod.register_forward_pre_hook(lambda _m, _a, _u=u: self._wait_ready(_u)))
        return u

    @staticmethod
    def _class_rank(key) -> tuple:
        from ttt_amd.infra.optimizers import GROUP_NAMES
        return (GROUP_NAMES.index(key), "") if key in GROUP_NAMES else (len(GROUP_NAMES), str(key))

    def _cast_inputs(self, module, args, kwargs):
        cast = lambda t: t.to(self.param_dtype) if isinstance(t, torch.Tensor) and t.is_floating_point() else t
        return tuple(cast(a) for a in args), {k: cast(v) for k, v in kwargs.items()}

    def master_parameters(self) -> List[torch.nn.Parameter]:
        """What the optimizer steps: this rank's fp32 master slices, one per (unit, optimizer class)."""
        return [m for _, m in self.named_master_parameters()]

    def named_master_parameters(self):
        """(name, master slice) pairs; the name is ``"<unit>.<class>"`` - e.g. ``"layers.3.<ttt_no_wd_bias>"`` - spelled so that the
        reference's NAME rules (optimizers.py:31-46: "ttt" / "ssm", "bias" / "norm" / "b1" / "b2") put it into the class it holds."""
        tag = {"ttt_no_wd": "<ttt_no_wd_bias>", "ttt_wd": "<ttt_wd>", "other_no_wd": "<other_no_wd_bias>", "other_wd": "<other_wd>"}
        return [(f"{u.prefix}.{tag.get(k, k)}", m) for u in self.units for k, m in u.masters.items()]

    def attach_optimizer(self, optimizer: torch.optim.Optimizer, gate: bool = True, extension=None, lr_scheduler=None):
        """Makes the reference's unchanged loop work on this holder (train.py:131-166: zero_grad, backward, clip, ``optimizer.step()``,
        ``lr_scheduler.step()``).  Step PRE-hook: ``finish_backward()`` (idempotent) and, with ``gate``, the look at the TTT-MLP
        backward's hand-over error word that must come before AdamW (one device synchronisation; all ranks agree by a MAX
        all-reduce): after a timed-out hand-over or with non-finite gradients the gradients are dropped, so that the step changes
        nothing (AdamW skips parameters without a gradient; ``last_step_skipped`` says so).  Step POST-hook: ``publish()``.

        A skipped step must not advance the schedule: the reference's loop calls ``lr_scheduler.step()`` unconditionally, so either
        pass ``lr_scheduler`` here - its ``step()`` then does nothing after a skipped optimizer step - or test ``last_step_skipped``
        before calling it.  ``checked_optimizer_step`` (train_step.py) takes the same decision itself BEFORE ``optimizer.step()``:
        it marks the optimizer (``_ttt_gate_done``) and the pre-hook does not synchronise a second time.  ``remove()`` takes the
        hooks off again and gives the optimizer / scheduler their own ``zero_grad`` / ``step`` back."""
        def pre(opt, args, kwargs):
            self.finish_backward()
            self.last_step_skipped = False
            if gate and not getattr(opt, "_ttt_gate_done", False):
                self.last_step_skipped = self._gate(extension)
            opt._ttt_gate_done = False
            return None

        def post(opt, args, kwargs):
            self.publish()

        # optimizer.zero_grad() also ends this holder's step (held bf16 gradients, pending counters, the "has a gradient" marks)
        stock = optimizer.zero_grad

        def zero_grad(set_to_none: bool = True):
            stock(set_to_none=set_to_none)
            self.zero_grad()

        optimizer.zero_grad = zero_grad
        sched_stock = None
        if lr_scheduler is not None:
            sched_stock = lr_scheduler.step

            def sched_step(*a, **k):
                if self.last_step_skipped:               # warm-up / decay do not advance on a step that changed nothing
                    return None
                return sched_stock(*a, **k)

            lr_scheduler.step = sched_step
        self._optimizers.append((optimizer, optimizer.register_step_pre_hook(pre), optimizer.register_step_post_hook(post), stock, lr_scheduler, sched_stock))
        return optimizer

    def _gate(self, extension) -> bool:
        err = 0
        if self._cuda or extension is not None:
            if extension is None:
                import test_time_training as extension
            err = int(extension.sweep_error())               # synchronises: everything the backward enqueued has run
        grads = [m.grad for u in self.units for m in u.masters.values() if m.grad is not None]
        dev = self.units[0].master.device
        finite = bool(torch.isfinite(torch.stack(torch._foreach_norm(grads)).sum())) if grads else True
        bad = torch.tensor([1 if (err != 0 or not finite) else 0], device=dev, dtype=torch.int32)
        if self.world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        if int(bad):
            self.zero_grad()
            if err:
                extension.sweep_error_clear()
            return True
        return False

    # ---------------------------------------------------------------------------------------------------------------- forward
    def _wait_ready(self, u: _Unit):
        if u.ready is not None:
            torch.cuda.current_stream().wait_event(u.ready)

    def publish(self):
        """After the optimizer step: masters -> bf16 compute parameters on every rank (one cast for all units, one in-place
        all-gather per unit on the side stream; a unit's forward waits for its own)."""
        with torch.no_grad():
            side = self._comm is not None
            if side:
                self._comm.wait_stream(torch.cuda.current_stream())       # the step's reads of the old parameters are queued in front
            with (torch.cuda.stream(self._comm) if side else _null()):
                torch._foreach_copy_([u.shard_view for u in self.units], [u.master for u in self.units])
                for u in self.units:
                    if self._communicate:
                        dist.all_gather_into_tensor(u.gathered, u.shard_view, group=self.group)
                    if side:
                        u.ready = torch.cuda.Event()
                        u.ready.record(self._comm)
        self._published = True

    # --------------------------------------------------------------------------------------------------------------- backward
    def _on_grad(self, u: _Unit):
        u.pending -= 1
        if u.pending == 0 and not self._defer:
            self._reduce(u)

    def _flat_buffer(self, n: int) -> torch.Tensor:
        """One of two persistent full-size fp32 buffers, in turn; the compute stream waits until the side stream has finished the
        reduce-scatter that last read it."""
        if not self._full:
            big = max(u.padded for u in self.units)
            dev = self.units[0].master.device
            self._full = [torch.empty(big, dtype=self.reduce_dtype, device=dev) for _ in range(2)]
            self._full_free = [None, None]
        k = self._turn = (self._turn + 1) & 1
        if self._full_free[k] is not None:
            torch.cuda.current_stream().wait_event(self._full_free[k])
        return self._full[k][:n]

    def _reduce(self, u: _Unit):
        grads = [p.grad for p in u.params]
        if all(g is None for g in grads):
            u.pending = len(u.params)
            return
        with torch.no_grad():
            # the bf16 -> fp32 cast of the unit's gradients runs on the compute stream (0.1 - 0.2 ms per unit): on the side stream it
            # would stream 0.5 GB beside the next layer's TTT backward; only the collective runs beside the backward
            persistent = self._communicate and self._cuda
            flat = self._flat_buffer(u.padded) if persistent else torch.empty(u.padded, dtype=self.reduce_dtype, device=u.master.device)
            if u.padded > u.numel or any(g is None for g in grads) or any(p.numel() % _ALIGN for p in u.params):
                flat.zero_()                                              # padding between / behind the parameters, unused parameters
            dst = [flat[o:o + p.numel()].view(p.shape) for p, o, g in zip(u.params, u.offsets, grads) if g is not None]
            torch._foreach_copy_(dst, [g for g in grads if g is not None])                # bf16 -> fp32, one multi-tensor launch
            if self._communicate:
                side = self._comm is not None
                if side:
                    self._comm.wait_stream(torch.cuda.current_stream())   # the flat gradient is complete when the side stream starts
                with (torch.cuda.stream(self._comm) if side else _null()):
                    first = not self._holds_grad(u)
                    if u.grad_shard is None:
                        u.grad_shard = torch.empty(u.shard, dtype=self.reduce_dtype, device=flat.device)      # once: kept across steps
                    shard = u.grad_shard if first else torch.empty_like(u.grad_shard)                          # (micro-batches: a temporary)
                    dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group)
                    if self.divide != 1.0:
                        shard.div_(self.divide)
                    if persistent:
                        ev = torch.cuda.Event()
                        ev.record(self._comm)
                        self._full_free[self._turn] = ev
                    self._accumulate(u, shard)
            else:
                shard = flat if self.divide == 1.0 else flat.div_(self.divide)
                self._accumulate(u, shard)
        if not self._defer:
            u.held.append(grads)      # freed in finish_backward, not here: the backward's allocation pattern stays that of the replica path
        for p in u.params:
            p.grad = None
        u.pending = len(u.params)

    @staticmethod
    def _holds_grad(u: _Unit) -> bool:
        """a reduced gradient of this step is already there (micro-batches accumulate into it); ``optimizer.zero_grad()`` and
        ``zero_grad()`` both end that"""
        return u.has_grad and all(m.grad is not None for m in u.masters.values())

    @classmethod
    def _accumulate(cls, u: _Unit, shard: torch.Tensor):
        """the unit's reduced gradient shard -> the gradients of its master slices (views of ONE shard-sized tensor)"""
        if not cls._holds_grad(u):
            u.grad_shard = shard
            for k, (a, b) in u.regions.items():
                u.masters[k].grad = shard[a:b]
            u.has_grad = True
        else:
            u.grad_shard.add_(shard)                                      # gradient accumulation over micro-batches

    def finish_backward(self):
        """Once per backward, before clipping / the optimizer: reduces units whose parameters did not all receive a gradient
        (a unit is normally reduced by the hook of its last gradient) and joins the side stream."""
        for u in self.units:
            if u.pending != len(u.params) or any(p.grad is not None for p in u.params):
                self._reduce(u)
        if self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)
        for u in self.units:
            u.held.clear()

    def zero_grad(self):
        for u in self.units:
            for m in u.masters.values():
                m.grad = None                                 # (u.grad_shard, the storage, is kept for the next step)
            u.has_grad = False
            u.held.clear()
            u.pending = len(u.params)
            for p in u.params:
                p.grad = None

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Global 2-norm over all ranks' shards (one scalar all-reduce), gradients scaled in place like
        ``torch.nn.utils.clip_grad_norm_``; returns the norm.  Finishes the backward first (idempotent), so that the reference's
        order - backward, clip, ``optimizer.step()`` - needs no extra call.  (``torch.nn.utils.clip_grad_norm_(model.parameters())``
        itself cannot serve: the module's parameters are bf16 views whose gradients have been reduced into fp32 SHARDS; FSDP2
        answers the same call through DTensor.  This method is the one-line change of an unchanged ``train.py``, INTEGRATION.md.)"""
        self.finish_backward()
        grads = [m.grad for u in self.units for m in u.masters.values() if m.grad is not None]
        if not grads and self.world == 1:
            return torch.zeros((), device=self.units[0].master.device)
        if not grads:                                         # this rank owns no element of any class: it still joins the all-reduce
            grads = [torch.zeros(1, device=self.units[0].master.device)]
        sq = torch.stack([n * n for n in torch._foreach_norm(grads)]).sum()
        if self.world > 1:
            dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=self.group)
        total = sq.sqrt()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        torch._foreach_mul_([u.grad_shard for u in self.units if self._holds_grad(u) and u.masters], coef)      # (one tensor per unit)
        return total

    # ------------------------------------------------------------------------------------------------------------ inspection
    def full_parameters(self, what: str = "param") -> Dict[str, torch.Tensor]:
        """{name: fp32 tensor} of the masters (``what="param"``) or of their gradients (``"grad"``), gathered from all ranks -
        for tests and checkpoints; collective."""
        out = {}
        for u in self.units:
            src = u.master if what == "param" else (u.grad_shard if u.has_grad else None)
            if src is None:
                continue
            full = torch.empty(u.padded, dtype=src.dtype, device=src.device)
            if self.world > 1:
                dist.all_gather_into_tensor(full, src.contiguous(), group=self.group)
            else:
                full.copy_(src)
            for n, p, o in zip(u.names, u.params, u.offsets):
                out[n] = full[o:o + p.numel()].view(p.shape).clone()
        return out

    # ------------------------------------------------------------------------------------------------- checkpoints (export / resume)
    def _flat_from_named(self, u: _Unit, named: Dict[str, torch.Tensor], what: str, strict: bool) -> Optional[torch.Tensor]:
        """this rank's fp32 shard of the unit's flat layout, filled from full per-parameter tensors (None if nothing is there)"""
        full = torch.zeros(u.padded, dtype=torch.float32, device=u.master.device)
        found = 0
        for n, p, o in zip(u.names, u.params, u.offsets):
            t = named.get(n)
            if t is None:
                if strict:
                    raise KeyError(f"FlatFSDP: no {what} for parameter {n!r}")
                continue
            if tuple(t.shape) != tuple(p.shape):
                raise ValueError(f"FlatFSDP: {what} of {n!r} has shape {tuple(t.shape)}, the parameter {tuple(p.shape)}")
            full[o:o + p.numel()].copy_(t.detach().reshape(-1).to(full.device, torch.float32))
            found += 1
        if not found:
            return None
        lo = self.rank * u.shard
        return full[lo:lo + u.shard].clone()

    def load_full_parameters(self, named: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """{name: full tensor} by the reference's parameter names (a checkpoint of the reference, `full_parameters()` of another
        run - any world size) -> this rank's fp32 master shards and, through `publish()`, the bf16 compute parameters of every
        rank; frozen parameters go into their replicated copies.  Every rank passes the same dictionary."""
        with torch.no_grad():
            for u in self.units:
                shard = self._flat_from_named(u, named, "value", strict)
                if shard is not None:
                    u.master.copy_(shard)
            frozen = {n: p for n, p in self.dit.named_parameters() if not p.requires_grad}
            for n, p in frozen.items():
                if n in named:
                    p.data.copy_(named[n].to(p.device, p.dtype))
                elif strict and p.is_floating_point():
                    raise KeyError(f"FlatFSDP: no value for frozen parameter {n!r}")
        self.publish()

    def optimizer_state_full(self, optimizer: torch.optim.Optimizer) -> Dict[str, Dict[str, torch.Tensor]]:
        """The optimizer's per-element state (AdamW: exp_avg, exp_avg_sq) gathered from all ranks and cut by parameter -
        {name: {"exp_avg": full fp32 tensor, "exp_avg_sq": ..., "step": scalar}}: the layout of an unsharded optimizer over the
        reference's parameters, independent of the world size it was trained with.  Collective."""
        out: Dict[str, Dict[str, torch.Tensor]] = {}
        for u in self.units:
            # the rank's shard of every per-element entry, assembled from the state of its master slices (a class this rank
            # owns nothing of, or an optimizer that has not stepped yet, contributes zeros / nothing)
            keys, scalars = [], {}
            for k, m in u.masters.items():
                for key, val in optimizer.state.get(m, {}).items():
                    if torch.is_tensor(val) and val.ndim > 0 and val.numel() == m.numel():
                        if key not in keys:
                            keys.append(key)
                    else:
                        scalars.setdefault(k, {})[key] = val
            if self.world > 1:                                # every rank must walk the same collectives: agree on the entries
                names = [None] * self.world
                dist.all_gather_object(names, keys, group=self.group)
                keys = sorted({k for ks in names for k in ks})
            per = {}
            for key in keys:
                mine = torch.zeros(u.shard, dtype=torch.float32, device=u.master.device)
                for k, m in u.masters.items():
                    val = optimizer.state.get(m, {}).get(key)
                    if val is not None:
                        a, b = u.regions[k]
                        mine[a:b].copy_(val.reshape(-1))
                full = torch.empty(u.padded, dtype=mine.dtype, device=mine.device)
                if self.world > 1:
                    dist.all_gather_into_tensor(full, mine, group=self.group)
                else:
                    full.copy_(mine)
                per[key] = full
            if self.world > 1:                                # scalar entries ("step") of a class live where the class has elements
                allsc = [None] * self.world
                dist.all_gather_object(allsc, {k: {kk: (float(v) if torch.is_tensor(v) else v) for kk, v in d.items()} for k, d in scalars.items()},
                                       group=self.group)
                merged = {}
                for d in allsc:
                    for k, dd in d.items():
                        merged.setdefault(k, dd)
                scalars = {k: {kk: torch.tensor(v, dtype=torch.float32) if isinstance(v, float) else v for kk, v in dd.items()} for k, dd in merged.items()}
            for n, p, o in zip(u.names, u.params, u.offsets):
                ent = {k: f[o:o + p.numel()].view(p.shape).clone() for k, f in per.items()}
                for key, val in scalars.get(self._group_of(n), {}).items():
                    ent[key] = val.clone() if torch.is_tensor(val) else val
                if ent:
                    out[n] = ent
        return out

    def load_optimizer_state_full(self, optimizer: torch.optim.Optimizer, state: Dict[str, Dict[str, torch.Tensor]], strict: bool = True) -> None:
        """Inverse of `optimizer_state_full` (any world size on either side).  Scalar entries ("step") are taken from the unit's
        first parameter that has them."""
        with torch.no_grad():
            for u in self.units:
                # per-element entries = tensors of the parameter's own shape ("step" is 0-dimensional)
                keys = {k for n, p in zip(u.names, u.params) for k, v in state.get(n, {}).items()
                        if torch.is_tensor(v) and v.ndim > 0 and tuple(v.shape) == tuple(p.shape)}
                shards = {}
                for k in sorted(keys):
                    shard = self._flat_from_named(u, {n: state[n][k] for n in u.names if n in state and k in state[n]}, f"optimizer state {k!r}", strict)
                    if shard is not None:
                        shards[k] = shard
                for cls, m in u.masters.items():
                    a, b = u.regions[cls]
                    new = {k: sh[a:b].clone() for k, sh in shards.items()}
                    for n in u.names:                         # scalar entries from the first parameter of this class that has them
                        if self._group_of(n) != cls:
                            continue
                        for k, v in state.get(n, {}).items():
                            if k not in keys and k not in new:
                                new[k] = v.clone().to(m.device) if torch.is_tensor(v) else v
                    if new:
                        optimizer.state[m] = new

    def remove(self):
        """Detach from the module (hooks removed, buffers released): the parameters keep their last bf16 values as views of
        nothing shared any more.  bench.py calls it between its runs so that the first model's memory is returned."""
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        self._root_hook.remove()
        for opt, h0, h1, stock_zero_grad, sched, sched_stock in self._optimizers:
            h0.remove()
            h1.remove()
            opt.zero_grad = stock_zero_grad              # (attach_optimizer replaced them on the instances)
            if sched is not None:
                sched.step = sched_stock
        self._optimizers.clear()
        if getattr(self.dit, "_master_holder", None) is self:
            del self.dit._master_holder
        for u in self.units:
            for m in u.masters.values():
                m.grad = None
            u.grad_shard = None
        self.units.clear()
        self._full, self._full_free = [], []


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
