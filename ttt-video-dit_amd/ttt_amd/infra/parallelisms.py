"""Process-group set-up and FSDP wrapping for one 8xMI355X node.

``apply_fsdp`` is the behavioural mirror of the reference's ``ttt/infra/parallelisms.py:155-175``:
FSDP2 ``fully_shard`` on every ``TransformerLayer`` and on the DiT root, ``reshard_after_forward``,
bf16 parameters / fp32 gradient reduction.  Backend string ``"nccl"`` is RCCL on ROCm; on one node
the collectives run over xGMI.  Tensor parallelism (reference :106-152) is a next-row item.
"""
from __future__ import annotations

import os
from datetime import timedelta

import torch
import torch.distributed as dist
from torch.distributed.device_mesh import init_device_mesh
from torch.distributed.fsdp import MixedPrecisionPolicy, fully_shard


def init_distributed(backend: str | None = None, timeout_s: int = 600):
    """One process per GPU (torchrun env: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")   # reference parallelisms.py:36
    if not dist.is_initialized():
        dist.init_process_group(backend, timeout=timedelta(seconds=timeout_s))
    return dist.get_rank(), dist.get_world_size()


def end_distributed():
    if dist.is_initialized():
        if torch.cuda.is_available():
            dist.barrier(device_ids=[torch.cuda.current_device()])
            torch.cuda.synchronize()
        else:
            dist.barrier()
        dist.destroy_process_group()


def get_dp_mesh(dp_shard: int | None = None, dp_replicate: int = 1):
    world = dist.get_world_size()
    dp_shard = dp_shard or world // dp_replicate
    assert dp_shard * dp_replicate == world, "world size must equal dp_replicate * dp_shard"
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    if dp_replicate > 1:   # HSDP (reference :160-162)
        return init_device_mesh(dev, (dp_replicate, dp_shard), mesh_dim_names=("dp_replicate", "dp_shard"))
    return init_device_mesh(dev, (dp_shard,), mesh_dim_names=("dp_shard",))


def apply_fsdp(model, dp_mesh, param_dtype=torch.bfloat16, reduce_dtype=torch.float32, reshard_after_forward: bool = True):
    """Shard every transformer layer, then the DiT root; nothing outside the DiT (reference :164-175).

    ``reshard_after_forward=True`` is the reference's setting (80-GB GPUs: the gathered bf16 parameters of a layer are
    freed after its forward and all-gathered again in backward).  On 288-GB MI355X the 14.5 GB of gathered bf16
    parameters can simply stay resident between forward and backward (``False``): one all-gather per layer and step
    instead of two (three with re-materialisation) over xGMI, reduce-scatter unchanged."""
    mp = MixedPrecisionPolicy(param_dtype=param_dtype, reduce_dtype=reduce_dtype)
    dit = model.dit if hasattr(model, "dit") else model
    for layer in dit.layers:
        fully_shard(layer, mesh=dp_mesh, mp_policy=mp, reshard_after_forward=reshard_after_forward)
    fully_shard(dit, mesh=dp_mesh, mp_policy=mp, reshard_after_forward=reshard_after_forward)
    return model


def apply_tp(model, tp_mesh, layout: str = "full"):
    """Tensor parallelism over ``tp_mesh`` (a 1-D ``DeviceMesh`` or a process group): ONE sample is worked on by the ranks of the
    group, as in the reference's ``apply_tp`` (``ttt/infra/parallelisms.py``:106-152; the TP ranks of a reference job see the same
    batch).  ``layout``:

      * ``"full"`` (default) - the reference's whole plan in its MI355X form: sequence-mixing work on a rank's HEAD shard over
        the full sequence (local-attention ``q / k / v`` projections + ``q_norm / k_norm`` + RoPE + attention: reference
        ``ColwiseParallel`` q/k/v, ``:109-113``; the TTT layer's ``wq / wk / wv``, per-head parameters, scan and backward:
        ``:118-120``, ``ttt_layer.py``:114-131, ``mlp_tk.py``:297-343), token-wise work on a rank's TOKEN shard (AdaLN,
        ``o`` / ``wo`` output projections, ``post_norm``, gates, the MLP: reference ``SequenceParallel`` norms ``:116-121`` and
        the sequence-sharded MLP, ``dit.py``:56-72,370-374; the final norm / AdaLN / projection, ``:125-127``).  Implemented by
        ``ttt_amd.infra.sequence_parallel`` with explicit differentiable RCCL collectives instead of DTensor redistribution
        (all-gather of the token shards in front of a sequence-mixing op, all-to-all heads -> tokens behind it);
      * ``"ttt_heads"`` - only the TTT layer is head-sharded (round 2's form; the rest of the block runs replicated).

    What is NOT mirrored, on purpose: parameters stay whole on every rank (the reference shards their storage through DTensor
    placements to fit 80-GB GPUs; 14.5 GB of bf16 weights are no constraint on 288 GB), so every rank holds PARTIAL parameter
    gradients - its tokens' share of the token-wise parameters, its heads' slices of the per-head ones - and
    ``tp_sync_gradients(model)`` must be called once per optimizer step after backward - unless FSDP2 shards the parameters over
    ALL ranks, TP groups included (``apply_parallelisms``: its reduce-scatter then sums the partial gradients).  ``reference
    shard_transformer_inputs`` (layer-group inputs kept sharded between checkpoints, ``dit.py``:494-498) is what the "full"
    layout does by construction: between sequence-mixing ops the activations only exist as token shards."""
    import torch.distributed as dist
    from ttt_amd.infra.sequence_parallel import SeqParallel

    dit = model.dit if hasattr(model, "dit") else model
    if layout == "ttt_heads":
        for layer in dit.layers:
            layer.seq_modeling_block.ssm.ttt.init_device_mesh(tp_mesh)
    elif layout == "full":
        group = tp_mesh if (tp_mesh is None or isinstance(tp_mesh, dist.ProcessGroup)) else tp_mesh.get_group()
        dit.sequence_parallel = SeqParallel(group)
    else:
        raise ValueError(f"apply_tp: unknown layout {layout!r}")
    dit._tp_layout = layout
    return tp_mesh


def get_world_mesh(tp_sharding: int = 1, dp_sharding: int | None = None, dp_replicate: int = 1):
    """The job's device mesh in the reference's order ``(dp_replicate, dp_shard, tp)`` with the TP ranks innermost = neighbours
    on the node (reference ``get_world_mesh``, ``parallelisms.py``:54-89; a dimension of size 1 is kept here, so that
    ``mesh["tp"]`` / ``mesh["dp_shard"]`` always exist)."""
    world = dist.get_world_size()
    dp_sharding = dp_sharding or world // (tp_sharding * dp_replicate)
    assert tp_sharding * dp_sharding * dp_replicate == world, \
        "world size must be equal to the product of tp_sharding, dp_sharding, and dp_replicate"
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    return init_device_mesh(dev, (dp_replicate, dp_sharding, tp_sharding), mesh_dim_names=("dp_replicate", "dp_shard", "tp"))


def apply_parallelisms(model, tp_sharding: int = 1, param_dtype=torch.bfloat16, reduce_dtype=torch.float32,
                       reshard_after_forward: bool = True, tp_layout_on_one_rank: bool = False):
    """TP x FSDP in one call, the counterpart of the reference's ``apply_parallelisms`` (``parallelisms.py``:92-104: ``apply_tp``
    on ``world_mesh["tp"]``, then ``apply_fsdp`` on the data-parallel dimensions).  Returns ``(world_mesh, dp_rank, dp_size)``;
    the ranks of one TP group must be fed the SAME sample (``dp_rank`` is what seeds / shards the data).

    The MI355X form of the 2-D plan: the "full" TP layout splits ACTIVATIONS (head shards / token shards) and leaves every rank
    with partial gradients of whole parameters; FSDP2 then shards parameters, gradients and optimizer state over ALL
    ``dp x tp`` ranks (the reference shards them over ``dp_shard`` only and adds DTensor TP placements on top, ``:106-175``).
    One reduce-scatter per layer does both jobs at once - it sums the partial gradients of a TP group and averages over the
    data-parallel groups: FSDP2's mean over ``W = dp x tp`` ranks is ``1 / tp`` of that, so the divide factor is set to ``dp``
    (``set_gradient_divide_factor``).  ``tp_sync_gradients`` becomes a no-op.  Per rank: ``1 / W`` of the fp32 masters and AdamW
    state instead of ``1 / dp``, no second gradient collective, no DTensor redistribution inside the layer."""
    from torch.distributed.fsdp import FSDPModule

    mesh = get_world_mesh(tp_sharding)
    world = dist.get_world_size()
    dp = world // tp_sharding
    dit = model.dit if hasattr(model, "dit") else model
    use_tp = tp_sharding > 1 or tp_layout_on_one_rank      # (one-rank groups: the layout's code path, for measurements)
    if use_tp:
        apply_tp(model, mesh["tp"], layout="full")
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    apply_fsdp(model, init_device_mesh(dev, (world,), mesh_dim_names=("dp_shard_tp",)), param_dtype=param_dtype,
               reduce_dtype=reduce_dtype, reshard_after_forward=reshard_after_forward)
    if use_tp:
        for mod in dit.modules():
            if isinstance(mod, FSDPModule):
                mod.set_gradient_divide_factor(float(dp))
                # plain SUM reduce-scatter + a division of the rank's shard (not the PreMulSum reduce op, which gloo does not
                # have and RCCL has never been asked for here)
                mod.set_force_sum_reduction_for_comms(True)
        dit._tp_grads_by_fsdp = True
    return mesh, dist.get_rank() // tp_sharding, dp


def tp_sync_gradients(model):
    """After backward, once per optimizer step: sum the ranks' partial parameter gradients over the TP group (nothing to do
    under ``apply_parallelisms``: FSDP2's reduce-scatter over all ``dp x tp`` ranks has summed them)."""
    dit = model.dit if hasattr(model, "dit") else model
    if getattr(dit, "_tp_grads_by_fsdp", False):
        return
    layout = getattr(dit, "_tp_layout", None)
    if layout == "full":
        for p in dit.parameters():
            if p.grad is not None and type(p.grad) is not torch.Tensor:
                raise RuntimeError("tp_sync_gradients: FSDP-sharded (DTensor) gradients - this tensor parallelism keeps whole "
                                   "parameters per rank and does not compose with FSDP2")
        dit.sequence_parallel.sum_gradients(dit)
    elif layout == "ttt_heads":
        for layer in dit.layers:
            layer.seq_modeling_block.ssm.ttt.tp_sync_gradients()


def enable_tuned_gemms(path: str | None = None) -> bool:
    """Plain library GEMMs (projections, MLP) go to hipBLASLt / rocBLAS through PyTorch; this loads a committed
    solution-selection file for the 5B GEMM shapes of the 3 s and 9 s configurations on gfx950 (produced with PyTorch
    TunableOp on an MI355X, ``tools/_run_r2i.sh``: the default heuristic picks 0.39 ms kernels for the 18048x3072x3072 projections where
    0.21 ms ones exist).  No tuning happens at run time; shapes that are not in the file, or a file written for another
    library version (its validator lines are checked by PyTorch), fall back to the default heuristic.  Returns whether
    the selections are active."""
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning_gfx950.csv")
    try:
        from torch.cuda import tunable
        if not (torch.cuda.is_available() and os.path.exists(path)):
            return False
        import tempfile
        tunable.enable(True)
        tunable.tuning_enable(False)
        tunable.set_filename(os.path.join(tempfile.gettempdir(), f"ttt_tunableop_{os.getpid()}.csv"))   # never write into the repo
        return bool(tunable.read_file(path))
    except Exception:
        return False


def init_model_parameters(model, initializer_range: float = 0.02):
    """N(0, 0.02) weights / zero biases everywhere except TTT modules, which initialise themselves
    (reference :178-196)."""
    from ttt_amd.models.ssm.ttt_layer import TTTBase

    for module in model.modules():
        if isinstance(module, TTTBase):
            module.init_weights()
            continue
        for name, p in module.named_parameters(recurse=False):
            if p is None:
                continue
            if "bias" in name:
                torch.nn.init.zeros_(p)
            else:
                torch.nn.init.normal_(p, mean=0.0, std=initializer_range)


class ReplicaMixedPrecision:
    """World-size-1 counterpart of ``apply_fsdp``'s mixed precision: fp32 master parameters, ``param_dtype`` compute
    copies inside the module, gradients converted to ``reduce_dtype`` for the optimizer - without FSDP2's machinery.

    Why: with one rank there is nothing to shard or gather, yet FSDP2 still walks every parameter of every layer each
    step (cast into an all-gather input, copy into the all-gather output, chunk-cat + cast of each gradient): about 3000
    small copy kernels and 190 ms of a 3.26 s step on one MI355X (``profiles/r1f_bench_kernel_stats.csv``:
    ``__amd_rocclr_copyBuffer`` 80 ms, ``bfloat16_copy`` 60 ms, ``bfloat16tofloat32_copy`` 38 ms, ``chunk_cat`` 11 ms).
    Here the same conversions are three multi-tensor launches per step.  Arithmetic is the FSDP path's: bf16 parameters
    and bf16 forward inputs at the root (FSDP2 ``cast_forward_inputs``), bf16 gradients widened to fp32, world-size-1
    "mean" = identity, fp32 optimizer state.

        rep = ReplicaMixedPrecision(model.dit)            # module parameters become bf16 copies
        opt = torch.optim.AdamW(rep.master_parameters(), ...)
        loss.backward(); rep.collect_grads(); clip_grad_norm_(rep.master_parameters(), 1.0); opt.step(); rep.publish()
    """

    def __init__(self, module: torch.nn.Module, param_dtype=torch.bfloat16, reduce_dtype=torch.float32):
        self.module, self.param_dtype, self.reduce_dtype = module, param_dtype, reduce_dtype
        self._compute, self._master, self._names = [], [], []
        seen = {}
        for name, p in module.named_parameters():
            if id(p) in seen or not p.is_floating_point():
                continue
            self._names.append(name)
            seen[id(p)] = True
            if p.requires_grad:
                master = torch.nn.Parameter(p.detach().to(torch.float32, copy=True), requires_grad=True)
                p.data = p.detach().to(param_dtype)
            else:
                # frozen (adapter "qkvo": 4.0 of 7.2 B parameters): nothing ever updates it, so no fp32 master is kept - 16 GB at
                # 5B that go to remat-free layers (round 4); the entry keeps the two lists aligned
                p.data = p.detach().to(param_dtype)
                master = p
            self._compute.append(p)
            self._master.append(master)
        self._hook = module.register_forward_pre_hook(self._cast_inputs, with_kwargs=True)
        module._master_holder = self                          # (ttt_amd.infra.optimizers.named_trainable finds the masters here)

    def _cast_inputs(self, module, args, kwargs):
        cast = lambda t: t.to(self.param_dtype) if isinstance(t, torch.Tensor) and t.is_floating_point() else t
        return tuple(cast(a) for a in args), {k: cast(v) for k, v in kwargs.items()}

    def master_parameters(self):
        return [m for m in self._master if m.requires_grad]

    def named_master_parameters(self):
        """(the module's parameter name, its fp32 master) of every trainable parameter: what the optimizer's name rules see"""
        return [(n, m) for n, m in zip(self._names, self._master) if m.requires_grad]

    def collect_grads(self):
        """Gradients of the compute copies -> ``reduce_dtype`` gradients of the masters (accumulating if one is already
        there), then dropped from the compute copies."""
        src, dst, acc_src, acc_dst = [], [], [], []
        for p, m in zip(self._compute, self._master):
            if p.grad is None:
                continue
            if m.grad is None:
                m.grad = torch.empty_like(m, dtype=self.reduce_dtype)
                src.append(p.grad)
                dst.append(m.grad)
            else:
                acc_src.append(p.grad)
                acc_dst.append(m.grad)
        if src:
            torch._foreach_copy_(dst, src)
        if acc_src:
            torch._foreach_add_(acc_dst, [g.to(self.reduce_dtype) for g in acc_src])
        for p in self._compute:
            p.grad = None

    def publish(self):
        """Masters -> compute copies (after the optimizer step)."""
        with torch.no_grad():
            pairs = [(p, m) for p, m in zip(self._compute, self._master) if m is not p]
            torch._foreach_copy_([p.data for p, _ in pairs], [m.data for _, m in pairs])

    def zero_grad(self):
        for m in self._master:
            m.grad = None
        for p in self._compute:
            p.grad = None
