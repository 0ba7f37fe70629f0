"""The end of a training step on the HIP path: gradient clipping and the optimizer step, GATED on the backward having been whole.

Why a helper: the TTT-MLP backward's cluster sweep (csrc/ttt_mfma_bwd4.hip) gives up LOUDLY when a partner workgroup is never
scheduled - it poisons that call's gradients with NaN and sets a host-mapped error word that makes the NEXT extension call
raise.  The host enqueues far ahead of the GPU, so by the time that next call is reached, ``clip_grad_norm_`` and
``optimizer.step()`` of the poisoned step are already queued: AdamW would run on NaN gradients and every parameter and moment
would be lost before the error surfaces (round-3 advisor finding).  Callers of the extension therefore MUST look at the word
after the backward and before ``optimizer.step()``; this function is that look: one device synchronisation per step, placed
where the reference's loop synchronises anyway (train.py reads ``loss.item()`` / the gradient norm for its log line).

Reference call sequence mirrored: train.py's ``clip_grad_norm_`` -> ``optimizer.step()`` -> ``optimizer.zero_grad()``.
"""
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def checked_optimizer_step(optimizer: torch.optim.Optimizer, parameters: Iterable[torch.nn.Parameter], max_norm: float,
                           process_group: Optional[dist.ProcessGroup] = None, extension=None, clip_fn=None, on_skip=None) -> Optional[torch.Tensor]:
    """Clip, verify, step.  Returns the total gradient norm, or ``None`` when the step was SKIPPED because a backward
    hand-over timed out on some rank (or the norm is not finite): the gradients are dropped (``zero_grad``), the error word is
    acknowledged on every rank, parameters and optimizer state are untouched, and the caller may run the batch again.
    Every rank takes the same decision (MAX all-reduce of the flag when a process group is initialised).  ``clip_fn(max_norm) ->
    norm`` replaces ``torch.nn.utils.clip_grad_norm_`` where the gradients are shards of a flat buffer (``FlatFSDP``)."""
    if extension is None:
        import test_time_training as extension
    params = [p for p in parameters if p.grad is not None]
    if clip_fn is not None:                                  # e.g. FlatFSDP.clip_grad_norm_: the norm over all ranks' shards
        total = clip_fn(max_norm)
    else:
        total = torch.nn.utils.clip_grad_norm_(params, max_norm) if params else None
    err = int(extension.sweep_error())                      # synchronises the device: everything the backward enqueued has run
    # the flag lives where the process group can reduce it: the current accelerator under RCCL, the host under gloo (a rank
    # without gradients has no norm tensor to borrow a device from)
    on = dist.is_available() and dist.is_initialized()
    rccl = on and "nccl" in str(dist.get_backend(process_group)) and torch.cuda.is_available()
    flag_dev = torch.device("cuda", torch.cuda.current_device()) if rccl else torch.device("cpu")
    if total is None:
        total = torch.zeros((), device=flag_dev)
    norm = total.full_tensor() if hasattr(total, "full_tensor") else total      # (FSDP2: the norm of sharded gradients is a DTensor)
    bad = torch.tensor([1 if (err != 0 or not bool(torch.isfinite(norm))) else 0], device=flag_dev, dtype=torch.int32)
    if on:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=process_group)
    if int(bad):
        optimizer.zero_grad(set_to_none=True)
        if on_skip is not None:
            on_skip()
        if err:
            extension.sweep_error_clear()
        return None
    optimizer._ttt_gate_done = True        # (a FlatFSDP step pre-hook would take the same decision again: one synchronisation is enough)
    try:
        optimizer.step()
    finally:
        optimizer._ttt_gate_done = False
    return total
