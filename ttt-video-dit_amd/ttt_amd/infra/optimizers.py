"""AdamW parameter groups and learning-rate schedules of the training loop - the public surface of the reference's
``ttt/infra/optimizers.py`` (``ParameterGroupManager`` :31-89, ``create_optimizer`` :92-133, ``create_specialized_optimizer``
:200-264, ``LRScheduleFunctions`` :267-331, ``create_basic_lr_scheduler`` :334-356, ``create_grouped_lr_scheduler`` :359-398,
``get_optimizer_and_scheduler`` :401-...), so that an unchanged ``train.py`` finds the same names.

The rules are the reference's: a parameter whose NAME contains "ttt" or "ssm" trains at ``ssm_lr``, every other one at ``base_lr``;
a name that contains "bias", "norm", "b1" or "b2" gets no weight decay, every other one 1e-4; AdamW betas (0.9, 0.95), eps 1e-8;
four groups in the order ttt_no_wd, ttt_wd, other_no_wd, other_wd, one LambdaLR schedule per group.

MI355X-first part: the model's trainable parameters may be held by ``FlatFSDP`` (``ttt_amd/infra/flat_fsdp.py``: flat sharded
fp32 masters) or ``ReplicaMixedPrecision`` (per-parameter fp32 masters).  Both expose ``named_master_parameters()`` - what the
optimizer steps, under names that carry the reference's name of the parameter(s) behind them - and ``named_trainable()`` here
prefers that over ``model.named_parameters()``; the grouping rules then apply unchanged.
"""
from __future__ import annotations

import functools
import math
from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, Iterable, List, Tuple

import torch
from torch import nn
from torch.optim.lr_scheduler import LambdaLR


class ScheduleType(str, Enum):
    COSINE = "cosine"
    LINEAR = "linear"


@dataclass(frozen=True)
class ScheduleConfig:
    schedule_type: ScheduleType
    warmup_steps: int
    total_steps: int
    lr_peak: float
    lr_end: float
    group_name: str


GROUP_NAMES = ("ttt_no_wd", "ttt_wd", "other_no_wd", "other_wd")      # the reference's group order (optimizers.py:165-197)


def named_trainable(model) -> List[Tuple[str, nn.Parameter]]:
    """(name, parameter the optimizer steps) pairs: the masters of a mixed-precision holder attached to the model
    (``model._master_holder`` / ``model.dit._master_holder``: FlatFSDP, ReplicaMixedPrecision) or the model's own parameters."""
    for m in (model, getattr(model, "dit", None)):
        holder = getattr(m, "_master_holder", None) if m is not None else None
        if holder is not None:
            # the holder's masters stand for the parameters of the module it wraps; trainable parameters of `model` OUTSIDE that
            # module are stepped as they are (the reference's model.named_parameters() would include them, optimizers.py:200-211)
            owned = {id(p) for p in m.parameters()}
            extras = [(n, p) for n, p in model.named_parameters() if p.requires_grad and id(p) not in owned]
            return list(holder.named_master_parameters()) + extras
    if hasattr(model, "named_master_parameters"):
        return list(model.named_master_parameters())
    return [(n, p) for n, p in model.named_parameters() if p.requires_grad]


class ParameterGroupManager:
    NO_WEIGHT_DECAY_PATTERNS = ["bias", "norm", "b1", "b2"]
    TTT_PARAMETER_PATTERNS = ["ttt", "ssm"]
    WEIGHT_DECAY_VALUE = 1e-4

    @classmethod
    def is_ttt_parameter(cls, param_name: str) -> bool:
        low = param_name.lower()
        return any(pat in low for pat in cls.TTT_PARAMETER_PATTERNS)

    @classmethod
    def should_skip_weight_decay(cls, param_name: str) -> bool:
        low = param_name.lower()
        return any(pat in low for pat in cls.NO_WEIGHT_DECAY_PATTERNS)

    @classmethod
    def group_of(cls, param_name: str) -> str:
        """one of GROUP_NAMES"""
        return ("ttt" if cls.is_ttt_parameter(param_name) else "other") + ("_no_wd" if cls.should_skip_weight_decay(param_name) else "_wd")

    @staticmethod
    def create_param_group(params: List[nn.Parameter], lr: float, weight_decay: float) -> Dict[str, Any]:
        if not params:
            raise ValueError("No parameters found for the group")
        return {"params": params, "lr": lr, "weight_decay": weight_decay}

    @classmethod
    def categorize_parameters(cls, model) -> Tuple[List, List, List, List]:
        """(ttt_no_wd, ttt_with_wd, other_no_wd, other_with_wd) of what the optimizer steps (``named_trainable``)."""
        by = {g: [] for g in GROUP_NAMES}
        for name, p in named_trainable(model):
            if p.requires_grad:
                by[cls.group_of(name)].append(p)
        return tuple(by[g] for g in GROUP_NAMES)


_ADAMW = {"betas": (0.9, 0.95), "eps": 1e-8}


def _adamw(groups, **kw):
    """fused AdamW on a HIP device (one multi-tensor launch per group), the stock implementation elsewhere"""
    first = next((p for g in groups for p in g["params"]), None)
    if first is not None and first.is_cuda:
        kw.setdefault("fused", True)
    return torch.optim.AdamW(groups, **_ADAMW, **kw)


def create_optimizer(model, learning_rate: float) -> torch.optim.AdamW:
    """Two groups (no weight decay / weight decay 1e-4), one learning rate - the reference's optimizer for models without a
    TTT layer (optimizers.py:92-133)."""
    no_wd, wd = [], []
    for name, p in named_trainable(model):
        (no_wd if ParameterGroupManager.should_skip_weight_decay(name) else wd).append(p)
    return _adamw([{"params": no_wd, "weight_decay": 0.0}, {"params": wd, "weight_decay": ParameterGroupManager.WEIGHT_DECAY_VALUE}],
                  lr=learning_rate)


def create_specialized_optimizer(model, base_lr: float, ssm_lr: float, final_lr: float, warmup_steps: int, total_steps: int,
                                 base_lr_schedule: ScheduleType, ssm_lr_schedule: ScheduleType, adapter_method: str
                                 ) -> Tuple[torch.optim.AdamW, List[ScheduleConfig]]:
    """The four AdamW groups by parameter name and their schedule configurations.  On a sharded holder a rank may own no
    element of a group (its parameters live in other ranks' shards): that group is still created - with a zero-element
    placeholder - so that every rank has the same groups and schedules."""
    assert adapter_method in ("sft", "qkvo"), f"Unsupported adapter method: '{adapter_method}'"
    cats = ParameterGroupManager.categorize_parameters(model)
    if not any(cats):
        raise ValueError("No trainable parameters found in the model")
    wd = ParameterGroupManager.WEIGHT_DECAY_VALUE
    spec = ((ssm_lr, 0.0, ssm_lr_schedule), (ssm_lr, wd, ssm_lr_schedule), (base_lr, 0.0, base_lr_schedule), (base_lr, wd, base_lr_schedule))
    sharded = any(hasattr(getattr(m, "_master_holder", None), "units") for m in (model, getattr(model, "dit", None)) if m is not None)
    groups, schedules = [], []
    ref = next(p for c in cats for p in c)
    for name, params, (lr, decay, sched) in zip(GROUP_NAMES, cats, spec):
        if not params:
            if not sharded:
                raise ValueError("No parameters found for the group")          # (the reference's behaviour)
            params = [nn.Parameter(torch.zeros(0, dtype=ref.dtype, device=ref.device))]
        groups.append(ParameterGroupManager.create_param_group(list(params), lr, decay))
        schedules.append(ScheduleConfig(sched, warmup_steps, total_steps, lr, final_lr, name))
    return _adamw(groups), schedules


class LRScheduleFunctions:
    """multipliers of the peak learning rate as functions of the step (LambdaLR)"""

    @staticmethod
    def cosine_decay_with_warmup(warmup_steps: int, decay_steps: int, lr_peak: float, lr_end: float, current_step: int) -> float:
        if lr_peak == 0 and lr_end == 0:          # frozen group
            return 1.0
        if current_step < warmup_steps:
            return float((current_step + 1) / warmup_steps)
        t = current_step - warmup_steps
        return (lr_end + (lr_peak - lr_end) * 0.5 * (1 + math.cos(math.pi * t / decay_steps))) / lr_peak

    @staticmethod
    def linear_decay_with_warmup(warmup_steps: int, total_steps: int, lr_peak: float, lr_end: float, current_step: int) -> float:
        if current_step < warmup_steps:
            return float((current_step + 1) / warmup_steps)
        frac = min((current_step - warmup_steps) / max(1, total_steps - warmup_steps), 1.0)
        return 1.0 - frac * (1.0 - lr_end / lr_peak)


def create_basic_lr_scheduler(optimizer, warmup_steps: int, total_steps: int, lr_peak: float, lr_end: float) -> LambdaLR:
    fn = functools.partial(LRScheduleFunctions.cosine_decay_with_warmup, warmup_steps, max(1, total_steps - warmup_steps), lr_peak, lr_end)
    return LambdaLR(optimizer, lr_lambda=fn)


def create_grouped_lr_scheduler(optimizer, schedule_configs: Iterable[ScheduleConfig]) -> LambdaLR:
    fns = []
    for c in schedule_configs:
        if c.schedule_type == ScheduleType.COSINE:
            fns.append(functools.partial(LRScheduleFunctions.cosine_decay_with_warmup, c.warmup_steps, max(1, c.total_steps - c.warmup_steps),
                                         c.lr_peak, c.lr_end))
        elif c.schedule_type == ScheduleType.LINEAR:
            fns.append(functools.partial(LRScheduleFunctions.linear_decay_with_warmup, c.warmup_steps, c.total_steps, c.lr_peak, c.lr_end))
        else:
            raise ValueError(f"Unsupported schedule type: '{c.schedule_type}'")
    return LambdaLR(optimizer, lr_lambda=fns)


def get_optimizer_and_scheduler(model, config: Any):
    """(optimizer, LambdaLR, schedule configuration(s)) from a job configuration with the reference's fields
    (``config.model.ssm_layer``, ``config.optimizer.lr / lr_ssm / lr_end / lr_schedule / lr_ssm_schedule``,
    ``config.training.warmup_steps / steps / adapter_method``; reference optimizers.py:401-...)."""
    o, t = config.optimizer, config.training
    if config.model.ssm_layer == "none":
        opt = create_optimizer(model, o.lr)
        return opt, create_basic_lr_scheduler(opt, t.warmup_steps, t.steps, o.lr, o.lr_end), \
            ScheduleConfig(o.lr_schedule, t.warmup_steps, t.steps, o.lr, o.lr_end, "standard")
    opt, cfgs = create_specialized_optimizer(model, o.lr, o.lr_ssm, o.lr_end, t.warmup_steps, t.steps, ScheduleType(o.lr_schedule),
                                             ScheduleType(o.lr_ssm_schedule), t.adapter_method)
    return opt, create_grouped_lr_scheduler(opt, cfgs), cfgs
