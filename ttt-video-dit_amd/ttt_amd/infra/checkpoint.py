"""Sharded checkpoints in the reference's on-disk layout.

The reference saves and restores training state with ``torch.distributed.checkpoint`` (``ttt/infra/checkpoint.py``:
``Checkpointer.save`` :93-108, ``load`` :61-91, ``load_pretrained`` :47-59): one DCP directory whose top-level entries are
``model``, ``optimizer``, ``lr_scheduler``, ``data_module`` and ``metadata``; a *pretrained* directory may also be a bare
model state dict (the converted CogVideoX weights) - both must load.  Because the module tree and the parameter names of
``ttt_amd`` are the reference's, a directory written by either code base loads into the other.

Differences in form, not in format: the training-loop collaborators (learning-rate scheduler, data sampler, logger) are
optional here - a benchmark or a fine-tuning script that has none of them passes ``None`` - and every rank-collective step is
in one place (``_gather`` / ``_scatter``), so that FSDP2-sharded and unsharded models go through the same code.  Works
without a process group too (DCP then writes a single-rank checkpoint).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch.distributed.checkpoint as dcp
from torch.distributed.checkpoint.api import CheckpointException
from torch.distributed.checkpoint.state_dict import (StateDictOptions, get_model_state_dict, get_state_dict,
                                                     set_model_state_dict, set_state_dict)

MODEL_KEY, OPTIMIZER_KEY, SCHEDULER_KEY, DATA_KEY, METADATA_KEY = "model", "optimizer", "lr_scheduler", "data_module", "metadata"


class Checkpointer:
    """``Checkpointer(model, optimizer, lr_scheduler=None, data_sampler=None, log=None)``.

    ``data_sampler`` is whatever object carries the data position (``state_dict()`` / ``load_state_dict()``; the reference
    passes its data module and uses ``.sampler``), ``log`` a callable taking one string."""

    def __init__(self, model, optimizer=None, lr_scheduler=None, data_sampler=None, log=None):
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.data_sampler = getattr(data_sampler, "sampler", data_sampler)
        self.metadata: Dict[str, Any] = {"wandb_id": None}
        self._log = log or (lambda msg: None)

    # -- state <-> flat dictionaries of (sharded) tensors ------------------------------------------------------------------
    def _gather(self) -> Dict[str, Any]:
        if self.optimizer is not None:
            model_sd, optim_sd = get_state_dict(self.model, self.optimizer)
        else:
            model_sd, optim_sd = get_model_state_dict(self.model), {}
        return {MODEL_KEY: model_sd, OPTIMIZER_KEY: optim_sd,
                SCHEDULER_KEY: self.lr_scheduler.state_dict() if self.lr_scheduler is not None else {},
                DATA_KEY: self.data_sampler.state_dict() if self.data_sampler is not None else {},
                METADATA_KEY: dict(self.metadata)}

    def _scatter(self, state: Dict[str, Any]) -> None:
        if self.optimizer is not None:
            set_state_dict(self.model, self.optimizer, model_state_dict=state[MODEL_KEY], optim_state_dict=state[OPTIMIZER_KEY])
        else:
            set_model_state_dict(self.model, model_state_dict=state[MODEL_KEY], options=StateDictOptions(strict=True))
        if self.lr_scheduler is not None and state.get(SCHEDULER_KEY):
            self.lr_scheduler.load_state_dict(state[SCHEDULER_KEY])
        if self.data_sampler is not None and state.get(DATA_KEY):
            self.data_sampler.load_state_dict(state[DATA_KEY])
        self.metadata = dict(state.get(METADATA_KEY) or self.metadata)

    # -- public surface (reference names) ------------------------------------------------------------------------------------
    def save(self, path: str) -> None:
        """Everything needed to resume, as one DCP directory (reference :93-108)."""
        self._log(f"Saving state at {path}.")
        state = self._gather()
        if not state[SCHEDULER_KEY]:
            state[SCHEDULER_KEY] = {}
        dcp.save(state_dict={k: v for k, v in state.items() if v or k in (MODEL_KEY, OPTIMIZER_KEY)}, checkpoint_id=path)
        self._log("Completed saving state.")

    def load(self, path: str) -> None:
        """Resume: model, optimizer and whatever else this checkpointer was given (reference :61-91).  Entries this
        checkpointer has no owner for (a scheduler state in a directory loaded without a scheduler) are left on disk."""
        self._log(f"Loading in state from {path}.")
        state = {k: v for k, v in self._gather().items() if v or k == MODEL_KEY}
        if self.optimizer is None:
            state.pop(OPTIMIZER_KEY, None)
        dcp.load(state_dict=state, checkpoint_id=path)
        self._scatter({**{OPTIMIZER_KEY: {}}, **state})
        self._log("Completed loading in state.")

    def load_pretrained(self, path: str) -> None:
        """Model weights only, strict, from either layout: a bare model state dict (converted CogVideoX weights) or a full
        training checkpoint of an earlier stage, whose weights sit under ``model`` (reference :47-59)."""
        self._log(f"Loading in state from {path}.")
        weights = get_model_state_dict(self.model)
        try:
            dcp.load(state_dict=weights, checkpoint_id=path)
        except (CheckpointException, RuntimeError, KeyError, ValueError):      # flat keys absent: a full training checkpoint
            nested = {MODEL_KEY: get_model_state_dict(self.model)}
            try:
                dcp.load(state_dict=nested, checkpoint_id=path)
            except CheckpointException as exc:      # (derives from BaseException) neither layout matches this model
                raise RuntimeError(f"{path} holds the weights of a different model: {exc}") from exc
            weights = nested[MODEL_KEY]
        set_model_state_dict(self.model, model_state_dict=weights, options=StateDictOptions(strict=True))

    def set_wandb(self, wandb_id: Optional[str]) -> None:
        self.metadata = {"wandb_id": wandb_id}
