"""Write a DCP checkpoint directory with the REFERENCE's own Checkpointer around the reference's own DiT (build container only).

    TORCHDYNAMO_DISABLE=1 python tests/golden/gen_checkpoint_golden.py

Imports /root/reference (read-only; stubs of SURVEY.md section 8c), builds a tiny reference DiffusionTransformer + AdamW +
scheduler, takes one optimizer step, and lets ``ttt.infra.checkpoint.Checkpointer.save`` write ``tests/golden/ref_checkpoint/``
(a ``.metadata`` file + one ``.distcp`` shard, ~1 MB).  ``ref_checkpoint_expected.pt`` holds the parameter values and two
optimizer moments for the comparison in tests/test_checkpoint_gloo.py.  Nothing from the reference is copied: only numbers.
"""
import os
import shutil
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules["wandb"] = types.ModuleType("wandb")
import tomli  # noqa: E402

sys.modules["tomllib"] = tomli
sys.path.insert(0, "/root/reference")

from ttt.infra.checkpoint import Checkpointer  # noqa: E402
from ttt.models.cogvideo.dit import DiffusionTransformer  # noqa: E402
from ttt.models.configs import ModelConfig  # noqa: E402

CFG = dict(model_dim=64, num_heads=1, num_layers=2, mini_batch_size=16, latent_height=4, latent_width=4, compressed_num_frames=2,
           ssm_layer="ttt_linear", text_dim=16, time_embed_dim=32, attn_length=1, prefix_temporal_length=1, adapter_method="sft",
           scan_checkpoint_group_size=2)


class _Sampler:
    def state_dict(self):
        return {"epoch": 3, "position": 17}

    def load_state_dict(self, sd):
        pass


def main():
    torch.manual_seed(0)
    cfg = ModelConfig(**CFG)
    m = DiffusionTransformer(cfg)
    for layer in m.layers:
        layer.seq_modeling_block.ssm.ttt.use_kernel = False
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.normal_(0, 0.05)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
    g = torch.Generator().manual_seed(3)
    v, t = torch.randn(1, 2, 16, 8, 8, generator=g), torch.randn(1, 1, 16, 16, generator=g)
    m(v, t, torch.tensor([200])).square().mean().backward()
    opt.step()
    sched.step()
    logger = types.SimpleNamespace(write=lambda msg: None, wandb_logger=types.SimpleNamespace(job_id="ref-run-7"))
    data_module = types.SimpleNamespace(sampler=_Sampler())
    out = os.path.join(HERE, "ref_checkpoint")
    shutil.rmtree(out, ignore_errors=True)
    Checkpointer(m, opt, sched, data_module, logger).save(out)
    names = [k for k, _ in m.named_parameters()]
    state = opt.state_dict()["state"]
    torch.save({"cfg": CFG, "params": {k: p.detach().clone() for k, p in m.named_parameters()},
                "exp_avg": {names[i]: s["exp_avg"].clone() for i, s in state.items()},
                "scheduler_last_epoch": sched.last_epoch}, os.path.join(HERE, "ref_checkpoint_expected.pt"))
    print("wrote", out, os.listdir(out))


if __name__ == "__main__":
    main()
