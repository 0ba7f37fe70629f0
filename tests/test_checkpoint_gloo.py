"""Sharded checkpoints (ttt_amd/infra/checkpoint.py) on the CPU: the reference's DCP directory layout
(``model`` / ``optimizer`` / ... entries; ttt/infra/checkpoint.py:61-108) round-trips model and optimizer state, a checkpoint
written by two FSDP2 ranks (gloo) loads into one unsharded process, and ``load_pretrained`` accepts both a bare model state
dict and a full training checkpoint."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed.checkpoint as dcp
import torch.multiprocessing as mp

from oracle import cpu_ext
from ttt_amd.infra.checkpoint import Checkpointer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(seed):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    torch.manual_seed(seed)
    cfg = ModelConfig(model_dim=64, num_heads=1, num_layers=2, mini_batch_size=16, latent_height=4, latent_width=4,
                      compressed_num_frames=2, ssm_layer="ttt_linear", text_dim=16, time_embed_dim=32, attn_length=1,
                      prefix_temporal_length=1, adapter_method="sft", scan_checkpoint_group_size=2)
    m = DiffusionTransformer(cfg)
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.normal_(0, 0.05)
    return m


def _step(m, opt, seed=3):
    g = torch.Generator().manual_seed(seed)
    v, t = torch.randn(1, 2, 16, 8, 8, generator=g), torch.randn(1, 1, 16, 16, generator=g)
    opt.zero_grad()
    m(v, t, torch.tensor([200])).square().mean().backward()
    opt.step()


@pytest.fixture
def ext():
    cpu_ext.install()
    yield
    cpu_ext.uninstall()


def test_save_load_roundtrip_single_process(ext, tmp_path):
    m = _build(0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
    _step(m, opt)
    sched.step()
    ck = Checkpointer(m, opt, sched)
    ck.set_wandb("run-42")
    ck.save(str(tmp_path / "full"))

    m2 = _build(1)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-2)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, step_size=3, gamma=0.5)
    _step(m2, opt2, seed=9)                       # optimizer state exists and differs
    ck2 = Checkpointer(m2, opt2, sched2)
    ck2.load(str(tmp_path / "full"))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and len(s1) > 10
    for k in s1:
        assert torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) and torch.equal(s1[k]["exp_avg_sq"], s2[k]["exp_avg_sq"])
    assert sched2.last_epoch == sched.last_epoch and ck2.metadata == {"wandb_id": "run-42"}
    _step(m, opt), _step(m2, opt2)                # and training continues identically
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_load_pretrained_accepts_both_layouts(ext, tmp_path):
    m = _build(0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    _step(m, opt)
    Checkpointer(m, opt).save(str(tmp_path / "stage1"))                       # full training checkpoint: weights under "model"
    dcp.save(state_dict=m.state_dict(), checkpoint_id=str(tmp_path / "bare"))   # converted weights: flat keys
    for name in ("stage1", "bare"):
        m2 = _build(5)
        Checkpointer(m2).load_pretrained(str(tmp_path / name))
        for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(a, b), (name, k)
    with pytest.raises(RuntimeError):                                         # strict: a model with other keys must not load silently
        from ttt_amd.models.cogvideo.dit import MLP
        from ttt_amd.models.configs import ModelConfig
        Checkpointer(MLP(ModelConfig(model_dim=64, num_heads=1, num_layers=1))).load_pretrained(str(tmp_path / "bare"))


# ---- written by two FSDP2 ranks, read by one process ---------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import cpu_ext as ce
    from ttt_amd.infra.checkpoint import Checkpointer as CK
    from ttt_amd.infra.parallelisms import apply_fsdp, end_distributed, get_dp_mesh, init_distributed
    ce.install()
    init_distributed("gloo")
    m = _build(0)
    apply_fsdp(m, get_dp_mesh(), param_dtype=torch.float32)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    _step(m, opt)                                     # both ranks step on the same sample: the single-process reference below
    CK(m, opt).save(os.path.join(out_dir, "sharded"))
    end_distributed()


@pytest.mark.timeout(600)
def test_checkpoint_of_two_fsdp_ranks_loads_unsharded(ext, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ref = _build(0)
    ref_opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    _step(ref, ref_opt)
    m = _build(7)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    _step(m, opt, seed=11)
    Checkpointer(m, opt).load(str(tmp_path / "sharded"))
    for (k, a), (_, b) in zip(ref.state_dict().items(), m.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), k
    s1, s2 = ref_opt.state_dict()["state"], opt.state_dict()["state"]
    for k in s1:
        assert torch.allclose(s1[k]["exp_avg"], s2[k]["exp_avg"], rtol=1e-4, atol=1e-8)


# ---- a directory written by the REFERENCE's own Checkpointer (tests/golden/gen_checkpoint_golden.py) --------------------------
def _golden_dir():
    return os.path.join(ROOT, "tests", "golden", "ref_checkpoint")


def test_reference_written_checkpoint_loads(ext):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    exp = torch.load(os.path.join(ROOT, "tests", "golden", "ref_checkpoint_expected.pt"))
    torch.manual_seed(123)
    m = DiffusionTransformer(ModelConfig(**exp["cfg"]))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
    _step(m, opt, seed=21)                                       # optimizer state exists, everything differs from the checkpoint

    class Sampler:
        got = None

        def state_dict(self):
            return {"epoch": 0, "position": 0}

        def load_state_dict(self, sd):
            Sampler.got = dict(sd)

    ck = Checkpointer(m, opt, sched, Sampler())
    ck.load(_golden_dir())
    params = dict(m.named_parameters())
    assert set(params) == set(exp["params"])
    for k, v in exp["params"].items():
        assert torch.equal(params[k].detach(), v), k
    names = [k for k, _ in m.named_parameters()]
    state = opt.state_dict()["state"]
    for i, s in state.items():
        assert torch.equal(s["exp_avg"], exp["exp_avg"][names[i]]), names[i]
    assert sched.last_epoch == exp["scheduler_last_epoch"]
    assert Sampler.got == {"epoch": 3, "position": 17} and ck.metadata == {"wandb_id": "ref-run-7"}
    # the same directory as a "pretrained" source for a bare model (weights under "model")
    m2 = DiffusionTransformer(ModelConfig(**exp["cfg"]))
    Checkpointer(m2).load_pretrained(_golden_dir())
    for k, v in exp["params"].items():
        assert torch.equal(dict(m2.named_parameters())[k].detach(), v), k
