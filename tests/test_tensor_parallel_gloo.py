"""Tensor parallelism through the reference's API surface (``apply_tp(model, tp_mesh)`` / ``TTTBase.init_device_mesh``:
``ttt/infra/parallelisms.py``:106-152, ``ttt/models/ssm/ttt_layer.py``:114-131) on CPU over gloo: world size 2, a 1-D DeviceMesh as
the reference passes it.  Two layouts: ``"full"`` - the reference's whole plan (head shards for local attention and the TTT layer,
token shards for AdaLN / output projections / norms / gates / MLP / final layer) - and ``"ttt_heads"`` (round 2: only the TTT
layer sharded).  Outputs and - after ``tp_sync_gradients`` - every gradient must equal the single-process DiT.  Dual-form scan and kernel plumbing
(HIP extension replaced by the oracle-backed stand-in), single- and multi-scene."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"mlp_dual_3scene": ("ttt_mlp", 4, 7, 3, False), "linear_kernel_1scene": ("ttt_linear", 2, 3, 1, True)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(case):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    ssm, nh, frames, scenes, use_kernel = CASES[case]
    torch.manual_seed(5)
    cfg = ModelConfig(model_dim=64 * nh, num_heads=nh, num_layers=2, mini_batch_size=16, latent_height=8, latent_width=8,
                      compressed_num_frames=frames, ssm_layer=ssm, text_dim=32, time_embed_dim=64, attn_length=2,
                      prefix_temporal_length=1, adapter_method="sft", scan_checkpoint_group_size=2)
    m = DiffusionTransformer(cfg)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 0.05)
            elif p.ndim == 1:
                p.add_(0.05 * torch.randn_like(p))
    for layer in m.layers:
        layer.seq_modeling_block.ssm.ttt.use_kernel = use_kernel
    g = torch.Generator().manual_seed(3)
    return m, (torch.randn(1, frames, 16, 8, 8, generator=g), torch.randn(1, scenes, 16, 32, generator=g), torch.tensor([412]))


def _worker(rank, world, port, case, out_dir, layout="ttt_heads"):
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from torch.distributed.device_mesh import init_device_mesh
    from oracle import cpu_ext
    from ttt_amd.infra.parallelisms import apply_tp, end_distributed, init_distributed, tp_sync_gradients
    cpu_ext.install()
    init_distributed("gloo")
    m, inputs = _build(case)
    apply_tp(m, init_device_mesh("cpu", (world,), mesh_dim_names=("tp",)), layout=layout)
    out = m(*inputs)
    out.square().mean().backward()
    tp_sync_gradients(m)
    if rank == 1:     # any rank must hold the complete answer
        torch.save({"out": out.detach(), "grads": {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}},
                   os.path.join(out_dir, "tp.pt"))
    end_distributed()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("layout", ["full", "ttt_heads"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_head_sharded_tp_matches_single_process(case, layout, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), case, str(tmp_path), layout), nprocs=2, join=True)
    from oracle import cpu_ext
    cpu_ext.install()
    try:
        m, inputs = _build(case)
        out = m(*inputs)
        out.square().mean().backward()
    finally:
        cpu_ext.uninstall()
    got = torch.load(os.path.join(tmp_path, "tp.pt"))
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(got["out"], out.detach()) < 1e-5
    ref = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    # (a parameter that cannot influence the loss - the text gates of the last layer - has an all-zero gradient in the
    # single-process graph and none in the token-sharded one, whose last layer never forms the text rows)
    assert set(got["grads"]) <= set(ref)
    for k in set(ref) - set(got["grads"]):
        assert float(ref[k].abs().max()) == 0.0, k
    bad = {k: rel(got["grads"][k], ref[k]) for k in got["grads"] if float(ref[k].norm()) > 0 and not rel(got["grads"][k], ref[k]) < 2e-4}
    assert not bad, bad


def test_init_device_mesh_rejects_indivisible_heads():
    import torch.distributed as dist
    from ttt_amd.models.configs import ModelConfig
    from ttt_amd.models.ssm.ttt_layer import TTTWrapper
    sys.path.insert(0, os.path.join(ROOT, "ttt-video-dit_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = TTTWrapper(ModelConfig(model_dim=128, num_heads=2, num_layers=1, mini_batch_size=16, latent_height=4, latent_width=4,
                                   compressed_num_frames=2, ssm_layer="ttt_linear"))
        m.ttt.init_device_mesh(dist.group.WORLD)      # one rank: accepted, forward unchanged
        assert m.ttt.tp_mesh is not None
    finally:
        dist.destroy_process_group()


def _inputs_for(dp_rank, case):
    """sample of data-parallel rank ``dp_rank`` (the ranks of a TP group share it)"""
    _, nh, frames, scenes, _ = CASES[case]
    g = torch.Generator().manual_seed(100 + dp_rank)
    return torch.randn(1, frames, 16, 8, 8, generator=g), torch.randn(1, scenes, 16, 32, generator=g), torch.tensor([412 + 97 * dp_rank])


def _worker_2d(rank, world, port, case, out_dir, tp, reshard=False):
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import cpu_ext
    from ttt_amd.infra.parallelisms import apply_parallelisms, end_distributed, init_distributed, tp_sync_gradients
    cpu_ext.install()
    init_distributed("gloo")
    m, _ = _build(case)
    m.remat_free_layers, m.remat_keep = 1, ("attn", "scan")     # layer 1 re-materialised (its collectives run again in backward)
    mesh, dp_rank, dp = apply_parallelisms(m, tp_sharding=tp, param_dtype=torch.float32, reshard_after_forward=reshard)
    assert (dp_rank, dp) == (rank // tp, world // tp) and mesh["tp"].size() == tp
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.05)
    losses = []
    for _ in range(2):                       # second step: the sharded update of step one went into the next all-gather
        opt.zero_grad(set_to_none=True)
        loss = m(*_inputs_for(dp_rank, case)).square().mean()
        loss.backward()
        tp_sync_gradients(m)                 # (no-op here: the reduce-scatter did it)
        grads = {k: p.grad.full_tensor().clone() for k, p in m.named_parameters() if p.grad is not None}
        opt.step()
        losses.append(float(loss))
    final = {k: p.full_tensor().detach().clone() for k, p in m.named_parameters()}
    if rank == world - 1:
        torch.save({"grads": grads, "final": final}, os.path.join(out_dir, "tp2d.pt"))
    torch.save(losses, os.path.join(out_dir, f"loss{rank}.pt"))
    end_distributed()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,tp,reshard", [(2, 2, False), (4, 2, False), (2, 2, True)])
def test_tp_times_fsdp_step_matches_data_parallel_reference(world, tp, reshard, tmp_path):
    """``apply_parallelisms`` (reference ``parallelisms.py``:92-104): "full" TP layout inside groups of ``tp`` ranks, FSDP2 over
    ALL ranks with the divide factor ``dp`` - two SGD steps against a single-process statement of data parallelism over the
    ``dp`` samples (gradient = mean over the samples).  fp32 policy: the comparison is about the plan, not about bf16."""
    case = "mlp_dual_3scene"
    # (reshard = the reference's FSDP setting: parameters gathered again in backward, beside the re-materialised layer's collectives)
    mp.spawn(_worker_2d, args=(world, _free_port(), case, str(tmp_path), tp, reshard), nprocs=world, join=True)
    dp = world // tp
    from oracle import cpu_ext
    cpu_ext.install()
    try:
        m, _ = _build(case)
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        ref_losses = []
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            ls = []
            for r in range(dp):
                loss = m(*_inputs_for(r, case)).square().mean()
                (loss / dp).backward()
                ls.append(float(loss.detach()))
            ref_g = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
            opt.step()
            ref_losses.append(ls)
    finally:
        cpu_ext.uninstall()
    got = torch.load(os.path.join(tmp_path, "tp2d.pt"))
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for rank in range(world):               # every rank of a TP group reports its group's sample loss, both steps
        ls = torch.load(os.path.join(tmp_path, f"loss{rank}.pt"))
        for step in range(2):
            assert abs(ls[step] - ref_losses[step][rank // tp]) <= 2e-5 * abs(ref_losses[step][rank // tp]), (rank, step, ls, ref_losses)
    bad = {k: rel(got["grads"][k], ref_g[k]) for k in got["grads"] if k in ref_g and float(ref_g[k].norm()) > 0
           and not rel(got["grads"][k], ref_g[k]) < 5e-4}
    assert not bad, bad
    ref_final = dict(m.named_parameters())
    badp = {k: rel(v, ref_final[k].detach()) for k, v in got["final"].items() if not rel(v, ref_final[k].detach()) < 1e-4}
    assert not badp, badp
