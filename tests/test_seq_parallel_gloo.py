"""Sequence parallelism (ttt_amd/infra/sequence_parallel.py) on CPU over gloo: T ranks, each holding 1/T of the tokens
for the token-wise work and 1/T of the heads for attention / the TTT scan, must reproduce the single-process forward of the
same DiT and, through the differentiable collectives + ``sum_gradients``, its loss and every parameter gradient.  World sizes 2 and 3 (3 = token shards that need padding), single- and multi-scene, TTT-MLP and TTT-Linear, the
dual-form PyTorch scan and the kernel plumbing (HIP extension replaced by the oracle-backed stand-in, as in
test_fsdp_gloo.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # name: (world, num_heads, ssm_layer, frames, scenes, use_kernel)
    "mlp_3scene_w2": (2, 2, "ttt_mlp", 7, 3, False),
    "mlp_3scene_w3_padded": (3, 3, "ttt_mlp", 7, 3, False),
    "linear_1scene_w2_kernel": (2, 2, "ttt_linear", 3, 1, True),
    "linear_2scene_w2_kernel": (2, 4, "ttt_linear", 5, 2, True),     # (the TTT-MLP kernel boundary is bf16-only, as in the reference)
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(case):
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    world, nh, ssm, frames, scenes, use_kernel = CASES[case]
    torch.manual_seed(7)
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
    g = torch.Generator().manual_seed(11)
    inputs = (torch.randn(2, frames, 16, 8, 8, generator=g), torch.randn(2, scenes, 16, 32, generator=g), torch.tensor([300, 650]))
    return m.eval(), inputs


def _loss(out):
    return (out * torch.linspace(-1, 1, out.numel()).view_as(out)).sum() / out.numel() + out.square().mean()


def _worker(rank, world, port, case, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from oracle import cpu_ext
    from ttt_amd.infra.sequence_parallel import SeqParallel
    if CASES[case][5]:
        cpu_ext.install()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, inputs = _build(case)
    sp = m.sequence_parallel = SeqParallel()
    with torch.no_grad():
        out = m(*inputs)
    torch.save(out, os.path.join(out_dir, f"sp_{rank}.pt"))
    # training through the same layout: every rank evaluates the loss on the gathered output, parameter gradients are
    # partial per rank (its tokens / its heads) and summed over the group
    loss = _loss(m(*inputs))
    loss.backward()
    sp.sum_gradients(m, bucket_bytes=1 << 14)          # several buckets
    if rank == 0:
        torch.save({"loss": loss.detach(), "grads": {n: p.grad for n, p in m.named_parameters() if p.grad is not None}},
                   os.path.join(out_dir, "sp_train.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("case", list(CASES))
def test_sequence_parallel_forward_equals_single_process(case, tmp_path):
    world = CASES[case][0]
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import cpu_ext
    if CASES[case][5]:
        cpu_ext.install()
    try:
        m, inputs = _build(case)
        with torch.no_grad():
            ref = m(*inputs)
        ref_loss = _loss(m(*inputs))
        ref_loss.backward()
        ref_grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    finally:
        if CASES[case][5]:
            cpu_ext.uninstall()
    outs = [torch.load(os.path.join(str(tmp_path), f"sp_{r}.pt")) for r in range(world)]
    for o in outs:
        err = float((o - ref).norm() / ref.norm())
        assert err < 1e-5, (case, err)
    assert torch.equal(outs[0], outs[-1])      # every rank ends with the same full output
    tr = torch.load(os.path.join(str(tmp_path), "sp_train.pt"))
    assert abs(float(tr["loss"]) - float(ref_loss.detach())) < 1e-5 * max(1.0, abs(float(ref_loss.detach())))
    # (a parameter that cannot influence the loss - the text gates of the last layer - has an all-zero gradient in one
    # graph and none in the other)
    zero = torch.zeros(())
    errs = []
    for n, g in ref_grads.items():
        got = tr["grads"].get(n)
        if got is None:
            assert float(g.abs().max()) == 0.0, n
            continue
        errs.append((float((got - g).norm() / g.norm().clamp_min(1e-12)) if float(g.norm()) > 0 else float(got.abs().max()), n))
    assert set(tr["grads"]) <= set(ref_grads)
    assert max(errs)[0] < 2e-4, max(errs)


def _sample(m, case):
    from ttt_amd.models.cogvideo.sampling import DiscreteDenoiser, VPSDEDPMPP2MSampler
    world, nh, ssm, frames, scenes, _ = CASES[case]
    smp = VPSDEDPMPP2MSampler(denoiser=DiscreteDenoiser(m, num_idx=1000, quantize_c_noise=False, dtype=torch.float32),
                              discretization_config={}, guider_config={"scale": 6, "exp": 5, "num_steps": 3}, device="cpu", num_steps=3)
    torch.manual_seed(5)                       # the sampler's noise draws must agree across ranks
    noise = torch.randn(1, frames, 16, 8, 8)
    text, neg = torch.randn(1, scenes, 16, 32), torch.randn(1, scenes, 16, 32)
    with torch.no_grad():
        return smp(noise, {"crossattn": text}, {"crossattn": neg})


def _sample_worker(rank, world, port, case, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from ttt_amd.infra.sequence_parallel import SeqParallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, _ = _build(case)
    m.sequence_parallel = SeqParallel()
    torch.save(_sample(m, case), os.path.join(out_dir, f"sample_{rank}.pt"))
    dist.destroy_process_group()


def test_sampler_over_sequence_parallel_model(tmp_path):
    """The mirrored DPM-Solver++ sampler (batched guidance pair) driving a sequence-parallel DiT: same video as one process."""
    case = "mlp_3scene_w2"
    mp.spawn(_sample_worker, args=(2, _free_port(), case, str(tmp_path)), nprocs=2, join=True)
    for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    m, _ = _build(case)
    ref = _sample(m, case)
    a, b = (torch.load(os.path.join(str(tmp_path), f"sample_{r}.pt")) for r in range(2))
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert float((a - ref).norm() / ref.norm()) < 1e-4
