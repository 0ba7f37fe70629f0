"""Sampling-path measurement ("Sampling (cfg 5): latent frames/s"; 8f #3) on one MI355X.

    python tools/sample_bench.py [--video-length 3sec] [--steps 3] [--layers 42] [--sequential]

CogVideoX-5B + TTT-MLP in bf16 with the evaluation settings of the reference (configs/eval/ttt-mlp/*.toml:
mini_batch_size = 16, no scan checkpoints), random-init weights and synthetic text embeddings, driven by the mirrored
DPM-Solver++(2M) sampler (ttt_amd/models/cogvideo/sampling.py).  A denoising step is one network evaluation on the
classifier-free-guidance pair: one batch of two here, two batch-1 calls with --sequential (the reference's order,
cogvideo/utils.py:478-492).  Prints one JSON line: seconds per denoising step and latent frames/s, with the 50-step
projection.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttt-video-dit_amd"))

TEXT_LEN = {"3sec": 498, "9sec": 502, "18sec": 500, "30sec": 497, "63sec": 458}   # configs/eval/ttt-mlp/*.toml:16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--video-length", default="3sec", choices=list(TEXT_LEN))
    ap.add_argument("--steps", type=int, default=3, help="denoising steps to time (after one untimed step)")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--sequential", action="store_true", help="evaluate the guidance pair one sample at a time")
    ap.add_argument("--impl", default="auto", choices=["auto", "generic", "mfma"])
    ap.add_argument("--sequence-parallel", action="store_true",
                    help="launched with torch.distributed.run on N GPUs: one video sampled by N ranks (token shards for the "
                         "token-wise work, head shards for attention / the TTT scan; ttt_amd/infra/sequence_parallel.py)")
    ap.add_argument("--no-warmup", action="store_true", help="skip the untimed step (long videos: the timed steps then include one-time costs)")
    a = ap.parse_args()

    import test_time_training as ext
    from ttt_amd.infra.parallelisms import enable_tuned_gemms, init_model_parameters
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.cogvideo.sampling import DiscreteDenoiser, VPSDEDPMPP2MSampler
    from ttt_amd.models.configs import ModelConfig

    rank = 0
    if a.sequence_parallel:
        import torch.distributed as dist
        rank = int(os.environ.get("RANK", "0"))
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl")                     # RCCL over xGMI
    else:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
    ext.load_library()
    ext.set_impl(a.impl)
    enable_tuned_gemms()
    over = {} if a.layers is None else {"num_layers": a.layers}
    cfg = ModelConfig.get_preset("5B", a.video_length, ssm_layer="ttt_mlp", adapter_method="sft", mini_batch_size=16,
                                 scan_checkpoint_group_size=10 ** 6, **over)
    frames, text_len = cfg.compressed_num_frames, TEXT_LEN[a.video_length]
    scenes = max((frames - 1) // 12, 1)
    with torch.device("meta"):
        net = DiffusionTransformer(cfg)
    net.to_empty(device=dev)
    torch.manual_seed(1234)
    with torch.no_grad():
        init_model_parameters(net)
        for layer in net.layers:
            layer.seq_modeling_block.rotary.init_freqs()
            layer.seq_modeling_block.ssm.init_freqs()
    net = net.to(torch.bfloat16).eval()
    if a.sequence_parallel:
        from ttt_amd.infra.sequence_parallel import SeqParallel
        net.sequence_parallel = SeqParallel()               # same weights (same seed) and same inputs on every rank
    for layer in net.layers:                                # rotary tables stay fp32 (reference cast_rotary_freqs)
        layer.seq_modeling_block.rotary.init_freqs()
        layer.seq_modeling_block.ssm.init_freqs()

    L = frames * 1350 + scenes * text_len
    NC = L // 16
    impl = ext.resolved_impl(2, cfg.num_heads, NC, 16, 64, NC, torch.bfloat16, mlp=True, backward=False)

    def run(n_steps):
        sampler = VPSDEDPMPP2MSampler(
            denoiser=DiscreteDenoiser(net, num_idx=1000, quantize_c_noise=False, dtype=torch.bfloat16, batch_samples=not a.sequential),
            discretization_config={"shift_scale": 1.0}, guider_config={"scale": 6, "exp": 5, "num_steps": n_steps},
            device=dev, num_steps=n_steps)
        torch.manual_seed(99)                                # the sampler's noise draws must agree across ranks
        g = torch.Generator(device=dev).manual_seed(7)
        noise = torch.randn(1, frames, 16, 60, 90, device=dev, generator=g)
        text = torch.randn(1, scenes, text_len, cfg.text_dim, device=dev, generator=g).bfloat16()
        neg = torch.randn(1, scenes, text_len, cfg.text_dim, device=dev, generator=g).bfloat16()
        with torch.no_grad():
            out = sampler(noise, {"crossattn": text}, {"crossattn": neg})
        torch.cuda.synchronize()
        return out

    if not a.no_warmup:
        run(1)                                               # warm-up: GEMM selection, allocator
    t0 = time.perf_counter()
    out = run(a.steps)
    dt = (time.perf_counter() - t0) / a.steps
    assert torch.isfinite(out).all()
    if rank != 0:
        return
    print(json.dumps({"metric": "sampling_denoising_step_seconds", "value": round(dt, 4), "unit": "s/step (cond+uncond)",
                      "latent_frames_per_s": round(frames / dt, 2), "projected_50_step_video_s": round(50 * dt, 1),
                      "config": {"workload": f"CogVideoX-5B+TTT-MLP sampling, {a.video_length}, CS=16, CFG pair "
                                             + ("sequential" if a.sequential else "batched")
                                             + (f", sequence-parallel over {int(os.environ.get('WORLD_SIZE', '1'))} ranks" if a.sequence_parallel else ""),
                                 "layers": cfg.num_layers, "timed_steps": a.steps, "warmup": not a.no_warmup, "tokens": L, "mini_batches": NC, "scan_impl": impl},
                      "dtype": "bf16", "data": "synthetic", "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == "__main__":
    main()
