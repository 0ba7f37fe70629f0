#!/usr/bin/env python
"""Round 5: the TTT-MLP layer (projections -> pre -> scan -> post-norm -> output projection, both scan directions) at the 5B / 9 s
geometry as one piece and as a pipeline over parts of the sequence (ttt_amd/models/ssm/pipeline.py); interleaved rounds in one
process, medians, forward alone and forward + backward; outputs / gradients compared.

    python tools/ttt_layer_bench.py [--parts 0,2,3,4] [--video-length 9sec] [--rounds 5]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="0,2,3,4")
    ap.add_argument("--video-length", default="9sec")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tuning-file", default=None, help="GEMM solution selections to load instead of the committed ttt_amd/infra/gemm_tuning_gfx950.csv")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE", help="library debug option(s) for the whole run (e.g. scan_pair=0)")
    a = ap.parse_args()
    import test_time_training as ext
    from bench import TEXT_LEN, TOKENS_PER_FRAME
    from ttt_amd.infra.parallelisms import enable_tuned_gemms
    from ttt_amd.models.cogvideo.utils import SequenceMetadata
    from ttt_amd.models.configs import ModelConfig
    from ttt_amd.models.ssm.ttt_layer import TTTWrapper
    ext.load_library()
    for kv in a.debug_option:
        ext.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    dev = torch.device("cuda:0")
    tuned = enable_tuned_gemms(a.tuning_file)
    cfg = ModelConfig.get_preset("5B", a.video_length, ssm_layer="ttt_mlp", adapter_method="qkvo")
    frames, tl = cfg.compressed_num_frames, TEXT_LEN[a.video_length]
    scenes = max((frames - 1) // 12, 1)
    n_vid = frames * TOKENS_PER_FRAME
    L = n_vid + scenes * tl
    torch.manual_seed(0)
    layer = TTTWrapper(cfg).to(dev).to(torch.bfloat16)
    layer.ttt.init_weights()
    layer.init_freqs()
    meta = SequenceMetadata(text_length=tl, seq_text_length=tl * scenes, num_frames=frames, num_chunks=scenes, tokens_per_frame=TOKENS_PER_FRAME,
                            latent_height=60, latent_width=90, t_emb=None)
    if meta.is_multiscene:
        meta.init_multiscene_offsets()
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(1, L, cfg.model_dim, device=dev, generator=g).bfloat16().requires_grad_(True)
    dy = torch.randn(1, L, cfg.model_dim, device=dev, generator=g).bfloat16() * 0.1
    params = [p for p in layer.parameters() if p.requires_grad]
    parts = [int(v) for v in a.parts.split(",")]
    res = {"L": L, "tuned_gemms": bool(tuned), "by_parts": {}}

    def fwd(n, reverse):
        layer.ttt.pipeline_parts = n
        with torch.no_grad():
            return layer(x, meta, reverse)

    def fwd_bwd(n, reverse):
        layer.ttt.pipeline_parts = n
        y = layer(x, meta, reverse)
        return y, torch.autograd.grad(y, [x] + params, dy)

    t = {n: {"fwd": [], "fwd_rev": [], "fwd_bwd": []} for n in parts}
    for _ in range(a.rounds):
        for n in parts:
            t[n]["fwd"].append(timeit(lambda: fwd(n, False)))
            t[n]["fwd_rev"].append(timeit(lambda: fwd(n, True)))
            t[n]["fwd_bwd"].append(timeit(lambda: fwd_bwd(n, False)))
    ref = {rev: fwd_bwd(parts[0], rev) for rev in (False, True)}
    rl2 = lambda p, q: float((p.double() - q.double()).norm() / q.double().norm().clamp_min(1e-30))
    for n in parts:
        med = {k: sorted(v)[len(v) // 2] for k, v in t[n].items()}
        ent = {"median_ms": med}
        if n != parts[0]:
            for rev in (False, True):
                y, gr = fwd_bwd(n, rev)
                y0, gr0 = ref[rev]
                ent["reverse" if rev else "forward"] = {"out_equal": bool(torch.equal(y, y0)), "out_rel_l2": rl2(y, y0),
                                                        "grads_equal": bool(all(torch.equal(p, q) for p, q in zip(gr, gr0))),
                                                        "grads_worst_rel_l2": max(rl2(p, q) for p, q in zip(gr, gr0))}
        res["by_parts"][str(n)] = ent
    print(json.dumps(res))


if __name__ == "__main__":
    main()
