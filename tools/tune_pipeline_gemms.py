#!/usr/bin/env python
"""TunableOp search for the GEMM shapes of the pipelined TTT layer forward (ttt_amd/models/ssm/pipeline.py): the projections of a
PART of the sequence are `addmm(bias, x[r0:r1], W^T)` over the token runs the part covers, i.e. the 3072 x 3072 projections at the
row counts of the plan.  Writes a TunableOp file; merge its new lines into ttt_amd/infra/gemm_tuning_gfx950.csv.

    python tools/tune_pipeline_gemms.py OUT.csv [--video-length 9sec,3sec] [--parts 4,6]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def row_counts(video_length, n_parts):
    from bench import TEXT_LEN, TOKENS_PER_FRAME
    from ttt_amd.models.cogvideo.utils import SequenceMetadata
    from ttt_amd.models.configs import ModelConfig
    from ttt_amd.models.ssm.pipeline import plan_parts
    from ttt_amd.models.ssm.ttt_layer import reversal_map, scene_permutation
    cfg = ModelConfig.get_preset("5B", video_length, ssm_layer="ttt_mlp", adapter_method="qkvo")
    frames, tl = cfg.compressed_num_frames, TEXT_LEN[video_length]
    scenes = max((frames - 1) // 12, 1)
    L = frames * TOKENS_PER_FRAME + scenes * tl
    meta = SequenceMetadata(text_length=tl, seq_text_length=tl * scenes, num_frames=frames, num_chunks=scenes, tokens_per_frame=TOKENS_PER_FRAME,
                            latent_height=60, latent_width=90, t_emb=None)
    if meta.is_multiscene:
        meta.init_multiscene_offsets()
    ms = set()
    for rev in (False, True):
        seq = scene_permutation(meta, L) if meta.is_multiscene else None
        if rev:
            r = reversal_map(meta, L)
            seq = r if seq is None else r[seq]
        for _, _, runs in plan_parts(seq, L, cfg.mini_batch_size, min(cfg.scan_checkpoint_group_size, L // cfg.mini_batch_size), n_parts):
            ms.update(r1 - r0 for r0, r1 in runs)
    return sorted(ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--video-length", default="9sec")
    ap.add_argument("--parts", default="4")
    ap.add_argument("--min-rows", type=int, default=0, help="skip the row counts below this (the text runs)")
    ap.add_argument("--occupy", type=int, default=0, metavar="CUS", help="tune BESIDE a kernel that holds this many CUs (round 6: the pair scan takes 96; "
                    "a stream-K selection made on the free chip runs its 256 workgroups in two waves on the 160 CUs that are left)")
    a = ap.parse_args()
    from torch.cuda import tunable
    from ttt_amd.infra.parallelisms import enable_tuned_gemms
    if not a.occupy:           # (a search beside a CU holder starts from nothing: shapes already in the committed file would not be searched again)
        enable_tuned_gemms()
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(30)
    tunable.set_filename(os.path.abspath(a.out))
    dev = torch.device("cuda:0")
    D = 3072
    w = (torch.randn(D, D, device=dev) * 0.02).bfloat16()
    b = torch.zeros(D, device=dev, dtype=torch.bfloat16)
    done = set()
    for vl in a.video_length.split(","):
        for n in (int(v) for v in a.parts.split(",")):
            for m in row_counts(vl, n):
                if m in done or m < a.min_rows:
                    continue
                done.add(m)
                x = torch.randn(m, D, device=dev).bfloat16()
                out = torch.empty(m, D, device=dev, dtype=torch.bfloat16)
                if a.occupy:           # 35 s of CU holders on a side stream (100 ms each), the search runs beside them
                    import test_time_training as ext
                    ext.load_library()
                    side = torch.cuda.Stream()
                    torch.cuda.synchronize()
                    for _ in range(350):
                        ext.debug_occupy_cus(a.occupy, 150 * 1024, 100000, stream=side)
                torch.addmm(b, x, w.t(), out=out)
                torch.cuda.synchronize()
                print("tuned rows", m, flush=True)
    tunable.tuning_enable(False)


if __name__ == "__main__":
    main()
