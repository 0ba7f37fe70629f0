#!/usr/bin/env python
"""What ONE remat-free TransformerLayer of CogVideoX-5B + TTT-MLP keeps for its backward.

    python tools/saved_tensor_audit.py [--video-length 9sec] [--adapter qkvo] > profiles/r6*_saved_tensor_audit.json

A one-layer DiT at the real width and sequence length runs ONE forward under `torch.autograd.graph.saved_tensors_hooks`; every tensor an
autograd node saves is recorded with its bytes, its STORAGE (two nodes saving views of one buffer pay once) and the innermost frame
of this repository on the call stack (which node saved it).  Reported: the list by storage, largest first, with how many nodes share
each storage, the total (= what a remat-free layer costs), and the total one would expect from bench.py's sizing (peak memory per
free layer).  Parameters are excluded (they are resident anyway).
"""
import argparse
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ttt-video-dit_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--video-length", default="9sec")
    ap.add_argument("--adapter", default="qkvo")
    ap.add_argument("--pipeline-parts", type=int, default=None)
    a = ap.parse_args()
    import test_time_training as ext
    from bench import TEXT_LEN, TOKENS_PER_FRAME
    from ttt_amd.infra.parallelisms import enable_tuned_gemms, init_model_parameters
    from ttt_amd.models.cogvideo.dit import DiffusionTransformer
    from ttt_amd.models.configs import ModelConfig
    ext.load_library()
    enable_tuned_gemms()
    dev = torch.device("cuda:0")
    cfg = ModelConfig.get_preset("5B", a.video_length, ssm_layer="ttt_mlp", adapter_method=a.adapter, num_layers=2, remat_free_layers=2)      # the SECOND layer is audited: its input needs a gradient
    frames, tl = cfg.compressed_num_frames, TEXT_LEN[a.video_length]
    scenes = max((frames - 1) // 12, 1)
    L = frames * TOKENS_PER_FRAME + scenes * tl
    with torch.device("meta"):
        m = DiffusionTransformer(cfg)
    m.to_empty(device=dev)
    torch.manual_seed(0)
    with torch.no_grad():
        init_model_parameters(m)
        for layer in m.layers:
            layer.seq_modeling_block.ssm.ttt.init_weights()
    m = m.to(torch.bfloat16)
    for mod in m.modules():
        if hasattr(mod, "init_freqs"):
            mod.init_freqs()
        if a.pipeline_parts is not None and hasattr(mod, "pipeline_parts"):
            mod.pipeline_parts, mod.pipeline_parts_auto = a.pipeline_parts, False
    params = {p.untyped_storage().data_ptr() for p in m.parameters()} | {b.untyped_storage().data_ptr() for b in m.buffers()}
    g = torch.Generator(device=dev).manual_seed(1)
    vid = torch.randn(1, frames, 16, 60, 90, device=dev, generator=g).bfloat16()
    text = torch.randn(1, scenes, tl, cfg.text_dim, device=dev, generator=g).bfloat16()
    ts = torch.tensor([417], device=dev)

    layer = m.layers[1]
    records, active = [], [False]

    def where():
        for fr in reversed(traceback.extract_stack()):
            if "ttt_amd" in fr.filename and "saved_tensor_audit" not in fr.filename:
                return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
        return "?"

    def pack(t):
        if active[0] and isinstance(t, torch.Tensor) and t.is_cuda and t.untyped_storage().data_ptr() not in params:
            records.append({"shape": list(t.shape), "dtype": str(t.dtype).replace("torch.", ""), "bytes": t.numel() * t.element_size(),
                            "storage": t.untyped_storage().data_ptr(), "storage_bytes": t.untyped_storage().nbytes(), "where": where()})
        return t

    orig_forward = layer.forward

    def audited(*args, **kw):
        active[0] = True
        try:
            return orig_forward(*args, **kw)
        finally:
            active[0] = False

    layer.forward = audited
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        out = m(vid, text, ts)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - base
    by_storage = {}
    for r in records:
        e = by_storage.setdefault(r["storage"], {"bytes": 0, "storage_bytes": r["storage_bytes"], "saved_by": [], "views": []})
        e["bytes"] = max(e["bytes"], r["bytes"])
        e["saved_by"].append(r["where"])
        e["views"].append(f"{r['dtype']}{r['shape']}")
    rows = sorted(by_storage.values(), key=lambda e: -e["storage_bytes"])
    unit = L * cfg.model_dim * 2                                   # one [L, D] bf16 tensor
    total = sum(e["storage_bytes"] for e in rows)
    naive = sum(r["bytes"] for r in records)
    res = {"workload": f"one TransformerLayer, 5B width, {a.video_length} (L = {L}), adapter {a.adapter}", "L": L, "unit_LD_bf16_bytes": unit,
           "saved_tensors": len(records), "distinct_storages": len(rows), "bytes_by_storage": total, "GiB_by_storage": round(total / 2 ** 30, 3),
           "units_LD_bf16": round(total / unit, 2), "bytes_if_every_save_were_a_copy": naive,
           "allocated_after_forward_minus_before_GiB": round(held / 2 ** 30, 3),
           "storages": [{"GiB": round(e["storage_bytes"] / 2 ** 30, 4), "units_LD_bf16": round(e["storage_bytes"] / unit, 3), "n_saves": len(e["saved_by"]),
                         "views": sorted(set(e["views"]))[:4], "saved_by": sorted(set(e["saved_by"]))[:6]} for e in rows if e["storage_bytes"] >= 1 << 20]}
    del out
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
