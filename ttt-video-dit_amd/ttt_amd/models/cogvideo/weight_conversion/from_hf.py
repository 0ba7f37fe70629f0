"""CogVideoX-5B HuggingFace (diffusers) checkpoint -> this package's state-dict keys (reference
``ttt/models/cogvideo/weight_conversion/from_hf.py``:13-143).

The reference walks an if/elif chain of substring tests per key; here the same correspondence is a table of
(diffusers suffix -> module path) rules, applied to the top-level tensors and, with the layer index substituted, to
``transformer_blocks.<n>.*``.  Keys the table does not know (e.g. diffusers' positional-embedding buffers) are skipped,
as in the reference, and reported.  TTT parameters have no pretrained counterpart: they keep their initialisation
(``strict=False`` load, reference :127).

    python -m ttt_amd.models.cogvideo.weight_conversion.from_hf --pretrained_weights_dir D --ssm_type ttt_mlp --final_save_path OUT
"""
from __future__ import annotations

import argparse
import os
import re
from typing import Dict, List, Tuple

import torch

# diffusers name (without .weight / .bias) -> our module path under ``dit.``
_TOP_LEVEL = {
    "patch_embed.proj": "patch_embedding.vid_proj",
    "patch_embed.text_proj": "patch_embedding.text_proj",
    "norm_final": "transformer_norm",
    "norm_out.norm": "final_layer.norm",
    "norm_out.linear": "final_layer.adaLN_modulation.1",
    "proj_out": "final_layer.linear",
    "time_embedding.linear_1": "time_embed.0",
    "time_embedding.linear_2": "time_embed.2",
}
# inside ``transformer_blocks.<n>.`` -> inside ``dit.layers.<n>.``
_PER_LAYER = {
    "attn1.norm_q": "seq_modeling_block.q_norm",
    "attn1.norm_k": "seq_modeling_block.k_norm",
    "attn1.to_q": "seq_modeling_block.q",
    "attn1.to_k": "seq_modeling_block.k",
    "attn1.to_v": "seq_modeling_block.v",
    "attn1.to_out.0": "seq_modeling_block.o",
    "ff.net.0.proj": "mlp.layer1",
    "ff.net.2": "mlp.layer2",
    "norm1.linear": "pre_seq_adaLN_modulation.1",
    "norm1.norm": "pre_seq_layernorm",
    "norm2.linear": "pre_mlp_adaLN_modulation.1",
    "norm2.norm": "pre_mlp_layernorm",
}
_BLOCK = re.compile(r"^transformer_blocks\.(\d+)\.(.+)\.(weight|bias)$")
_TOP = re.compile(r"^(.+)\.(weight|bias)$")


def map_key(hf_key: str) -> str | None:
    """Our state-dict key for one diffusers key, or None if the tensor has no counterpart."""
    m = _BLOCK.match(hf_key)
    if m:
        target = _PER_LAYER.get(m.group(2))
        return None if target is None else f"dit.layers.{int(m.group(1))}.{target}.{m.group(3)}"
    m = _TOP.match(hf_key)
    if m:
        target = _TOP_LEVEL.get(m.group(1))
        return None if target is None else f"dit.{target}.{m.group(2)}"
    return None


def convert_state_dict(hf_tensors: Dict[str, torch.Tensor], dtype=torch.bfloat16) -> Tuple[Dict[str, torch.Tensor], List[str]]:
    """(converted state dict, skipped diffusers keys)."""
    out, skipped = {}, []
    for key, tensor in hf_tensors.items():
        target = map_key(key)
        if target is None:
            skipped.append(key)
        else:
            out[target] = tensor.to(dtype)
    return out, skipped


def load_hf_weights(model: torch.nn.Module, hf_tensors: Dict[str, torch.Tensor], dtype=torch.bfloat16) -> List[str]:
    """Load the pretrained DiT tensors into a ``CogVideoX`` model; returns the model keys that stayed at their
    initialisation (the TTT parameters, gates and anything else without a pretrained counterpart)."""
    state, _ = convert_state_dict(hf_tensors, dtype)
    unexpected = sorted(set(state) - set(model.state_dict()))
    if unexpected:
        raise KeyError(f"converted keys missing from the model: {unexpected[:5]} ...")
    missing, _ = model.load_state_dict(state, strict=False)
    return list(missing)


def read_safetensors(directory: str) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    tensors = {}
    for name in sorted(os.listdir(directory)):
        if name.startswith("diffusion_pytorch_model") and name.endswith(".safetensors"):
            with safe_open(os.path.join(directory, name), framework="pt", device="cpu") as f:
                for key in f.keys():
                    tensors[key] = f.get_tensor(key)
    if not tensors:
        raise FileNotFoundError(f"no diffusion_pytorch_model*.safetensors under {directory}")
    return tensors


def main(final_save_path: str, ssm_layer: str, path_to_weights: str) -> None:
    from ttt_amd.models.cogvideo.model import CogVideoX
    from ttt_amd.models.configs import ModelConfig

    cfg = ModelConfig.get_preset("5B", "3sec", ssm_layer=ssm_layer, adapter_method="sft")
    model = CogVideoX(cfg, 0, 1).to(torch.bfloat16)
    kept = load_hf_weights(model, read_safetensors(path_to_weights))
    print(f"{len(kept)} parameters keep their initialisation (TTT layers, gates)")
    os.makedirs(final_save_path, exist_ok=True)
    import torch.distributed.checkpoint as dcp
    dcp.save(model.state_dict(), checkpoint_id=final_save_path, no_dist=True)     # same DCP layout torch_save_to_dcp writes


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Convert CogVideoX HuggingFace safetensors to a DCP checkpoint of this package.")
    ap.add_argument("--final_save_path", required=True)
    ap.add_argument("--ssm_type", required=True)
    ap.add_argument("--pretrained_weights_dir", required=True)
    a = ap.parse_args()
    main(a.final_save_path, a.ssm_type, a.pretrained_weights_dir)
