"""HuggingFace -> package weight-key conversion against the key map produced by executing the reference's
own conversion loop (tests/golden/gen_hf_keymap_golden.py -> hf_keymap.json)."""
import json
import os

import torch

from helpers import GOLDEN, load_golden
from ttt_amd.models.cogvideo.model import CogVideoX
from ttt_amd.models.cogvideo.weight_conversion.from_hf import convert_state_dict, load_hf_weights, map_key
from ttt_amd.models.configs import ModelConfig


def _golden():
    return json.load(open(os.path.join(GOLDEN, "hf_keymap.json")))


def test_key_map_equals_reference():
    gold = _golden()
    assert sum(v is None for v in gold.values()) == 2
    for hf_key, ref_key in gold.items():
        assert map_key(hf_key) == ref_key, hf_key


def test_mapped_keys_exist_in_the_5b_model():
    with torch.device("meta"):
        model = CogVideoX(ModelConfig.get_preset("5B", "3sec", ssm_layer="ttt_mlp", adapter_method="sft"))
    have = set(model.state_dict())
    for ref_key in _golden().values():
        assert ref_key is None or ref_key in have, ref_key
    # every non-TTT, non-gate parameter of a layer is covered by the table: nothing pretrained is left behind
    covered = {v.replace(".layers.17.", ".layers.N.") for v in _golden().values() if v and ".layers.17." in v}
    layer = {k.replace(".layers.17.", ".layers.N.") for k in have if ".layers.17." in k}
    left = {k for k in layer - covered if ".ssm." not in k and "gating" not in k and "gate" not in k.split(".")[-2]}
    assert not left, left


def test_load_into_a_small_model():
    g = load_golden("dit_mlp_3scene.pt")
    cfg = ModelConfig(**{**g["cfg"], "num_layers": 2})
    model = CogVideoX(cfg)
    sd = model.state_dict()
    # invert the table over this model's keys to fabricate a diffusers-style checkpoint with the right shapes
    inverse = {}
    names = {"dit.patch_embedding.vid_proj": "patch_embed.proj", "dit.patch_embedding.text_proj": "patch_embed.text_proj",
             "dit.transformer_norm": "norm_final", "dit.final_layer.norm": "norm_out.norm",
             "dit.final_layer.adaLN_modulation.1": "norm_out.linear", "dit.final_layer.linear": "proj_out",
             "dit.time_embed.0": "time_embedding.linear_1", "dit.time_embed.2": "time_embedding.linear_2"}
    per_layer = {"seq_modeling_block.q_norm": "attn1.norm_q", "seq_modeling_block.k_norm": "attn1.norm_k",
                 "seq_modeling_block.q": "attn1.to_q", "seq_modeling_block.k": "attn1.to_k", "seq_modeling_block.v": "attn1.to_v",
                 "seq_modeling_block.o": "attn1.to_out.0", "mlp.layer1": "ff.net.0.proj", "mlp.layer2": "ff.net.2",
                 "pre_seq_adaLN_modulation.1": "norm1.linear", "pre_seq_layernorm": "norm1.norm",
                 "pre_mlp_adaLN_modulation.1": "norm2.linear", "pre_mlp_layernorm": "norm2.norm"}
    gen = torch.Generator().manual_seed(0)
    hf = {}
    for k, v in sd.items():
        stem, leaf = k.rsplit(".", 1)
        if stem in names:
            hf_key = f"{names[stem]}.{leaf}"
        elif stem.startswith("dit.layers."):
            n, rest = stem[len("dit.layers."):].split(".", 1)
            if rest not in per_layer:
                continue
            hf_key = f"transformer_blocks.{n}.{per_layer[rest]}.{leaf}"
        else:
            continue
        hf[hf_key] = torch.randn(v.shape, generator=gen)
        inverse[hf_key] = k
    hf["patch_embed.pos_embedding"] = torch.zeros(3)
    state, skipped = convert_state_dict(hf, dtype=torch.float32)
    assert skipped == ["patch_embed.pos_embedding"]
    assert {inverse[k] for k in hf if k in inverse} == set(state)
    kept = load_hf_weights(model, hf, dtype=torch.float32)
    new = model.state_dict()
    for hf_key, k in inverse.items():
        assert torch.equal(new[k], hf[hf_key])
    assert kept and all(".ssm." in k or "gat" in k for k in kept), [k for k in kept if ".ssm." not in k and "gat" not in k]
