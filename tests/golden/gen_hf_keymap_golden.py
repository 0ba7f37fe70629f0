"""Golden key map for the HuggingFace weight conversion, made by EXECUTING THE REFERENCE's own conversion loop
(/root/reference/ttt/models/cogvideo/weight_conversion/from_hf.py:13-143) on synthetic diffusers-style keys with file I/O,
model construction and DCP export stubbed out (build container only):

    python tests/golden/gen_hf_keymap_golden.py

Writes hf_keymap.json = {diffusers key: reference state-dict key or null}.  Only key strings are saved.
"""
import contextlib
import importlib.util
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules["wandb"] = types.ModuleType("wandb")
import tomli  # noqa: E402

sys.modules["tomllib"] = tomli
sys.path.insert(0, "/root/reference")

top = ["patch_embed.proj", "patch_embed.text_proj", "norm_final", "norm_out.norm", "norm_out.linear", "proj_out",
       "time_embedding.linear_1", "time_embedding.linear_2"]
per_layer = ["attn1.norm_q", "attn1.norm_k", "attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "ff.net.0.proj", "ff.net.2",
             "norm1.linear", "norm1.norm", "norm2.linear", "norm2.norm"]
keys = [f"{t}.{s}" for t in top for s in ("weight", "bias")]
for n in (0, 1, 17, 41):
    keys += [f"transformer_blocks.{n}.{t}.{s}" for t in per_layer for s in ("weight", "bias")]
keys += ["patch_embed.pos_embedding", "some.unknown.weight"]        # no counterpart: must be skipped
# every synthetic tensor carries its own index so the captured mapping can be read back from the values
tensors = {k: torch.full((1,), float(i)) for i, k in enumerate(keys)}

spec = importlib.util.spec_from_file_location("ref_from_hf", "/root/reference/ttt/models/cogvideo/weight_conversion/from_hf.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

captured = {}


class FakeFile:
    def __init__(self, ks):
        self.ks = ks

    def keys(self):
        return self.ks

    def get_tensor(self, k):
        return tensors[k]


@contextlib.contextmanager
def fake_safe_open(filename, framework="pt", device="cpu"):
    yield FakeFile(keys if "00001" in filename else [])


class FakeModel:
    def __init__(self, *a, **k):
        pass

    def to(self, *a, **k):
        return self

    def load_state_dict(self, sd, strict=True):
        captured.update(sd)

    def state_dict(self):
        return {}


ref.safe_open = fake_safe_open
ref.CogVideoX = FakeModel
ref.torch_save_to_dcp = lambda src, dst: None
ref.main(os.path.join("/tmp", "hf_golden_out"), "ttt_mlp", "/nonexistent")

by_index = {int(v.float().item()): k for k, v in captured.items()}
golden = {k: by_index.get(i) for i, k in enumerate(keys)}
json.dump(golden, open(os.path.join(HERE, "hf_keymap.json"), "w"), indent=1)
print(len(golden), "keys,", sum(v is None for v in golden.values()), "skipped")
