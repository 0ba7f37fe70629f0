"""Round-6 golden vector, produced by EXECUTING THE REFERENCE (build container only):

    TORCHDYNAMO_DISABLE=1 python tests/golden/gen_golden_r6.py

``dit_mlp64_3scene_long_lastrow.pt`` - the reference's own 2-layer DiffusionTransformer (TTT-MLP, mini-batches of 64, checkpoint
groups of 2) on a 3-scene sample LONG ENOUGH for the pipelined TTT layer forward that ``bench.py`` times
(``ttt_amd/models/ssm/pipeline.py``: four parts need at least eight checkpoint groups): 7 latent frames of 8 x 12 = 96 tokens + 3 x 96 text
tokens = 960 tokens = 15 mini-batches = 8 checkpoint groups; scene lengths 384 / 288 / 288, so the third scene starts at token 672, not
a multiple of 64, and the eta rows of a tile differ (hazard C2) - every eta tile is replaced by its LAST ROW before the op sees it,
exactly what ``TkMLP`` hands to ttt-tk (``gen_golden_r2.kernel_contract``).  Stored: the output and every parameter gradient of the
reference's fp32 run (what ttt_layer.py:314-334 computes inside dit.py:224-266).  Only numbers are saved.
"""
import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden_r2 as G2  # noqa: E402  (sets up the import stubs and sys.path for /root/reference)
import torch  # noqa: E402

if __name__ == "__main__":
    case = G2.dit_mlp64_multiscene_lastrow_case(seed=61, latent=(16, 24), frames=7, text_len=96)
    L = 7 * 96 + 3 * 96
    assert L % 64 == 0 and -(-(L // 64) // 2) >= 8, L
    case["tokens"], case["mini_batches"], case["checkpoint_groups"] = L, L // 64, -(-(L // 64) // 2)
    path = os.path.join(HERE, "dit_mlp64_3scene_long_lastrow.pt")
    torch.save(case, path)
    print("wrote", path, os.path.getsize(path), "bytes;", len(case["grads"]), "gradients; out", tuple(case["out"].shape))
    # the reference's own bf16-autocast run of this model against its fp32 run (the yardstick of the bf16 tolerances)
    yard = G2.dit_bf16_yardstick(["dit_mlp64_3scene_long_lastrow.pt"], lastrow=True)
    torch.save(yard, os.path.join(HERE, "dit_bf16_yardstick_r6.pt"))
    print("wrote dit_bf16_yardstick_r6.pt")
