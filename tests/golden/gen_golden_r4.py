"""Round-4 fixture, generated in the build container by EXECUTING the reference (needs /root/reference):

  tkmlp_replay.pt - the exact positional argument lists the reference's own autograd wrapper ``TkMLP``
      (ttt/models/ssm/mlp_tk.py:80-133 ``_forward_core``, :156-275 ``_backward_core``) hands ``test_time_training.ttt_forward``
      (15 tensors + G) and ``ttt_backward`` (42 tensors + G) for one small seeded case, recorded at the extension boundary
      (every tensor as passed: values, dtype, shape, contiguity; output / scratch buffers by dtype and shape only), together
      with what the reference's ops path (ttt/models/ssm/ops/ttt_mlp.py:9-99, fp32 arithmetic on the same bf16-rounded
      activations) returns for it: output and all ten gradients as ``TkMLP.backward`` returns them.
      tests/test_parity_r4_gpu.py replays the lists on an MI355X through the real ``test_time_training`` binding of
      libttt_hip.so: "train.py is drop-in" as an executed call, not as an argument-count comparison (round-3 verdict,
      missing #3).  (The wrapper's extension calls are served by the oracle-backed stand-in while recording; its results are
      NOT stored - the device produces its own checkpoints / XQW and feeds them to its own backward, as TkMLP's ctx would.)

Usage:  python tests/golden/gen_golden_r4.py
"""
import importlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")


def import_reference():
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    try:
        import tomllib  # noqa: F401
    except ImportError:
        import tomli
        sys.modules["tomllib"] = tomli
    sys.path.insert(0, REF)
    return importlib.import_module("ttt.models.ssm.mlp_tk"), importlib.import_module("ttt.models.ssm.ops")


def main():
    from oracle import cpu_ext, ttt_oracle as O
    fake = cpu_ext.install()
    rec = {}

    def recorder(name, fn, n_inputs):
        def call(*args):
            lst = []
            for i, a in enumerate(args):
                if isinstance(a, torch.Tensor):
                    assert a.is_contiguous(), (name, i)
                    # inputs with their values; output / scratch buffers by dtype and shape
                    if name == "backward" and 6 <= i <= 10:      # the forward's own outputs (TkMLP's ctx): the replay feeds the device's
                        lst.append({"role": "fwd_out", "index": 10 + (i - 6), "dtype": a.dtype, "shape": tuple(a.shape)})
                    elif i < n_inputs[0] or n_inputs[1] <= i < n_inputs[2]:
                        lst.append({"role": "in", "value": a.detach().clone()})
                    else:
                        lst.append({"role": "out", "dtype": a.dtype, "shape": tuple(a.shape)})
                else:
                    lst.append({"role": "int", "value": int(a)})
            rec[name] = lst
            return fn(*args)
        return call

    # ttt_forward: 10 inputs, then 4 checkpoint buffers + XQW (outputs), G
    fake.ttt_forward = recorder("forward", fake.ttt_forward, (10, 0, 0))
    # ttt_backward: 11 inputs (6 + 4 checkpoints + XQW), 16 remat scratch buffers, 5 upstream gradients (inputs), 10 outputs, G
    fake.ttt_backward = recorder("backward", fake.ttt_backward, (11, 27, 32))
    mlp_tk, ops = import_reference()

    B, NH, NC, CS, F, G = 1, 2, 5, 64, 64, 2
    d = O.make_inputs("mlp", B, NH, NC, CS, F, seed=777)
    bf = lambda t: t.to(torch.bfloat16)
    tile = lambda w: torch.tile(w.unsqueeze(0), dims=(B, 1, 1, 1))
    leaves = {k: d[k].clone().requires_grad_(True) for k in ("ln_w", "ln_b", "W1", "b1", "W2", "b2", "XQ", "XV", "XK", "eta")}

    def run(fn, cast):
        for v in leaves.values():
            v.grad = None
        a = {k: (cast(v) if k in ("XQ", "XV", "XK", "eta") else v) for k, v in leaves.items()}
        out = fn(a, [tile(a[k]) for k in ("W1", "b1", "W2", "b2")])
        out.float().backward(d["dOut"])
        return out.detach().float(), {k: v.grad.detach().clone() for k, v in leaves.items()}

    out_k, g_k = run(lambda a, st: mlp_tk.TkMLP.apply(a["ln_w"], a["ln_b"], *st, a["XQ"], a["XV"], a["XK"], a["eta"], G), bf)
    out_o, g_o = run(lambda a, st: ops.ttt_mlp(a["XK"], a["XQ"], a["XV"], a["eta"], a["ln_w"], a["ln_b"], *st, G).permute(0, 3, 1, 2, 4),
                     lambda t: bf(t).float())
    assert len(rec["forward"]) == 16 and len(rec["backward"]) == 43
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print("reference TkMLP (on the stand-in) vs reference ops path: out", rel(out_k, out_o),
          {k: round(rel(g_k[k].sum(-2) if k == "eta" else g_k[k], g_o[k].sum(-2) if k == "eta" else g_o[k]), 5) for k in g_o})
    g_o["eta_last_row_sum"] = g_o.pop("eta").sum(-2)          # kernels put the whole eta gradient into the last row (mlp_tk.py:280)
    torch.save({"dims": dict(B=B, NH=NH, NC=NC, CS=CS, F=F, G=G), "forward_args": rec["forward"], "backward_args": rec["backward"],
                "ops_out": out_o, "ops_grads": g_o,
                "note": "ops_grads are w.r.t. the TkMLP.apply arguments (ln_w, ln_b [NH,F]; W1.. [NH,..] untiled); XQ/XV/XK/eta kernel layout"},
               os.path.join(HERE, "tkmlp_replay.pt"))
    print("wrote tkmlp_replay.pt", os.path.getsize(os.path.join(HERE, "tkmlp_replay.pt")), "bytes")


if __name__ == "__main__":
    main()
