"""The REFERENCE's own autograd wrapper on top of this repo's extension interface (build container only: needs the mounted
reference checkout, skipped elsewhere).

``ttt/models/ssm/mlp_tk.py`` does ``import test_time_training`` and calls ``ttt_forward`` with 15 tensors + G
(mlp_tk.py:116-133) and ``ttt_backward`` with 42 tensors + G (mlp_tk.py:227-275).  Here that very code - ``TkMLP.apply`` - runs
against the CPU stand-in of the extension (oracle/cpu_ext.py: the same positional-buffer contract as the ctypes binding,
backed by the oracle; TEST INFRASTRUCTURE), and is compared with the reference's ops path (``ops/ttt_mlp.py``) on the same
inputs: the argument lists line up, the buffers the wrapper allocates have the shapes the interface documents, outputs and
all ten gradients agree within the bf16 op-boundary tolerance.  The real binding's signatures
(``ttt-video-dit_amd/test_time_training/__init__.py``) are checked for the same arity, so what passes here is what the
reference would call on the GPU box.
"""
import importlib
import inspect
import os
import sys
import types

import pytest
import torch

from helpers import rel_l2
from oracle import cpu_ext as fake_ext
from oracle import ttt_oracle as O

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ttt")), reason="reference checkout not mounted")


@pytest.fixture
def reference_modules():
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    try:
        import tomllib  # noqa: F401
    except ImportError:
        import tomli
        sys.modules["tomllib"] = tomli
    sys.path.insert(0, REF)
    fake_ext.install()
    try:
        mlp_tk = importlib.import_module("ttt.models.ssm.mlp_tk")
        ops = importlib.import_module("ttt.models.ssm.ops")
        yield mlp_tk, ops
    finally:
        fake_ext.uninstall()
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k == "ttt" or k.startswith("ttt."):
                del sys.modules[k]
        for k in ("wandb", "tomllib"):
            if k in saved_mods:
                sys.modules[k] = saved_mods[k]
            else:
                sys.modules.pop(k, None)


def test_real_binding_has_the_reference_arity():
    """16 / 43 positional parameters: what mlp_tk.py:116-133 / 227-275 pass (the Linear pair: 12 / 22)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_ttt_binding_sig", os.path.join(root, "ttt-video-dit_amd", "test_time_training", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                     # importing does not load the library
    n = lambda f: len(inspect.signature(f).parameters)
    assert (n(mod.ttt_forward), n(mod.ttt_backward), n(mod.ttt_linear_forward), n(mod.ttt_linear_backward)) == (16, 43, 12, 22)


@pytest.mark.parametrize("B,NH,NC,G", [(1, 2, 5, 2), (2, 3, 4, 4)])
def test_reference_tkmlp_apply_on_this_interface(reference_modules, B, NH, NC, G):
    mlp_tk, ops = reference_modules
    CS, F = 64, 64
    d = O.make_inputs("mlp", B, NH, NC, CS, F, seed=4242 + NC)
    bf = lambda t: t.to(torch.bfloat16)
    tile = lambda w: torch.tile(w.unsqueeze(0), dims=(B, 1, 1, 1))
    leaves = {k: d[k].clone().requires_grad_(True) for k in ("ln_w", "ln_b", "W1", "b1", "W2", "b2", "XQ", "XV", "XK", "eta")}

    def run(fn, cast):
        for v in leaves.values():
            v.grad = None
        a = {k: (cast(v) if k in ("XQ", "XV", "XK", "eta") else v) for k, v in leaves.items()}
        st = [tile(a[k]) for k in ("W1", "b1", "W2", "b2")]
        out = fn(a, st)
        out.float().backward(d["dOut"])
        return out.detach().float(), {k: v.grad.detach().clone() for k, v in leaves.items()}

    # the reference wrapper: returns the kernel layout [B,NH,NC,CS,F]
    out_k, g_k = run(lambda a, st: mlp_tk.TkMLP.apply(a["ln_w"], a["ln_b"], *st, a["XQ"], a["XV"], a["XK"], a["eta"], G), bf)
    # the reference ops path on the bf16-rounded activations, fp32 arithmetic: [B,NC,CS,NH,F] -> kernel layout
    out_o, g_o = run(lambda a, st: ops.ttt_mlp(a["XK"], a["XQ"], a["XV"], a["eta"], a["ln_w"], a["ln_b"], *st, G).permute(0, 3, 1, 2, 4),
                     lambda t: bf(t).float())
    assert out_k.shape == (B, NH, NC, CS, F)
    assert rel_l2(out_k, out_o) < 1e-2
    errs = {k: rel_l2(g_k[k].sum(-2) if k == "eta" else g_k[k], g_o[k].sum(-2) if k == "eta" else g_o[k]) for k in g_o}
    bad = {k: v for k, v in errs.items() if not v < 3e-2}
    assert not bad, (bad, errs)
