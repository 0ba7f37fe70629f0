"""Host-side mirror vs golden vectors produced by the reference's own modules (CPU, fp32).

Two modes are checked against the same goldens (generated with the reference's use_kernel=False):
  * ``use_kernel=False``  - our dual-form PyTorch path: must match in every case;
  * ``use_kernel=True``   - the kernel plumbing (TkMLP/HipLinear wrappers, last-row eta), with the
    HIP extension replaced by the oracle-backed stand-in of tests/fake_ext.py: must match in the
    single-scene cases (rows of eta identical => primal == dual).  In multi-scene cases the kernel
    contract differs from the dual form by design (SURVEY.md hazard C2) - asserted too.
"""
import pytest
import torch

from oracle import cpu_ext as fake_ext
from helpers import load_golden, rel_l2
from ttt_amd.models.cogvideo.dit import DiffusionTransformer
from ttt_amd.models.cogvideo.utils import SequenceMetadata
from ttt_amd.models.configs import ModelConfig
from ttt_amd.models.ssm.ttt_layer import TTTWrapper, scene_permutation


@pytest.fixture
def fake_extension():
    fake_ext.install()
    yield
    fake_ext.uninstall()


def _build_wrapper(g, use_kernel, dtype=torch.float32):
    cfg = ModelConfig(**g["cfg"])
    m = TTTWrapper(cfg).to(dtype)
    missing, unexpected = m.load_state_dict(g["state_dict"], strict=True)
    m.ttt.use_kernel = use_kernel
    meta = SequenceMetadata(t_emb=torch.zeros(1, 512), **g["meta"])
    if meta.is_multiscene:
        meta.init_multiscene_offsets()
    return m, meta


def _run_wrapper(m, meta, g):
    x = g["x"].clone().requires_grad_(True)
    y = m(x, meta)
    y.backward(g["dy"])
    return y.detach(), x.grad, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name", ["mod_mlp_cfg1.pt", "mod_lin_cfg1.pt", "mod_mlp_multi.pt", "mod_lin_multi.pt"])
def test_ttt_wrapper_dual_form_matches_reference(name):
    g = load_golden(name)
    m, meta = _build_wrapper(g, use_kernel=False)
    y, dx, grads = _run_wrapper(m, meta, g)
    assert rel_l2(y, g["y"]) < 2e-5
    assert rel_l2(dx, g["dx"]) < 2e-4
    assert set(grads) == set(g["grads"])
    for k, ref in g["grads"].items():
        assert rel_l2(grads[k], ref) < 5e-4, k


@pytest.mark.parametrize("name", ["mod_mlp_cfg1.pt", "mod_lin_cfg1.pt"])
def test_ttt_wrapper_kernel_plumbing_single_scene(name, fake_extension):
    g = load_golden(name)
    # TkMLP requires bf16 activations (mlp_tk.py:89); run the module in bf16 for MLP, fp32 for linear
    bf16 = g["ssm_layer"] == "ttt_mlp"
    m, meta = _build_wrapper(g, use_kernel=True)
    x = g["x"].clone().requires_grad_(True)
    if bf16:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = m(x, meta)
        tol_y, tol_g = 2e-2, 6e-2
    else:
        y = m(x, meta)
        tol_y, tol_g = 2e-5, 5e-4
    y.float().backward(g["dy"])
    assert rel_l2(y.float(), g["y"]) < tol_y
    assert rel_l2(x.grad, g["dx"]) < tol_g
    for k, ref in g["grads"].items():
        p = dict(m.named_parameters())[k]
        assert p.grad is not None, k
        assert rel_l2(p.grad, ref) < tol_g, k


def test_kernel_contract_differs_in_multiscene(fake_extension):
    g = load_golden("mod_lin_multi.pt")
    m, meta = _build_wrapper(g, use_kernel=True)
    y = m(g["x"], meta)
    assert rel_l2(y, g["y"]) > 1e-3   # last-row eta != dual form once interleave permutes rows


def test_scene_permutation_roundtrip():
    meta = SequenceMetadata(text_length=16, seq_text_length=48, num_frames=7, num_chunks=3, tokens_per_frame=16,
                            latent_height=4, latent_width=4, t_emb=torch.zeros(1))
    meta.init_multiscene_offsets()
    assert (meta.base_offset, meta.init_offset) == (48, 64)
    p = scene_permutation(meta, 160)
    assert sorted(p.tolist()) == list(range(160))
    # scene 0: text 0..15 then 3 frames (48 tokens) ; scene 1: text 16..31 then 2 frames
    assert p[:16].tolist() == list(range(16)) and p[16] == 48 and p[64] == 16 and p[80] == 96


@pytest.mark.parametrize("name", ["dit_mlp_3scene.pt", "dit_lin_1scene.pt"])
def test_dit_matches_reference(name):
    g = load_golden(name)
    cfg = ModelConfig(**g["cfg"])
    m = DiffusionTransformer(cfg)
    m.load_state_dict(g["state_dict"], strict=True)
    for mod in m.modules():
        if hasattr(mod, "use_kernel"):
            mod.use_kernel = False
    out = m(g["video"], g["text"], g["timesteps"])
    assert rel_l2(out, g["out"]) < 5e-5
    out.backward(g["dout"])
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    for k, ref in g["grads"].items():
        assert k in grads, k
        assert rel_l2(grads[k], ref) < 2e-3, k


def test_state_dict_keys_5b_meta():
    with torch.device("meta"):
        m = DiffusionTransformer(ModelConfig.get_preset("5B", "3sec"))
    sd = m.state_dict()
    assert len(sd) == 1948
    assert sum(p.numel() for p in m.parameters()) == 7_230_178_848
    pre = "layers.0.seq_modeling_block.ssm.ttt."
    assert tuple(sd[pre + "W1"].shape) == (48, 64, 256) and tuple(sd[pre + "learnable_ttt_lr_weight"].shape) == (48, 1, 3072)
    assert tuple(sd[pre + "learnable_ttt_lr_bias"].shape) == (48, 1) and tuple(sd[pre + "ttt_norm_weight"].shape) == (48, 64)


def test_hip_path_fails_loudly_without_gpu():
    """No silent CPU fallback: the kernel path on CPU tensors must raise."""
    fake_ext.uninstall()
    g = load_golden("mod_lin_cfg1.pt")
    m, meta = _build_wrapper(g, use_kernel=True)
    with pytest.raises(RuntimeError):
        m(g["x"], meta)


def test_remat_free_layers_changes_memory_policy_only():
    """Keeping activations for the first layers (remat_free_layers, the 288-GB setting) vs re-materialising every layer
    (the reference's setting): identical outputs and gradients."""
    g = load_golden("dit_lin_1scene.pt")
    res = []
    for n_free in (0, 1, 99):
        m = DiffusionTransformer(ModelConfig(**g["cfg"]))
        m.load_state_dict(g["state_dict"], strict=True)
        for mod in m.modules():
            if hasattr(mod, "use_kernel"):
                mod.use_kernel = False
        m.remat_free_layers = n_free
        out = m(g["video"], g["text"], g["timesteps"])
        out.backward(g["dout"])
        res.append((out.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    for out, grads in res[1:]:
        assert torch.equal(out, res[0][0])
        for k, v in grads.items():
            assert torch.equal(v, res[0][1][k]), k
