"""The wave-level kernel bodies of csrc/ttt_lin16_body.h (TTT-Linear, mini-batches of 16) executed on the CPU by the
lane-level wave emulator of tests/emul (64 host threads = 64 lanes; MFMA / transposed-LDS-read / DPP semantics of gfx950),
against the fp64 oracle.  The same template bodies are instantiated with the device backend in csrc/ttt_mfma16.hip, so this
checks the kernels' index algebra and arithmetic without a GPU (tolerances as for the GPU parity tests: SURVEY.md 8c)."""
import ctypes
import os
import shutil
import subprocess

import pytest
import torch

from helpers import rel_l2, tile_states
from oracle import ttt_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/amdclang++"


class Params(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("XQ", "XK", "XV", "eta", "ln_w", "ln_b", "W1", "b1", "W1c", "b1c", "out", "dOut", "dW1_last", "db1_last",
                 "scratch_w", "scratch_b", "dln_w", "dln_b", "dW1", "db1", "deta", "dXQ", "dXK", "dXV")] + \
               [(n, ctypes.c_int) for n in ("NH", "NC", "G", "K")] + [("eps", ctypes.c_float)]


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(CLANG):
        pytest.skip("host clang of the ROCm toolchain not available")
    build = os.path.join(HERE, "emul", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "liblin16_emul.so")
    srcs = [os.path.join(HERE, "emul", f) for f in ("lin16_emul.cpp", "wave_emul.h")] + \
           [os.path.join(ROOT, "ttt-video-dit_amd", "csrc", f) for f in ("ttt_lin16_body.h", "ttt_mlp16_body.h", "ttt_wave_types.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([CLANG, "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-Wno-psabi",
                               "-I", os.path.join(ROOT, "ttt-video-dit_amd", "csrc"), "-I", os.path.join(HERE, "emul"),
                               srcs[0], "-o", so])
    lib = ctypes.CDLL(so)
    assert lib.emul_lin16_params_size() == ctypes.sizeof(Params)
    return lib


def _inputs(B, NH, NC, seed):
    d = O.make_inputs("linear", B, NH, NC, 16, 64, seed=seed)
    for k in ("XQ", "XK", "XV", "eta", "dOut"):
        d[k] = d[k].to(torch.bfloat16).to(torch.float32)            # the kernels see bf16 activations
    return d


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _forward(lib, d, G):
    B, NH, NC = d["XQ"].shape[:3]
    K = -(-NC // G)
    bf = lambda t: t.to(torch.bfloat16).contiguous()
    XQ, XK, XV = bf(d["XQ"]), bf(d["XK"]), bf(d["XV"])
    eta = bf(d["eta"][:, :, :, -1, :, None])
    st = tile_states(d, B)
    W1, b1 = st["W1"].float().contiguous(), st["b1"].float().contiguous()
    lw, lb = d["ln_w"].float().contiguous(), d["ln_b"].float().contiguous()
    W1c, b1c = torch.full((B, NH, K, 64, 64), float("nan")), torch.full((B, NH, K, 1, 64), float("nan"))
    out = torch.full((B, NH, NC, 16, 64), float("nan"), dtype=torch.bfloat16)
    p = Params()
    for n, t in dict(XQ=XQ, XK=XK, XV=XV, eta=eta, ln_w=lw, ln_b=lb, W1=W1, b1=b1, W1c=W1c, b1c=b1c, out=out).items():
        setattr(p, n, t.data_ptr())
    p.NH, p.NC, p.G, p.K, p.eps = NH, NC, G, K, 1e-8
    lib.emul_lin16_forward(ctypes.byref(p), B * NH)
    keep = (XQ, XK, XV, eta, lw, lb, W1, b1)
    return out, (W1c, b1c), keep


def _oracle(d, G):
    d64 = {k: v.double() for k, v in d.items()}
    st = tile_states(d64, d64["XQ"].shape[0])
    le = d64["eta"][:, :, :, -1, :, None]
    out, cks, _ = O.linear_forward(d64["XQ"], d64["XK"], d64["XV"], le, d64["ln_w"], d64["ln_b"], st["W1"], st["b1"], G)
    g = O.linear_backward(d64["XQ"], d64["XK"], d64["XV"], le, d64["ln_w"], d64["ln_b"], cks, G, d64["dOut"])
    return out, cks, g


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 2, 5, 2), (2, 1, 6, 4)])
def test_emulated_linear_forward_vs_oracle(emul, shape):
    B, NH, NC, G = shape
    d = _inputs(B, NH, NC, seed=17 + NC)
    out, cks, _ = _forward(emul, d, G)
    ro, rc, _ = _oracle(d, G)
    assert rel_l2(out, ro) < 1e-2
    assert rel_l2(cks[0], rc[0]) < 1e-2 and rel_l2(cks[1], rc[1]) < 1e-2


def _backward(lib, d, G, fwd):
    out, (W1c, b1c), (XQ, XK, XV, eta, lw, lb, W1, b1) = fwd
    B, NH, NC = XQ.shape[:3]
    K = -(-NC // G)
    dOut = d["dOut"].to(torch.bfloat16).contiguous()
    z = lambda *s: torch.zeros(*s)
    nan = lambda *s, dt=torch.float32: torch.full(s, float("nan"), dtype=dt)
    g = dict(dln_w=nan(B, NH, 1, 64), dln_b=nan(B, NH, 1, 64), dW1=nan(B, NH, 64, 64), db1=nan(B, NH, 1, 64),
             dlast_eta=nan(B, NH, NC, 16, 1, dt=torch.bfloat16), dXQ=nan(B, NH, NC, 16, 64, dt=torch.bfloat16),
             dXK=nan(B, NH, NC, 16, 64, dt=torch.bfloat16), dXV=nan(B, NH, NC, 16, 64, dt=torch.bfloat16))
    dWl, dbl = z(B, NH, 64, 64), z(B, NH, 1, 64)
    scr_w, scr_b = torch.zeros(B * NH * G * 16 * 1024, dtype=torch.uint8), nan(B, NH, G, 1, 64)
    p = Params()
    for n, t in dict(XQ=XQ, XK=XK, XV=XV, eta=eta, ln_w=lw, ln_b=lb, W1c=W1c, b1c=b1c, dOut=dOut, dW1_last=dWl, db1_last=dbl,
                     scratch_w=scr_w, scratch_b=scr_b, dln_w=g["dln_w"], dln_b=g["dln_b"], dW1=g["dW1"], db1=g["db1"],
                     deta=g["dlast_eta"], dXQ=g["dXQ"], dXK=g["dXK"], dXV=g["dXV"]).items():
        setattr(p, n, t.data_ptr())
    p.NH, p.NC, p.G, p.K, p.eps = NH, NC, G, K, 1e-8
    lib.emul_lin16_backward(ctypes.byref(p), B * NH)
    return g


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 2, 5, 2), (2, 1, 6, 4), (1, 1, 7, 3), (1, 1, 4, 1), (1, 1, 3, 3)])
def test_emulated_linear_backward_vs_oracle(emul, shape):
    """Reverse sweep of the TTT-Linear scan (mini-batch 16) on the wave emulator vs fp64 autograd-equivalent oracle:
    single step, even / odd group sizes, ragged last group, one step per group, one group for the whole sequence."""
    B, NH, NC, G = shape
    d = _inputs(B, NH, NC, seed=29 + NC)
    fwd = _forward(emul, d, G)
    g = _backward(emul, d, G, fwd)
    _, _, rg = _oracle(d, G)
    errs = {k: rel_l2(g[k], rg[k].reshape(g[k].shape) if k in ("dln_w", "dln_b") else rg[k]) for k in g}
    print("emulated linear backward errors", shape, {k: round(v, 5) for k, v in errs.items()})
    assert all(v < 3e-2 for v in errs.values()), errs


# ---------------------------------------------------------------------------------------------- TTT-MLP, mini-batch 16
class MlpParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("XQ", "XK", "XV", "eta", "ln_w", "ln_b", "W1", "b1", "W2", "b2", "W1c", "b1c", "W2c", "b2c", "out")] + \
               [(n, ctypes.c_int) for n in ("NH", "NC", "G", "K")] + [("eps", ctypes.c_float)]


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 2, 5, 2), (2, 1, 4, 4)])
def test_emulated_mlp_scan16_forward_vs_oracle(emul, shape):
    """The 8-wave / two-barriers-per-step TTT-MLP forward scan body (csrc/ttt_mlp16_body.h) on the multi-wave emulator (512
    host threads) vs the fp64 oracle: output, checkpoints."""
    assert emul.emul_mlp16_params_size() == ctypes.sizeof(MlpParams)
    B, NH, NC, G = shape
    d = O.make_inputs("mlp", B, NH, NC, 16, 64, seed=41 + NC)
    for k in ("XQ", "XK", "XV", "eta", "dOut"):
        d[k] = d[k].to(torch.bfloat16).to(torch.float32)
    K = -(-NC // G)
    bf = lambda t: t.to(torch.bfloat16).contiguous()
    XQ, XK, XV = bf(d["XQ"]), bf(d["XK"]), bf(d["XV"])
    eta = bf(d["eta"][:, :, :, -1, :, None])
    st = {k: v.float().contiguous() for k, v in tile_states(d, B).items()}
    lw, lb = d["ln_w"].float().contiguous(), d["ln_b"].float().contiguous()
    nan = lambda *s: torch.full(s, float("nan"))
    cks = (nan(B, NH, K, 64, 256), nan(B, NH, K, 1, 256), nan(B, NH, K, 256, 64), nan(B, NH, K, 1, 64))
    out = torch.full((B, NH, NC, 16, 64), float("nan"), dtype=torch.bfloat16)
    p = MlpParams()
    for n, t in dict(XQ=XQ, XK=XK, XV=XV, eta=eta, ln_w=lw, ln_b=lb, W1=st["W1"], b1=st["b1"], W2=st["W2"], b2=st["b2"],
                     W1c=cks[0], b1c=cks[1], W2c=cks[2], b2c=cks[3], out=out).items():
        setattr(p, n, t.data_ptr())
    p.NH, p.NC, p.G, p.K, p.eps = NH, NC, G, K, 1e-8
    msg = ctypes.create_string_buffer(256)
    races = emul.emul_mlp16_forward(ctypes.byref(p), B * NH, msg, 256)
    assert races == 0, f"LDS race between waves (hazard analysis of the two-barrier schedule violated): {msg.value.decode()}"
    d64 = {k: v.double() for k, v in d.items()}
    s64 = tile_states(d64, B)
    ro, rc, _ = O.mlp_forward(d64["XQ"], d64["XK"], d64["XV"], d64["eta"][:, :, :, -1, :, None], d64["ln_w"], d64["ln_b"],
                              s64["W1"], s64["b1"], s64["W2"], s64["b2"], G)
    assert rel_l2(out, ro) < 1e-2
    for c, r in zip(cks, rc):
        assert rel_l2(c, r) < 1e-2


@pytest.mark.parametrize("shape,M,N,K", [(0, 16, 16, 32), (1, 16, 16, 16), (2, 32, 32, 16)])
def test_emulated_mfma_shapes_are_matmuls(emul, shape, M, N, K):
    """The emulator's MFMA fragment layouts (the same lane / register maps the kernels are written against) really compute
    D = A B: 16x16x32, 16x16x16 and the 32x32x16 shape of the CS = 64 kernels."""
    g = torch.Generator().manual_seed(shape)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    D = torch.full((M, N), float("nan"))
    emul.emul_mfma_selftest(shape, ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(B.data_ptr()), ctypes.c_void_p(D.data_ptr()))
    ref = A.bfloat16().float() @ B.bfloat16().float()
    assert torch.allclose(D, ref, atol=1e-5, rtol=1e-5)


def test_race_detector_flags_missing_barriers(emul):
    """The emulator's LDS race detector on a two-wave exchange: clean with both barriers, and every way of breaking it
    (missing barrier before the read, two writers in one epoch, rewrite before the readers are done) is reported."""
    msg = ctypes.create_string_buffer(256)
    assert emul.emul_race_selftest(0, msg, 256) == 0
    # (which of the two racing accesses the host threads perform first decides whether a read/write race is reported as
    # "read after write" or "write after read" - it is the same race)
    for mode, kinds in ((1, ("read after write", "write after read")), (2, ("write after write",)),
                        (3, ("write after read", "read after write"))):
        assert emul.emul_race_selftest(mode, msg, 256) > 0
        assert any(k in msg.value.decode() for k in kinds), (mode, msg.value.decode())
