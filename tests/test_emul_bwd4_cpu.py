"""The deriver-wave body of the revision-4 TTT-MLP backward (csrc/ttt_bwd4_aux_body.h) executed on the CPU by the lane-level
wave emulator (tests/emul): one reverse step of one hidden slice against plain torch statements of the same step.

Checks the new index algebra of round 3 without a GPU: the reversed state update W2_i = W2_{i+1} + (eta X2_i)^T gZ2_i,
W1_i = W1_{i+1} + (eta K_i)^T gZ1_i (forward: ops/ttt_mlp.py:48-52 with the opposite sign), gX2 = gZ2 W2_i^T, the derived
activations, and the fragment images the sweep's compute waves read (layouts of csrc/ttt_mfma_dev.h).  bf16 operands, fp32
accumulation: tolerance 1e-2 on every array (a layout error shows as O(1))."""
import ctypes
import os
import subprocess

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/amdclang++"


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(CLANG):
        pytest.skip("host clang of the ROCm toolchain not available")
    build = os.path.join(HERE, "emul", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libbwd4_emul.so")
    srcs = [os.path.join(HERE, "emul", f) for f in ("bwd4_emul.cpp", "wave_emul.h")] + \
           [os.path.join(ROOT, "ttt-video-dit_amd", "csrc", f) for f in ("ttt_bwd4_aux_body.h", "ttt_wave_types.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([CLANG, "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-Wno-psabi",
                               "-I", os.path.join(ROOT, "ttt-video-dit_amd", "csrc"), "-I", os.path.join(HERE, "emul"),
                               srcs[0], "-o", so])
    return ctypes.CDLL(so)


def _pi(s, h, e):
    return 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)


def _idx(a, b, s):
    return (a * 2 + b) * 2 + s


def frag_map(kind):
    """index tensors (row, col) [8 fragments, 64 lanes, 8 elements] of a fragment array over a [64][64] matrix X[row][col]:
    'T'  [ti][nj][s]: X[t = 32 ti + pi][n = 32 nj + c]      'N' [nj][ti][s]: X[t = 32 ti + c][n = 32 nj + pi]"""
    rows = torch.zeros(8, 64, 8, dtype=torch.long)
    cols = torch.zeros(8, 64, 8, dtype=torch.long)
    for a in range(2):
        for b in range(2):
            for s in range(2):
                for lane in range(64):
                    h, c = lane >> 5, lane & 31
                    for e in range(8):
                        if kind == "T":          # first index = row block (contraction rows in registers), second = lane block
                            rows[_idx(a, b, s), lane, e], cols[_idx(a, b, s), lane, e] = 32 * a + _pi(s, h, e), 32 * b + c
                        else:                    # N: array index [nj][ti]: rows = t on the lane, cols = n in registers
                            rows[_idx(a, b, s), lane, e], cols[_idx(a, b, s), lane, e] = 32 * b + c, 32 * a + _pi(s, h, e)
    return rows, cols


def encode(X, kind):
    r, c = frag_map(kind)
    return X[r, c].to(torch.bfloat16).contiguous()


def decode(arr, kind):
    r, c = frag_map(kind)
    X = torch.full((64, 64), float("nan"))
    X[r, c] = arr.float()
    assert not torch.isnan(X).any()
    return X


def gelu3(x):
    a, c3 = 0.79788456, 0.044715
    u = a * x * (1 + c3 * x * x)
    t = torch.tanh(u)
    du = a * (1 + 3 * c3 * x * x)
    d2u = 6 * a * c3 * x
    y = 0.5 * x * (1 + t)
    dy = 0.5 * (1 + t) + 0.5 * x * (1 - t * t) * du
    d2y = (1 - t * t) * du + 0.5 * x * ((1 - t * t) * d2u - 2 * t * (1 - t * t) * du * du)
    return y, dy, d2y


@pytest.mark.parametrize("seed", [0, 1])
def test_deriver_reverse_step_on_the_emulator(emul, seed):
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16)
    W1n = 0.05 * torch.randn(64, 64, generator=g)                    # [f][n]  state after the step
    W2n = 0.05 * torch.randn(64, 64, generator=g)                    # [n][f]
    Z1 = bf(1.5 * torch.randn(64, 64, generator=g))                  # [t][n]
    Z1b = bf(1.5 * torch.randn(64, 64, generator=g))
    K = bf(torch.nn.functional.normalize(torch.randn(64, 64, generator=g), dim=-1))
    G = bf(torch.randn(64, 64, generator=g))                         # gZ2 [t][f]
    eta = (0.02 * torch.rand(64, generator=g) + 0.005).float()       # large, so that the update is visible

    # ---- expected, plain statements (fp32; operands rounded where the kernel rounds them) ----------------
    X2, D1, D2 = (bf(t).float() for t in gelu3(Z1.float()))
    W2i = W2n + bf(eta[:, None] * X2).float().T @ G.float()
    gX2 = G.float() @ bf(W2i).float().T
    gZ1 = gX2 * D1
    M = gX2 * D2
    W1i = W1n + K.float().T @ bf(eta[:, None] * gZ1).float()
    X2b, D1b, _ = gelu3(Z1b.float())

    # ---- emulator -----------------------------------------------------------------------------------------
    W1, W2 = W1n.clone().contiguous(), W2n.clone().contiguous()
    z1f, z1bf = encode(Z1.float(), "T"), encode(Z1b.float(), "T")
    n_lds = emul.emul_bwd4_lds_bytes()
    lds = torch.zeros(n_lds, dtype=torch.uint8)
    gsl = torch.zeros(16 * 1024, dtype=torch.uint8)
    msg = ctypes.create_string_buffer(256)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    races = emul.emul_bwd4_aux_step(P(W1), P(W2), P(z1f), P(z1bf), P(K.contiguous()), P(G.contiguous()), P(eta), P(lds), P(gsl), msg, 256)
    assert races == 0, msg.value.decode()

    def arr(buf, k):                        # k-th 8 KiB fragment array of a byte buffer
        return buf[k * 8192:(k + 1) * 8192].view(torch.bfloat16).reshape(8, 64, 8)

    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    got = {
        "W2_i (fp32 state)": (W2, W2i),
        "R1 gZ1 (N)": (decode(arr(lds, 0), "N"), gZ1), "R1 gelu'(Z1) (N)": (decode(arr(lds, 1), "N"), D1), "R1 X2 (N)": (decode(arr(lds, 2), "N"), X2),
        "R2 W2_i (rows n, lane f)": (decode(arr(lds, 3), "T"), W2i),
        "R3 gelu'(Z1b) (T)": (decode(arr(lds, 4), "T"), D1b), "R3 X2b (T)": (decode(arr(lds, 5), "T"), X2b),
        "R3 W2_i^T (rows f, lane n)": (decode(arr(lds, 6), "T"), W2i.T),
        "R4 gelu'(Z1) (T)": (decode(arr(lds, 7), "T"), D1), "R4 M (T)": (decode(arr(lds, 8), "T"), M), "R4 X2 (T)": (decode(arr(lds, 9), "T"), X2),
    }
    # the tail kernel's share: gZ1 in the T orientation; W1 is not the deriver's business any more (the tail rebuilds it per group)
    got["tail gZ1 (T)"] = (decode(arr(gsl, 0), "T"), gZ1)
    assert torch.equal(W1, W1n) and not gsl[8192:].any()
    errs = {k: rel(a, b) for k, (a, b) in got.items()}
    # the reversed update must be VISIBLE at this eta (else the state checks prove nothing)
    assert rel(W2n, W2i) > 5e-2 and rel(W1n, W1i) > 5e-2
    bad = {k: v for k, v in errs.items() if not v < 1e-2}
    assert not bad, (bad, errs)
