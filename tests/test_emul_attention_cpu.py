"""The segment-attention workgroup bodies of csrc/attn_body.h (revision 2 of the forward and dQ kernels) executed on the CPU by
the lane-level wave emulator of tests/emul (8 emulated waves per workgroup, MFMA 32x32x16 / ds_read_b64_tr_b16 / workgroup
barrier semantics of gfx950, LDS race detector) against the fp64 attention oracle on the same bf16-rounded inputs.  The same
template bodies are instantiated with the device backend in csrc/attn_v2.hip.  Tolerances as for the GPU parity tests
(tests/test_attention_gpu.py): outputs rel-L2 <= 1e-2, gradients <= 2e-2."""
import ctypes
import math
import os
import subprocess

import pytest
import torch

from helpers import rel_l2
from oracle import attn_oracle as AO

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLANG = "/opt/rocm/lib/llvm/bin/amdclang++"

_P, _L, _I, _F = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_float


class FwdParams(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("Q", "K", "V", "O", "LSE")] + \
               [(n, _L) for n in ("q_sb", "q_sh", "q_ss", "k_sb", "k_sh", "k_ss", "v_sb", "v_sh", "v_ss", "o_sb", "o_sh", "o_ss")] + \
               [("B", _I), ("NH", _I), ("S", _I), ("scale", _F)]


class BwdParams(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("Q", "K", "V", "O", "dO", "LSE", "Delta", "dQ", "dK", "dV")] + \
               [(f"{t}_{s}", _L) for t in ("q", "k", "v", "o", "do", "dq", "dk", "dv") for s in ("sb", "sh", "ss")] + \
               [("B", _I), ("NH", _I), ("S", _I), ("scale", _F)]


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(CLANG):
        pytest.skip("host clang of the ROCm toolchain not available")
    build = os.path.join(HERE, "emul", "_build")
    os.makedirs(build, exist_ok=True)
    so = os.path.join(build, "libattn_emul.so")
    csrc = os.path.join(ROOT, "ttt-video-dit_amd", "csrc")
    srcs = [os.path.join(HERE, "emul", f) for f in ("attn_emul.cpp", "wave_emul.h")] + \
           [os.path.join(csrc, f) for f in ("attn_body.h", "attn_types.h", "ttt_wave_types.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([CLANG, "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-Wno-psabi",
                               "-I", csrc, "-I", os.path.join(HERE, "emul"), srcs[0], "-o", so])
    lib = ctypes.CDLL(so)
    assert lib.emul_attn_fwd_params_size() == ctypes.sizeof(FwdParams)
    assert lib.emul_attn_bwd_params_size() == ctypes.sizeof(BwdParams)
    return lib


def _make(B, NH, S, seed, layout):
    """bf16 q, k, v, dO as [B, NH, S, 64] views: of [B, S, NH, 64] memory (what the block produces) or contiguous."""
    g = torch.Generator().manual_seed(seed)
    mk = lambda s: (torch.randn(B, S, NH, 64, generator=g) * s).bfloat16()
    q, k, v, do = mk(1.5), mk(1.5), mk(1.0), mk(1.0)
    f = (lambda t: t.transpose(1, 2)) if layout == "bshd" else (lambda t: t.transpose(1, 2).contiguous())
    return f(q), f(k), f(v), f(do)


def _strides(p, name, t):
    sb, sh, ss, sd = t.stride()
    assert sd == 1
    for s, v in (("sb", sb), ("sh", sh), ("ss", ss)):
        setattr(p, f"{name}_{s}", v)


def _forward(lib, q, k, v):
    B, NH, S, _ = q.shape
    out = torch.full((B, S, NH, 64), float("nan"), dtype=torch.bfloat16).transpose(1, 2)
    lse = torch.full((B, NH, S), float("nan"))
    p = FwdParams()
    p.Q, p.K, p.V, p.O, p.LSE = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr()
    for n, t in (("q", q), ("k", k), ("v", v), ("o", out)):
        _strides(p, n, t)
    p.B, p.NH, p.S, p.scale = B, NH, S, 1 / math.sqrt(64)
    msg = ctypes.create_string_buffer(256)
    races = lib.emul_attn_forward(ctypes.byref(p), msg, 256)
    assert races == 0, msg.value.decode()
    return out, lse


def _oracle(q, k, v, do):
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    out, lse = AO.attention(q64, k64, v64)
    out.backward(do.double())
    return out.detach(), lse.detach(), q64.grad, k64.grad, v64.grad


# ragged tail inside one tile, ragged over several tiles and blocks (both workgroup->head mappings), exact multiples of 64 / 256
SHAPES = [(1, 2, 40, "bshd"), (2, 3, 300, "bshd"), (1, 8, 128, "bhsd"), (1, 1, 577, "bshd")]


@pytest.mark.parametrize("B,NH,S,layout", SHAPES)
def test_emulated_attention_forward_vs_oracle(emul, B, NH, S, layout):
    q, k, v, do = _make(B, NH, S, 7 + S, layout)
    out, lse = _forward(emul, q, k, v)
    ro, rl, *_ = _oracle(q, k, v, do)
    assert not torch.isnan(out.float()).any()
    assert rel_l2(out, ro) < 1e-2
    assert (lse.double() - rl).abs().max() < 2e-2


def test_emulated_attention_forward_rescale_path(emul):
    """one key dominates from a late tile on: the running-max rescale of the online softmax (as tests/test_attention_gpu.py)"""
    B, NH, S = 1, 1, 330
    q, k, v, do = _make(B, NH, S, 3, "bshd")
    k = k.clone()
    k[:, :, 290] = q[:, :, 17] * 6.0
    out, _ = _forward(emul, q, k, v)
    ro, _ = AO.attention(q.double(), k.double(), v.double())
    assert rel_l2(out, ro) < 1e-2


def _bwd_params(q, k, v, do, ro, rl):
    """the backward kernels' inputs: the forward's bf16 output and fp32 LSE, Delta = rowsum(dO * O) as attn_delta_kernel forms
    it; outputs NaN-filled [B, NH, S, 64] views of [B, S, NH, 64] memory"""
    B, NH, S, _ = q.shape
    o = ro.to(torch.bfloat16)
    lse = rl.float().contiguous()
    delta = (do.float() * o.float()).sum(-1).contiguous()
    outs = [torch.full((B, S, NH, 64), float("nan"), dtype=torch.bfloat16).transpose(1, 2) for _ in range(3)]
    p = BwdParams()
    p.Q, p.K, p.V, p.O, p.dO, p.LSE, p.Delta = (t.data_ptr() for t in (q, k, v, o, do, lse, delta))
    p.dQ, p.dK, p.dV = (t.data_ptr() for t in outs)
    for n, t in (("q", q), ("k", k), ("v", v), ("o", o), ("do", do), ("dq", outs[0]), ("dk", outs[1]), ("dv", outs[2])):
        _strides(p, n, t)
    p.B, p.NH, p.S, p.scale = B, NH, S, 1 / math.sqrt(64)
    return p, outs, (o, lse, delta)


@pytest.mark.parametrize("B,NH,S,layout", SHAPES)
def test_emulated_attention_dq_vs_oracle(emul, B, NH, S, layout):
    q, k, v, do = _make(B, NH, S, 11 + S, layout)
    ro, rl, rq, _, _ = _oracle(q, k, v, do)
    p, (dq, _, _), keep = _bwd_params(q, k, v, do, ro, rl)
    msg = ctypes.create_string_buffer(256)
    races = emul.emul_attn_dq(ctypes.byref(p), msg, 256)
    assert races == 0, msg.value.decode()
    assert not torch.isnan(dq.float()).any()
    assert rel_l2(dq, rq) < 2e-2


# dK / dV body: 2 = revision 1's arithmetic, 3 = accumulators started from -LSE / scale and -Delta, 4 = the same with 12 waves
@pytest.mark.parametrize("variant", [2, 3, 4])
@pytest.mark.parametrize("B,NH,S,layout", [(1, 2, 40, "bshd"), (2, 3, 300, "bshd"), (1, 8, 128, "bhsd"), (1, 1, 800, "bshd")])
def test_emulated_attention_dkdv_vs_oracle(emul, B, NH, S, layout, variant):
    q, k, v, do = _make(B, NH, S, 13 + S, layout)
    ro, rl, _, rk, rv = _oracle(q, k, v, do)
    p, (_, dk, dv), keep = _bwd_params(q, k, v, do, ro, rl)
    msg = ctypes.create_string_buffer(256)
    races = emul.emul_attn_dkdv(ctypes.byref(p), variant, msg, 256)
    assert races == 0, msg.value.decode()
    assert not torch.isnan(dk.float()).any() and not torch.isnan(dv.float()).any()
    errs = (rel_l2(dk, rk), rel_l2(dv, rv))
    print(variant, (B, NH, S), errs)
    assert max(errs) < 2e-2, errs


def test_emulated_dkdv_variants_agree(emul):
    """the accumulator-initialised form against revision 1's arithmetic on the same inputs: differences only at the level of
    the bf16 rounding of P / dS (a few 1e-3 relative), and the 8- and 12-wave decompositions of the SAME arithmetic bit-equal"""
    q, k, v, do = _make(1, 2, 450, 99, "bshd")
    ro, rl, *_ = _oracle(q, k, v, do)
    res = {}
    for variant in (2, 3, 4):
        p, (_, dk, dv), keep = _bwd_params(q, k, v, do, ro, rl)
        assert emul.emul_attn_dkdv(ctypes.byref(p), variant, None, 0) == 0
        res[variant] = (dk.float().clone(), dv.float().clone())
    assert torch.equal(res[3][0], res[4][0]) and torch.equal(res[3][1], res[4][1])
    assert rel_l2(res[3][0], res[2][0]) < 5e-3 and rel_l2(res[3][1], res[2][1]) < 5e-3


@pytest.mark.parametrize("B,NH,S,layout", [(1, 2, 40, "bshd"), (1, 1, 300, "bshd"), (1, 1, 128, "bhsd"), (1, 1, 577, "bshd")])
def test_two_tiles_per_stage_is_bit_identical(emul, B, NH, S, layout):
    """dQ and dK / dV with TWO tiles of 64 per LDS stage (half the workgroup barriers; csrc/attn_body.h dq_staged / dkdv_staged,
    not instantiated on the device yet): the same arithmetic in the same order, so the same bits as the one-tile form - one tile in all (40), an
    odd tile count (300: 5 = a half-filled last stage), a ragged tail (577: 10 tiles), exactly one stage (128) - and no LDS
    race between the stage being filled and the stage being read."""
    q, k, v, do = _make(B, NH, S, 21 + S, layout)
    ro, rl, *_ = _oracle(q, k, v, do)
    res = {}
    for nsub in (1, 2, -1, -2):           # (negative: the same with XOR-swizzled LDS tiles - another layout, the same numbers)
        p, (dq, dk, dv), keep = _bwd_params(q, k, v, do, ro, rl)
        msg = ctypes.create_string_buffer(256)
        assert emul.emul_attn_dq_n(ctypes.byref(p), nsub, msg, 256) == 0, msg.value.decode()
        assert emul.emul_attn_dkdv_n(ctypes.byref(p), 4, nsub, msg, 256) == 0, msg.value.decode()
        res[nsub] = [t.float().clone() for t in (dq, dk, dv)]
        assert not any(torch.isnan(t).any() for t in res[nsub])
    for nsub in (2, -1, -2):
        for a, b in zip(res[1], res[nsub]):
            assert torch.equal(a, b), nsub
    # dK / dV with three and four tiles per stage (one workgroup per CU: up to 148 KiB of LDS; round-4 A/B candidates)
    for nsub in (3, 4):
        p, (dq, dk, dv), keep = _bwd_params(q, k, v, do, ro, rl)
        msg = ctypes.create_string_buffer(256)
        assert emul.emul_attn_dkdv_n(ctypes.byref(p), 4, nsub, msg, 256) == 0, msg.value.decode()
        for a, b in zip(res[1][1:], (dk.float(), dv.float())):
            assert torch.equal(a, b), nsub


def test_lds_bank_model_of_the_backward_bodies(emul):
    """The emulator's LDS bank model (tests/emul/wave_emul.h: lane groups and banks of MI355X_MICROARCH.md) on the shipped dQ and
    dK / dV bodies.  What the layout was designed for holds - the 16-byte row fragments of the stride-72 tiles and the staging
    stores are conflict-free - and what the counters show is explained: every transposed read of those tiles is 2-way conflicted
    (rows r and r + 2 of its 4-row groups sit 8 banks apart, as do its two half-groups).  The model's bank-conflict share of all LDS
    passes - dQ 22.2 %, dK / dV 24.0 % - is what rocprofv3 measured on the device (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,
    profiles/r3p_wait_lds_summary.txt: 6.4 of 28.8 points = 22.2 %, 10.1 of 42.3 = 23.9 %): the model can be trusted to judge a
    layout before it is timed."""
    q, k, v, do = _make(1, 1, 768, 5, "bshd")
    ro, rl, *_ = _oracle(q, k, v, do)
    for kernel, share in ((0, 6.4 / 28.8), (1, 10.1 / 42.3)):
        p, outs, keep = _bwd_params(q, k, v, do, ro, rl)
        out = (ctypes.c_long * 9)()
        assert emul.emul_attn_bank_model(ctypes.byref(p), kernel, out) == 0
        rd, wr, tr = out[0:3], out[3:6], out[6:9]
        print("kernel", kernel, "reads", rd, "stores", wr, "transposed", tr)
        assert rd[1] == 0 and wr[1] == 0                      # plain reads and stores: no conflict passes
        assert tr[1] == tr[0] // 2 and tr[0] == 4 * tr[2]     # transposed reads: 2 groups x 2 passes each, half of them replays
        total = rd[0] + wr[0] + tr[0]
        assert abs(tr[1] / total - share) < 0.01, (tr[1] / total, share)


def test_swizzled_tiles_are_conflict_free_both_ways(emul):
    """The XOR-swizzled [64][64] tile of csrc/attn_body.h (tile_off<true>) under the bank model: row fragments, transposed reads
    and staging stores of the dQ and dK / dV bodies all conflict-free - the 22 - 24 % of the LDS passes that the stride-72 layout
    spends on replays are gone (what that is worth in time has to be measured: the guide's warning is that a two-phase loop hides
    LDS-read conflicts)."""
    q, k, v, do = _make(1, 1, 768, 5, "bshd")
    ro, rl, *_ = _oracle(q, k, v, do)
    for kernel in (2, 3):
        p, outs, keep = _bwd_params(q, k, v, do, ro, rl)
        out = (ctypes.c_long * 9)()
        assert emul.emul_attn_bank_model(ctypes.byref(p), kernel, out) == 0
        print("kernel", kernel, list(out))
        assert out[1] == 0 and out[4] == 0 and out[7] == 0


@pytest.mark.parametrize("B,NH,S,layout", [(1, 2, 40, "bshd"), (1, 1, 300, "bshd"), (1, 1, 577, "bhsd"), (1, 1, 700, "bshd")])
def test_dq_with_64_query_rows_per_wave_is_bit_identical(emul, B, NH, S, layout):
    """dq_wide<NSUB, 2> (csrc/attn_body.h, round 4): a wave owns TWO blocks of 32 query rows, so every K / V fragment read from LDS
    feeds two MFMAs (half the LDS bytes per MFMA); a workgroup covers 512 rows.  A query row's arithmetic and its order over the
    keys are dq()'s: same bits - one partly filled workgroup (40, 300), a ragged key tail inside the second workgroup (577), rows
    of the second query block past the end (700) - and no LDS race, with one and with two key tiles per stage."""
    q, k, v, do = _make(B, NH, S, 77 + S, layout)
    ro, rl, *_ = _oracle(q, k, v, do)
    p, (dq, dk, dv), keep = _bwd_params(q, k, v, do, ro, rl)
    msg = ctypes.create_string_buffer(256)
    assert emul.emul_attn_dq_n(ctypes.byref(p), 1, msg, 256) == 0, msg.value.decode()
    ref = dq.float().clone()
    for nsub in (1, 2):
        p, (dq, dk, dv), keep = _bwd_params(q, k, v, do, ro, rl)
        assert emul.emul_attn_dq_wide(ctypes.byref(p), nsub, msg, 256) == 0, msg.value.decode()
        assert not torch.isnan(dq.float()).any()
        assert torch.equal(dq.float(), ref), nsub
