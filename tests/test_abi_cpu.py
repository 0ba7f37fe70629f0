"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol that
include/ttt_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ttt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ttt_hip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import test_time_training as ext
    if not os.path.exists(ext.library_path()):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(ext.library_path())
    names = _declared_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ttt_hip.h but not exported"
    assert sorted(ext.EXPORTED_SYMBOLS) == names
    lib.ttt_hip_abi_version.restype = ctypes.c_int
    assert lib.ttt_hip_abi_version() == 5


def test_argument_validation_without_gpu():
    """Dimension / impl validation happens before any launch, so it can be exercised on CPU."""
    import test_time_training as ext
    lib = ext.load_library()
    d = ext._Dims(1, 2, 4, 64, 64, 2, 0, 0, 1e-8)
    assert lib.ttt_hip_resolve_impl(ctypes.byref(d), 1, 0) in (1, 2)
    bad = ext._Dims(1, 2, 4, 24, 64, 2, 0, 0, 1e-8)        # CS=24 unsupported by every kernel family
    assert lib.ttt_hip_resolve_impl(ctypes.byref(bad), 1, 0) == -1
    neg = ext._Dims(0, 2, 4, 64, 64, 2, 0, 0, 1e-8)
    assert lib.ttt_hip_resolve_impl(ctypes.byref(neg), 1, 0) == -1
    assert b"dimension" in lib.ttt_hip_last_error()
    assert lib.ttt_hip_mlp_forward_workspace(ctypes.byref(d)) > 0 or lib.ttt_hip_resolve_impl(ctypes.byref(d), 1, 0) == 2


def test_cpu_tensors_are_rejected():
    import torch
    import test_time_training as ext
    x = torch.zeros(1, 1, 2, 16, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="HIP device"):
        ext.ttt_forward(x, x, x, x, *[x] * 11, 1)


def test_forward_workspace_is_the_pair_scan_ring():
    """ABI 5 (round 6): the TTT-MLP forward at mini-batches of 64 on the MFMA scan asks for its ring of state records - per (b,h) four records of
    65.25 KiB (pack(W1'), pack(W2'), b1', b2') + two 128-byte flag lines (csrc/ttt_mfma2.hip) -, every other forward for nothing."""
    import test_time_training as ext
    lib = ext.load_library()
    d = ext._Dims(1, 48, 804, 64, 64, 16, 0, 2, 1e-8)                 # bf16, impl = MFMA
    assert lib.ttt_hip_mlp_forward_workspace(ctypes.byref(d)) == 48 * (4 * (64 * 1024 + 1024 + 256) + 256)
    d16 = ext._Dims(2, 48, 21948, 16, 64, 21948, 0, 2, 1e-8)          # the sampling geometry: mini-batches of 16
    assert lib.ttt_hip_mlp_forward_workspace(ctypes.byref(d16)) == 0
    assert lib.ttt_hip_linear_forward_workspace(ctypes.byref(d16)) == 0


def test_pipeline_part_plan_tapers_and_covers_every_group(monkeypatch):
    """ttt_amd/models/ssm/pipeline.py: the parts of the pipelined layer forward are whole checkpoint groups, cover the scan exactly, and taper
    towards the end by default (round 6); TTT_PIPELINE_WEIGHTS overrides."""
    from ttt_amd.models.ssm.pipeline import part_group_counts, plan_parts
    monkeypatch.delenv("TTT_PIPELINE_WEIGHTS", raising=False)
    assert part_group_counts(51, 5) == [16, 16, 11, 6, 2] and part_group_counts(51, 4) == [16, 17, 12, 6]
    for K, n in ((51, 5), (18, 5), (8, 4), (343, 8), (165, 5), (10, 5), (300, 7)):
        c = part_group_counts(K, n)
        assert len(c) == n and sum(c) == K and min(c) >= 1, (K, n, c)
        assert c[-1] <= c[0]
    monkeypatch.setenv("TTT_PIPELINE_WEIGHTS", "equal")
    assert part_group_counts(51, 4) == [13, 13, 13, 12]
    monkeypatch.setenv("TTT_PIPELINE_WEIGHTS", "1,1,2")
    assert part_group_counts(8, 3) == [2, 2, 4]
    monkeypatch.delenv("TTT_PIPELINE_WEIGHTS")
    parts = plan_parts(None, 804 * 64, 64, 16, 5)
    assert [p[0] for p in parts] == [0, 256, 512, 688, 784] and sum(p[1] for p in parts) == 804
    assert parts[-1][2] == [(784 * 64, 804 * 64)]

