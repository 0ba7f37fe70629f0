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
