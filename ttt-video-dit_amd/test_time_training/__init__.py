"""Drop-in replacement for the reference's un-vendored ``test_time_training`` CUDA extension
(ttt-tk), backed by hand-written HIP kernels for MI355X / gfx950 behind the C ABI of
``include/ttt_hip.h``.

``ttt_forward`` / ``ttt_backward`` take exactly the positional tensors the reference passes at
``ttt/models/ssm/mlp_tk.py:116-133`` and ``:227-275``: every buffer (outputs, checkpoints,
re-materialisation scratch) is allocated by the caller, results are written in place, nothing
is returned, kernels are enqueued on the current torch stream without synchronising.

``ttt_linear_forward`` / ``ttt_linear_backward`` expose the TTT-Linear kernels with the tensor
contract of the reference's Triton launch sites (``ttt/models/ssm/linear_triton.py:98-129``,
``:203-246``).

There is no CPU path and no fallback: if ``lib/libttt_hip.so`` is missing or a tensor is not on
a HIP device the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libttt_hip.so")

IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA = 0, 1, 2
_IMPL_NAMES = {"auto": IMPL_AUTO, "generic": IMPL_GENERIC, "mfma": IMPL_MFMA}

# LayerNorm epsilon of the reference's ops path (ops/utils.py:4,21); the Triton kernels use 1e-6
# (kernels/linear_forward.py:111) - SURVEY.md hazard C1.  Adjustable for experiments.
_state = {"impl": IMPL_AUTO, "eps": 1e-8}


class _Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "NH", "NC", "CS", "F", "G", "act_dtype", "impl")] + [("eps", ctypes.c_float)]


def _ptr_struct(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": [(f, ctypes.c_void_p) for f in fields]})


MLP_FWD_FIELDS = ("XQ", "XK", "XV", "last_eta", "ttt_norm_weight", "ttt_norm_bias", "W1_init", "b1_init", "W2_init",
                  "b2_init", "W1_checkpoints", "b1_checkpoints", "W2_checkpoints", "b2_checkpoints", "XQW")
MLP_BWD_FIELDS = ("XQ", "XK", "XV", "last_eta", "ttt_norm_weight", "ttt_norm_bias", "W1_checkpoints", "b1_checkpoints",
                  "W2_checkpoints", "b2_checkpoints", "XQW", "W1_init_group", "b1_init_group", "W2_init_group",
                  "b2_init_group", "x_hat_ln_group", "std_ln_group", "X2_group", "Z1_group", "Z1_bar_group", "X2_bar_group",
                  "grad_l_wrt_Z2_group", "grad_l_wrt_Z1_group", "x_hat_fused_group", "grad_x_hat_fused_group",
                  "grad_output_fused_group", "std_fused_group", "grad_L_W1_last", "grad_L_b1_last", "grad_L_W2_last",
                  "grad_L_b2_last", "grad_L_XQW", "grad_L_ttt_norm_weight", "grad_L_ttt_norm_bias", "grad_L_W1_init",
                  "grad_L_b1_init", "grad_L_W2_init", "grad_L_b2_init", "grad_L_last_eta", "grad_L_XQ", "grad_L_XK", "grad_L_XV")
LIN_FWD_FIELDS = ("XQ", "XK", "XV", "last_eta", "ttt_norm_weight", "ttt_norm_bias", "W1_init", "b1_init",
                  "W1_checkpoints", "b1_checkpoints", "XQW")
LIN_BWD_FIELDS = ("XQ", "XK", "XV", "last_eta", "ttt_norm_weight", "ttt_norm_bias", "W1_checkpoints", "b1_checkpoints",
                  "grad_L_W1_last", "grad_L_b1_last", "grad_L_XQW", "W1_init_group", "b1_init_group",
                  "grad_L_ttt_norm_weight", "grad_L_ttt_norm_bias", "grad_L_W1_init", "grad_L_b1_init", "grad_L_last_eta",
                  "grad_L_XQ", "grad_L_XK", "grad_L_XV")

_MlpFwd = _ptr_struct("_MlpFwd", MLP_FWD_FIELDS)
_MlpBwd = _ptr_struct("_MlpBwd", MLP_BWD_FIELDS)
_LinFwd = _ptr_struct("_LinFwd", LIN_FWD_FIELDS)
_LinBwd = _ptr_struct("_LinBwd", LIN_BWD_FIELDS)

# TTT_HIP_ABI_VERSION of include/ttt_hip.h this binding was written against (2: return codes -3 / -10 / -11 / -12 of the TTT-MLP
# entry points, the round-1 debug exports ttt_hip_debug_variant / _helpers gone; 3: ttt_hip_pre_backward_ld / ttt_hip_attn_pre_backward_ld,
# ttt_hip_mlp_forward_chunk, ttt_hip_pre_forward_range / ttt_hip_post_forward_range; 4 (round 6): ttt_hip_stream_create_masked /
# ttt_hip_stream_destroy / ttt_hip_debug_placement_probe added, the tensor entry points unchanged; 5 (round 6): the TTT-MLP forward has a
# workspace - the ring of state records of the pair scan - and ttt_hip_mlp_forward_chunk uses its workspace arguments)
ABI_VERSION = 5

# every extern "C" symbol declared in include/ttt_hip.h
EXPORTED_SYMBOLS = (
    "ttt_hip_mlp_forward_workspace", "ttt_hip_mlp_backward_workspace", "ttt_hip_linear_forward_workspace",
    "ttt_hip_linear_backward_workspace", "ttt_hip_mlp_forward", "ttt_hip_mlp_forward_chunk", "ttt_hip_mlp_backward", "ttt_hip_linear_forward",
    "ttt_hip_linear_backward", "ttt_hip_resolve_impl", "ttt_hip_abi_version", "ttt_hip_last_error",
    "ttt_hip_debug_timing", "ttt_hip_debug_groups_per_chunk", "ttt_hip_debug_dump", "ttt_hip_debug_option", "ttt_hip_debug_sweep_error", "ttt_hip_sweep_error_clear", "ttt_hip_debug_occupy_cus",
    "ttt_hip_stream_create_masked", "ttt_hip_stream_destroy", "ttt_hip_debug_placement_probe",
    "ttt_hip_pre_forward", "ttt_hip_pre_forward_range", "ttt_hip_post_forward_range", "ttt_hip_pre_backward_partials", "ttt_hip_pre_backward", "ttt_hip_pre_backward_ld", "ttt_hip_post_partials",
    "ttt_hip_post_forward", "ttt_hip_post_backward", "ttt_hip_gate_forward", "ttt_hip_gate_backward_partials",
    "ttt_hip_gate_backward", "ttt_hip_attn_forward", "ttt_hip_attn_backward",
    "ttt_hip_attn_pre_forward", "ttt_hip_attn_pre_partials", "ttt_hip_attn_pre_backward", "ttt_hip_attn_pre_backward_ld",
    "ttt_hip_adaln_forward", "ttt_hip_adaln_backward_partials", "ttt_hip_adaln_backward",
    "ttt_hip_resgate_forward", "ttt_hip_resgate_backward_partials", "ttt_hip_resgate_backward",
)

_lib: Optional[ctypes.CDLL] = None


def library_path() -> str:
    return _LIB_PATH


def load_library() -> ctypes.CDLL:
    """dlopen libttt_hip.so (built by ``__graft_entry__.build()`` / ``csrc/build.sh``).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"test_time_training: HIP library not found at {_LIB_PATH}; build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    lib = ctypes.CDLL(_LIB_PATH)
    for n in ("mlp_forward", "mlp_backward", "linear_forward", "linear_backward"):
        getattr(lib, f"ttt_hip_{n}_workspace").restype = ctypes.c_size_t
        getattr(lib, f"ttt_hip_{n}_workspace").argtypes = [ctypes.c_void_p]
        getattr(lib, f"ttt_hip_{n}").restype = ctypes.c_int
        getattr(lib, f"ttt_hip_{n}").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.ttt_hip_resolve_impl.restype = ctypes.c_int
    lib.ttt_hip_resolve_impl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.ttt_hip_abi_version.restype = ctypes.c_int
    lib.ttt_hip_last_error.restype = ctypes.c_char_p
    if lib.ttt_hip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"test_time_training: libttt_hip.so ABI version {lib.ttt_hip_abi_version()}, this binding needs {ABI_VERSION}: rebuild (csrc/build.sh)")
    _lib = lib
    return lib


def debug_timing(buf: Optional[torch.Tensor]) -> None:
    """DEBUG: per-phase cycle counters of workgroup 0 are accumulated into ``buf`` (16 x int64, zeroed, on device)."""
    lib = load_library()
    lib.ttt_hip_debug_timing.argtypes = [ctypes.c_void_p]
    lib.ttt_hip_debug_timing(buf.data_ptr() if buf is not None else None)


def debug_groups_per_chunk(groups: int) -> None:
    """DEBUG: force the MFMA backward's chunk size in checkpoint groups (0 = automatic)."""
    lib = load_library()
    lib.ttt_hip_debug_groups_per_chunk.argtypes = [ctypes.c_int]
    lib.ttt_hip_debug_groups_per_chunk(int(groups))


def debug_option(name: str, value: int) -> None:
    """DEBUG / A-B knobs by name (see ttt_hip_debug_option in include/ttt_hip.h): ``groups_per_chunk``, ``fast_records``."""
    lib = load_library()
    lib.ttt_hip_debug_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.ttt_hip_debug_option.restype = ctypes.c_int
    if lib.ttt_hip_debug_option(name.encode(), int(value)) != 0:
        raise ValueError(f"unknown debug option {name!r}")


def sweep_error() -> int:
    """0, or 1 + (b,h) of a cluster-form backward workgroup whose hand-over partner never arrived (synchronises)."""
    lib = load_library()
    lib.ttt_hip_debug_sweep_error.restype = ctypes.c_uint
    return int(lib.ttt_hip_debug_sweep_error())


def debug_occupy_cus(workgroups: int, lds_bytes: int, microseconds: int, stream=None) -> None:
    """DEBUG (stress tests): hold ``workgroups`` CUs' worth of LDS for ``microseconds`` on ``stream`` (default: the current one)."""
    lib = load_library()
    st = (stream or torch.cuda.current_stream()).cuda_stream
    lib.ttt_hip_debug_occupy_cus.restype = ctypes.c_int
    if lib.ttt_hip_debug_occupy_cus(int(workgroups), int(lds_bytes), int(microseconds), ctypes.c_void_p(st)) != 0:
        raise RuntimeError("ttt_hip_debug_occupy_cus: bad arguments or launch failure")


def masked_stream(cu_mask_words) -> "torch.cuda.ExternalStream":
    """A stream whose kernels run only on the compute units of ``cu_mask_words`` (sequence of 32-bit words, bit i of word w = logical CU
    32 w + i; ``ttt_hip_stream_create_masked``), as a ``torch.cuda.ExternalStream`` of the current device.  The HIP stream lives as long as
    the process (a handful per process: the sweep / side streams of a backward, a communication stream)."""
    lib = load_library()
    words = [int(w) & 0xFFFFFFFF for w in cu_mask_words]
    arr = (ctypes.c_uint * len(words))(*words)
    out = ctypes.c_void_p()
    lib.ttt_hip_stream_create_masked.restype = ctypes.c_int
    if lib.ttt_hip_stream_create_masked(arr, len(words), ctypes.byref(out)) != 0:
        lib.ttt_hip_last_error.restype = ctypes.c_char_p
        raise RuntimeError(lib.ttt_hip_last_error().decode())
    return torch.cuda.ExternalStream(out.value)


def placement_probe(workgroups: int, stream=None, lds_bytes: int = 150 * 1024, microseconds: int = 200):
    """DEBUG: where ``workgroups`` one-wave workgroups of ``stream`` run - a list of (xcd, shader engine, shader array, cu) per workgroup
    (each holds ``lds_bytes`` of LDS for ``microseconds``, so that every CU takes one).  Synchronises."""
    lib = load_library()
    st = stream or torch.cuda.current_stream()
    out = torch.zeros(workgroups, dtype=torch.int32, device="cuda")
    lib.ttt_hip_debug_placement_probe.restype = ctypes.c_int
    with torch.cuda.stream(st):
        rc = lib.ttt_hip_debug_placement_probe(ctypes.c_void_p(out.data_ptr()), int(workgroups), int(lds_bytes), int(microseconds), ctypes.c_void_p(st.cuda_stream))
    if rc != 0:
        raise RuntimeError("ttt_hip_debug_placement_probe: bad arguments or launch failure")
    st.synchronize()
    w = out.cpu().tolist()
    return [((v >> 16) & 0xF, (v >> 13) & 0x7, (v >> 12) & 0x1, (v >> 8) & 0xF) for v in w]


def sweep_error_clear() -> None:
    """Acknowledge a hand-over time-out (synchronises): TTT-MLP calls are accepted again."""
    load_library().ttt_hip_sweep_error_clear()


def sweep_fast_count() -> int:
    """DEBUG statistic: cluster workgroup launches that proved same-XCD placement and published plain (L2-resident) records."""
    lib = load_library()
    lib.ttt_hip_debug_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.ttt_hip_debug_option.restype = ctypes.c_int
    return -2 - int(lib.ttt_hip_debug_option(b"sweep_fast_count", 0))


def debug_dump(buf: Optional[torch.Tensor]) -> None:
    """DEBUG: step-0 intermediates of workgroup 0 of the revision-2 forward go to ``buf`` (>= 120000 fp32 on device)."""
    lib = load_library()
    lib.ttt_hip_debug_dump.argtypes = [ctypes.c_void_p]
    lib.ttt_hip_debug_dump(buf.data_ptr() if buf is not None else None)


def set_impl(name: str) -> None:
    """Select the kernel family: 'auto' (MFMA when the geometry allows), 'generic', 'mfma'."""
    _state["impl"] = _IMPL_NAMES[name]


def get_impl() -> str:
    return {v: k for k, v in _IMPL_NAMES.items()}[_state["impl"]]


def set_ln_eps(eps: float) -> None:
    _state["eps"] = float(eps)


def get_ln_eps() -> float:
    return _state["eps"]


# ------------------------------------------------------------------------------------------------
def _check(t: torch.Tensor, name: str, shape, dtype) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")


def _check5(XQ) -> None:
    if not isinstance(XQ, torch.Tensor) or XQ.ndim != 5:
        raise RuntimeError("XQ: expected a 5-D tensor [B, NH, NC, CS, F]")


def _dims(B, NH, NC, CS, F, G, act_dtype) -> _Dims:
    if act_dtype == torch.bfloat16:
        code = 0
    elif act_dtype == torch.float32:
        code = 1
    else:
        raise RuntimeError(f"activations must be bfloat16 or float32, got {act_dtype}")
    if G < 1:
        raise RuntimeError("checkpoint_group_size must be >= 1")
    return _Dims(B, NH, NC, CS, F, G, code, _state["impl"], _state["eps"])


# Workspaces (the TTT-MLP backward's step records: 2.2 GB at 48 heads) are kept per (device, stream) and grown on demand instead of
# being drawn from torch's caching allocator at every call (84 times per training step at 9 s, in the phase where HBM is fullest:
# round-3 verdict, weak #8).  Calls on one stream are ordered, so they can share the buffer; calls on different streams (two
# autograd threads) get their own.  The key is the raw stream handle; a stream that was destroyed and whose handle the runtime
# hands out again would inherit a buffer that torch's allocator associates with the old stream - harmless for ordering (the
# buffer is only ever used on the stream of the key) but the entry of a dead stream would stay: at most `_WS_MAX` entries are kept,
# the least recently used one goes first.  ``release_workspaces()`` returns the memory (sampling does, see
# ttt_amd/models/cogvideo/sampling.py).
_ws_cache = {}
_WS_MAX = 4


def _workspace(device, stream: int, nbytes: int):
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), int(stream))
    buf = _ws_cache.pop(key, None)                  # (re-inserted below: dict order = recency)
    if buf is None or buf.numel() < nbytes:
        del buf                                     # let the old block go before the larger one is requested
        while len(_ws_cache) >= _WS_MAX:
            _ws_cache.pop(next(iter(_ws_cache)))
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _ws_cache[key] = buf
    return buf


def release_workspaces() -> None:
    """Drop the cached kernel workspaces (they are re-created by the next call that needs one)."""
    _ws_cache.clear()


def _launch(fn_name: str, dims: _Dims, args, device) -> None:
    lib = load_library()
    ws_bytes = getattr(lib, fn_name + "_workspace")(ctypes.byref(dims))
    stream = torch.cuda.current_stream(device).cuda_stream
    ws = _workspace(device, stream, ws_bytes) if ws_bytes else None
    with torch.cuda.device(device):
        rc = getattr(lib, fn_name)(ctypes.byref(dims), ctypes.byref(args), ws.data_ptr() if ws is not None else None,
                                   ws_bytes, ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(lib.ttt_hip_last_error().decode())


def resolved_impl(B, NH, NC, CS, F, G, act_dtype=torch.bfloat16, mlp=True, backward=False) -> str:
    """Name of the kernel family a call with these dims would run ('generic' / 'mfma')."""
    lib = load_library()
    d = _dims(B, NH, NC, CS, F, G, act_dtype)
    r = lib.ttt_hip_resolve_impl(ctypes.byref(d), int(mlp), int(backward))
    return {1: "generic", 2: "mfma"}.get(r, "unsupported")


# ------------------------------------------------------------------------------------------------
def ttt_forward(XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, W2_init, b2_init,
                W1_checkpoints, b1_checkpoints, W2_checkpoints, b2_checkpoints, XQW_batch, checkpoint_group_size):
    """TTT-MLP forward scan; argument list of the reference call site mlp_tk.py:116-133."""
    _check5(XQ)
    B, NH, NC, CS, F = XQ.shape
    G = int(checkpoint_group_size)
    K = -(-NC // G)
    H = 4 * F
    act, f32 = XQ.dtype, torch.float32
    t = dict(XQ=XQ, XK=XK, XV=XV, last_eta=last_eta, ttt_norm_weight=ttt_norm_weight, ttt_norm_bias=ttt_norm_bias,
             W1_init=W1_init, b1_init=b1_init, W2_init=W2_init, b2_init=b2_init, W1_checkpoints=W1_checkpoints,
             b1_checkpoints=b1_checkpoints, W2_checkpoints=W2_checkpoints, b2_checkpoints=b2_checkpoints, XQW=XQW_batch)
    spec = dict(XQ=((B, NH, NC, CS, F), act), XK=((B, NH, NC, CS, F), act), XV=((B, NH, NC, CS, F), act),
                last_eta=((B, NH, NC, CS, 1), act), ttt_norm_weight=((1, NH, 1, F), f32), ttt_norm_bias=((1, NH, 1, F), f32),
                W1_init=((B, NH, F, H), f32), b1_init=((B, NH, 1, H), f32), W2_init=((B, NH, H, F), f32),
                b2_init=((B, NH, 1, F), f32), W1_checkpoints=((B, NH, K, F, H), f32), b1_checkpoints=((B, NH, K, 1, H), f32),
                W2_checkpoints=((B, NH, K, H, F), f32), b2_checkpoints=((B, NH, K, 1, F), f32), XQW=((B, NH, NC, CS, F), act))
    for k, (shape, dt) in spec.items():
        _check(t[k], k, shape, dt)
    args = _MlpFwd(*[t[k].data_ptr() for k in MLP_FWD_FIELDS])
    _launch("ttt_hip_mlp_forward", _dims(B, NH, NC, CS, F, G, act), args, XQ.device)


def ttt_forward_chunk(XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_state, b1_state, W2_state, b2_state,
                      W1_checkpoints, b1_checkpoints, W2_checkpoints, b2_checkpoints, XQW_batch, checkpoint_group_size, step0, nsteps):
    """The TTT-MLP forward over steps [step0, step0 + nsteps) of the sequence the (whole-sequence) tensors describe - an extension
    beside ``ttt_forward``'s 15-tensor call (``ttt_hip_mlp_forward_chunk``): started from the fp32 state in ``*_state``
    ([B,NH,F,H], [B,NH,1,H], [B,NH,H,F], [B,NH,1,F]), which it REPLACES by the state after its last step, so that consecutive
    calls walk the sequence with the bits of the one-call forward.  Parts start at checkpoint-group boundaries."""
    _check5(XQ)
    B, NH, NC, CS, F = XQ.shape
    G = int(checkpoint_group_size)
    K = -(-NC // G)
    H = 4 * F
    act, f32 = XQ.dtype, torch.float32
    t = dict(XQ=XQ, XK=XK, XV=XV, last_eta=last_eta, ttt_norm_weight=ttt_norm_weight, ttt_norm_bias=ttt_norm_bias,
             W1_init=W1_state, b1_init=b1_state, W2_init=W2_state, b2_init=b2_state, W1_checkpoints=W1_checkpoints,
             b1_checkpoints=b1_checkpoints, W2_checkpoints=W2_checkpoints, b2_checkpoints=b2_checkpoints, XQW=XQW_batch)
    spec = dict(XQ=((B, NH, NC, CS, F), act), XK=((B, NH, NC, CS, F), act), XV=((B, NH, NC, CS, F), act),
                last_eta=((B, NH, NC, CS, 1), act), ttt_norm_weight=((1, NH, 1, F), f32), ttt_norm_bias=((1, NH, 1, F), f32),
                W1_init=((B, NH, F, H), f32), b1_init=((B, NH, 1, H), f32), W2_init=((B, NH, H, F), f32),
                b2_init=((B, NH, 1, F), f32), W1_checkpoints=((B, NH, K, F, H), f32), b1_checkpoints=((B, NH, K, 1, H), f32),
                W2_checkpoints=((B, NH, K, H, F), f32), b2_checkpoints=((B, NH, K, 1, F), f32), XQW=((B, NH, NC, CS, F), act))
    for k, (shape, dt) in spec.items():
        _check(t[k], k, shape, dt)
    args = _MlpFwd(*[t[k].data_ptr() for k in MLP_FWD_FIELDS])
    dims = _dims(B, NH, NC, CS, F, G, act)
    lib = load_library()
    stream = torch.cuda.current_stream(XQ.device).cuda_stream
    ws_bytes = lib.ttt_hip_mlp_forward_workspace(ctypes.byref(dims))       # the pair scan's ring of state records (round 6)
    ws = _workspace(XQ.device, stream, ws_bytes) if ws_bytes else None
    with torch.cuda.device(XQ.device):
        rc = lib.ttt_hip_mlp_forward_chunk(ctypes.byref(dims), ctypes.byref(args), ctypes.c_int(int(step0)), ctypes.c_int(int(nsteps)),
                                           _p(W1_state), _p(b1_state), _p(W2_state), _p(b2_state), ctypes.c_void_p(ws.data_ptr() if ws is not None else None),
                                           ctypes.c_size_t(ws_bytes), ctypes.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(lib.ttt_hip_last_error().decode())


def ttt_backward(XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_checkpoints, b1_checkpoints, W2_checkpoints,
                 b2_checkpoints, XQW_batch, W1_init_group, b1_init_group, W2_init_group, b2_init_group, x_hat_ln_group,
                 std_ln_group, X2_group, Z1_group, Z1_bar_group, X2_bar_group, grad_l_wrt_Z2_group, grad_l_wrt_Z1_group,
                 x_hat_fused_group, grad_x_hat_fused_group, grad_output_fused_group, std_fused_group, grad_L_W1_last,
                 grad_L_b1_last, grad_L_W2_last, grad_L_b2_last, grad_L_XQW_batch, grad_L_ttt_norm_weight,
                 grad_L_ttt_norm_bias, grad_L_W1_init, grad_L_b1_init, grad_L_W2_init, grad_L_b2_init, grad_L_last_eta,
                 grad_L_XQ, grad_L_XK, grad_L_XV, checkpoint_group_size):
    """TTT-MLP backward; the 42 tensors + 1 int of the reference call site mlp_tk.py:227-275."""
    _check5(XQ)
    B, NH, NC, CS, F = XQ.shape
    G = int(checkpoint_group_size)
    K = -(-NC // G)
    H = 4 * F
    act, f32, bf = XQ.dtype, torch.float32, torch.bfloat16
    vals = (XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_checkpoints, b1_checkpoints, W2_checkpoints,
            b2_checkpoints, XQW_batch, W1_init_group, b1_init_group, W2_init_group, b2_init_group, x_hat_ln_group,
            std_ln_group, X2_group, Z1_group, Z1_bar_group, X2_bar_group, grad_l_wrt_Z2_group, grad_l_wrt_Z1_group,
            x_hat_fused_group, grad_x_hat_fused_group, grad_output_fused_group, std_fused_group, grad_L_W1_last,
            grad_L_b1_last, grad_L_W2_last, grad_L_b2_last, grad_L_XQW_batch, grad_L_ttt_norm_weight,
            grad_L_ttt_norm_bias, grad_L_W1_init, grad_L_b1_init, grad_L_W2_init, grad_L_b2_init, grad_L_last_eta,
            grad_L_XQ, grad_L_XK, grad_L_XV)
    t = dict(zip(MLP_BWD_FIELDS, vals))
    A = ((B, NH, NC, CS, F), act)
    spec = dict(XQ=A, XK=A, XV=A, last_eta=((B, NH, NC, CS, 1), act),
                ttt_norm_weight=((1, NH, 1, F), f32), ttt_norm_bias=((1, NH, 1, F), f32),
                W1_checkpoints=((B, NH, K, F, H), f32), b1_checkpoints=((B, NH, K, 1, H), f32),
                W2_checkpoints=((B, NH, K, H, F), f32), b2_checkpoints=((B, NH, K, 1, F), f32), XQW=A,
                W1_init_group=((B, NH, G, F, H), f32), b1_init_group=((B, NH, G, 1, H), f32),
                W2_init_group=((B, NH, G, H, F), f32), b2_init_group=((B, NH, G, 1, F), f32),
                x_hat_ln_group=((B, NH, G, CS, F), bf), std_ln_group=((B, NH, G, CS, 1), f32),
                X2_group=((B, NH, G, CS, H), bf), Z1_group=((B, NH, G, CS, H), bf), Z1_bar_group=((B, NH, G, CS, H), bf),
                X2_bar_group=((B, NH, G, CS, H), bf), grad_l_wrt_Z2_group=((B, NH, G, CS, F), bf),
                grad_l_wrt_Z1_group=((B, NH, G, CS, H), bf), x_hat_fused_group=((B, NH, G, CS, F), bf),
                grad_x_hat_fused_group=((B, NH, G, CS, F), bf), grad_output_fused_group=((B, NH, G, CS, F), bf),
                std_fused_group=((B, NH, G, CS, 1), f32),
                grad_L_W1_last=((B, NH, F, H), f32), grad_L_b1_last=((B, NH, 1, H), f32),
                grad_L_W2_last=((B, NH, H, F), f32), grad_L_b2_last=((B, NH, 1, F), f32), grad_L_XQW=A,
                grad_L_ttt_norm_weight=((B, NH, 1, F), f32), grad_L_ttt_norm_bias=((B, NH, 1, F), f32),
                grad_L_W1_init=((B, NH, F, H), f32), grad_L_b1_init=((B, NH, 1, H), f32),
                grad_L_W2_init=((B, NH, H, F), f32), grad_L_b2_init=((B, NH, 1, F), f32),
                grad_L_last_eta=((B, NH, NC, CS, 1), act), grad_L_XQ=A, grad_L_XK=A, grad_L_XV=A)
    # The sixteen re-materialisation buffers of the reference contract (mlp_tk.py:192-210) may be None HERE (the reference
    # always passes them; this repo's fused autograd node does not allocate what no kernel touches): the MFMA backward works in
    # its own workspace, the generic kernels need the four *_init_group buffers (the C ABI refuses NULL there).
    scratch = MLP_BWD_FIELDS[11:27]
    for k, (shape, dt) in spec.items():
        if k in scratch and t[k] is None:
            continue
        _check(t[k], k, shape, dt)
    args = _MlpBwd(*[(t[k].data_ptr() if t[k] is not None else None) for k in MLP_BWD_FIELDS])
    _launch("ttt_hip_mlp_backward", _dims(B, NH, NC, CS, F, G, act), args, XQ.device)


def ttt_linear_forward(XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, W1_checkpoints,
                       b1_checkpoints, XQW_batch, checkpoint_group_size):
    """TTT-Linear forward scan (replaces ttt_linear_scan_forward, linear_triton.py:98-129)."""
    _check5(XQ)
    B, NH, NC, CS, F = XQ.shape
    G = int(checkpoint_group_size)
    K = -(-NC // G)
    act, f32 = XQ.dtype, torch.float32
    vals = (XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_init, b1_init, W1_checkpoints, b1_checkpoints, XQW_batch)
    t = dict(zip(LIN_FWD_FIELDS, vals))
    A = ((B, NH, NC, CS, F), act)
    spec = dict(XQ=A, XK=A, XV=A, last_eta=((B, NH, NC, CS, 1), act), ttt_norm_weight=((NH, F), f32),
                ttt_norm_bias=((NH, F), f32), W1_init=((B, NH, F, F), f32), b1_init=((B, NH, 1, F), f32),
                W1_checkpoints=((B, NH, K, F, F), f32), b1_checkpoints=((B, NH, K, 1, F), f32), XQW=A)
    for k, (shape, dt) in spec.items():
        _check(t[k], k, shape, dt)
    args = _LinFwd(*[t[k].data_ptr() for k in LIN_FWD_FIELDS])
    _launch("ttt_hip_linear_forward", _dims(B, NH, NC, CS, F, G, act), args, XQ.device)


def ttt_linear_backward(XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_checkpoints, b1_checkpoints,
                        grad_L_W1_last, grad_L_b1_last, grad_L_XQW_batch, W1_init_group, b1_init_group,
                        grad_L_ttt_norm_weight, grad_L_ttt_norm_bias, grad_L_W1_init, grad_L_b1_init, grad_L_last_eta,
                        grad_L_XQ, grad_L_XK, grad_L_XV, checkpoint_group_size):
    """TTT-Linear backward (replaces ttt_linear_scan_backward, linear_triton.py:203-246)."""
    _check5(XQ)
    B, NH, NC, CS, F = XQ.shape
    G = int(checkpoint_group_size)
    K = -(-NC // G)
    act, f32 = XQ.dtype, torch.float32
    vals = (XQ, XK, XV, last_eta, ttt_norm_weight, ttt_norm_bias, W1_checkpoints, b1_checkpoints, grad_L_W1_last,
            grad_L_b1_last, grad_L_XQW_batch, W1_init_group, b1_init_group, grad_L_ttt_norm_weight, grad_L_ttt_norm_bias,
            grad_L_W1_init, grad_L_b1_init, grad_L_last_eta, grad_L_XQ, grad_L_XK, grad_L_XV)
    t = dict(zip(LIN_BWD_FIELDS, vals))
    A = ((B, NH, NC, CS, F), act)
    spec = dict(XQ=A, XK=A, XV=A, last_eta=((B, NH, NC, CS, 1), act), ttt_norm_weight=((NH, F), f32),
                ttt_norm_bias=((NH, F), f32), W1_checkpoints=((B, NH, K, F, F), f32), b1_checkpoints=((B, NH, K, 1, F), f32),
                grad_L_W1_last=((B, NH, F, F), f32), grad_L_b1_last=((B, NH, 1, F), f32), grad_L_XQW=A,
                W1_init_group=((B, NH, G, F, F), f32), b1_init_group=((B, NH, G, 1, F), f32),
                grad_L_ttt_norm_weight=((B, NH, 1, F), f32), grad_L_ttt_norm_bias=((B, NH, 1, F), f32),
                grad_L_W1_init=((B, NH, F, F), f32), grad_L_b1_init=((B, NH, 1, F), f32),
                grad_L_last_eta=((B, NH, NC, CS, 1), act), grad_L_XQ=A, grad_L_XK=A, grad_L_XV=A)
    for k, (shape, dt) in spec.items():
        _check(t[k], k, shape, dt)
    args = _LinBwd(*[t[k].data_ptr() for k in LIN_BWD_FIELDS])
    _launch("ttt_hip_linear_backward", _dims(B, NH, NC, CS, F, G, act), args, XQ.device)


# ------------------------------------------------------------------------------------------------
# Fused pre- / post-processing kernels (include/ttt_hip.h, "Fused pre- / post-processing").  Thin wrappers: the
# caller (ttt_amd/models/ssm/fused.py) allocates every tensor; bf16 activations, fp32 parameters / tables.
def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _req(t, name, dtype):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous {dtype} tensor on a HIP device")


def _call(fn, *args, device):
    lib = load_library()
    f = getattr(lib, fn)
    f.restype = ctypes.c_int
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    with torch.cuda.device(device):
        rc = f(*args, stream)
    if rc != 0:
        raise RuntimeError(lib.ttt_hip_last_error().decode())


def _req_maps(rope, src, pos, L, F, n_pos):
    """Token maps / RoPE table of the fused pre kernels: the kernel indexes ``rope[pos[t]]`` and ``x[src[t]]`` unchecked, so
    the table must cover every position (the reference's apply_rotary_emb raises a shape error for a video longer than
    config.compressed_num_frames, ssm/utils.py:82-108) and the maps must be int32 of length L.  ``n_pos`` = 1 + the largest
    position the maps address, as a host integer (the module caches it with the maps; no device synchronisation here).
    With ``pos`` given and ``n_pos`` unknown (a direct user of the binding, or maps that lost the module's cache through
    ``.to()`` / ``clone()``) the bound is taken from the map itself, once per map tensor (one synchronising ``max()``, cached on
    the tensor); with no ``pos`` the kernels rotate nothing (every token is text: position -1) and there is nothing to check."""
    for t, n in ((src, "src"), (pos, "pos")):
        if t is None:
            continue
        _req(t, n, torch.int32)
        if t.numel() != L:
            raise RuntimeError(f"{n}: expected {L} entries, got {t.numel()}")
    if rope is None:
        return
    _req(rope, "rope", torch.float32)
    if rope.numel() % F != 0:
        raise RuntimeError(f"rope: expected [n_pos, {F // 2}, 2] (cos, sin) pairs, got {tuple(rope.shape)}")
    n_rows = rope.numel() // F
    if pos is None:
        return
    if n_pos is None:
        n_pos = getattr(pos, "_ttt_max_pos", None)
        if n_pos is None:
            n_pos = int(pos.max()) + 1
            try:
                pos._ttt_max_pos = n_pos
            except Exception:
                pass
    if n_pos > n_rows:
        raise RuntimeError(f"rope table has {n_rows} positions but the sequence addresses {n_pos} (video longer than "
                           f"config.compressed_num_frames?)")


def pre_forward(XQ_raw, XK_raw, XV_raw, rope, src, pos, ln_w, ln_b, XQ, XK, XV, NH, n_pos=None, t0=0, tn=None):
    """``t0``, ``tn``: the scan positions [t0, t0 + tn) only (a part of the sequence; default: all of it)"""
    B, L, D = XQ_raw.shape
    _req_maps(rope, src, pos, L, D // NH, n_pos)
    for t, n in ((XQ_raw, "XQ_raw"), (XK_raw, "XK_raw"), (XV_raw, "XV_raw"), (XQ, "XQ"), (XK, "XK"), (XV, "XV")):
        _req(t, n, torch.bfloat16)
    for t, n in ((ln_w, "ln_w"), (ln_b, "ln_b")):
        _req(t, n, torch.float32)
    _call("ttt_hip_pre_forward_range", B, L, NH, D // NH, _p(XQ_raw), _p(XK_raw), _p(XV_raw), _p(rope), _p(src), _p(pos), _p(ln_w), _p(ln_b),
          _p(XQ), _p(XK), _p(XV), int(t0), int(L - t0 if tn is None else tn), device=XQ_raw.device)


def pre_backward_partials(NH):
    return load_library().ttt_hip_pre_backward_partials(int(NH))


def _req_rows(t, name, shape, ld):
    """a bf16 [B, L, D] tensor on a HIP device whose token rows are `ld` elements apart (a column block of a wider buffer)"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.bfloat16 or tuple(t.shape) != tuple(shape) \
            or t.stride(2) != 1 or t.stride(1) != ld or (shape[0] > 1 and t.stride(0) != shape[1] * ld) or t.data_ptr() % 16:
        raise RuntimeError(f"{name}: expected a bf16 {tuple(shape)} tensor on a HIP device with token rows {ld} elements apart")


def pre_backward(XQ_raw, XK_raw, XV_raw, rope, src, pos, ln_w, dXQ, dXK, dXV, dXQ_raw, dXK_raw, dXV_raw, dlnw_part, dlnb_part, NH, ld_out=None):
    """``ld_out``: row stride (elements) of the three raw-gradient outputs - None: contiguous [B, L, D] tensors; 3 * D: the column
    blocks of one [B, L, 3 D] buffer (the q / k / v projections' weight gradients are then one GEMM, ttt_amd/infra/fused_linear.py)."""
    B, L, D = XQ_raw.shape
    for t, n in ((XQ_raw, "XQ_raw"), (XK_raw, "XK_raw"), (XV_raw, "XV_raw"), (dXQ, "dXQ"), (dXK, "dXK"), (dXV, "dXV")):
        _req(t, n, torch.bfloat16)
    for t, n in ((dXQ_raw, "dXQ_raw"), (dXK_raw, "dXK_raw"), (dXV_raw, "dXV_raw")):
        if ld_out is None:
            _req(t, n, torch.bfloat16)
        else:
            _req_rows(t, n, (B, L, D), int(ld_out))
    for t, n in ((ln_w, "ln_w"), (dlnw_part, "dlnw_part"), (dlnb_part, "dlnb_part")):
        _req(t, n, torch.float32)
    _call("ttt_hip_pre_backward_ld", B, L, NH, D // NH, _p(XQ_raw), _p(XK_raw), _p(XV_raw), _p(rope), _p(src), _p(pos), _p(ln_w),
          _p(dXQ), _p(dXK), _p(dXV), _p(dXQ_raw), _p(dXK_raw), _p(dXV_raw), ctypes.c_int64(D if ld_out is None else int(ld_out)),
          _p(dlnw_part), _p(dlnb_part), device=XQ_raw.device)


def post_partials(B, L):
    return load_library().ttt_hip_post_partials(int(B), int(L))


def post_forward(Y, src, w, b, out, eps, t0=0, tn=None):
    """``t0``, ``tn``: the scan positions [t0, t0 + tn) only (default: all)"""
    B, NH, L, F = Y.shape
    _req(Y, "Y", torch.bfloat16); _req(out, "out", torch.bfloat16); _req(w, "w", torch.float32); _req(b, "b", torch.float32)
    _call("ttt_hip_post_forward_range", B, L, NH, F, ctypes.c_float(eps), _p(Y), _p(src), _p(w), _p(b), _p(out),
          int(t0), int(L - t0 if tn is None else tn), device=Y.device)


def post_backward(Y, dOut, src, w, dY, dw_part, db_part, eps):
    B, NH, L, F = Y.shape
    _req(Y, "Y", torch.bfloat16); _req(dOut, "dOut", torch.bfloat16); _req(dY, "dY", torch.bfloat16)
    _req(w, "w", torch.float32); _req(dw_part, "dw_part", torch.float32); _req(db_part, "db_part", torch.float32)
    _call("ttt_hip_post_backward", B, L, NH, F, ctypes.c_float(eps), _p(Y), _p(dOut), _p(src), _p(w), _p(dY), _p(dw_part), _p(db_part),
          device=Y.device)


def gate_forward(res, y, tanh_text, tanh_video, out, n_text):
    B, L, D = res.shape
    _req(res, "res", torch.bfloat16); _req(y, "y", torch.bfloat16); _req(out, "out", torch.bfloat16)
    _req(tanh_text, "tanh_text", torch.float32); _req(tanh_video, "tanh_video", torch.float32)
    _call("ttt_hip_gate_forward", B, L, D, int(n_text), _p(res), _p(y), _p(tanh_text), _p(tanh_video), _p(out), device=res.device)


def gate_backward_partials(D):
    return load_library().ttt_hip_gate_backward_partials(int(D))


def gate_backward(g, y, tanh_text, tanh_video, dy, dtanh_part, n_text):
    B, L, D = g.shape
    _req(g, "g", torch.bfloat16); _req(y, "y", torch.bfloat16); _req(dy, "dy", torch.bfloat16)
    _req(tanh_text, "tanh_text", torch.float32); _req(tanh_video, "tanh_video", torch.float32); _req(dtanh_part, "dtanh_part", torch.float32)
    _call("ttt_hip_gate_backward", B, L, D, int(n_text), _p(g), _p(y), _p(tanh_text), _p(tanh_video), _p(dy), _p(dtanh_part), device=g.device)


# ------------------------------------------------------------------------------------------------
# Segment self-attention (include/ttt_hip.h, "Segment self-attention"): strided [B, NH, S, 64] bf16 views, no copies.
class _AttnTensor(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("stride_b", ctypes.c_int64), ("stride_h", ctypes.c_int64), ("stride_s", ctypes.c_int64)]


class _AttnFwd(ctypes.Structure):
    _fields_ = [("Q", _AttnTensor), ("K", _AttnTensor), ("V", _AttnTensor), ("O", _AttnTensor), ("LSE", ctypes.c_void_p),
                ("B", ctypes.c_int32), ("NH", ctypes.c_int32), ("S", ctypes.c_int32), ("D", ctypes.c_int32), ("scale", ctypes.c_float)]


class _AttnBwd(ctypes.Structure):
    _fields_ = [(n, _AttnTensor) for n in ("Q", "K", "V", "O", "dO", "dQ", "dK", "dV")] + [
        ("LSE", ctypes.c_void_p), ("Delta", ctypes.c_void_p),
        ("B", ctypes.c_int32), ("NH", ctypes.c_int32), ("S", ctypes.c_int32), ("D", ctypes.c_int32), ("scale", ctypes.c_float)]


def _attn_tensor(t, name, shape):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.bfloat16:
        raise RuntimeError(f"{name}: expected a bfloat16 tensor on a HIP device (there is no CPU path)")
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if t.stride(3) != 1:
        raise RuntimeError(f"{name}: the head dimension must be contiguous")
    return _AttnTensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def attn_forward(q, k, v, out, lse, scale):
    """O = softmax(q k^T * scale) v for [B, NH, S, 64] bf16 views (any batch/head/token strides); lse [B,NH,S] fp32 or None."""
    B, NH, S, D = q.shape
    a = _AttnFwd(_attn_tensor(q, "q", (B, NH, S, D)), _attn_tensor(k, "k", (B, NH, S, D)), _attn_tensor(v, "v", (B, NH, S, D)),
                 _attn_tensor(out, "out", (B, NH, S, D)), lse.data_ptr() if lse is not None else None, B, NH, S, D, float(scale))
    if lse is not None:
        _req(lse, "lse", torch.float32)
    _call("ttt_hip_attn_forward", ctypes.byref(a), device=q.device)


def attn_backward(q, k, v, out, dout, lse, delta, dq, dk, dv, scale):
    B, NH, S, D = q.shape
    sh = (B, NH, S, D)
    _req(lse, "lse", torch.float32); _req(delta, "delta", torch.float32)
    a = _AttnBwd(_attn_tensor(q, "q", sh), _attn_tensor(k, "k", sh), _attn_tensor(v, "v", sh), _attn_tensor(out, "out", sh),
                 _attn_tensor(dout, "dout", sh), _attn_tensor(dq, "dq", sh), _attn_tensor(dk, "dk", sh), _attn_tensor(dv, "dv", sh),
                 lse.data_ptr(), delta.data_ptr(), B, NH, S, D, float(scale))
    _call("ttt_hip_attn_backward", ctypes.byref(a), device=q.device)


def attn_pre_forward(q_raw, k_raw, wq, bq, wk, bk, cos, sin, q, k, NH, n_text, eps):
    """Fused per-head LayerNorm(64) + RoPE of the attention's q and k; [B, S, NH*64] bf16 in and out."""
    B, S, D = q_raw.shape
    for t, n in ((q_raw, "q_raw"), (k_raw, "k_raw"), (q, "q"), (k, "k")):
        _req(t, n, torch.bfloat16)
    for t, n in ((wq, "wq"), (bq, "bq"), (wk, "wk"), (bk, "bk"), (cos, "cos"), (sin, "sin")):
        _req(t, n, torch.float32)
    if D != NH * 64 or cos.shape[-1] != 64 or cos.shape[0] < S - n_text:
        raise RuntimeError("attn_pre_forward: head_dim must be 64 and the rope tables must cover the video tokens")
    _call("ttt_hip_attn_pre_forward", B, S, NH, int(n_text), ctypes.c_float(eps), _p(q_raw), _p(k_raw), _p(wq), _p(bq), _p(wk), _p(bk),
          _p(cos), _p(sin), _p(q), _p(k), device=q_raw.device)


def attn_pre_partials(B, S, NH):
    return load_library().ttt_hip_attn_pre_partials(int(B), int(S), int(NH))


def attn_pre_backward(q_raw, k_raw, dq, dk, wq, wk, cos, sin, dq_raw, dk_raw, part, NH, n_text, eps, ld_out=None):
    """``ld_out``: as in ``pre_backward`` (dq_raw / dk_raw as column blocks of one [B, S, 3 D] buffer whose third block is dV)."""
    B, S, D = q_raw.shape
    for t, n in ((q_raw, "q_raw"), (k_raw, "k_raw")):
        _req(t, n, torch.bfloat16)
    for t, n in ((dq_raw, "dq_raw"), (dk_raw, "dk_raw")):
        if ld_out is None:
            _req(t, n, torch.bfloat16)
        else:
            _req_rows(t, n, (B, S, D), int(ld_out))
    for t, n in ((wq, "wq"), (wk, "wk"), (cos, "cos"), (sin, "sin"), (part, "part")):
        _req(t, n, torch.float32)
    tq, tk = _attn_tensor(dq, "dq", (B, NH, S, 64)), _attn_tensor(dk, "dk", (B, NH, S, 64))
    _call("ttt_hip_attn_pre_backward_ld", B, S, NH, int(n_text), ctypes.c_float(eps), _p(q_raw), _p(k_raw), ctypes.byref(tq), ctypes.byref(tk),
          _p(wq), _p(wk), _p(cos), _p(sin), _p(dq_raw), _p(dk_raw), ctypes.c_int64(D if ld_out is None else int(ld_out)), _p(part),
          device=q_raw.device)


# ------------------------------------------------------------------------------------------------
# TransformerLayer glue (include/ttt_hip.h, "TransformerLayer glue"): bf16 activations, fp32 parameter vectors.
def adaln_forward(vid, text, w, b, shift, scale1p, out, eps):
    B, Lv, D = vid.shape
    Lt = text.shape[1]
    for t, n in ((vid, "vid"), (text, "text"), (out, "out")):
        _req(t, n, torch.bfloat16)
    for t, n in ((w, "w"), (b, "b"), (shift, "shift"), (scale1p, "scale1p")):
        _req(t, n, torch.float32)
    if tuple(out.shape) != (B, Lt + Lv, D) or tuple(shift.shape) != (B, 2, D) or tuple(scale1p.shape) != (B, 2, D):
        raise RuntimeError("adaln_forward: out must be [B, Lt+Lv, D], shift / scale1p [B, 2, D]")
    _call("ttt_hip_adaln_forward", B, Lt, Lv, D, ctypes.c_float(eps), _p(vid), _p(text), _p(w), _p(b), _p(shift), _p(scale1p), _p(out),
          device=vid.device)


def adaln_backward_partials():
    return load_library().ttt_hip_adaln_backward_partials()


def adaln_backward(vid, text, dout, w, b, scale1p, dvid, dtext, part, eps):
    B, Lv, D = vid.shape
    Lt = text.shape[1]
    for t, n in ((vid, "vid"), (text, "text"), (dout, "dout"), (dvid, "dvid"), (dtext, "dtext")):
        _req(t, n, torch.bfloat16)
    for t, n in ((w, "w"), (b, "b"), (scale1p, "scale1p"), (part, "part")):
        _req(t, n, torch.float32)
    _call("ttt_hip_adaln_backward", B, Lt, Lv, D, ctypes.c_float(eps), _p(vid), _p(text), _p(dout), _p(w), _p(b), _p(scale1p),
          _p(dvid), _p(dtext), _p(part), device=vid.device)


def resgate_forward(vid, text, y, gate, ovid, otext):
    B, Lv, D = vid.shape
    Lt = text.shape[1]
    for t, n in ((vid, "vid"), (text, "text"), (y, "y"), (ovid, "ovid"), (otext, "otext")):
        _req(t, n, torch.bfloat16)
    _req(gate, "gate", torch.float32)
    if tuple(y.shape) != (B, Lt + Lv, D) or tuple(gate.shape) != (B, 2, D):
        raise RuntimeError("resgate_forward: y must be [B, Lt+Lv, D], gate [B, 2, D]")
    _call("ttt_hip_resgate_forward", B, Lt, Lv, D, _p(vid), _p(text), _p(y), _p(gate), _p(ovid), _p(otext), device=vid.device)


def resgate_backward_partials(D):
    return load_library().ttt_hip_resgate_backward_partials(int(D))


def resgate_backward(dvid, dtext, y, gate, dy, dgate_part):
    B, Lv, D = dvid.shape
    Lt = dtext.shape[1]
    for t, n in ((dvid, "dvid"), (dtext, "dtext"), (y, "y"), (dy, "dy")):
        _req(t, n, torch.bfloat16)
    _req(gate, "gate", torch.float32); _req(dgate_part, "dgate_part", torch.float32)
    _call("ttt_hip_resgate_backward", B, Lt, Lv, D, _p(dvid), _p(dtext), _p(y), _p(gate), _p(dy), _p(dgate_part), device=dvid.device)
