"""ctypes binding of libmpnn_amd.so (the C ABI declared in include/mpnn_amd.h).

There is no CPU fallback: if the shared object is missing and cannot be built, or a tensor
is not a contiguous device tensor, the call raises.
"""
import ctypes
import os
import re

import torch

from . import build as _build

_HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "mpnn_amd.h")

c_f = ctypes.c_void_p     # const float* / float*
c_i = ctypes.c_void_p     # const int32_t* / int32_t*
c_v = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int
usz = ctypes.c_size_t

_SIGNATURES = {
    "mpnn_version": (ctypes.c_int, []),
    "mpnn_last_error_string": (ctypes.c_char_p, []),
    "mpnn_init": (ctypes.c_int, []),
    "mpnn_csr_workspace_bytes": (usz, [i64]),
    "mpnn_csr_count": (ctypes.c_int, [c_f, c_f, i64, i32, i32, c_i, c_v, usz, c_v]),
    "mpnn_csr_fill": (ctypes.c_int, [c_f, c_f, i64, i32, i32, c_i, c_i, c_f, c_f, c_v]),
    "mpnn_edge_message_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, i64, i64, i32, i32, i32, c_v]),
    "mpnn_edge_message_bwd_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f,
                                                 i64, i64, i32, i32, i32, c_v]),
    "mpnn_edge_message_agg_bwd_da_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_f, c_i, c_i, c_f, c_f,
                                                        i64, i64, i32, i32, i32, c_v]),
    "mpnn_edge_message_agg_bwd_dgate_f32": (ctypes.c_int, [c_f, c_f, c_f, c_i, c_i, c_f, c_i, c_i, c_f,
                                                           i64, i64, i32, i32, i32, c_v]),
    "mpnn_att_gate_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_f, i64, i64, i32, i32, c_v]),
    "mpnn_att_gate_bwd_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_f, c_f, i64, i64, i32, i32, c_v]),
    "mpnn_tower_chain_f32": (ctypes.c_int, [c_f, c_f, c_f, i32, i32, i32, c_v]),
    "mpnn_tower_chain_bwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, i32, i32, i32, c_v]),
    "mpnn_masked_bn_workspace_bytes": (usz, [i32]),
    "mpnn_masked_bn_fwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, ctypes.c_float, i32,
                                              c_v, usz, c_v]),
    "mpnn_masked_bn_bwd_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, ctypes.c_float,
                                              i32, c_f, c_v, usz, c_v]),
    "mpnn_plan_tiles_host": (i64, [c_v, i64, i32, c_v]),
    "mpnn_message_aggregate_tile_atoms": (ctypes.c_int, []),
    "mpnn_message_aggregate_max_types": (ctypes.c_int, []),
    "mpnn_message_aggregate_max_row_tiles": (ctypes.c_int, []),
    "mpnn_message_aggregate_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_i, c_f, i64, i64, i32, i32, i32, c_v]),
    "mpnn_message_aggregate_bwd_da_f32": (ctypes.c_int, [c_f, c_f, c_i, c_i, c_i, c_i, c_f, i64, i64, i32, i32, i32, c_v]),
    "mpnn_bilinear_message_f32": (ctypes.c_int, [c_f, c_f, c_f, i64, i32, i32, c_v]),
    "mpnn_segsum_f32": (ctypes.c_int, [c_f, c_i, c_f, c_f, i64, i32, c_v]),
    "mpnn_segsum_bwd_f32": (ctypes.c_int, [c_f, c_i, c_f, c_f, i64, i32, c_v]),
    "mpnn_segsum_gather_f32": (ctypes.c_int, [c_f, c_i, c_i, c_f, c_f, i64, i32, c_v]),
    "mpnn_gru_update_f32": (ctypes.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, i64, i32, c_v]),
    "mpnn_gru_bwd_workspace_bytes": (usz, [i64, i32]),
    "mpnn_gru_update_bwd_f32": (ctypes.c_int, [c_f] * 7 + [c_f] * 6 + [c_v, usz, i64, i32, c_v]),
}

_lib = None


class MpnnError(RuntimeError):
    pass


def declared_symbols():
    """Every function name include/mpnn_amd.h declares (used by the CPU-side export test)."""
    with open(_HEADER) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(mpnn_[a-z0-9_]+)\s*\(", text)))


def library_path():
    return _build.LIB


def load():
    """Load (building first if stale/missing) the C-ABI library; raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        path = _build.build()          # raises when hipcc is absent
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError => the .so is out of date: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MpnnError("mpnn_amd kernels need device tensors; got a %s tensor (no CPU fallback)" % t.device)
    if not t.is_contiguous():
        raise MpnnError("mpnn_amd kernels need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise MpnnError("expected dtype %s, got %s" % (dtype, t.dtype))
    return ctypes.c_void_p(t.data_ptr())


def fptr(t):
    return ptr(t, torch.float32)


def iptr(t):
    return ptr(t, torch.int32)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(code, what):
    if code != 0:
        msg = load().mpnn_last_error_string().decode("utf-8", "replace")
        raise MpnnError("%s failed (%d): %s" % (what, code, msg))
