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

_SCALARS = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
            "float": ctypes.c_float, "double": ctypes.c_double}


def _ctype(decl, is_return=False):
    """ctypes type of one C declaration of the header ("const float* h", "int64_t V", "void* stream" ...)."""
    decl = decl.strip()
    if "*" in decl:
        if is_return and re.match(r"const\s+char\s*\*", decl):
            return ctypes.c_char_p
        return ctypes.c_void_p                         # every pointer argument: a device (or host) address
    words = [w for w in re.split(r"\s+", decl) if w not in ("const", "unsigned")]
    base = words[0]
    if base not in _SCALARS:
        raise RuntimeError("include/mpnn_amd.h: no ctypes mapping for %r" % decl)
    return _SCALARS[base]


def header_signatures():
    """{name: (restype, [argtypes])} of every prototype in include/mpnn_amd.h -- the header is the single statement of
    the ABI; the binding is derived from it, so an argument added, removed or retyped there cannot drift here."""
    with open(_HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)          # preprocessor lines
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    sigs = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mpnn_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        ret = re.sub(r"\b(extern|static|inline)\b", " ", ret).strip()
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        sigs[name] = (_ctype(ret, is_return=True), argtypes)
    return sigs


_lib = None


class MpnnError(RuntimeError):
    pass


def declared_symbols():
    """Every function name include/mpnn_amd.h declares (used by the CPU-side export test)."""
    return sorted(header_signatures())


def library_path():
    return _build.LIB


def load():
    """Load the C-ABI library, building it first when it is missing or STALE (its manifest differs from the hash of the
    sources, headers and flags now in the tree: build._stale()); raises if that is impossible -- a library that was not
    built from this tree is never loaded silently."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if _build._stale():
        try:
            path = _build.build()      # raises when hipcc is absent or a source does not compile
        except Exception as e:
            raise MpnnError("libmpnn_amd.so is missing or was not built from the sources in this tree, and rebuilding "
                            "it failed (%s); run `python -m mpnn_amd.build`" % e)
    lib = ctypes.CDLL(path)
    for name, (res, args) in header_signatures().items():
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
