"""ctypes binding of ``libmmeeg_hip.so`` (C ABI: include/mmeeg_hip.h).

Plain pointers and sizes cross the boundary; device pointers are borrowed from
torch tensors and every call is asynchronous on torch's current HIP stream.
There is no fallback: a missing library or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict

import torch

_LIB_NAME = "libmmeeg_hip.so"
_lib = None

# Signatures are parsed from include/mmeeg_hip.h (single source of truth):
# pointer -> c_void_p, int -> c_int, float -> c_float, uint32_t -> c_uint32,
# int64_t -> c_int64; the trailing hipStream_t is supplied by call().
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float,
       "u": ctypes.c_uint32, "l": ctypes.c_int64}
_SIGS: Dict[str, str] = {}


def header_path() -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "include", "mmeeg_hip.h")


def parse_header(path: str = None) -> Dict[str, str]:
    """{'mm_name': 'ppii...'} for every ``int mm_*(..., hipStream_t stream);``."""
    text = open(path or header_path()).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\bint\s+(mm_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
        if not args or "hipStream_t" not in args[-1]:
            continue
        code = ""
        for a in args[:-1]:
            if "*" in a:
                code += "p"
            elif a.startswith("int64_t"):
                code += "l"
            elif a.startswith("uint32_t"):
                code += "u"
            elif a.startswith("float"):
                code += "f"
            elif a.startswith("int"):
                code += "i"
            else:
                raise ValueError(f"{name}: cannot map argument '{a}'")
        sigs[name] = code
    return sigs


class HipLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    """in-tree library; MMEEG_HIP_LIB points tools/kbench.py at an ablation build instead"""
    return os.environ.get("MMEEG_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load():
    """Load the shared library once; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found: build it with multimodal_eeg_fmri_amd/csrc/build.sh "
            "(or __graft_entry__.build()). This package has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(path)
    lib.mm_last_error.restype = ctypes.c_char_p
    lib.mm_abi_version.restype = ctypes.c_int
    _SIGS.update(parse_header())
    missing = [n for n in _SIGS if not hasattr(lib, n)]
    if missing:
        raise HipLibraryError(f"{path} is stale: missing symbols {missing}; rebuild it")
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [_CT[c] for c in sig] + [ctypes.c_void_p]
    _lib = lib
    return lib


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if not x.is_cuda:
            raise HipLibraryError("HIP path needs GPU tensors (got a CPU tensor); "
                                  "there is no CPU fallback in this package")
        if not x.is_contiguous():
            raise HipLibraryError("HIP path needs contiguous tensors")
        return x.data_ptr()
    return int(x)


def call(name: str, *args):
    """Invoke ``name`` on torch's current stream; raise on a non-zero code."""
    lib = load()
    fn = getattr(lib, name)
    sig = _SIGS[name]
    if len(args) != len(sig):
        raise HipLibraryError(f"{name}: expected {len(sig)} args, got {len(args)}")
    conv = [(_ptr(a) if c == "p" else a) for a, c in zip(args, sig)]
    conv.append(torch.cuda.current_stream().cuda_stream)
    rc = fn(*conv)
    if rc != 0:
        raise HipLibraryError(f"{name} failed ({rc}): {lib.mm_last_error().decode()}")
