"""ctypes binding of ``libmmeeg_hip.so`` (C ABI: include/mmeeg_hip.h).

Plain pointers and sizes cross the boundary; device pointers are borrowed from
torch tensors and every call is asynchronous on torch's current HIP stream.
There is no fallback: a missing library or a non-zero return code raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List

import torch

_LIB_NAME = "libmmeeg_hip.so"
_lib = None

# signature codes: p = device/host pointer, i = int, f = float, u = uint32, l = int64
# (the trailing hipStream_t is implicit for every entry except those in _NO_STREAM)
_SIGS: Dict[str, str] = {
    "mm_pack_nct_bf16": "ppiiii",
    "mm_unpack_ntc_f32": "ppiiii",
    "mm_prep_conv_weight": "pppiiiii",
    "mm_conv1d_fwd": "ppiiiiiippippippppfu",
    "mm_conv1d_wgrad": "pppiiiiiip",
    "mm_bn_finalize": "pppppppiffi",
    "mm_bn_act_fwd": "pppppppiiiiiifui",
    "mm_bn_act_bwd_reduce": "pppppppiiiiiifui",
    "mm_bn_act_bwd_apply": "ppppppppiiiiiifuii",
    "mm_layernorm_fwd": "ppppppiif",
    "mm_layernorm_bwd": "ppppppppppii",
    "mm_attn_fwd": "pppiiiif",
    "mm_attn_bwd": "ppppppiiiif",
    "mm_colsum": "pppii",
    "mm_meanpool_fwd": "ppiii",
    "mm_meanpool_bwd": "ppiii",
    "mm_act_bwd": "ppppiiifu",
    "mm_small_linear_fwd": "pppppiiiiifu",
    "mm_cast_bf16": "ppl",
    "mm_cast_f32": "ppl",
    "mm_add_pe": "ppppiiifu",
    "mm_conv3d_direct_fwd": "pppppiiiiii",
    "mm_conv3d_direct_wgrad": "ppppiiiii",
    "mm_conv3d_fwd": "pppiiiiiippp",
    "mm_conv3d_wgrad": "pppiiiiiip",
    "mm_pool3d_bn_act_fwd": "ppppppppiiiiiiifui",
    "mm_pool3d_bn_act_bwd_reduce": "pppppppiiiiiiifui",
    "mm_pool3d_bn_act_bwd_apply": "ppppppppiiiiiiifuii",
    "mm_prep_conv3d_weight": "pppiiii",
    "mm_proj_head_fwd": "ppppppppiiifu",
    "mm_proj_head_bwd": "pppppppppppiiifu",
    "mm_clip_loss": "pppppppppiiifi",
    "mm_bridge_fwd": "ppppppppi",
    "mm_adamw_clip": "ppppppplffffffi",
    "mm_sumsq": "ppl",
    "mm_scale_inplace": "ppl",
}
_NO_STREAM: List[str] = []
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float,
       "u": ctypes.c_uint32, "l": ctypes.c_int64}


class HipLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


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
    for name, sig in _SIGS.items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue                      # checked by tests/test_abi.py against the header
        fn.restype = ctypes.c_int
        fn.argtypes = [_CT[c] for c in sig] + ([] if name in _NO_STREAM else [ctypes.c_void_p])
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
    if name not in _NO_STREAM:
        conv.append(torch.cuda.current_stream().cuda_stream)
    rc = fn(*conv)
    if rc != 0:
        raise HipLibraryError(f"{name} failed ({rc}): {lib.mm_last_error().decode()}")
