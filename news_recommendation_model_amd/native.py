"""ctypes binding of libnrm_hotpath.so (C ABI: include/nrm_hotpath.h).

There is deliberately NO fallback: if the library is missing or a call fails, a RuntimeError is
raised -- the product path never silently runs anything but the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRM_HOTPATH_LIB") or os.path.join(_HERE, "libnrm_hotpath.so")   # override: diagnostic builds (scripts/_diag)
ABI_VERSION = 6

_c_fp = ctypes.c_void_p      # device pointers travel as integers
_c_i, _c_l = ctypes.c_int, ctypes.c_long

# name -> (restype, argtypes); must list every symbol include/nrm_hotpath.h declares
SIGNATURES = {
    "nrm_abi_version": (_c_i, []),
    "nrm_last_error": (ctypes.c_char_p, []),
    "nrm_build_flags": (_c_i, []),
    "nrm_source_digest": (ctypes.c_char_p, []),
    "nrm_build_info": (ctypes.c_char_p, []),
    "nrm_pwattn_packed_floats": (_c_l, [_c_i]),
    "nrm_pwattn_pack_wp": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp]),
    "nrm_pwattn_fwd": (_c_i, [_c_fp] * 9 + [_c_i] * 5 + [_c_fp]),
    "nrm_pwattn_bwd_dz": (_c_i, [_c_fp] * 7 + [_c_i] * 5 + [_c_fp]),
    "nrm_pwattn_bwd_rw_supported": (_c_i, [_c_i, _c_i]),
    "nrm_pwattn_bwd_rw_packed_floats": (_c_l, [_c_i, _c_i]),
    "nrm_pwattn_bwd_rw_pack": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp]),
    "nrm_pwattn_bwd_rw_dtdh": (_c_i, [_c_fp] * 6 + [_c_i] * 5 + [_c_fp]),
    "nrm_pwattn_bwd_dp_supported": (_c_i, [_c_i, _c_i]),
    "nrm_pwattn_bwd_dp_packed_floats": (_c_l, [_c_i, _c_i]),
    "nrm_pwattn_bwd_dp_pack": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_fp, _c_fp]),
    "nrm_pwattn_bwd_dp_dtdh": (_c_i, [_c_fp] * 6 + [_c_i] * 4 + [_c_fp]),
    "nrm_pwattn_bwd_nsplit": (_c_i, [_c_i] * 5),
    "nrm_pwattn_bwd_contract": (_c_i, [_c_fp] * 4 + [_c_i] + [_c_fp] * 3 + [_c_i] * 7 + [_c_fp]),
    "nrm_gemm_packed_floats": (_c_l, [_c_i, _c_i, _c_i]),
    "nrm_gemm_nt_bf16_supported": (_c_i, [_c_i, _c_i, _c_i]),
    "nrm_gemm_pack": (_c_i, [_c_fp, _c_l, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "nrm_gemm_nt": (_c_i, [_c_fp, _c_i, _c_i, _c_fp, _c_i, _c_i, _c_fp, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_i, _c_i, _c_fp]),
    "nrm_slab_reduce": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_fp, _c_l, _c_l, _c_fp, _c_l, _c_l, ctypes.c_float,
                               _c_fp, _c_fp, _c_fp]),
    "nrm_bn_finalize": (_c_i, [_c_i, _c_fp, _c_fp, _c_fp, _c_i, _c_i, ctypes.c_float, ctypes.c_float, _c_fp]),
    "nrm_mul_bwd": (_c_i, [_c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_fp, _c_i, _c_i, _c_i, _c_fp]),
    "nrm_gemm_tn_nsplit": (_c_i, [_c_i, _c_i, _c_i, _c_i]),
    "nrm_gemm_tn": (_c_i, [_c_fp, _c_i, _c_i, _c_fp, _c_i, _c_i, _c_i, _c_fp, _c_i, _c_fp, _c_fp, _c_l, _c_i, _c_fp]),
    "nrm_gemm_pack_multi": (_c_i, [_c_fp, _c_i, _c_fp]),
    "nrm_slab_reduce_multi": (_c_i, [_c_fp, _c_i, _c_fp]),
    "nrm_colreduce": (_c_i, [_c_i] + [_c_fp] * 6 + [_c_i] * 3 + [_c_fp]),
    "nrm_bn_apply": (_c_i, [_c_fp] * 6 + [_c_i] * 3 + [_c_fp]),
    "nrm_bn_backward": (_c_i, [_c_fp] * 9 + [_c_i] * 4 + [_c_fp]),
    "nrm_concat_cols": (_c_i, [_c_fp, _c_fp, _c_fp, _c_i, _c_fp, _c_i, _c_l, _c_fp]),
    "nrm_pool_bmm": (_c_i, [_c_fp, _c_l, _c_l, _c_l, _c_fp, _c_i, _c_fp] + [_c_i] * 5 + [_c_fp]),
    "nrm_pool_rowdot": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp] + [_c_i] * 4 + [_c_fp, _c_i, _c_fp]),
    "nrm_small_linear_relu_fwd": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp, _c_fp, _c_l, _c_i, _c_i, _c_i, _c_fp]),
    "nrm_small_linear_relu_bwd": (_c_i, [_c_fp, _c_i, _c_fp, _c_fp, _c_fp, _c_i, _c_l, _c_i, _c_i, _c_fp, _c_fp]),
    "nrm_loss_fwd_bwd": (_c_i, [_c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_fp, _c_l, ctypes.c_float, _c_i, _c_i, _c_fp, _c_fp, _c_i, _c_fp, _c_fp, _c_fp]),
    "nrm_adam_step": (_c_i, [_c_fp] * 4 + [_c_l] + [ctypes.c_float] * 5 + [_c_i, _c_i, _c_fp]),
    "nrm_adam_step_dev": (_c_i, [_c_fp] * 4 + [_c_l] + [ctypes.c_float] * 5 + [_c_fp, _c_i, _c_fp]),
    "nrm_gather_flat": (_c_i, [_c_fp, _c_fp, _c_fp, _c_i, _c_fp, _c_l, _c_fp]),
    "nrm_row_auc": (_c_i, [_c_fp] * 3 + [_c_i, _c_i] + [_c_fp] * 3),
    "nrm_frontend_fwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,  _c_fp, _c_i, _c_i,  _c_fp, _c_fp, _c_i,
                                 _c_fp, _c_i, _c_i,  _c_fp, _c_fp, _c_fp, _c_fp,  _c_i, _c_i, _c_i, _c_i, _c_i,
                                 _c_fp, _c_i, _c_fp, _c_i, _c_fp, _c_fp]),
    "nrm_frontend_cat_ws_ints": (_c_l, [_c_i, _c_l, _c_i]),
    "nrm_frontend_cat_grad": (_c_i, [_c_fp, _c_i, _c_i, _c_fp, _c_i] * 2 + [_c_i] * 5 + [_c_fp, _c_fp, _c_fp]),
    "nrm_frontend_bwd": (_c_i, [_c_fp, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,  _c_fp, _c_i, _c_fp, _c_fp]
                         + [_c_i] * 10 + [_c_fp] * 8 + [_c_fp]),
}



class PackDesc(ctypes.Structure):
    """nrm_pack_desc of include/nrm_hotpath.h"""
    _fields_ = [("src", ctypes.c_void_p), ("src2", ctypes.c_void_p), ("sign2", ctypes.c_float),
                ("row_stride", ctypes.c_long), ("col_stride", ctypes.c_long), ("nrows", ctypes.c_int), ("ncols", ctypes.c_int),
                ("packed", ctypes.c_void_p), ("mma", ctypes.c_int)]


class SlabDesc(ctypes.Structure):
    """nrm_slab_desc of include/nrm_hotpath.h"""
    _fields_ = [("ws", ctypes.c_void_p), ("nsplit", ctypes.c_int), ("nj", ctypes.c_int), ("ldws", ctypes.c_int), ("ni", ctypes.c_int),
                ("out", ctypes.c_void_p), ("out_istride", ctypes.c_long), ("out_jstride", ctypes.c_long),
                ("out2", ctypes.c_void_p), ("out2_istride", ctypes.c_long), ("out2_jstride", ctypes.c_long), ("sign2", ctypes.c_float),
                ("vec", ctypes.c_void_p), ("vec_out", ctypes.c_void_p)]


_lib = None
_lock = threading.Lock()


def load():
    """Load (once) and return the ctypes library; raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m news_recommendation_model_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if lib.nrm_abi_version() != ABI_VERSION:
            raise RuntimeError(f"ABI mismatch: library {lib.nrm_abi_version()} != binding {ABI_VERSION}; rebuild")
        flags = lib.nrm_build_flags()
        if flags and os.environ.get("NRM_ALLOW_DIAG_LIB") != "1":
            raise RuntimeError(
                f"{LIB_PATH} is a timing-diagnostic build (nrm_build_flags() = {flags:#x}: kernels with parts of their work "
                "removed, results are wrong by construction); set NRM_ALLOW_DIAG_LIB=1 to load it for timing experiments")
        # provenance: the binary must have been built from the kernel sources lying next to it (they travel with it to the GPU box)
        if os.environ.get("NRM_ALLOW_STALE_LIB") != "1" and not os.environ.get("NRM_HOTPATH_LIB"):
            from . import build as _build
            if os.path.isdir(_build.CSRC):
                have, want = lib.nrm_source_digest().decode(), _build.sources_digest()
                if have != want:
                    raise RuntimeError(f"{LIB_PATH} was built from other kernel sources (library {have[:16]}..., sources {want[:16]}...): "
                                       "rebuild with `python -m news_recommendation_model_amd.build` (NRM_ALLOW_STALE_LIB=1 loads it anyway)")
        _lib = lib
    return _lib


def provenance():
    """{'library_sources_sha256', 'build_info'} of the loaded library."""
    lib = load()
    return {"library_sources_sha256": lib.nrm_source_digest().decode(), "build_info": lib.nrm_build_info().decode()}


# Optional per-launch timing (bench.py): when a list is installed here every kernel-launching call is
# bracketed by two events recorded on the stream the kernels are launched on (torch's current stream).
# kernel_event_tags (a set, or None = every call) limits the bracketing: an event pair costs a few microseconds of GPU time
# per launch (1.5 % of a C3 step, 3-4x on the small shapes when all ~600 launches of a step are bracketed).
kernel_events = None
kernel_event_tags = None


def call(name, *args, tag=None):
    """Call an int-returning entry point; non-zero return raises with the library's message."""
    lib = load()
    if kernel_events is not None and (kernel_event_tags is None or (tag or name) in kernel_event_tags):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        kernel_events.append((tag or name, e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.nrm_last_error().decode()}")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
