"""Loader for the native library (libptx355.so).  There is no fallback: if the HIP library is missing
or cannot be loaded, importing the device API fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libptx355.so")


class PtxStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("bounces", C.c_uint64), ("shadow_rays", C.c_uint64)]


# Every symbol include/ptx355.h declares: (name, restype, argtypes)
_P = C.c_void_p
_U = C.c_uint32
SYMBOLS = [
    ("ptx_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("ptx_destroy", None, [_P]),
    ("ptx_last_error", C.c_char_p, [_P]),
    ("ptx_upload_scene", C.c_int, [_P, _P, _U, _P, _U, _P, _U, _P, _U, _P, _U, _P, _U, _P, _U, _P, _U]),
    ("ptx_resize", C.c_int, [_P, _U, _U]),
    ("ptx_set_tile", C.c_int, [_P, _U, _U, _U, _U]),
    ("ptx_set_partition", C.c_int, [_P, _U, _U]),
    ("ptx_clear", C.c_int, [_P]),
    ("ptx_trace", C.c_int, [_P, _P]),
    ("ptx_trace_n", C.c_int, [_P, _P, _U, _U, _U]),
    ("ptx_trace_n_static", C.c_int, [_P, _P, _U, _U, _U]),
    ("ptx_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("ptx_sync", C.c_int, [_P]),
    ("ptx_read_accum", C.c_int, [_P, _P]),
    ("ptx_write_accum", C.c_int, [_P, _P, _P]),
    ("ptx_read_output", C.c_int, [_P, _P]),
    ("ptx_read_primary_ids", C.c_int, [_P, _P]),
    ("ptx_bvh_build", C.c_int, [_P, _P, _U, _P, C.POINTER(_U), _P]),
    ("ptx_debug_capture", C.c_int, [_P, _U]),
    ("ptx_read_debug_radiance", C.c_int, [_P, _P, _U]),
    ("ptx_enable_moments", C.c_int, [_P, C.c_int]),
    ("ptx_read_moments", C.c_int, [_P, _P]),
    ("ptx_device_accum", _P, [_P]),
    ("ptx_device_moments", _P, [_P]),
    ("ptx_merge_moments", C.c_int, [_P, _U, _U, _U, _P, _P, _U, _P]),
    ("ptx_tonemap", C.c_int, [_P, _U, _U, _P]),
    ("ptx_comm_unique_id", C.c_int, [_P]),
    ("ptx_comm_init", C.c_int, [_P, _P, _U, _U]),
    ("ptx_comm_adopt", C.c_int, [_P, _P]),
    ("ptx_comm_free", C.c_int, [_P]),
    ("ptx_merge_sample_split", C.c_int, [_P, _P, _P]),
    ("ptx_assemble_tiles", C.c_int, [_P, _P]),
    ("ptx_slice_of", None, [C.c_uint64, _U, _U, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("ptx_post_process", C.c_int, [_P, _P]),
    ("ptx_read_stats", C.c_int, [_P, C.POINTER(PtxStats)]),
    ("ptx_trace_kernel_name", C.c_char_p, [_P]),
    ("ptx_last_launch_info", C.c_char_p, [_P]),
    ("ptx_launch_plan", C.c_char_p, [_P, _U]),
    ("ptx_kernel_time", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(_U)]),
    ("ptx_kernel_times", C.c_int, [_P, _P, _U, C.POINTER(_U)]),
    ("ptx_kat_math", C.c_int, [_P, C.c_int, _U, _P, _P, _P]),
    ("ptx_kat_exhaustive", C.c_int, [_P, C.c_int, _P, _P]),
    ("ptx_kat_trace", C.c_int, [_P, _P, _U, _P, _P, _P, _P, _P]),
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc, gfx950). There is no CPU fallback.")
        # PyTorch wheels bundle their own HIP runtime.  When a process uses both (multi-GPU runs use
        # torch.distributed for RCCL), torch has to be loaded first so that both resolve to one runtime;
        # loaded the other way round torch later reports "No HIP GPUs are available".
        # (PTX355_NO_TORCH=1: a process that must not load torch's bundled libraries, e.g. one that brings its own RCCL.)
        if os.environ.get("PTX355_NO_TORCH", "") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
