"""TEST INFRASTRUCTURE — ctypes front end of the CPU oracle (oracle/pt_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from . import host_prep as hp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")


class PtoScene(C.Structure):
    _fields_ = [
        ("figs", C.c_void_p), ("nfig", C.c_uint32),
        ("mats", C.c_void_p), ("nmat", C.c_uint32),
        ("nodes", C.c_void_p), ("nnode", C.c_uint32),
        ("prims", C.c_void_p), ("nprim", C.c_uint32),
        ("lights", C.c_void_p), ("nlight", C.c_uint32),
        ("infinite", C.c_void_p), ("ninf", C.c_uint32),
        ("tex_pixels", C.c_void_p), ("npix", C.c_uint32),
        ("tex_infos", C.c_void_p), ("ntex", C.c_uint32),
    ]


COUNTER_FIELDS = (["paths", "bounces", "shadow_rays", "node_visits", "aabb_tests"]
                  + [f"prim_tests_{i}" for i in range(17)] + [f"sdf_evals_{i}" for i in range(17)]
                  + ["fractal_inner", "eval_brdf", "sample_brdf", "sample_glass", "nee", "plane_tests_infinite"])


def build(force: bool = False) -> str:
    """Compile oracle/libpt_oracle.so with the committed Makefile if it is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("pt_oracle.cpp", "dmath.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpt_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.pto_render.restype = C.c_int
        _lib.pto_pcg_hash.restype = C.c_uint32
        _lib.pto_pcg_hash.argtypes = [C.c_uint32]
        _lib.pto_max_threads.restype = C.c_int
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """Keeps the numpy buffers alive and exposes them as a PtoScene."""

    def __init__(self, sb: hp.SceneBuffers):
        self.sb = sb
        self._keep = [np.ascontiguousarray(x) for x in
                      (sb.figs, sb.mats, sb.nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)]
        k = self._keep
        self.c = PtoScene(_p(k[0]), len(k[0]), _p(k[1]), len(k[1]), _p(k[2]), len(k[2]), _p(k[3]), len(k[3]),
                          _p(k[4]), len(k[4]), _p(k[5]), len(k[5]), _p(k[6]), len(k[6]), _p(k[7]), len(k[7]))


def render(scene: OracleScene, cam: np.ndarray, n: int, first_frame: int = 0, first_sample: int = 1,
           tile=None, accum: Optional[np.ndarray] = None, want_radiance: bool = False, want_m2: bool = False,
           m2: Optional[np.ndarray] = None, counters: bool = False, threads: int = 0) -> dict:
    """Runs n samples per pixel over `tile` (x0, y0, w, h; default full frame) the way n successive
    frames of the reference would (sample k: sample_count = first_sample + k, frame_index = first_frame + k)."""
    W, H = int(cam[0]["width"]), int(cam[0]["height"])
    x0, y0, tw, th = tile if tile is not None else (0, 0, W, H)
    if accum is None:
        accum = np.zeros((H, W, 4), dtype=np.float32)
    rgba8 = np.zeros((H, W, 4), dtype=np.uint8)
    pid = np.full((H, W), 0xFFFFFFFF, dtype=np.uint32)
    pt = np.zeros((H, W), dtype=np.float32)
    bc = np.zeros((H, W), dtype=np.uint32)
    rad = np.zeros((n, th, tw, 3), dtype=np.float32) if want_radiance else None
    if want_m2 and m2 is None:
        m2 = np.zeros((H, W, 3), dtype=np.float32)
    cnt = (C.c_uint64 * len(COUNTER_FIELDS))() if counters else None
    rc = lib().pto_render(C.byref(scene.c), _p(cam), C.c_uint32(first_frame), C.c_uint32(first_sample), C.c_uint32(n),
                          C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(tw), C.c_uint32(th), _p(accum), _p(rgba8), _p(pid),
                          _p(pt), _p(rad), _p(m2), _p(bc), cnt, C.c_int(threads))
    if rc != 0:
        raise RuntimeError(f"pto_render failed: {rc}")
    out = {"accum": accum, "rgba8": rgba8, "primary_id": pid, "primary_t": pt, "bounces": bc, "radiance": rad, "m2": m2}
    if counters:
        out["counters"] = {k: int(v) for k, v in zip(COUNTER_FIELDS, cnt)}
    return out


MATH_OPS = {"sin": 0, "cos": 1, "asin": 2, "acos": 3, "log": 4, "log2": 5, "exp2": 6, "atan2": 7, "pow": 8,
            "min": 9, "max": 10, "sqrt": 11, "div": 12, "fma_aba": 13, "rcp": 14}


def math_op(name: str, a, b=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    out = np.zeros_like(a)
    lib().pto_math(C.c_int(MATH_OPS[name]), C.c_uint32(a.size), _p(a), _p(b), _p(out))
    return out


def rand_sequence(px: int, py: int, frame: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float32)
    lib().pto_rand_sequence(C.c_uint32(px), C.c_uint32(py), C.c_uint32(frame), C.c_uint32(n), _p(out))
    return out


def intersect(scene: OracleScene, cam: np.ndarray, fig_idx: int, rays: np.ndarray) -> np.ndarray:
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    out = np.zeros((len(rays), 10), dtype=np.float32)
    lib().pto_intersect(C.byref(scene.c), _p(cam), C.c_uint32(fig_idx), C.c_uint32(len(rays)), _p(rays), _p(out))
    return out


def trace(scene: OracleScene, cam: np.ndarray, rays: np.ndarray, max_t: Optional[np.ndarray] = None):
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    n = len(rays)
    hid = np.zeros(n, dtype=np.uint32)
    ht = np.zeros(n, dtype=np.float32)
    sh = np.zeros(n, dtype=np.uint32) if max_t is not None else None
    mt = np.ascontiguousarray(max_t, dtype=np.float32) if max_t is not None else None
    lib().pto_trace(C.byref(scene.c), _p(cam), C.c_uint32(n), _p(rays), _p(mt), _p(hid), _p(ht), _p(sh))
    return hid, ht, sh


def tonemap(cam: np.ndarray, rgb: np.ndarray) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((len(rgb), 4), dtype=np.uint8)
    lib().pto_tonemap(_p(cam), C.c_uint32(len(rgb)), _p(rgb), _p(out))
    return out


EFFECTS = {"none": 0, "negative": 1, "sepia": 2, "grayscale": 3, "fxaa": 4, "oil": 5, "bw": 6, "comic": 7, "casting": 8}


def post_params(width: int, height: int, effects, oil_radius: int = 3, comic_levels: int = 4) -> np.ndarray:
    """AppState::build_post_params (app/state.rs:448-466): the 16-word uniform of the post-process pass."""
    p = np.zeros(16, dtype=np.uint32)
    p[0], p[1] = width, height
    ids = [EFFECTS[e] if isinstance(e, str) else int(e) for e in effects][:8]
    p[2], p[3] = len(ids), oil_radius
    p[4:4 + len(ids)] = ids
    p[12] = comic_levels
    return p


def post_process(accum: np.ndarray, params: np.ndarray, threads: int = 0) -> np.ndarray:
    """The reference's post-process pass over an (h, w, 4) f32 accumulation image; returns (h, w, 4) u8."""
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    h, w = accum.shape[:2]
    params = np.ascontiguousarray(params, dtype=np.uint32)
    out = np.zeros((h, w, 4), dtype=np.uint8)
    if lib().pto_post_process(_p(accum), C.c_uint32(w), C.c_uint32(h), _p(params), _p(out), C.c_int(threads)) != 0:
        raise ValueError("post_process: unusable parameters")
    return out


def unpack2x16float(packed_bits: np.ndarray) -> np.ndarray:
    pk = np.ascontiguousarray(packed_bits, dtype=np.uint32)
    out = np.zeros((pk.size, 2), dtype=np.float32)
    lib().pto_unpack2x16float(C.c_uint32(pk.size), _p(pk), _p(out))
    return out


# Fixed per-operation flop costs used for the algorithmic-work figure (FMA = 2).
FLOP_COST = {
    "aabb": 24, "prim": {0: 20, 1: 14, 2: 30, 3: 60, 4: 60, 5: 0, 6: 25, 7: 50, 8: 20, 9: 0, 10: 0, 11: 35,
                          12: 45, 13: 45, 14: 0, 15: 300, 16: 200},
    "sdf": {5: 15, 14: 60, 9: 0, 10: 0}, "fractal_inner": 35, "eval_brdf": 80, "sample_brdf": 120,
    "sample_glass": 60, "nee": 90,
}


def counted_flops(cn: dict) -> int:
    f = cn["aabb_tests"] * FLOP_COST["aabb"]
    for t, c in FLOP_COST["prim"].items():
        f += cn[f"prim_tests_{t}"] * c
    for t, c in FLOP_COST["sdf"].items():
        f += cn[f"sdf_evals_{t}"] * c
    f += cn["fractal_inner"] * FLOP_COST["fractal_inner"]
    f += cn["eval_brdf"] * FLOP_COST["eval_brdf"] + cn["sample_brdf"] * FLOP_COST["sample_brdf"]
    f += cn["sample_glass"] * FLOP_COST["sample_glass"] + cn["nee"] * FLOP_COST["nee"]
    return f
