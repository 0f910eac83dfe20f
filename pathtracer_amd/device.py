"""ctypes view of the ptx355 C ABI (include/ptx355.h): one PathTracer = one ptx_ctx = one GPU."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .records import CAMERA_DT, MATERIAL_DT, NODE_DT, SHAPE_DT, TEXINFO_DT


class PtxError(RuntimeError):
    pass


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dt) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=dt)
    return a


class PathTracer:
    """Mirrors the reference's render seam: upload scene buffers, size the frame, then one `trace` per
    frame (or `trace_n` for a batch) and read the accumulation / output images back."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.ptx_create(device, C.byref(h))
        if rc != 0 or not h:
            raise PtxError(f"ptx_create(device={device}) failed with {rc}: no usable HIP device")
        self.h = h
        self.device = device
        self.width = self.height = 0
        self.tile = (0, 0, 0, 0)
        self.moments_enabled = False

    def close(self):
        if getattr(self, "h", None):
            self.lib.ptx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int, what: str):
        if rc != 0:
            raise PtxError(f"{what} failed ({rc}): {self.lib.ptx_last_error(self.h).decode()}")

    def upload_scene(self, figs, mats, nodes, prims, lights, infinite, tex_pixels, tex_infos):
        figs, mats, nodes = _arr(figs, SHAPE_DT), _arr(mats, MATERIAL_DT), _arr(nodes, NODE_DT)
        prims, lights, infinite = _arr(prims, np.uint32), _arr(lights, np.uint32), _arr(infinite, np.uint32)
        tex_pixels, tex_infos = _arr(tex_pixels, np.uint32), _arr(tex_infos, TEXINFO_DT)
        self._ck(self.lib.ptx_upload_scene(self.h, _p(figs), len(figs), _p(mats), len(mats), _p(nodes), len(nodes),
                                           _p(prims), len(prims), _p(lights), len(lights), _p(infinite), len(infinite),
                                           _p(tex_pixels), len(tex_pixels), _p(tex_infos), len(tex_infos)), "ptx_upload_scene")

    def resize(self, width: int, height: int):
        self._ck(self.lib.ptx_resize(self.h, width, height), "ptx_resize")
        self.width, self.height = width, height
        self.tile = (0, 0, width, height)

    def set_tile(self, x0: int, y0: int, w: int, h: int):
        self._ck(self.lib.ptx_set_tile(self.h, x0, y0, w, h), "ptx_set_tile")
        self.tile = (x0, y0, w, h)

    def set_partition(self, rank: int, world: int):
        self._ck(self.lib.ptx_set_partition(self.h, rank, world), "ptx_set_partition")

    def clear(self):
        self._ck(self.lib.ptx_clear(self.h), "ptx_clear")

    def trace(self, cam: np.ndarray):
        cam = _arr(cam, CAMERA_DT)
        self._ck(self.lib.ptx_trace(self.h, _p(cam)), "ptx_trace")

    def trace_n(self, cam: np.ndarray, first_frame: int, first_sample: int, n: int):
        cam = _arr(cam, CAMERA_DT)
        self._ck(self.lib.ptx_trace_n(self.h, _p(cam), first_frame, first_sample, n), "ptx_trace_n")

    def trace_n_static(self, cam: np.ndarray, first_frame: int, first_sample: int, n: int):
        """ptx_trace_n_static: the same frames as trace_n from a call that allocates nothing."""
        cam = _arr(cam, CAMERA_DT)
        self._ck(self.lib.ptx_trace_n_static(self.h, _p(cam), first_frame, first_sample, n), "ptx_trace_n_static")

    def set_option(self, key: str, value: int):
        self._ck(self.lib.ptx_set_option(self.h, key.encode(), int(value)), "ptx_set_option")

    def set_options(self, **kv):
        for k, v in kv.items():
            self.set_option(k, v)

    DEFAULT_OPTIONS = dict(kernel_family=-1, upload_path=-1, claim_tile=0, claim_guided=-1, claim_paced=-1, claim_cap=-1, claim_ranges=0, split_samples=1, split_small=-1, split_chunk=0, share_walk=0, share_post=0,
                           share_march=0, share_leaf=0, walk_mode=-1, grid_per_cu=0, flights=0, sorted_pool=-1, pool_over=-1, host_waits_per_frame=-1, pool_paths=0, arith_mode=0, pool_guided=-1, pool_finish=-1, pool_spread=-1)

    def reset_options(self):
        self.set_options(**self.DEFAULT_OPTIONS)

    def sync(self):
        self._ck(self.lib.ptx_sync(self.h), "ptx_sync")

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._ck(self.lib.ptx_read_accum(self.h, _p(out)), "ptx_read_accum")
        return out

    def write_accum(self, rgba: np.ndarray, m2: Optional[np.ndarray] = None):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        assert rgba.shape == (self.height, self.width, 4)
        if m2 is not None:
            m2 = np.ascontiguousarray(m2, dtype=np.float32)
            assert m2.shape == (self.height, self.width, 3)
        self._ck(self.lib.ptx_write_accum(self.h, _p(rgba), _p(m2) if m2 is not None else None), "ptx_write_accum")

    def read_output(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._ck(self.lib.ptx_read_output(self.h, _p(out)), "ptx_read_output")
        return out

    def read_primary_ids(self) -> np.ndarray:
        out = np.empty((self.height, self.width), dtype=np.uint32)
        self._ck(self.lib.ptx_read_primary_ids(self.h, _p(out)), "ptx_read_primary_ids")
        return out

    def bvh_build(self, aabbs: np.ndarray):
        """Bvh::build (src/accel/bvh.rs:37-213) on the GPU over n boxes {min xyz, max xyz}: (nodes, primitive order),
        the arrays the reference's sequential builder gives."""
        aabbs = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 6)
        n = len(aabbs)
        nodes = np.empty(max(1, 2 * n), dtype=NODE_DT)
        prims = np.empty(n, dtype=np.uint32)
        nn = C.c_uint32()
        self._ck(self.lib.ptx_bvh_build(self.h, _p(aabbs), n, _p(nodes), C.byref(nn), _p(prims)), "ptx_bvh_build")
        return nodes[: nn.value], prims

    def debug_capture(self, max_samples: int):
        self._ck(self.lib.ptx_debug_capture(self.h, max_samples), "ptx_debug_capture")

    def read_debug_radiance(self, n: int) -> np.ndarray:
        out = np.empty((n, self.tile[3], self.tile[2], 3), dtype=np.float32)
        self._ck(self.lib.ptx_read_debug_radiance(self.h, _p(out), n), "ptx_read_debug_radiance")
        return out

    def enable_moments(self, on: bool = True):
        self._ck(self.lib.ptx_enable_moments(self.h, 1 if on else 0), "ptx_enable_moments")
        self.moments_enabled = bool(on)

    def read_moments(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._ck(self.lib.ptx_read_moments(self.h, _p(out)), "ptx_read_moments")
        return out

    def device_accum_ptr(self) -> int:
        return int(self.lib.ptx_device_accum(self.h) or 0)

    def device_moments_ptr(self) -> int:
        return int(self.lib.ptx_device_moments(self.h) or 0)

    def merge_moments(self, first_pixel: int, n_pixels: int, n_a: int, other_mean4_ptr: int, other_m2_ptr: int, n_b: int, cam):
        cam = _arr(cam, CAMERA_DT)
        self._ck(self.lib.ptx_merge_moments(self.h, first_pixel, n_pixels, n_a, C.c_void_p(other_mean4_ptr),
                                            C.c_void_p(other_m2_ptr) if other_m2_ptr else None, n_b, _p(cam)), "ptx_merge_moments")

    def tonemap(self, first_pixel: int, n_pixels: int, cam):
        cam = _arr(cam, CAMERA_DT)
        self._ck(self.lib.ptx_tonemap(self.h, first_pixel, n_pixels, _p(cam)), "ptx_tonemap")

    @staticmethod
    def comm_unique_id() -> bytes:
        """128 opaque bytes for ptx_comm_init, made by one rank and carried to the others by the host."""
        buf = C.create_string_buffer(128)
        rc = _lib.load().ptx_comm_unique_id(buf)
        if rc != 0:
            raise PtxError(f"ptx_comm_unique_id failed ({rc}): RCCL could not be loaded")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._ck(self.lib.ptx_comm_init(self.h, C.c_char_p(unique_id), rank, world), "ptx_comm_init")

    def comm_free(self):
        self._ck(self.lib.ptx_comm_free(self.h), "ptx_comm_free")

    def merge_sample_split(self, counts, cam):
        """ptx_merge_sample_split: slice exchange, rank-ordered Chan merge, merged slices to every rank, tonemap."""
        cam = _arr(cam, CAMERA_DT)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        self._ck(self.lib.ptx_merge_sample_split(self.h, _p(counts), _p(cam)), "ptx_merge_sample_split")

    def assemble_tiles(self, cam):
        cam = _arr(cam, CAMERA_DT)
        self._ck(self.lib.ptx_assemble_tiles(self.h, _p(cam)), "ptx_assemble_tiles")

    def post_process(self, params):
        """params: 16 uint32 words (host.build_post_params / ptx_post_params).  Overwrites the rgba8 output."""
        params = np.ascontiguousarray(params, dtype=np.uint32)
        assert params.size == 16
        self._ck(self.lib.ptx_post_process(self.h, _p(params)), "ptx_post_process")

    def stats(self) -> dict:
        s = _lib.PtxStats()
        self._ck(self.lib.ptx_read_stats(self.h, C.byref(s)), "ptx_read_stats")
        return {"paths": s.paths, "bounces": s.bounces, "shadow_rays": s.shadow_rays}

    def kernel_name(self) -> str:
        return self.lib.ptx_trace_kernel_name(self.h).decode()

    def launch_info(self) -> str:
        """What the launch policy chose for the latest trace call: kernel, job form, reservation size, counters, pools, flights."""
        return self.lib.ptx_last_launch_info(self.h).decode()

    def launch_plan(self, n: int) -> str:
        """What a trace_n of n samples would run now (the launch policy's row for that length), without launching."""
        return self.lib.ptx_launch_plan(self.h, int(n)).decode()

    def kernel_time(self) -> Tuple[float, int]:
        ms, n = C.c_float(), C.c_uint32()
        self._ck(self.lib.ptx_kernel_time(self.h, C.byref(ms), C.byref(n)), "ptx_kernel_time")
        return float(ms.value), int(n.value)

    def kernel_times(self) -> np.ndarray:
        """Durations (ms) of the trace calls since the last kernel_times(), one by one."""
        buf, n = np.zeros(65536, dtype=np.float32), C.c_uint32()
        self._ck(self.lib.ptx_kernel_times(self.h, _p(buf), buf.size, C.byref(n)), "ptx_kernel_times")
        return buf[: min(n.value, buf.size)].copy()

    KAT_OPS = {"sin": 0, "cos": 1, "asin": 2, "acos": 3, "log": 4, "log2": 5, "exp2": 6, "atan2": 7, "pow": 8,
               "min": 9, "max": 10, "sqrt": 11, "div": 12, "fma_aba": 13, "rcp": 14}

    def kat_exhaustive(self, op: str) -> Tuple[int, int]:
        """(number of binary32 inputs on which the kernels' 'rcp' / 'sqrt' differs from its definition, one such input's bits)."""
        n, first = C.c_uint64(), C.c_uint32()
        self._ck(self.lib.ptx_kat_exhaustive(self.h, {"rcp": 0, "sqrt": 1}[op], C.byref(n), C.byref(first)), "ptx_kat_exhaustive")
        return int(n.value), int(first.value)

    def kat_math(self, op: str, a, b=None) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
        out = np.empty_like(a)
        self._ck(self.lib.ptx_kat_math(self.h, self.KAT_OPS[op], a.size, _p(a), _p(b), _p(out)), "ptx_kat_math")
        return out

    def kat_trace(self, cam, rays, max_t=None):
        cam = _arr(cam, CAMERA_DT)
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = len(rays)
        hid, ht = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.float32)
        sh = np.empty(n, dtype=np.uint32) if max_t is not None else None
        mt = np.ascontiguousarray(max_t, dtype=np.float32) if max_t is not None else None
        self._ck(self.lib.ptx_kat_trace(self.h, _p(cam), n, _p(rays), _p(mt), _p(hid), _p(ht), _p(sh)), "ptx_kat_trace")
        return hid, ht, sh
