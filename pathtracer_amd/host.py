"""Python face of the host-side mirror (include/pth_host.h): scene ingest, camera uniform, accumulator
and the frame loop of the reference's `update_and_render` (src/app/rendering.rs:72-120) over a
`device.PathTracer`."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

from .records import CAMERA_DT, MATERIAL_DT, NODE_DT, SHAPE_DT, TEXINFO_DT

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpthost.so")
_P, _U = C.c_void_p, C.c_uint32


class PthAccumulator(C.Structure):
    _fields_ = [("sample_count", _U), ("dirty", _U), ("frame_index", _U)]


SYMBOLS = [
    ("pth_scene_load", _P, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("pth_scene_new", _P, []),
    ("pth_scene_save", C.c_int, [_P, C.c_char_p]),
    ("pth_scene_free", None, [_P]),
    ("pth_last_error", C.c_char_p, [_P]),
    ("pth_scene_append_shapes", C.c_int, [_P, _P, _P, _U]),
    ("pth_scene_append_heightfield", C.c_int, [_P, _U, C.c_float, C.c_float, C.c_float, _U, _U]),
    ("pth_scene_set_shape_texture", C.c_int, [_P, _U, C.c_char_p]),
    ("pth_scene_shape_count", _U, [_P]),
    ("pth_scene_set_camera", C.c_int, [_P, _P, _P, C.c_float]),
    ("pth_scene_set_sky", C.c_int, [_P, _P, C.c_float]),
    ("pth_scene_texture_count", _U, [_P]),
    ("pth_scene_texture_path", C.c_char_p, [_P, _U]),
    ("pth_scene_provide_texture", C.c_int, [_P, _U, _P, _U, _U]),
    ("pth_scene_build", C.c_int, [_P]),
    ("pth_bvh_build", C.c_int, [_P, _P, _U, _P, C.POINTER(_U), _P]),
    ("pth_scene_build_with", C.c_int, [_P, _P, _P]),
    ("pth_scene_figures", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_materials", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_bvh_nodes", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_bvh_prims", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_lights", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_infinite", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_tex_pixels", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_tex_infos", _P, [_P, C.POINTER(_U)]),
    ("pth_scene_camera", C.c_int, [_P, _U, _U, _U, _U, _P]),
    ("pth_scene_set_render_settings", C.c_int, [_P, _U, _U, _U, C.c_float, C.c_float]),
    ("pth_accumulator_init", None, [C.POINTER(PthAccumulator)]),
    ("pth_accumulator_reset", None, [C.POINTER(PthAccumulator)]),
    ("pth_accumulator_advance", C.c_int, [C.POINTER(PthAccumulator), C.POINTER(_U)]),
    ("pth_dispatch_size", _U, [_U, _U]),
    ("pth_build_post_params", None, [_U, _U, _P, _U, _U, _U, _P]),
    ("pth_f32_to_f16_bits", C.c_uint16, [C.c_float]),
]
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run pathtracer_amd/build.py")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _decode_image(path: str) -> np.ndarray:
    from PIL import Image  # decode only; the reference uses the `image` crate (src/io/texture_atlas.rs:47-49)
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8))


class Scene:
    """A loaded scene and, after `build()`, the eight buffers the render seam uploads."""

    def __init__(self, handle):
        self.lib = load()
        self.h = handle
        self.built = False

    @classmethod
    def load(cls, yaml_path: str, resource_root: Optional[str] = None) -> "Scene":
        lib = load()
        err = C.create_string_buffer(512)
        h = lib.pth_scene_load(yaml_path.encode(), resource_root.encode() if resource_root else None, err, 512)
        if not h:
            raise RuntimeError(f"scene load failed: {err.value.decode()}")
        return cls(h)

    @classmethod
    def empty(cls) -> "Scene":
        return cls(load().pth_scene_new())

    def close(self):
        if getattr(self, "h", None):
            self.lib.pth_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def save(self, yaml_path: str):
        """AppState::save_scene (src/app/scene_ops.rs:115-124): camera + every shape as YAML."""
        if self.lib.pth_scene_save(self.h, yaml_path.encode()) != 0:
            raise RuntimeError(f"pth_scene_save({yaml_path}) failed")

    def append_shapes(self, figs: np.ndarray, mats: np.ndarray):
        figs = np.ascontiguousarray(figs, dtype=SHAPE_DT)
        mats = np.ascontiguousarray(mats, dtype=MATERIAL_DT)
        assert len(figs) == len(mats)
        if self.lib.pth_scene_append_shapes(self.h, figs.ctypes.data_as(_P), mats.ctypes.data_as(_P), len(figs)) != 0:
            raise RuntimeError(self.lib.pth_last_error(self.h).decode())
        self.built = False

    def shape_count(self) -> int:
        return int(self.lib.pth_scene_shape_count(self.h))

    def set_shape_texture(self, shape_index: int, name: str):
        if self.lib.pth_scene_set_shape_texture(self.h, shape_index, name.encode()) != 0:
            raise RuntimeError("set_shape_texture failed")
        self.built = False

    def set_camera(self, position, rotation, fov: float):
        p = np.ascontiguousarray(position, dtype=np.float32)
        r = np.ascontiguousarray(rotation, dtype=np.float32)
        self.lib.pth_scene_set_camera(self.h, p.ctypes.data_as(_P), r.ctypes.data_as(_P), float(fov))

    def set_sky(self, color, brightness: float):
        c = np.ascontiguousarray(color, dtype=np.float32)
        self.lib.pth_scene_set_sky(self.h, c.ctypes.data_as(_P), float(brightness))

    @classmethod
    def from_fixture(cls, path: str) -> "Scene":
        """Rebuilds a scene from a data/scenes/*.npz fixture (authored shapes + camera block + decoded
        textures), i.e. what Scene.load produces from the scene file the fixture was made from."""
        d = np.load(path, allow_pickle=False)
        sc = cls.empty()
        sc.append_shapes(d["figs"], d["mats"])
        cam = d["camera"]
        sc.set_camera(cam[0:3], cam[3:6], float(cam[6]))
        sc.set_render_settings(max_bounces=int(cam[8]), fractal_march_steps=int(cam[15]), tone_mapper=int(cam[14]),
                               firefly_clamp=float(cam[9]), exposure=float(cam[7]))
        sc.set_sky(cam[10:13], float(cam[13]))
        ntex = int(d["n_textures"])
        for t in range(ntex):
            for si in d[f"tex{t}_shapes"]:
                sc.set_shape_texture(int(si), f"fixture_texture_{t}")
        paths = sc.texture_paths()
        for t in range(ntex):
            img = np.ascontiguousarray(d[f"tex{t}_rgba8"], dtype=np.uint8)
            i = paths.index(f"fixture_texture_{t}")
            sc.lib.pth_scene_provide_texture(sc.h, i, img.ctypes.data_as(_P), img.shape[1], img.shape[0])
        return sc

    def append_heightfield(self, quads: int, extent: float = 20.0, y0: float = -0.9, amp: float = 0.3,
                           seed: int = 0x5EED355, glass_every: int = 64):
        if self.lib.pth_scene_append_heightfield(self.h, quads, extent, y0, amp, seed, glass_every) != 0:
            raise RuntimeError("append_heightfield failed")
        self.built = False

    def texture_paths(self) -> List[str]:
        return [self.lib.pth_scene_texture_path(self.h, i).decode() for i in range(self.lib.pth_scene_texture_count(self.h))]

    def provide_textures(self, decoder=_decode_image):
        for i, p in enumerate(self.texture_paths()):
            try:
                img = decoder(p)
            except Exception:
                continue  # like the reference's load failure: warn-and-skip (src/app/state.rs:340-342)
            self.lib.pth_scene_provide_texture(self.h, i, img.ctypes.data_as(_P), img.shape[1], img.shape[0])

    def set_render_settings(self, max_bounces=None, fractal_march_steps=None, tone_mapper=None, firefly_clamp=None, exposure=None):
        u = lambda v: 0xFFFFFFFF if v is None else int(v)
        f = lambda v: float("nan") if v is None else float(v)
        self.lib.pth_scene_set_render_settings(self.h, u(max_bounces), u(fractal_march_steps), u(tone_mapper), f(firefly_clamp), f(exposure))

    def build(self, decode_textures: bool = True, tracer=None) -> "Scene":
        """pth_scene_build; with a device.PathTracer as `tracer` the BVH is built on that GPU (ptx_bvh_build),
        which gives the same node and primitive arrays as the host builder."""
        if decode_textures:
            self.provide_textures()
        if tracer is None:
            rc = self.lib.pth_scene_build(self.h)
        else:
            fn = C.cast(tracer.lib.ptx_bvh_build, _P)
            rc = self.lib.pth_scene_build_with(self.h, fn, tracer.h)
        if rc != 0:
            raise RuntimeError(f"pth_scene_build failed: {self.lib.pth_last_error(self.h).decode()}")
        self.built = True
        return self

    def _get(self, fn, dt) -> np.ndarray:
        n = _U()
        p = fn(self.h, C.byref(n))
        if n.value == 0 or not p:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dt).copy()

    def buffers(self) -> dict:
        assert self.built, "call build() first"
        L = self.lib
        return {
            "figs": self._get(L.pth_scene_figures, SHAPE_DT), "mats": self._get(L.pth_scene_materials, MATERIAL_DT),
            "nodes": self._get(L.pth_scene_bvh_nodes, NODE_DT), "prims": self._get(L.pth_scene_bvh_prims, np.uint32),
            "lights": self._get(L.pth_scene_lights, np.uint32), "infinite": self._get(L.pth_scene_infinite, np.uint32),
            "tex_pixels": self._get(L.pth_scene_tex_pixels, np.uint32), "tex_infos": self._get(L.pth_scene_tex_infos, TEXINFO_DT),
        }

    def camera(self, width: int, height: int, frame_index: int = 0, sample_count: int = 0) -> np.ndarray:
        cam = np.zeros(1, dtype=CAMERA_DT)
        if self.lib.pth_scene_camera(self.h, width, height, frame_index, sample_count, cam.ctypes.data_as(_P)) != 0:
            raise RuntimeError("pth_scene_camera failed")
        return cam

    def upload_to(self, tracer):
        """Hands the buffers straight from the host library's memory to ptx_upload_scene."""
        assert self.built, "call build() first"
        L = self.lib
        n = [_U() for _ in range(8)]
        ptrs = [fn(self.h, C.byref(k)) for fn, k in zip(
            (L.pth_scene_figures, L.pth_scene_materials, L.pth_scene_bvh_nodes, L.pth_scene_bvh_prims, L.pth_scene_lights,
             L.pth_scene_infinite, L.pth_scene_tex_pixels, L.pth_scene_tex_infos), n)]
        args = []
        for p, k in zip(ptrs, n):
            args += [C.c_void_p(p), k.value]
        tracer._ck(tracer.lib.ptx_upload_scene(tracer.h, *args), "ptx_upload_scene")


def bvh_build(aabbs: np.ndarray):
    """Bvh::build (src/accel/bvh.rs:37-213) on the host over n boxes {min xyz, max xyz}: (nodes, primitive order)."""
    aabbs = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 6)
    n = len(aabbs)
    nodes = np.empty(max(1, 2 * n), dtype=NODE_DT)
    prims = np.empty(n, dtype=np.uint32)
    nn = _U()
    rc = load().pth_bvh_build(None, aabbs.ctypes.data_as(_P), n, nodes.ctypes.data_as(_P), C.byref(nn), prims.ctypes.data_as(_P))
    if rc != 0:
        raise RuntimeError(f"pth_bvh_build failed ({rc})")
    return nodes[: nn.value], prims


class Accumulator:
    """src/render/accumulator.rs:6-41 plus the frame loop's free-running frame index."""

    def __init__(self):
        self.lib = load()
        self.a = PthAccumulator()
        self.lib.pth_accumulator_init(C.byref(self.a))

    @property
    def sample_count(self) -> int:
        return int(self.a.sample_count)

    @property
    def frame_index(self) -> int:
        return int(self.a.frame_index)

    def reset(self):
        self.lib.pth_accumulator_reset(C.byref(self.a))

    def advance(self):
        f = _U()
        needs_clear = bool(self.lib.pth_accumulator_advance(C.byref(self.a), C.byref(f)))
        return needs_clear, int(f.value)


# PostEffect::as_u32 (src/render/post_process.rs:17-30)
POST_EFFECTS = {"none": 0, "negative": 1, "sepia": 2, "grayscale": 3, "fxaa": 4, "oil": 5, "bw": 6, "comic": 7, "casting": 8}
DEFAULT_OIL_RADIUS, DEFAULT_COMIC_LEVELS = 3, 4  # src/constants.rs:28-29


def build_post_params(width: int, height: int, effects, oil_radius: int = DEFAULT_OIL_RADIUS,
                      comic_levels: int = DEFAULT_COMIC_LEVELS) -> np.ndarray:
    """AppState::build_post_params (src/app/state.rs:448-466): the post-process pass's 16-word uniform."""
    ids = np.array([POST_EFFECTS[e] if isinstance(e, str) else int(e) for e in effects], dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    load().pth_build_post_params(width, height, ids.ctypes.data_as(_P), len(ids), oil_radius, comic_levels, out.ctypes.data_as(_P))
    return out


def save_checkpoint(path: str, tracer, samples: int, frame_index: int):
    """Progressive-render checkpoint (no reference counterpart: its accumulation lives only on the GPU): the
    per-pixel count / mean / M2 moments as a plain .npz (no pickling), enough to resume bit-exactly with
    load_checkpoint and trace_n(cam, frame_index, samples + 1, more)."""
    tracer.sync()
    m2 = tracer.read_moments() if tracer.moments_enabled else np.zeros((0,), dtype=np.float32)
    np.savez_compressed(path, accum=tracer.read_accum(), m2=m2, samples=np.uint32(samples), frame_index=np.uint32(frame_index),
                        size=np.array([tracer.width, tracer.height], dtype=np.uint32))


def load_checkpoint(path: str, tracer):
    """Loads a save_checkpoint file into a tracer that already has its scene; returns (samples, frame_index)."""
    d = np.load(path, allow_pickle=False)
    w, h = (int(v) for v in d["size"])
    has_m2 = d["m2"].size > 0
    tracer.enable_moments(has_m2)
    tracer.resize(w, h)
    tracer.write_accum(d["accum"], d["m2"] if has_m2 else None)
    return int(d["samples"]), int(d["frame_index"])


class FrameLoop:
    """The render half of AppState::update_and_render (src/app/rendering.rs:72-128): per frame,
    accumulator.advance(), camera uniform with (frame_index, sample_count), optional clear, dispatch, and the
    post-process dispatch when effects are active."""

    def __init__(self, scene: Scene, tracer, width: int, height: int):
        self.scene, self.tracer = scene, tracer
        self.width, self.height = width, height
        self.acc = Accumulator()
        self.active_effects = []  # AppState::active_effects
        self.oil_radius, self.comic_levels = DEFAULT_OIL_RADIUS, DEFAULT_COMIC_LEVELS
        scene.upload_to(tracer)
        tracer.resize(width, height)

    def _post(self):
        if self.active_effects:  # rendering.rs:122-130
            self.tracer.post_process(build_post_params(self.width, self.height, self.active_effects, self.oil_radius, self.comic_levels))

    def reset(self):
        self.acc.reset()

    def render_frame(self):
        needs_clear, frame = self.acc.advance()
        cam = self.scene.camera(self.width, self.height, frame, self.acc.sample_count)
        if needs_clear:
            self.tracer.clear()
        self.tracer.trace(cam)
        self._post()

    def render_frames(self, n: int):
        """n frames as one batched launch: identical pixels to n calls of render_frame()."""
        if n <= 0:
            return
        needs_clear, frame = self.acc.advance()
        first_sample = self.acc.sample_count
        for _ in range(n - 1):
            self.acc.advance()
        cam = self.scene.camera(self.width, self.height, frame, first_sample)
        if needs_clear:
            self.tracer.clear()
        self.tracer.trace_n(cam, frame, first_sample, n)
        self._post()
