"""TEST INFRASTRUCTURE — numpy-float32 restatement of the reference's host-side data preparation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It is
the independent checker for the product's C++ host library: both must produce bit-identical scene
buffers.  Every function cites the reference file it follows (paths relative to the reference
checkout).  Arithmetic is done on np.float32 scalars so every intermediate rounds to binary32 as the
reference's Rust f32 code does (Rust never fuses multiply-add on its own).

Third-party arithmetic that is not in the reference checkout (Cargo.lock pins): glam 0.29.3
(Quat::from_euler / Quat * Vec3 / normalize_or_zero), tobj 4.0.3 (OBJ parse), serde_yml 0.0.12,
image 0.25.9.  Their published algorithms are restated here; none has a reference-side test, so
those boundaries are "parity unpinned" (exact for rotation-free cameras such as cornell_box.yaml).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
import yaml

f32 = np.float32

# --- reference ABI records (shape.rs:223-248, material.rs:108-119, bvh.rs:12-19,
#     texture_atlas.rs:10-18, camera.rs:150-171) ---------------------------------------------------
SHAPE_DT = np.dtype([
    ("shape_type", "<u4"), ("material_idx", "<u4"), ("radius", "<f4"), ("radius2", "<f4"),
    ("position", "<f4", 3), ("height", "<f4"),
    ("normal", "<f4", 3), ("csg_op", "<u4"),
    ("rotation", "<f4", 3), ("texture_scale", "<f4"),
    ("v0", "<f4", 3), ("pad2", "<u4"),
    ("v1", "<f4", 3), ("pad3", "<u4"),
    ("v2", "<f4", 3), ("pad4", "<u4"),
])
MATERIAL_DT = np.dtype([
    ("base_color", "<f4", 3), ("metallic", "<f4"),
    ("emission", "<f4", 3), ("roughness", "<f4"),
    ("emission_strength", "<f4"), ("ior", "<f4"), ("transmission", "<f4"), ("texture_id", "<i4"),
])
NODE_DT = np.dtype([("aabb_min", "<f4", 3), ("left_or_prim", "<u4"), ("aabb_max", "<f4", 3), ("prim_count", "<u4")])
TEXINFO_DT = np.dtype([("width", "<u4"), ("height", "<u4"), ("offset", "<u4"), ("pad", "<u4")])
CAMERA_DT = np.dtype([
    ("position", "<f4", 3), ("focal_length", "<f4"),
    ("right", "<f4", 3), ("aspect", "<f4"),
    ("up", "<f4", 3), ("exposure", "<f4"),
    ("forward", "<f4", 3), ("frame_index", "<u4"),
    ("width", "<u4"), ("height", "<u4"), ("sample_count", "<u4"), ("max_bounces", "<u4"),
    ("tone_mapper", "<u4"), ("fractal_march_steps", "<u4"), ("firefly_clamp", "<f4"), ("skybox_brightness", "<f4"),
    ("skybox_color", "<f4", 3), ("pad2", "<f4"),
])
assert SHAPE_DT.itemsize == 112 and MATERIAL_DT.itemsize == 48 and NODE_DT.itemsize == 32
assert TEXINFO_DT.itemsize == 16 and CAMERA_DT.itemsize == 112

# shape.rs:12-30
SHAPE_TYPES = ["sphere", "plane", "cube", "cylinder", "cone", "torus", "disc", "triangle", "skybox",
               "mandelbulb", "julia", "ellipsoid", "paraboloid", "hyperboloid", "mebius", "pyramid", "tetrahedron"]
TYPE_ID = {n: i for i, n in enumerate(SHAPE_TYPES)}

# constants.rs:10-27
BVH_NUM_BINS = 12
BVH_LEAF_MAX_PRIMS = 4
AABB_EPS = f32(0.0001)


@dataclass
class Material:  # material.rs:9-39, defaults 41-55
    base_color: Tuple[float, float, float] = (0.8, 0.8, 0.8)
    metallic: float = 0.0
    roughness: float = 0.5
    emission: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    emission_strength: float = 0.0
    ior: float = 1.5
    transmission: float = 0.0
    texture_id: int = -1

    def is_emissive(self) -> bool:  # material.rs:97-100
        e = [f32(x) for x in self.emission]
        return f32(self.emission_strength) > 0 and (e[0] > 0 or e[1] > 0 or e[2] > 0)


@dataclass
class Shape:  # shape.rs:104-176
    shape_type: str = "sphere"
    name: Optional[str] = None
    negative: bool = False
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    normal: Tuple[float, float, float] = (0.0, 1.0, 0.0)
    radius: float = 1.0
    radius2: float = 0.0
    height: float = 0.0
    rotation: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    v0: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    v1: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    v2: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    power: float = 8.0
    max_iterations: int = 12
    texture: Optional[str] = None
    texture_scale: Optional[float] = None
    uv0: Tuple[float, float] = (0.0, 0.0)
    uv1: Tuple[float, float] = (0.0, 0.0)
    uv2: Tuple[float, float] = (0.0, 0.0)
    material: Material = field(default_factory=Material)


@dataclass
class CameraConfig:  # scene.rs:72-141, constants.rs:17-27
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)  # serde default for a present `camera:` map
    rotation: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    fov: float = 60.0
    exposure: float = 1.0
    max_bounces: int = 16
    firefly_clamp: float = 100.0
    skybox_color: Tuple[float, float, float] = (0.5, 0.7, 1.0)
    skybox_brightness: float = 0.3
    tone_mapper: int = 0
    fractal_march_steps: int = 256


@dataclass
class ModelRef:  # scene.rs:143-162
    path: str = ""
    position: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    rotation: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    scale: float = 1.0
    material: Material = field(default_factory=Material)


@dataclass
class Scene:
    camera: CameraConfig
    shapes: List[Shape]
    models: List[ModelRef]


def _mat_from(d: Optional[dict]) -> Material:
    m = Material()
    for k, v in (d or {}).items():
        if not hasattr(m, k):
            raise KeyError(f"unknown material key {k}")
        setattr(m, k, tuple(v) if isinstance(v, list) else v)
    return m


def _resolve(scene_dir: str, rel: str, root: Optional[str]) -> str:
    # constants.rs:91-108: as-is from the CWD, else relative to the scene file, else unchanged.
    for cand in ([os.path.join(root, rel)] if root else []) + [rel, os.path.join(scene_dir, rel)]:
        if os.path.exists(cand):
            return cand
    return rel


def load_scene(path: str, resource_root: Optional[str] = None) -> Scene:
    """scene/loader.rs:12-41.  `resource_root` plays the role of the process CWD the reference is
    started from (scene files name textures/models as `resources/...`)."""
    with open(path) as fh:
        doc = yaml.safe_load(fh) or {}
    if "camera" in doc and doc["camera"] is not None:
        cam = CameraConfig()
        for k, v in doc["camera"].items():
            if not hasattr(cam, k):
                raise KeyError(f"unknown camera key {k}")
            setattr(cam, k, tuple(v) if isinstance(v, list) else v)
    else:
        cam = CameraConfig(position=(0.0, 2.0, -10.0))  # CameraConfig::default, scene.rs:126-141
    shapes = []
    for sd in doc.get("shapes", doc.get("figures", [])) or []:
        s = Shape()
        for k, v in sd.items():
            if k == "type":
                s.shape_type = v
            elif k == "material":
                s.material = _mat_from(v)
            elif hasattr(s, k):
                setattr(s, k, tuple(v) if isinstance(v, list) else v)
            else:
                raise KeyError(f"unknown shape key {k}")
        if s.shape_type not in TYPE_ID:
            raise ValueError(f"unknown shape type {s.shape_type}")
        shapes.append(s)
    models = []
    for md in doc.get("models", []) or []:
        m = ModelRef()
        for k, v in md.items():
            if k == "material":
                m.material = _mat_from(v)
            elif hasattr(m, k):
                setattr(m, k, tuple(v) if isinstance(v, list) else v)
            else:
                raise KeyError(f"unknown model key {k}")
        models.append(m)
    scene_dir = os.path.dirname(os.path.abspath(path))
    for s in shapes:
        if s.texture:
            s.texture = _resolve(scene_dir, s.texture, resource_root)
    for m in models:
        m.path = _resolve(scene_dir, m.path, resource_root)
    return Scene(cam, shapes, models)


# --- model/obj_loader.rs -------------------------------------------------------------------------
def _parse_mtl(path: str) -> dict:
    mats, cur = {}, None
    if not os.path.exists(path):
        return mats
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "newmtl":
                cur = {}
                mats[t[1]] = cur
            elif cur is not None and t[0] in ("Kd", "Ka", "Ks"):
                cur[t[0]] = tuple(f32(x) for x in t[1:4])
            elif cur is not None and t[0] in ("Ns", "d", "Ni"):
                cur[t[0]] = f32(t[1])
            elif cur is not None and t[0] == "map_Kd":
                cur["map_Kd"] = t[1]
    return mats


def _obj_material_to_pbr(om: dict, fallback: Material) -> Material:  # obj_loader.rs:178-214
    m = Material(**fallback.__dict__)
    if "Kd" in om:
        m.base_color = tuple(float(x) for x in om["Kd"])
    elif "Ka" in om:
        m.base_color = tuple(float(x) for x in om["Ka"])
    if "Ks" in om:
        s = om["Ks"]
        inten = (s[0] + s[1] + s[2]) / f32(3.0)
        m.metallic = float(min(max(inten, f32(0.0)), f32(1.0)))
    if "Ns" in om:
        r = f32(1.0) - np.sqrt(om["Ns"] / f32(1000.0))
        m.roughness = float(min(max(r, f32(0.04)), f32(1.0)))
    if "d" in om and om["d"] < f32(1.0):
        m.transmission = float(f32(1.0) - om["d"])
    if "Ni" in om and om["Ni"] > 0:
        m.ior = float(om["Ni"])
    return m


def load_obj(path: str, position, scale: float, default_material: Material) -> List[Shape]:
    """obj_loader.rs:48-175 over tobj's GPU_LOAD_OPTIONS (triangulated, single index).  Supports the
    subset the bundled models use: v / vt / f (fan-triangulated) / usemtl / mtllib with Kd-style MTL."""
    verts, uvs = [], []
    groups = []  # (material name or None, has_uv, [(vi, ti) * 3])
    cur_mat, cur = None, None
    mtl = {}
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                verts.append((f32(t[1]), f32(t[2]), f32(t[3])))
            elif t[0] == "vt":
                uvs.append((f32(t[1]), f32(t[2]) if len(t) > 2 else f32(0)))
            elif t[0] == "mtllib":
                mtl = _parse_mtl(os.path.join(os.path.dirname(path), t[1]))
            elif t[0] == "usemtl":
                cur_mat, cur = t[1], None
            elif t[0] in ("o", "g"):
                cur = None
            elif t[0] == "f":
                if cur is None:
                    cur = (cur_mat, [])
                    groups.append(cur)
                idx = []
                for tok in t[1:]:
                    p = tok.split("/")
                    vi = int(p[0])
                    vi = vi - 1 if vi > 0 else len(verts) + vi
                    ti = None
                    if len(p) > 1 and p[1]:
                        ti = int(p[1])
                        ti = ti - 1 if ti > 0 else len(uvs) + ti
                    idx.append((vi, ti))
                for k in range(1, len(idx) - 1):
                    cur[1].extend([idx[0], idx[k], idx[k + 1]])
    sc = f32(scale)

    def rv(i):  # read_vertex, obj_loader.rs:239-245
        v = verts[i]
        return np.array([v[0] * sc, v[1] * sc, v[2] * sc], dtype=f32)

    bb_min = np.full(3, np.finfo(f32).max, dtype=f32)
    bb_max = np.full(3, np.finfo(f32).min, dtype=f32)
    for _, tri in groups:
        for vi, _ in tri:
            v = rv(vi)
            bb_min = np.minimum(bb_min, v)
            bb_max = np.maximum(bb_max, v)
    center = (bb_min + bb_max) * f32(0.5)
    offset = np.array(position, dtype=f32) - center
    name = os.path.splitext(os.path.basename(path))[0]
    out = []
    for mat_name, tri in groups:
        if mat_name is not None and mat_name in mtl:
            mat = _obj_material_to_pbr(mtl[mat_name], default_material)
            tex = mtl[mat_name].get("map_Kd")
            if tex is not None and not os.path.exists(tex):
                tex = os.path.join(os.path.dirname(path), tex)
        else:
            mat, tex = Material(**default_material.__dict__), None
        has_uv = any(ti is not None for _, ti in tri)
        for k in range(0, len(tri), 3):
            vs = [rv(tri[k + j][0]) + offset for j in range(3)]

            def uv(j):  # read_uv, obj_loader.rs:247-255 (V flipped)
                ti = tri[k + j][1]
                if not has_uv or ti is None:
                    return (0.0, 0.0)
                return (float(uvs[ti][0]), float(f32(1.0) - uvs[ti][1]))

            out.append(Shape(shape_type="triangle", name=name, radius=0.0, power=0.0, max_iterations=0,
                             v0=tuple(float(x) for x in vs[0]), v1=tuple(float(x) for x in vs[1]),
                             v2=tuple(float(x) for x in vs[2]), uv0=uv(0), uv1=uv(1), uv2=uv(2),
                             texture=tex, material=Material(**mat.__dict__)))
    return out


def expand_models(scene: Scene) -> List[Shape]:
    """app/state.rs:122-140: scene shapes followed by each model's triangles."""
    shapes = list(scene.shapes)
    for m in scene.models:
        shapes.extend(load_obj(m.path, m.position, m.scale, m.material))
    return shapes


# --- io/texture_atlas.rs + app/state.rs:328-348 ---------------------------------------------------
def build_texture_atlas(shapes: List[Shape], loader=None):
    """Slot 0 is the 1x1 gray 0xFF808080; textures get ids 1.. in first-seen order; `offset` counts
    pixels.  `loader(path) -> (h, w, 4) uint8` defaults to Pillow (the reference decodes with the
    `image` crate; JPEG decoders may differ by +-1 LSB: parity unpinned for JPEG textures)."""
    pixels = [np.array([0xFF808080], dtype=np.uint32)]
    infos = [(1, 1, 0, 0)]
    cache = {}
    if loader is None:
        def loader(p):
            from PIL import Image
            return np.asarray(Image.open(p).convert("RGBA"), dtype=np.uint8)
    npx = 1
    for s in shapes:
        if s.texture and s.texture not in cache:
            try:
                img = loader(s.texture)
            except Exception:
                continue  # state.rs:340-342: warn and leave the material's texture_id alone
            h, w = img.shape[:2]
            i = img.astype(np.uint32)
            pixels.append(((i[..., 3] << 24) | (i[..., 2] << 16) | (i[..., 1] << 8) | i[..., 0]).reshape(-1))
            cache[s.texture] = len(infos)
            infos.append((w, h, npx, 0))
            npx += w * h
    return np.concatenate(pixels).astype(np.uint32), np.array(infos, dtype=TEXINFO_DT), cache


# --- scene/shape.rs:250-322 ----------------------------------------------------------------------
def f32_to_f16_bits(val) -> int:
    bits = int(np.array([val], dtype=f32).view(np.uint32)[0])
    sign = (bits >> 16) & 0x8000
    exponent = (bits >> 23) & 0xFF
    mantissa = bits & 0x007FFFFF
    if exponent == 0:
        return sign
    if exponent == 0xFF:
        return sign | (0x7E00 if mantissa != 0 else 0x7C00)
    new_exp = exponent - 127 + 15
    if new_exp >= 31:
        return sign | 0x7C00
    if new_exp <= 0:
        if new_exp < -10:
            return sign
        m = (mantissa | 0x00800000) >> (1 - new_exp + 13)
        return (sign | m) & 0xFFFF
    return (sign | (new_exp << 10) | (mantissa >> 13)) & 0xFFFF


def pack_f16x2(a, b) -> int:
    return f32_to_f16_bits(a) | (f32_to_f16_bits(b) << 16)


def normalize_or_zero(v):
    """glam 0.29.3 Vec3::normalize_or_zero (scalar path): rcp = 1/sqrt(x*x + y*y + z*z); v*rcp if
    rcp is finite and > 0, else zero."""
    x, y, z = (f32(c) for c in v)
    d = x * x + y * y + z * z
    with np.errstate(divide="ignore", invalid="ignore"):
        rcp = f32(1.0) / np.sqrt(d)
    if np.isfinite(rcp) and rcp > 0:
        return (x * rcp, y * rcp, z * rcp)
    return (f32(0), f32(0), f32(0))


def build_gpu_data(shapes: List[Shape], tex_cache: dict):
    """app/state.rs:350-377 with GpuShape::from_shape (shape.rs:251-278) and GpuMaterial::from
    (material.rs:121-134)."""
    n = len(shapes)
    gs = np.zeros(n, dtype=SHAPE_DT)
    gm = np.zeros(n, dtype=MATERIAL_DT)
    lights = []
    for i, s in enumerate(shapes):
        m = s.material
        gm[i]["base_color"] = m.base_color
        gm[i]["metallic"] = m.metallic
        gm[i]["emission"] = m.emission
        gm[i]["roughness"] = max(f32(m.roughness), f32(0.04))
        gm[i]["emission_strength"] = m.emission_strength
        gm[i]["ior"] = m.ior
        gm[i]["transmission"] = m.transmission
        gm[i]["texture_id"] = m.texture_id
        if s.texture and s.texture in tex_cache:
            gm[i]["texture_id"] = tex_cache[s.texture]
        g = gs[i]
        g["shape_type"] = TYPE_ID[s.shape_type]
        g["material_idx"] = i
        g["radius"] = s.radius
        g["radius2"] = s.radius2
        g["position"] = s.position
        g["height"] = s.height
        g["normal"] = normalize_or_zero(s.normal)
        g["csg_op"] = 1 if s.negative else 0
        g["rotation"] = s.rotation
        g["texture_scale"] = 1.0 if s.texture_scale is None else s.texture_scale
        if s.shape_type in ("mandelbulb", "julia"):
            g["v0"] = (s.power, float(s.max_iterations), 0.0)
        else:
            g["v0"] = s.v0
        g["v1"] = s.v1
        g["v2"] = s.v2
        g["pad2"] = pack_f16x2(s.uv0[0], s.uv0[1])
        g["pad3"] = pack_f16x2(s.uv1[0], s.uv1[1])
        g["pad4"] = pack_f16x2(s.uv2[0], s.uv2[1])
        if m.is_emissive():
            lights.append(i)
    return gs, gm, np.array(lights, dtype=np.uint32)


# --- accel/aabb.rs -------------------------------------------------------------------------------
def _pad(mn, mx):  # aabb.rs:69-81
    mn, mx = mn.copy(), mx.copy()
    for a in range(3):
        if mx[a] - mn[a] < AABB_EPS:
            mn[a] = mn[a] - AABB_EPS
            mx[a] = mx[a] + AABB_EPS
    return mn, mx


def shape_aabb(s: Shape):  # aabb.rs:105-174
    pos = np.array(s.position, dtype=f32)
    r, r2, h = f32(s.radius), f32(s.radius2), f32(s.height)
    t = s.shape_type
    V = lambda x, y, z: np.array([x, y, z], dtype=f32)
    if t in ("sphere", "cube", "tetrahedron"):
        e = V(r, r, r)
        return pos - e, pos + e
    if t == "cylinder":
        e = V(r, h * f32(0.5), r)
        return pos - e, pos + e
    if t in ("cone", "paraboloid", "pyramid"):
        return pos - V(r, f32(0), r), pos + V(r, h, r)
    if t == "torus":
        ext = r + r2
        e = V(ext, r2, ext)
        return pos - e, pos + e
    if t == "disc":
        e = V(r, r, r)
        return _pad(pos - e, pos + e)
    if t == "triangle":
        v0, v1, v2 = (np.array(v, dtype=f32) for v in (s.v0, s.v1, s.v2))
        mn = np.minimum(np.minimum(v0, v1), v2)
        mx = np.maximum(np.maximum(v0, v1), v2)
        return _pad(mn, mx)
    if t in ("mandelbulb", "julia", "mebius"):
        k = r * f32(1.5)
        e = V(k, k, k)
        return pos - e, pos + e
    if t == "ellipsoid":
        e = V(r, max(h, r), max(r2, r))
        return pos - e, pos + e
    if t == "hyperboloid":
        hh = h * f32(0.5)
        e = V(r + hh, hh, r + hh)
        return pos - e, pos + e
    big = V(1e6, 1e6, 1e6)
    return -big, big


def _area(mn, mx):  # aabb.rs:46-49
    with np.errstate(invalid="ignore", over="ignore"):
        d = mx - mn
        return f32(2.0) * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])


# --- accel/bvh.rs ----------------------------------------------------------------------------------
def bvh_build(aabbs):
    """Bvh::build (bvh.rs:37-57).  `aabbs` is a list of (min, max) float32 triples.  Returns
    (nodes, prim_indices) in the reference's flattened layout."""
    n = len(aabbs)
    if n == 0:
        return np.zeros(1, dtype=NODE_DT), np.zeros(0, dtype=np.uint32)
    mins = np.array([a[0] for a in aabbs], dtype=f32)
    maxs = np.array([a[1] for a in aabbs], dtype=f32)
    cents = (mins + maxs) * f32(0.5)  # Aabb::center
    indices = list(range(n))
    build = []  # [mn, mx, left, right, first, count]
    INF = f32(np.inf)

    def find_best_split(idx, pmin, pmax):  # bvh.rs:111-172
        best_cost, best_axis, best_split = INF, 0, f32(0.0)
        for axis in range(3):
            mn, mx = pmin[axis], pmax[axis]
            extent = mx - mn
            if abs(extent) < f32(1e-8):
                continue
            bmin = np.full((BVH_NUM_BINS, 3), INF, dtype=f32)
            bmax = np.full((BVH_NUM_BINS, 3), -INF, dtype=f32)
            bcnt = [0] * BVH_NUM_BINS
            inv_extent = f32(BVH_NUM_BINS) / extent
            for i in idx:
                v = (cents[i][axis] - mn) * inv_extent
                b = 0 if not (v > 0) else (BVH_NUM_BINS - 1 if v >= BVH_NUM_BINS else int(v))
                bmin[b] = np.minimum(bmin[b], mins[i])
                bmax[b] = np.maximum(bmax[b], maxs[i])
                bcnt[b] += 1
            right_area = [f32(0)] * (BVH_NUM_BINS - 1)
            right_count = [0] * (BVH_NUM_BINS - 1)
            rmin, rmax, rc = np.full(3, INF, dtype=f32), np.full(3, -INF, dtype=f32), 0
            for i in range(BVH_NUM_BINS - 1, 0, -1):
                rmin, rmax = np.minimum(rmin, bmin[i]), np.maximum(rmax, bmax[i])
                rc += bcnt[i]
                right_area[i - 1] = _area(rmin, rmax)
                right_count[i - 1] = rc
            lmin, lmax, lc = np.full(3, INF, dtype=f32), np.full(3, -INF, dtype=f32), 0
            bin_width = extent / f32(BVH_NUM_BINS)
            for i in range(BVH_NUM_BINS - 1):
                lmin, lmax = np.minimum(lmin, bmin[i]), np.maximum(lmax, bmax[i])
                lc += bcnt[i]
                if lc == 0 or right_count[i] == 0:
                    continue
                cost = f32(lc) * _area(lmin, lmax) + f32(right_count[i]) * right_area[i]
                if cost < best_cost:
                    best_cost, best_axis = cost, axis
                    best_split = mn + f32(i + 1) * bin_width
        return best_axis, best_split

    def partition(start, end, axis, split):  # bvh.rs:174-186
        lo, hi = start, end
        while lo < hi:
            if cents[indices[lo]][axis] < split:
                lo += 1
            else:
                hi -= 1
                indices[lo], indices[hi] = indices[hi], indices[lo]
        return lo

    def rec(start, end):  # bvh.rs:59-109
        idx = indices[start:end]
        pmin = np.full(3, INF, dtype=f32)
        pmax = np.full(3, -INF, dtype=f32)
        for i in idx:
            pmin, pmax = np.minimum(pmin, mins[i]), np.maximum(pmax, maxs[i])
        node = len(build)
        count = end - start
        if count <= BVH_LEAF_MAX_PRIMS:
            build.append([pmin, pmax, None, None, start, count])
            return node
        axis, split = find_best_split(idx, pmin, pmax)
        raw_mid = partition(start, end, axis, split)
        mid = (start + end) // 2 if raw_mid in (start, end) else raw_mid
        build.append([pmin, pmax, None, None, 0, 0])
        l = rec(start, mid)
        r = rec(mid, end)
        build[node][2], build[node][3] = l, r
        return node

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    rec(0, n)
    out = []

    def flatten(i):  # bvh.rs:188-213
        mn, mx, l, r, first, count = build[i]
        oi = len(out)
        if count > 0:
            out.append((mn, first, mx, count))
        else:
            out.append([mn, 0, mx, 0])
            flatten(l)
            out[oi][1] = len(out)
            flatten(r)

    flatten(0)
    nodes = np.zeros(len(out), dtype=NODE_DT)
    for i, (mn, lp, mx, pc) in enumerate(out):
        nodes[i]["aabb_min"], nodes[i]["left_or_prim"], nodes[i]["aabb_max"], nodes[i]["prim_count"] = mn, lp, mx, pc
    return nodes, np.array(indices, dtype=np.uint32)


def build_bvh(shapes: List[Shape]):  # app/state.rs:543-567
    finite, infinite = [], []
    for i, s in enumerate(shapes):
        if s.shape_type == "plane":
            infinite.append(i)
        elif s.shape_type == "skybox":
            pass
        else:
            finite.append(i)
    nodes, prims = bvh_build([shape_aabb(shapes[i]) for i in finite])
    prims = np.array([finite[p] for p in prims], dtype=np.uint32)
    return nodes, prims, np.array(infinite, dtype=np.uint32)


# --- camera/camera.rs:80-128 ---------------------------------------------------------------------
def _sincos32(x):
    """f32 sin/cos as the correctly rounded value of the double-precision libm result."""
    return f32(math.sin(float(x))), f32(math.cos(float(x)))


def _quat_mul(a, b):  # glam scalar Quat::mul_quat, (x, y, z, w)
    x0, y0, z0, w0 = a
    x1, y1, z1, w1 = b
    return (w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
            w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1,
            w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1,
            w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1)


def _quat_rotate(q, v):  # glam scalar Quat * Vec3
    x, y, z, w = q
    b = (x, y, z)
    b2 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2]
    s1 = w * w - b2
    vb = (v[0] * b[0] + v[1] * b[1] + v[2] * b[2]) * f32(2.0)
    cr = (b[1] * v[2] - v[1] * b[2], b[2] * v[0] - v[2] * b[0], b[0] * v[1] - v[0] * b[1])  # b.cross(v)
    w2 = w * f32(2.0)
    return tuple(v[i] * s1 + b[i] * vb + cr[i] * w2 for i in range(3))


def camera_to_gpu(cfg: CameraConfig, width: int, height: int, frame_index: int = 0, sample_count: int = 0):
    """Camera::from_config + Camera::to_gpu.  yaw = rotation[1], pitch = rotation[0] (camera.rs:29-35);
    orientation = Quat::from_euler(YXZ, yaw, pitch, 0) restated as Ry(yaw) * Rx(pitch)."""
    deg = f32(math.pi) / f32(180.0)  # f32::to_radians multiplies by (PI / 180) in f32
    yaw, pitch = f32(cfg.rotation[1]) * deg, f32(cfg.rotation[0]) * deg
    sy, cy = _sincos32(yaw * f32(0.5))
    sx, cx = _sincos32(pitch * f32(0.5))
    z0, one = f32(0.0), f32(1.0)
    q = _quat_mul(_quat_mul((z0, sy, z0, cy), (sx, z0, z0, cx)), (z0, z0, z0, one))
    fwd = _quat_rotate(q, (z0, z0, one))
    right = _quat_rotate(q, (one, z0, z0))
    up = _quat_rotate(q, (z0, one, z0))
    cam = np.zeros(1, dtype=CAMERA_DT)
    c = cam[0]
    c["position"] = cfg.position
    half = f32(cfg.fov) * deg * f32(0.5)
    c["focal_length"] = f32(1.0) / f32(math.tan(float(half)))
    c["right"], c["up"], c["forward"] = right, up, fwd
    c["aspect"] = f32(width) / f32(height)
    c["exposure"] = cfg.exposure
    c["frame_index"], c["sample_count"] = frame_index, sample_count
    c["width"], c["height"] = width, height
    c["max_bounces"] = cfg.max_bounces
    c["tone_mapper"] = cfg.tone_mapper
    c["fractal_march_steps"] = cfg.fractal_march_steps
    c["firefly_clamp"] = cfg.firefly_clamp
    c["skybox_brightness"] = cfg.skybox_brightness
    c["skybox_color"] = cfg.skybox_color
    return cam


@dataclass
class SceneBuffers:
    figs: np.ndarray
    mats: np.ndarray
    nodes: np.ndarray
    prims: np.ndarray
    lights: np.ndarray
    infinite: np.ndarray
    tex_pixels: np.ndarray
    tex_infos: np.ndarray
    camera: CameraConfig


def prepare(path: str, resource_root: Optional[str] = None, extra_shapes: Optional[List[Shape]] = None,
            tex_loader=None) -> SceneBuffers:
    """AppState::new's data path (app/state.rs:114-146) from a scene file to the eight scene buffers."""
    scene = load_scene(path, resource_root)
    shapes = expand_models(scene)
    if extra_shapes:
        shapes.extend(extra_shapes)
    return prepare_shapes(shapes, scene.camera, tex_loader)


def prepare_shapes(shapes: List[Shape], camera: CameraConfig, tex_loader=None) -> SceneBuffers:
    pix, infos, cache = build_texture_atlas(shapes, tex_loader)
    gs, gm, lights = build_gpu_data(shapes, cache)
    nodes, prims, infinite = build_bvh(shapes)
    return SceneBuffers(gs, gm, nodes, prims, lights, infinite, pix, infos, camera)


# --- data/scenes/*.npz fixtures ---------------------------------------------------------------------
def _half_to_float(h: int) -> float:
    return float(np.array([h], dtype=np.uint16).view(np.float16)[0])


def scene_from_fixture(path: str):
    """Rebuilds (shapes, camera, texture loader) from a fixture written by tools/make_scene_fixtures.py:
    authored shapes in record containers, the camera block and decoded textures."""
    d = np.load(path, allow_pickle=False)
    figs, mats = d["figs"], d["mats"]
    shapes = []
    for g, m in zip(figs, mats):
        t = SHAPE_TYPES[int(g["shape_type"])]
        s = Shape(shape_type=t, negative=bool(g["csg_op"] == 1), position=tuple(float(x) for x in g["position"]),
                  normal=tuple(float(x) for x in g["normal"]), radius=float(g["radius"]), radius2=float(g["radius2"]),
                  height=float(g["height"]), rotation=tuple(float(x) for x in g["rotation"]),
                  v1=tuple(float(x) for x in g["v1"]), v2=tuple(float(x) for x in g["v2"]),
                  texture_scale=float(g["texture_scale"]))
        if t in ("mandelbulb", "julia"):
            s.power, s.max_iterations = float(g["v0"][0]), int(g["v0"][1])
        else:
            s.v0 = tuple(float(x) for x in g["v0"])
        for name, fld in (("uv0", "pad2"), ("uv1", "pad3"), ("uv2", "pad4")):
            b = int(g[fld])
            setattr(s, name, (_half_to_float(b & 0xFFFF), _half_to_float(b >> 16)))
        s.material = Material(tuple(float(x) for x in m["base_color"]), float(m["metallic"]), float(m["roughness"]),
                              tuple(float(x) for x in m["emission"]), float(m["emission_strength"]), float(m["ior"]),
                              float(m["transmission"]), int(m["texture_id"]))
        shapes.append(s)
    c = d["camera"]
    cam = CameraConfig(position=tuple(c[0:3]), rotation=tuple(c[3:6]), fov=float(c[6]), exposure=float(c[7]),
                       max_bounces=int(c[8]), firefly_clamp=float(c[9]), skybox_color=tuple(c[10:13]),
                       skybox_brightness=float(c[13]), tone_mapper=int(c[14]), fractal_march_steps=int(c[15]))
    images = {}
    for t in range(int(d["n_textures"])):
        name = f"fixture_texture_{t}"
        images[name] = d[f"tex{t}_rgba8"]
        for si in d[f"tex{t}_shapes"]:
            shapes[int(si)].texture = name
    return shapes, cam, (lambda p: images[p])


def prepare_fixture(path: str) -> SceneBuffers:
    shapes, cam, loader = scene_from_fixture(path)
    return prepare_shapes(shapes, cam, loader)
