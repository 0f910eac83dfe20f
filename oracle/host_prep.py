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
import re
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


# --- model/obj_loader.rs over tobj 4.0.3 -----------------------------------------------------------
# tobj is a Cargo.lock pin, not part of the reference checkout: `load_obj(path, &GPU_LOAD_OPTIONS)` (triangulate,
# single_index) is restated here from its published parser.  What the reference's scenes depend on: a numeric token
# that Rust's `f32::from_str` / `isize::from_str` rejects fails the whole load (LoadError), and the reference then
# logs the error and renders the scene WITHOUT that model (app/state.rs:122-140; resources/models/aelita/aelita.obj
# has such a token); `vt` reads two values and ignores a third; a model (mesh) ends at `o` / `g` and where `usemtl`
# changes the material; vertices are merged per (v, vt, vn) triple in first-seen order and a mesh's texcoords hold
# only the vertices that have one; a failed MTL file leaves the model without any materials.
class ObjLoadError(Exception):
    pass


_RUST_FLOAT = re.compile(r"[+-]?(?:inf|infinity|nan|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)\Z", re.IGNORECASE)
_RUST_INT = re.compile(r"[+-]?[0-9]+\Z")


def rust_f32(tok: str):
    """`f32::from_str`: the decimal value rounded ONCE to binary32 (through binary64 it is rounded twice), or None."""
    if not _RUST_FLOAT.match(tok):
        return None
    body = tok.lstrip("+-").lower()
    if body in ("inf", "infinity", "nan"):
        return f32(float(tok))
    from fractions import Fraction
    if tok[0] == "-":  # the sign is applied to the rounded magnitude: "-0.000000" is -0.0
        return -rust_f32(tok[1:])
    exact = Fraction(tok)
    with np.errstate(over="ignore"):
        near = f32(float(exact))
    if not np.isfinite(near):
        return near
    # `near` is within one binary32 step of the exact value: pick the closest of it and its neighbours, ties to even
    cands = [near, np.nextafter(near, f32(np.inf)), np.nextafter(near, f32(-np.inf))]
    best = None
    for c in cands:
        if not np.isfinite(c):
            continue
        d = abs(Fraction(float(c)) - exact)
        even = (int(np.array(c, dtype=f32).view(np.uint32)) & 1) == 0
        if best is None or d < best[0] or (d == best[0] and even and not best[2]):
            best = (d, c, even)
    return best[1]


def _rust_lines(text: str):
    """BufRead::lines: split at "\\n" only, one trailing "\\r" dropped."""
    return [l[:-1] if l.endswith("\r") else l for l in text.split("\n")]


def _floatn(words, n):
    """tobj parse_floatn: the first n of `words`, all of them valid; None when one is not or fewer than n are there."""
    out = []
    for w in words[:n]:
        v = rust_f32(w)
        if v is None:
            return None
        out.append(v)
    return out if len(out) == n else None


def _parse_mtl(path: str):
    """tobj load_mtl: ([material dicts], {name: index}); raises ObjLoadError on what makes tobj return an error."""
    try:
        fh = open(path, encoding="utf-8", newline="")
    except OSError:
        raise ObjLoadError(f"cannot open {path}")
    mats, index, cur = [], {}, None
    with fh:
        try:
            lines = _rust_lines(fh.read())
        except UnicodeDecodeError:
            raise ObjLoadError(f"{path}: not UTF-8")
    for line in lines:
        t = line.split()
        if not t or t[0] == "#":
            continue
        if t[0] == "newmtl":
            name = line.strip()[6:].strip()
            if not name:
                raise ObjLoadError(f"{path}: newmtl without a name")
            cur = {"name": name}
            index[name] = len(mats)
            mats.append(cur)
        elif t[0] in ("Ka", "Kd", "Ks"):
            v = _floatn(t[1:], 3)
            if v is None:
                raise ObjLoadError(f"{path}: bad {t[0]}")
            if cur is not None:
                cur[t[0]] = tuple(v)
        elif t[0] in ("Ns", "Ni", "d"):
            v = _floatn(t[1:], 1)
            if v is None:
                raise ObjLoadError(f"{path}: bad {t[0]}")
            if cur is not None:
                cur[t[0]] = v[0]
        elif t[0] == "map_Kd":
            tex = line.strip()[6:].strip()
            if not tex:
                raise ObjLoadError(f"{path}: map_Kd without a file")
            if cur is not None:
                cur["map_Kd"] = tex
    return mats, index


def tobj_load(path: str):
    """-> (meshes, materials or None).  A mesh: {"positions": [f32...], "texcoords": [f32...], "indices": [int...],
    "material_id": int or None}.  Raises ObjLoadError where tobj::load_obj returns Err."""
    try:
        fh = open(path, encoding="utf-8", newline="")
    except OSError:
        raise ObjLoadError(f"Failed to load OBJ: {path}")
    with fh:
        try:
            lines = _rust_lines(fh.read())
        except UnicodeDecodeError:
            raise ObjLoadError(f"{path}: not UTF-8")
    pos, col, tex, nrm = [], [], [], []
    faces = []          # lists of (v, vt, vn), None where absent
    meshes = []
    materials, mat_index, mtl_failed = [], {}, False
    mat_id = None

    def export():
        index_map, m = {}, {"positions": [], "texcoords": [], "indices": [], "material_id": mat_id}

        def add(vert):
            i = index_map.get(vert)
            if i is None:
                v, vt, vn = vert
                if v is None or v < 0 or v * 3 + 2 >= len(pos):
                    raise ObjLoadError(f"{path}: face vertex out of bounds")
                m["positions"].extend(pos[v * 3:v * 3 + 3])
                if tex and vt is not None:
                    if vt < 0 or vt * 2 + 1 >= len(tex):
                        raise ObjLoadError(f"{path}: face texcoord out of bounds")
                    m["texcoords"].extend(tex[vt * 2:vt * 2 + 2])
                if nrm and vn is not None and (vn < 0 or vn * 3 + 2 >= len(nrm)):
                    raise ObjLoadError(f"{path}: face normal out of bounds")
                if col and v * 3 + 2 >= len(col):
                    raise ObjLoadError(f"{path}: face colour out of bounds")
                i = len(index_map)
                index_map[vert] = i
            m["indices"].append(i)

        for f in faces:
            if len(f) == 1:
                tris = [(f[0], f[0], f[0])]
            elif len(f) == 2:
                tris = [(f[0], f[1], f[1])]
            else:
                tris = [(f[0], f[k], f[k + 1]) for k in range(1, len(f) - 1)]
            for tri in tris:
                for vert in tri:
                    add(vert)
        faces.clear()
        meshes.append(m)

    for line in lines:
        t = line.split()
        if not t or t[0] == "#":
            continue
        if t[0] == "v":
            v = _floatn(t[1:], 3)
            if v is None:
                raise ObjLoadError(f"{path}: bad vertex position")
            pos.extend(v)
            for w in t[4:7]:  # vertex colours, kept as far as they parse
                c = rust_f32(w)
                if c is None:
                    break
                col.append(c)
        elif t[0] == "vt":
            v = _floatn(t[1:], 2)
            if v is None:
                raise ObjLoadError(f"{path}: bad texture coordinate")
            tex.extend(v)
        elif t[0] == "vn":
            v = _floatn(t[1:], 3)
            if v is None:
                raise ObjLoadError(f"{path}: bad normal")
            nrm.extend(v)
        elif t[0] in ("f", "l"):
            sizes = (len(pos) // 3, len(tex) // 2, len(nrm) // 3)
            face = []
            for tok in t[1:]:
                parts = tok.split("/")
                if len(parts) > 3:
                    raise ObjLoadError(f"{path}: bad face")
                vert = [None, None, None]
                for i, p in enumerate(parts):
                    if p == "":
                        continue
                    if not _RUST_INT.match(p):
                        raise ObjLoadError(f"{path}: bad face index")
                    x = int(p)
                    vert[i] = sizes[i] + x if x < 0 else x - 1  # 0 -> -1: out of bounds when used
                face.append(tuple(vert))
            faces.append(face)
        elif t[0] in ("o", "g"):
            if faces:
                export()
        elif t[0] == "mtllib":
            rest = line.strip().split(" ", 1)
            name = rest[1].strip() if len(rest) > 1 else ""
            try:
                mats, idx = _parse_mtl(os.path.join(os.path.dirname(path), name))
                for k, v in idx.items():
                    mat_index[k] = v + len(materials)
                materials.extend(mats)
            except ObjLoadError:
                mtl_failed = True
        elif t[0] == "usemtl":
            rest = line.strip().split(" ", 1)
            name = rest[1].strip() if len(rest) > 1 else ""
            if not name:
                raise ObjLoadError(f"{path}: usemtl without a name")
            new = mat_index.get(name)
            if new != mat_id and faces:
                export()
            mat_id = new
    export()  # the last mesh
    return meshes, (None if mtl_failed else materials)


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
    """obj_loader.rs:48-175 (`load_obj` + `build_triangles`) over `tobj_load`.  Raises ObjLoadError where the reference's
    `load_obj` returns Err."""
    meshes, materials = tobj_load(path)
    if materials is None:  # resolve_materials, obj_loader.rs:61-78
        materials = []
    sc = f32(scale)

    def rv(positions, i):  # read_vertex, obj_loader.rs:239-245
        return np.array([positions[i * 3] * sc, positions[i * 3 + 1] * sc, positions[i * 3 + 2] * sc], dtype=f32)

    bb_min = np.full(3, np.finfo(f32).max, dtype=f32)
    bb_max = np.full(3, np.finfo(f32).min, dtype=f32)
    for m in meshes:
        for i in m["indices"]:
            v = rv(m["positions"], i)
            bb_min = np.minimum(bb_min, v)
            bb_max = np.maximum(bb_max, v)
    center = (bb_min + bb_max) * f32(0.5)
    offset = np.array(position, dtype=f32) - center
    name = os.path.splitext(os.path.basename(path))[0]
    out = []
    for m in meshes:
        tc = m["texcoords"]
        has_uv = len(tc) > 0
        mid = m["material_id"]
        if mid is not None and mid < len(materials):
            om = materials[mid]
            mat = _obj_material_to_pbr(om, default_material)
            tex = om.get("map_Kd")
            if tex is not None and not os.path.exists(tex):  # resolve_texture_path, obj_loader.rs:216-237
                cand = os.path.join(os.path.dirname(path), tex)
                if os.path.exists(cand):
                    tex = cand
        else:
            mat, tex = Material(**default_material.__dict__), None

        def uv(i):  # read_uv, obj_loader.rs:247-255 (V flipped), by the merged vertex's index
            base = i * 2
            if not has_uv or base + 1 >= len(tc):
                return (0.0, 0.0)
            return (float(tc[base]), float(f32(1.0) - tc[base + 1]))

        idx = m["indices"]
        for k in range(0, len(idx) - 2, 3):
            vs = [rv(m["positions"], idx[k + j]) + offset for j in range(3)]
            out.append(Shape(shape_type="triangle", name=name, radius=0.0, power=0.0, max_iterations=0,
                             v0=tuple(float(x) for x in vs[0]), v1=tuple(float(x) for x in vs[1]),
                             v2=tuple(float(x) for x in vs[2]), uv0=uv(idx[k]), uv1=uv(idx[k + 1]), uv2=uv(idx[k + 2]),
                             texture=tex, material=Material(**mat.__dict__)))
    return out


def expand_models(scene: Scene) -> List[Shape]:
    """app/state.rs:122-140: scene shapes followed by each model's triangles."""
    shapes = list(scene.shapes)
    for m in scene.models:
        try:
            shapes.extend(load_obj(m.path, m.position, m.scale, m.material))
        except ObjLoadError:
            pass  # state.rs:139: the error is logged and the scene goes on without the model
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
