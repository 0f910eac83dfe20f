"""CPU tier: OBJ ingest (src/model/obj_loader.rs over tobj 4.0.3's `load_obj(.., &GPU_LOAD_OPTIONS)`).

tobj is a Cargo.lock pin outside the reference checkout; its parser is restated twice, independently, from its
published behaviour — the C++ host (`pth_scene_load`) and the numpy oracle (`oracle/host_prep.py`) — and the two are
compared bit for bit on hand-made files that exercise each rule.  Parity with tobj itself is unpinned (no Rust here);
the expectations below state the rule each case is about.  One rule matters to a scene the reference ships:
`resources/models/aelita/aelita.obj` has a texture coordinate `0.347156й`, which Rust's `f32::from_str` rejects, so
`tobj::load_obj` fails, the reference logs the error and renders aelita.yaml without the model (app/state.rs:122-140)."""
import glob
import os

import numpy as np
import pytest

from oracle import host_prep as hp
from pathtracer_amd import host
from util import REFERENCE

KEYS = ["figs", "mats", "nodes", "prims", "lights", "infinite", "tex_pixels", "tex_infos"]
SCENE = """camera:
  position: [0.0, 1.0, -5.0]
  rotation: [0.0, 0.0, 0.0]
  fov: 60.0
shapes:
- type: sphere
  position: [0.0, 0.0, 0.0]
  radius: 0.5
  material:
    base_color: [0.5, 0.5, 0.5]
    roughness: 0.8
models:
- path: m.obj
  position: [0.0, 0.0, 0.0]
  scale: 2.0
  material:
    base_color: [0.1, 0.2, 0.3]
    roughness: 0.6
"""
CUBE_V = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nv 1 0 1\n"


def _load_both(tmp_path, obj: bytes, mtl: str = None):
    d = str(tmp_path)
    with open(os.path.join(d, "scene.yaml"), "w") as f:
        f.write(SCENE)
    with open(os.path.join(d, "m.obj"), "wb") as f:
        f.write(obj)
    if mtl is not None:
        with open(os.path.join(d, "m.mtl"), "w") as f:
            f.write(mtl)
    sc = host.Scene.load(os.path.join(d, "scene.yaml"), d).build()
    sb = hp.prepare(os.path.join(d, "scene.yaml"), d)
    b = sc.buffers()
    for k in KEYS:
        assert b[k].tobytes() == getattr(sb, k).tobytes(), k
    return b


def _tris(b):
    return int((b["figs"]["shape_type"] == 7).sum())


def test_rust_float_grammar_and_single_rounding():
    ok = {"1.5": 0x3FC00000, "+2": 0x40000000, "5.": 0x40A00000, ".5": 0x3F000000, "1e3": 0x447A0000, "1E-3": 0x3A83126F,
          "-0.000000": 0x80000000, "inf": 0x7F800000, "-Infinity": 0xFF800000, "0.1": 0x3DCCCCCD, "3.4028236e38": 0x7F800000,
          "1e-46": 0x00000000, "16777217": 0x4B800000}
    for tok, bits in ok.items():
        assert int(np.array(hp.rust_f32(tok), dtype=np.float32).view(np.uint32)) == bits, tok
    assert np.isnan(hp.rust_f32("NaN"))
    for tok in ["0.347156й", "1.0f", "0x10", "", "-", ".", "e5", "1e", "1e+", "١٢", "1_0", " 1", "1 ", "1,5"]:
        assert hp.rust_f32(tok) is None, tok
    # rounded once: 1 + 2^-24 + 2^-60 lies above the midpoint of 1 and 1 + 2^-23, but binary64 rounds it onto the midpoint,
    # which then rounds to even (down)
    assert int(np.array(hp.rust_f32("1.00000005960464477626"), dtype=np.float32).view(np.uint32)) == 0x3F800001


def test_well_formed_file_every_construct(tmp_path):
    obj = (CUBE_V + "vt 0.25 0.75 0.0\nvt 1 0\nvt -0.000000 1.000000\nvn 0 0 1\n"
           "mtllib m.mtl\nusemtl red\nf 1/1/1 2/2/1 3/3/1\n"      # triangle
           "f 1/1 2/2 3/3 4/1\n"                                    # quad: fan of two
           "g second\nusemtl blue\nf -6 -5 -4 -3 -2\n"              # polygon with relative indices: fan of three
           "usemtl nowhere\nf 1//1 2//1 5//1\n").encode()          # unknown material: default
    mtl = "newmtl red\nKd 1 0 0\nNs 250\nnewmtl blue\nKa 0 0 1\nKs 0.5 0.5 0.5\nd 0.25\nNi 1.3\n"
    b = _load_both(tmp_path, obj, mtl)
    assert _tris(b) == 1 + 2 + 3 + 1
    m = b["mats"][1:]
    assert tuple(m[0]["base_color"]) == (1.0, 0.0, 0.0) and m[0]["roughness"] == np.float32(0.5)      # 1 - sqrt(250 / 1000)
    assert tuple(m[3]["base_color"]) == (0.0, 0.0, 1.0) and m[3]["metallic"] == np.float32(0.5) and m[3]["transmission"] == np.float32(0.75)
    assert m[3]["ior"] == np.float32(1.3)
    assert tuple(m[6]["base_color"]) == (np.float32(0.1), np.float32(0.2), np.float32(0.3))
    # the third texture coordinate of the first triangle is (-0.0, 1 - 1): the sign of the zero survives the f16 packing
    assert int(b["figs"]["pad4"][1]) & 0xFFFF == 0x8000


@pytest.mark.parametrize("bad_line", [
    "vt 0.577084 0.347156й",  # the token in aelita.obj
    "vt 0.5",                     # fewer than two values
    "v 1 2",                      # fewer than three
    "v 1 2 z",
    "vn 0 0",
    "f 1 2 x",
    "f 1.5 2 3",
    "f 1/1/1/1 2 3",              # more than three indices per vertex
    "f 1 2 7",                    # beyond the six vertices
    "f 0 1 2",                    # OBJ indices start at 1
    "f 1 2 -7",                   # relative index before the first vertex
    "f 1/9 2/1 3/1",              # texture coordinate index out of range (the file has one)
    "usemtl",
])
def test_a_malformed_line_fails_the_whole_model(tmp_path, bad_line):
    obj = (CUBE_V + "vt 0 0\nf 1 2 3\n" + bad_line + "\nf 4 5 6\n").encode()
    b = _load_both(tmp_path, obj)
    assert _tris(b) == 0 and len(b["figs"]) == 1  # the scene goes on with its own shapes only


def test_not_utf8_fails_the_model(tmp_path):
    b = _load_both(tmp_path, (CUBE_V + "# caf\xe9\nf 1 2 3\n").encode("latin-1"))
    assert _tris(b) == 0


def test_points_and_lines_become_degenerate_triangles(tmp_path):
    b = _load_both(tmp_path, (CUBE_V + "f 1 2 3\nl 4 5\nf 6\nf 1 2\n").encode())
    assert _tris(b) == 4
    g = b["figs"][1:]
    assert (g["v1"][1] == g["v2"][1]).all() and (g["v0"][2] == g["v1"][2]).all() and (g["v1"][2] == g["v2"][2]).all()


def test_comment_forms_blank_lines_and_unknown_keywords(tmp_path):
    b = _load_both(tmp_path, ("#no space\n# spaced\n\n   \ns off\nfoo bar\n" + CUBE_V + "f 1 2 3\r\nf 4 5 6\n").encode())
    assert _tris(b) == 2


def test_vertices_with_and_without_texture_coordinates_in_one_mesh(tmp_path):
    """tobj appends a texcoord only for merged vertices that have one, the reference reads them by merged-vertex index
    (read_uv, obj_loader.rs:247-255): in a mesh that mixes both kinds the coordinates shift — reproduced, not repaired."""
    b = _load_both(tmp_path, (CUBE_V + "vt 0.25 0.25\nvt 0.5 0.5\nf 1 2/1 3/2\n").encode())
    g = b["figs"][1]
    h = lambda u, v: int(hp.pack_f16x2(u, v))
    # merged vertices 0, 1, 2 = (1,-), (2,vt1), (3,vt2); mesh texcoords = [vt1, vt2]; vertex 0 reads vt1, vertex 1 reads vt2, vertex 2 nothing
    assert (int(g["pad2"]), int(g["pad3"]), int(g["pad4"])) == (h(0.25, 0.75), h(0.5, 0.5), h(0.0, 0.0))


def test_a_bad_mtl_file_drops_every_material(tmp_path):
    obj = (CUBE_V + "mtllib m.mtl\nusemtl red\nf 1 2 3\n").encode()
    b = _load_both(tmp_path, obj, "newmtl red\nKd 1 0 0\nnewmtl broken\nKd 1 0\n")
    assert tuple(b["mats"][1]["base_color"]) == (np.float32(0.1), np.float32(0.2), np.float32(0.3))
    b = _load_both(tmp_path, obj, "newmtl red\nKd 1 0 0\n")
    assert tuple(b["mats"][1]["base_color"]) == (1.0, 0.0, 0.0)


def test_same_material_again_does_not_split_a_mesh(tmp_path):
    """A mesh ends where `usemtl` CHANGES the material: texture coordinates are per mesh, so a face without them after a
    repeated `usemtl` still belongs to a mesh that has some (and reads zeros past their end), while a new mesh has none."""
    head = CUBE_V + "vt 0.25 0.25\nvt 0.5 0.5\nvt 0.75 0.75\nmtllib m.mtl\nusemtl red\nf 1/1 2/2 3/3\n"
    mtl = "newmtl red\nKd 1 0 0\nnewmtl blue\nKd 0 0 1\n"
    same = _load_both(tmp_path, (head + "usemtl red\nf 4 5 6\n").encode(), mtl)
    other = _load_both(tmp_path, (head + "usemtl blue\nf 4 5 6\n").encode(), mtl)
    assert _tris(same) == _tris(other) == 2
    assert tuple(same["mats"][2]["base_color"]) == (1.0, 0.0, 0.0) and tuple(other["mats"][2]["base_color"]) == (0.0, 0.0, 1.0)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not mounted")
def test_every_scene_file_of_the_reference():
    paths = sorted(glob.glob(f"{REFERENCE}/resources/scenes/*.yaml"))
    assert len(paths) >= 30
    for path in paths:
        sc = host.Scene.load(path, REFERENCE).build()
        sb = hp.prepare(path, REFERENCE)
        b = sc.buffers()
        for k in KEYS:
            assert b[k].tobytes() == getattr(sb, k).tobytes(), (os.path.basename(path), k)
        assert sc.camera(640, 360, 2, 3).tobytes() == hp.camera_to_gpu(sb.camera, 640, 360, 2, 3).tobytes()
        if path.endswith("aelita.yaml"):
            assert int((b["figs"]["shape_type"] == 7).sum()) == 0  # its model does not load (see the module docstring)
