"""CPU tier: pth_scene_save (AppState::save_scene + scene/exporter.rs) writes a file that pth_scene_load reads back to
the same scene: same upload buffers and camera uniform for every fixture scene, fields at their defaults left out,
numeric arrays in flow style.  The byte-level format of serde_yml cannot be produced here (no Rust toolchain):
parity unpinned at that level, pinned by the round trip."""
import os

import numpy as np
import pytest

from pathtracer_amd import host
from pathtracer_amd.records import MATERIAL_DT, SHAPE_DT
from util import fixture


def _buffers(sc):
    sc.build(decode_textures=False)
    b = sc.buffers()
    return {k: b[k].tobytes() for k in ("figs", "mats", "nodes", "prims", "lights", "infinite", "tex_pixels", "tex_infos")}, sc.camera(64, 36, 3, 4).tobytes()


@pytest.mark.parametrize("name", ["cornell_box", "teapot", "fractals", "all", "planet"])
def test_save_then_load_gives_the_same_scene(name, tmp_path):
    sc = host.Scene.from_fixture(fixture(name))
    path = str(tmp_path / f"{name}.yaml")
    sc.save(path)
    back = host.Scene.load(path)
    assert back.shape_count() == sc.shape_count()
    # the fixture's decoded textures, handed to the re-loaded scene under its (resolved) paths, first-seen order kept
    d = np.load(fixture(name), allow_pickle=False)
    assert len(back.texture_paths()) == len(sc.texture_paths())
    for i, orig in enumerate(sc.texture_paths()):
        img = np.ascontiguousarray(d[f"tex{int(orig.rsplit('_', 1)[1])}_rgba8"], dtype=np.uint8)
        back.lib.pth_scene_provide_texture(back.h, i, img.ctypes.data_as(host._P), img.shape[1], img.shape[0])
    a, cam_a = _buffers(sc)
    b, cam_b = _buffers(back)
    figs_a, figs_b = np.frombuffer(a["figs"], dtype=SHAPE_DT).copy(), np.frombuffer(b["figs"], dtype=SHAPE_DT).copy()
    for k in ("_pad2" if "_pad2" in SHAPE_DT.names else "pad2", "pad3", "pad4"):  # per-vertex UVs are not saved (shape.rs:169-175)
        if k in SHAPE_DT.names:
            figs_a[k] = 0
            figs_b[k] = 0
    assert figs_a.tobytes() == figs_b.tobytes()
    for k in ("mats", "nodes", "prims", "lights", "infinite", "tex_pixels", "tex_infos"):
        assert a[k] == b[k], k
    assert cam_a == cam_b
    # saving the loaded scene again gives the same text
    path2 = str(tmp_path / f"{name}_2.yaml")
    back.save(path2)
    assert open(path).read() == open(path2).read()


def test_format_defaults_left_out_flow_arrays_and_special_values(tmp_path):
    figs = np.zeros(3, dtype=SHAPE_DT)
    mats = np.zeros(3, dtype=MATERIAL_DT)
    mats["base_color"], mats["roughness"], mats["ior"], mats["texture_id"] = 0.8, 0.5, 1.5, -1   # Material::default
    figs["normal"] = (0.0, 1.0, 0.0)
    figs["radius"], figs["texture_scale"] = 1.0, 1.0
    figs[0]["shape_type"] = 0                                     # a sphere with every field at its default
    figs[1]["shape_type"], figs[1]["csg_op"] = 2, 1               # a negative cube
    figs[1]["position"], figs[1]["radius"] = (0.1, -2.5, 1e-7), 0.3
    mats[1]["emission"], mats[1]["emission_strength"], mats[1]["transmission"] = (1.0, 0.5, 0.25), 3.0, 1.0
    figs[2]["shape_type"] = 9                                     # Mandelbulb: power, max_iterations in v0
    figs[2]["v0"] = (6.0, 9.0, 0.0)
    figs[2]["radius2"] = np.float32(1) / np.float32(3)
    sc = host.Scene.empty()
    sc.append_shapes(figs, mats)
    sc.set_camera((0.0, 1.0, 5.5), (0.0, 0.0, 0.0), 45.0)
    path = str(tmp_path / "s.yaml")
    sc.save(path)
    text = open(path).read()
    lines = text.splitlines()
    assert lines[0] == "camera:" and "  position: [0.0, 1.0, 5.5]" in lines and "  fov: 45.0" in lines
    assert "  rotation" not in text and "exposure" not in text
    assert "- type: sphere" in lines
    i = lines.index("- type: sphere")
    assert lines[i + 1] == "- type: cube"                          # nothing but the type for the all-default sphere
    assert "  negative: true" in lines and "  position: [0.1, -2.5, 1e-07]" in lines and "  radius: 0.3" in lines
    assert "  material:" in lines and "    emission: [1.0, 0.5, 0.25]" in lines and "    transmission: 1.0" in lines
    assert "base_color" not in text and "roughness" not in text and "texture_id" not in text
    assert "- type: mandelbulb" in lines and "  power: 6.0" in lines and "  max_iterations: 9" in lines and "  radius2: 0.33333334" in lines
    back = host.Scene.load(path)
    a, cam_a = _buffers(sc)
    b, cam_b = _buffers(back)
    assert a == b and cam_a == cam_b


def test_unwritable_path_is_an_error(tmp_path):
    sc = host.Scene.empty()
    with pytest.raises(RuntimeError):
        sc.save(str(tmp_path / "no_such_dir" / "s.yaml"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources/scenes"), reason="reference checkout not mounted")
def test_reference_scene_files_survive_save_and_load(tmp_path):
    """Every scene file the reference ships: loaded, saved, loaded again -> the same upload buffers and camera."""
    import glob
    files = sorted(glob.glob("/root/reference/resources/scenes/*.yaml"))
    assert files
    for f in files:
        sc = host.Scene.load(f, "/root/reference")
        path = str(tmp_path / os.path.basename(f))
        sc.save(path)
        back = host.Scene.load(path, "/root/reference")
        a, cam_a = _buffers(sc)
        b, cam_b = _buffers(back)
        fa, fb = np.frombuffer(a.pop("figs"), dtype=SHAPE_DT).copy(), np.frombuffer(b.pop("figs"), dtype=SHAPE_DT).copy()
        for k in ("pad2", "pad3", "pad4"):   # UVs of model triangles are not saved (shape.rs:169-175)
            fa[k] = 0
            fb[k] = 0
        for k in SHAPE_DT.names:             # a vector of zeros is left out whatever their signs (is_zero_vec3): -0.0 reads back as 0.0
            if SHAPE_DT[k].base == np.float32:
                fa[k] += np.float32(0.0)
                fb[k] += np.float32(0.0)
        assert fa.tobytes() == fb.tobytes(), f
        assert a == b and cam_a == cam_b, f
