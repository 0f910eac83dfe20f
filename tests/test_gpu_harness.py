"""GPU tier: the C++ host (tools/pt355.cpp -> pathtracer_amd/bin/pt355) drives both C ABIs without Python in between:
scene file in, the reference's frame-loop ordering, rgba8 / accumulation files out, compared with the committed golden."""
import os
import subprocess

import numpy as np
import pytest

from pathtracer_amd import host
from util import GOLDEN_CASES, GOLDEN_DIR, ROOT, fixture, same_bits

BIN = os.path.join(ROOT, "pathtracer_amd", "bin", "pt355")


def test_harness_is_built_and_states_its_usage():
    assert os.path.exists(BIN), "run __graft_entry__.build()"
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 2 and "usage: pt355 render" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--batched"], ["--static"]])
def test_cpp_harness_renders_the_golden_frames(tmp_path, mode):
    name = "cornell_box"
    w, h, n = GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"{name}_{w}x{h}x{n}.npz"))
    scene_file = str(tmp_path / "scene.yaml")
    host.Scene.from_fixture(fixture(name)).save(scene_file)     # the reference's scene format, written by pth_scene_save
    out, acc = str(tmp_path / "out.rgba8"), str(tmp_path / "out.f32")
    r = subprocess.run([BIN, "render", scene_file, str(w), str(h), str(n), out, "--accum", acc] + mode, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert f"{w * h * n} paths" in r.stdout
    rgba8 = np.fromfile(out, dtype=np.uint8).reshape(h, w, 4)
    accum = np.fromfile(acc, dtype=np.float32).reshape(h, w, 4)
    assert (rgba8 == g["rgba8"]).all()
    assert same_bits(accum, g["accum"]) == 0


@pytest.mark.gpu
def test_cpp_harness_reports_errors(tmp_path):
    r = subprocess.run([BIN, "render", str(tmp_path / "missing.yaml"), "8", "8", "1", str(tmp_path / "o")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot load" in r.stderr
