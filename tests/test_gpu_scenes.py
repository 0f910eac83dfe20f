"""GPU tier: every scene file the reference ships (resources/scenes/*.yaml, 30 of them: one per primitive type, the textured
ones, the OBJ models, the benchmark scenes), as committed fixtures, rendered through the C ABI and by the CPU oracle on the
same seeds.  Bit-exact: primary ids, per-sample radiance, running mean, second moment, rgba8, work counters — first with
whole-pixel jobs (radiance capture), then with the library's default job form for the scene (sample jobs + fold)."""
import glob
import os

import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
from util import SCENE_DIR, same_bits

pytestmark = pytest.mark.gpu
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(SCENE_DIR, "*.npz")))
W, H, N = 96, 54, 3


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.close()


def test_the_fixture_set_is_the_reference_scene_set():
    assert len(NAMES) == 30
    types = set()
    for n in NAMES:
        types |= set(int(t) for t in np.load(os.path.join(SCENE_DIR, f"{n}.npz"))["figs"]["shape_type"])
    assert types == set(range(17))  # every primitive type is authored in some scene


@pytest.mark.parametrize("name", NAMES)
def test_scene_matches_oracle(pt, name):
    path = os.path.join(SCENE_DIR, f"{name}.npz")
    sc = host.Scene.from_fixture(path).build()
    sb = hp.prepare_fixture(path)
    b = sc.buffers()
    for k in ("figs", "mats", "nodes", "prims", "lights", "infinite", "tex_pixels", "tex_infos"):
        assert b[k].tobytes() == getattr(sb, k).tobytes(), k  # the product's host library and the numpy restatement agree
    cam = sc.camera(W, H)
    ref = orc.render(orc.OracleScene(sb), cam, N, want_radiance=True, want_m2=True, counters=True)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(W, H)
    pt.debug_capture(N)
    pt.trace_n(cam, 0, 1, N)
    assert (pt.read_primary_ids() == ref["primary_id"]).all()
    assert same_bits(pt.read_debug_radiance(N), ref["radiance"]) == 0
    assert same_bits(pt.read_accum(), ref["accum"]) == 0
    assert same_bits(pt.read_moments(), ref["m2"]) == 0
    assert (pt.read_output() == ref["rgba8"]).all()
    st, c = pt.stats(), ref["counters"]
    assert (st["paths"], st["bounces"], st["shadow_rays"]) == (c["paths"], c["bounces"], c["shadow_rays"])
    pt.debug_capture(0)
    pt.clear()  # the same frames as the reference dispatches them: one sample per call, the library's own job form
    for f in range(N):
        pt.trace_n(cam, f, f + 1, 1)
    assert (pt.read_primary_ids() == ref["primary_id"]).all()
    assert same_bits(pt.read_accum(), ref["accum"]) == 0
    assert same_bits(pt.read_moments(), ref["m2"]) == 0
    assert (pt.read_output() == ref["rgba8"]).all()
    st = pt.stats()
    assert (st["paths"], st["bounces"], st["shadow_rays"]) == (c["paths"], c["bounces"], c["shadow_rays"])
    pt.enable_moments(False)
