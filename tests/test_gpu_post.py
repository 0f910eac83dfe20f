"""GPU tier: the post-process pass (ptx_post_process) against the oracle, bit for bit, and resuming a render from
a written-back accumulation buffer (ptx_write_accum)."""
import os

import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer, PtxError
from util import GOLDEN_DIR, fixture, same_bits

pytestmark = pytest.mark.gpu

ALL = ["negative", "sepia", "grayscale", "fxaa", "oil", "bw", "comic", "casting"]


@pytest.fixture(scope="module")
def pt():
    p = PathTracer(0)
    yield p
    p.close()


def _image(w, h, seed):
    rng = np.random.default_rng(seed)
    a = (rng.uniform(-0.2, 1.4, (h, w, 4)) * rng.choice([0.02, 1.0, 1.0, 3.0], (h, w, 1))).astype(np.float32)
    a[..., 3] = 1.0
    return a


def _run(pt, a, params):
    h, w = a.shape[:2]
    pt.resize(w, h)
    pt.write_accum(a)
    pt.post_process(params)
    return pt.read_output()


def test_golden_chains(pt):
    g = np.load(os.path.join(GOLDEN_DIR, "post_process_53x31.npz"), allow_pickle=False)
    for i, params in enumerate(g["params"]):
        assert (_run(pt, g["accum"], params) == g["out"][i]).all(), params.tolist()


@pytest.mark.parametrize("w,h", [(1, 1), (2, 5), (16, 16), (17, 3), (33, 47), (130, 70)])
def test_every_effect_on_odd_sizes(pt, w, h):
    a = _image(w, h, w * 131 + h)
    for e in ALL:
        p = orc.post_params(w, h, [e], oil_radius=3, comic_levels=5)
        assert (_run(pt, a, p) == orc.post_process(a, p)).all(), e
    p = orc.post_params(w, h, ALL, oil_radius=2, comic_levels=3)
    assert (_run(pt, a, p) == orc.post_process(a, p)).all()


@pytest.mark.parametrize("radius", [0, 1, 7, 8, 9, 13])  # up to 8 from the LDS tile, beyond from global memory
def test_oil_radius_tile_and_global_paths(pt, radius):
    a = _image(75, 41, radius)
    p = orc.post_params(75, 41, ["oil", "fxaa"], oil_radius=radius)
    assert (_run(pt, a, p) == orc.post_process(a, p)).all()


def test_non_finite_and_degenerate_inputs(pt):
    a = _image(40, 24, 5)
    a[3, 4, :3] = np.nan
    a[10, 20, :3] = np.inf
    a[11, 21, :3] = -np.inf
    for chain, levels in ((["comic"], 0), (["casting", "sepia"], 4), (["fxaa", "oil"], 4), (["bw", "negative"], 4)):
        p = orc.post_params(40, 24, chain, comic_levels=levels)
        assert (_run(pt, a, p) == orc.post_process(a, p)).all(), chain


def test_frame_loop_applies_active_effects(pt):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    loop = host.FrameLoop(sc, pt, 96, 54)
    loop.active_effects = ["fxaa", "sepia"]
    loop.render_frames(5)
    accum = pt.read_accum()
    ref = orc.render(orc.OracleScene(hp.prepare_fixture(fixture("cornell_box"))), sc.camera(96, 54), 5)["accum"]
    assert same_bits(accum, ref) == 0
    assert (pt.read_output() == orc.post_process(ref, host.build_post_params(96, 54, ["fxaa", "sepia"]))).all()


def test_full_hd_post_process_properties(pt):
    w, h = 1920, 1080
    a = _image(w, h, 77)
    twice = _run(pt, a, orc.post_params(w, h, ["negative", "negative"]))   # 1 - (1 - c) == c only approximately...
    once = _run(pt, a, orc.post_params(w, h, []))
    assert np.abs(twice[..., :3].astype(int) - once[..., :3].astype(int)).max() <= 1
    gray = _run(pt, a, orc.post_params(w, h, ["grayscale"]))
    assert (gray[..., 0] == gray[..., 1]).all() and (gray[..., 1] == gray[..., 2]).all() and (gray[..., 3] == 255).all()
    bw = _run(pt, a, orc.post_params(w, h, ["grayscale", "bw"]))
    assert set(np.unique(bw[..., :3]).tolist()) <= {0, 255}
    # a tile-interior window of the full frame against the oracle (oil radius 3 + emboss)
    p = orc.post_params(w, h, ["oil", "casting"])
    assert (_run(pt, a, p)[500:564, 900:1028] == orc.post_process(a, p, threads=8)[500:564, 900:1028]).all()


def test_post_process_errors(pt):
    pt.resize(32, 32)
    with pytest.raises(PtxError):
        pt.post_process(orc.post_params(33, 32, ["negative"]))
    p = orc.post_params(32, 32, ["negative"])
    p[2] = 9
    with pytest.raises(PtxError):
        pt.post_process(p)


def test_resume_from_written_accumulation(pt):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    sc.upload_to(pt)
    w, h = 80, 45
    cam = sc.camera(w, h)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, 7)
    whole, whole_m2 = pt.read_accum(), pt.read_moments()
    pt.clear()
    pt.trace_n(cam, 0, 1, 3)
    part, part_m2 = pt.read_accum(), pt.read_moments()
    pt.clear()                      # "a new process": state comes back from the saved buffers
    pt.write_accum(part, part_m2)
    pt.trace_n(cam, 3, 4, 4)
    assert same_bits(pt.read_accum(), whole) == 0 and same_bits(pt.read_moments(), whole_m2) == 0
    pt.enable_moments(False)


def test_checkpoint_file_resumes_bit_exactly(pt, tmp_path):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    sc.upload_to(pt)
    w, h = 64, 40
    cam = sc.camera(w, h)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, 6)
    whole, whole_m2 = pt.read_accum(), pt.read_moments()
    pt.clear()
    pt.trace_n(cam, 0, 1, 2)
    ck = str(tmp_path / "render.npz")
    host.save_checkpoint(ck, pt, samples=2, frame_index=2)
    other = PathTracer(0)  # a fresh context, as after a restart
    try:
        sc.upload_to(other)
        samples, frame = host.load_checkpoint(ck, other)
        assert (samples, frame) == (2, 2) and (other.width, other.height) == (w, h)
        other.trace_n(cam, frame, samples + 1, 4)
        assert same_bits(other.read_accum(), whole) == 0 and same_bits(other.read_moments(), whole_m2) == 0
    finally:
        other.close()
    pt.enable_moments(False)
