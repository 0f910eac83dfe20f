"""GPU tier: launches of sample-granular jobs (fractal scenes) overlap -- every launch traces on a stream of its own
into a staging buffer of its own, the folds follow one another -- and the frames are those of launches run one after
the other: accumulation, M2, primary ids, rgba8 and counters, bit for bit, against the lockstep kernels and the oracle
golden.  The frame loop's ordering (src/app/rendering.rs:72-120: clear when the accumulator asks, one sample per
dispatch) is what is replayed."""
import numpy as np
import pytest

from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
from util import fixture, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.close()


def _frames(pt, sc, w, h, plan, **opts):
    """plan: list of ("clear",) / ("trace", first_frame, first_sample, n) steps; returns what a host reads afterwards."""
    pt.reset_options()
    pt.set_options(**opts)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    cam = sc.camera(w, h)
    for step in plan:
        if step[0] == "clear":
            pt.clear()
        else:
            pt.trace_n(cam, step[1], step[2], step[3])
    st = pt.stats()
    out = (pt.read_accum(), pt.read_moments(), pt.read_primary_ids(), pt.read_output(), (st["paths"], st["bounces"], st["shadow_rays"]))
    pt.enable_moments(False)
    pt.reset_options()
    return out


def _same(a, b):
    return same_bits(a[0], b[0]) == 0 and same_bits(a[1], b[1]) == 0 and (a[2] == b[2]).all() and (a[3] == b[3]).all() and a[4] == b[4]


def test_overlapping_single_sample_launches_equal_launches_in_turn(pt):
    sc = host.Scene.from_fixture(fixture("fractals")).build(decode_textures=False)
    w, h = 96, 54
    # twelve dispatches of one sample each, as the reference's frame loop issues them, without a host sync in between
    plan = [("trace", k, k + 1, 1) for k in range(12)]
    ref = _frames(pt, sc, w, h, [("trace", 0, 1, 12)], kernel_family=0)          # lockstep kernels, one launch
    one = _frames(pt, sc, w, h, plan, kernel_family=1, flights=1)                 # scheduled kernels, launches in turn
    many = _frames(pt, sc, w, h, plan, kernel_family=1)                           # ... overlapping (up to eight in flight)
    two = _frames(pt, sc, w, h, plan, kernel_family=1, flights=2)
    assert _same(one, ref) and _same(many, ref) and _same(two, ref)
    assert ref[4][0] == w * h * 12


def test_clear_and_other_calls_between_overlapping_launches(pt):
    sc = host.Scene.from_fixture(fixture("fractals")).build(decode_textures=False)
    w, h = 80, 45
    # a camera move in the middle of the sequence: the accumulator restarts (clear), sample counts start over
    plan = [("trace", k, k + 1, 1) for k in range(5)] + [("clear",)] + [("trace", 5 + k, k + 1, 1) for k in range(4)] + [("trace", 9, 5, 3)]
    ref = _frames(pt, sc, w, h, [("trace", 5, 1, 7)], kernel_family=0)
    got = _frames(pt, sc, w, h, plan, kernel_family=1)
    assert _same(got, ref)
    # chunks of one launch overlap too (split_chunk bounds the samples per chunk)
    chunks = _frames(pt, sc, w, h, [("trace", 5, 1, 7)], kernel_family=1, split_chunk=2)
    assert _same(chunks, ref)


def test_resize_while_launches_are_in_flight(pt):
    sc = host.Scene.from_fixture(fixture("fractals")).build(decode_textures=False)
    pt.reset_options()
    sc.upload_to(pt)
    pt.resize(64, 40)
    cam = sc.camera(64, 40)
    for k in range(6):
        pt.trace_n(cam, k, k + 1, 1)
    pt.resize(48, 32)  # frees the frame and the staging buffers: waits for what is in flight
    cam = sc.camera(48, 32)
    for k in range(3):
        pt.trace_n(cam, k, k + 1, 1)
    a = pt.read_accum()
    pt.clear()
    pt.trace_n(cam, 0, 1, 3)
    assert same_bits(pt.read_accum(), a) == 0 and np.isfinite(a).all()


@pytest.mark.parametrize("name", ["cornell_box", "planet", "all"])
def test_single_sample_dispatches_of_the_lockstep_kernels_overlap_too(pt, name):
    """Sample jobs + fold on the lockstep kernels (and on the scheduled kernels' scenes without fractals), folding in the lane
    for launches longer than `split_small`: the same frames, whatever the mix."""
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    w, h = 112, 63
    ref = _frames(pt, sc, w, h, [("trace", 0, 1, 11)], kernel_family=0, split_samples=0)
    singles = _frames(pt, sc, w, h, [("trace", k, k + 1, 1) for k in range(11)], kernel_family=0)
    mixed = _frames(pt, sc, w, h, [("trace", 0, 1, 2), ("trace", 2, 3, 5), ("trace", 7, 8, 1), ("trace", 8, 9, 2), ("trace", 10, 11, 1)], kernel_family=0,
                    split_small=2)  # launches of up to two samples as sample jobs, the five-sample one folds in the lane
    wide = _frames(pt, sc, w, h, [("trace", 0, 1, 4), ("trace", 4, 5, 4), ("trace", 8, 9, 3)], kernel_family=0, split_small=4, flights=3)
    default = _frames(pt, sc, w, h, [("trace", k, k + 1, 1) for k in range(11)])  # the family the library picks for the scene
    assert _same(singles, ref) and _same(mixed, ref) and _same(wide, ref) and _same(default, ref)
