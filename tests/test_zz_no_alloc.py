"""GPU tier, collected last: ptx_trace_n_static, the entry point that allocates nothing, against ptx_trace_n through
the C ABI: accumulation, moments, primary ids, rgba8 and work counters bit for bit, on two calls in a row (the second
on a smaller frame with fewer samples, so whatever the first left in the shared scratch array is in the way), for the
three kinds of launch: scene staged in LDS (Cornell box), tree walked from global memory with deep stack entries in
the scratch array (teapot), fractal scene (whole-pixel jobs instead of staged samples)."""
import numpy as np
import pytest

from pathtracer_amd import host
from pathtracer_amd.device import PathTracer, PtxError
from util import fixture, same_bits

pytestmark = pytest.mark.gpu


def _frames(pt, cam, n, static, first_frame=0):
    pt.clear()
    (pt.trace_n_static if static else pt.trace_n)(cam, first_frame, 1, n)
    st = pt.stats()
    return pt.read_accum(), pt.read_moments(), pt.read_primary_ids(), pt.read_output(), (st["paths"], st["bounces"], st["shadow_rays"])


def _same(a, b):
    assert same_bits(a[0], b[0]) == 0 and same_bits(a[1], b[1]) == 0
    assert (a[2] == b[2]).all() and (a[3] == b[3]).all() and a[4] == b[4]


@pytest.mark.parametrize("name,first,second", [("cornell_box", (200, 120, 5), (64, 36, 2)), ("teapot", (256, 144, 3), (96, 54, 1)),
                                               ("fractals", (64, 36, 3), (32, 18, 2)), ("all", (128, 72, 3), (40, 30, 2))])
def test_static_entry_point_equals_ptx_trace_n_on_two_calls_in_a_row(name, first, second):
    pt = PathTracer(0)
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    sc.upload_to(pt)
    pt.enable_moments(True)
    want = {}
    for w, h, n in (first, second):
        pt.resize(w, h)
        want[(w, h, n)] = _frames(pt, sc.camera(w, h), n, static=False)
    # the static calls, back to back: large then small, each against its own reference
    for w, h, n in (first, second):
        pt.resize(w, h)
        _same(_frames(pt, sc.camera(w, h), n, static=True), want[(w, h, n)])
    # continuing an accumulation across the two entry points
    w, h, n = second
    pt.resize(w, h)
    cam = sc.camera(w, h)
    pt.clear()
    pt.trace_n_static(cam, 0, 1, 1)
    if n > 1:
        pt.trace_n(cam, 1, 2, n - 1)
    assert same_bits(pt.read_accum(), want[second][0]) == 0
    pt.close()


def test_static_entry_point_with_two_contexts_taking_turns():
    """Every context on a GPU shares the one scratch array: used one after the other they do not disturb each other."""
    a, b = PathTracer(0), PathTracer(0)
    sa = host.Scene.from_fixture(fixture("teapot")).build(decode_textures=False)
    sb = host.Scene.from_fixture(fixture("cornell_box")).build()
    sa.upload_to(a)
    sb.upload_to(b)
    a.resize(160, 90)
    b.resize(96, 54)
    ca, cb = sa.camera(160, 90), sb.camera(96, 54)
    a.trace_n(ca, 0, 1, 4)
    b.trace_n(cb, 0, 1, 4)
    want_a, want_b = a.read_accum(), b.read_accum()
    a.clear()
    b.clear()
    for k in range(4):
        a.trace_n_static(ca, k, 1 + k, 1)
        a.sync()
        b.trace_n_static(cb, k, 1 + k, 1)
        b.sync()
    assert same_bits(a.read_accum(), want_a) == 0 and same_bits(b.read_accum(), want_b) == 0
    a.close()
    b.close()


def test_static_entry_point_refuses_a_launch_beyond_its_scratch_array():
    pt = PathTracer(0)
    sc = host.Scene.from_fixture(fixture("teapot")).build(decode_textures=False)
    sc.upload_to(pt)
    pt.set_option("kernel_family", 0)   # lockstep kernel on a mesh: one workgroup per 16x16 block, grid follows the frame
    w, h = 2048, 1200                   # 9600 workgroups = 38 400 waves > 4096
    pt.resize(w, h)
    cam = sc.camera(w, h)
    with pytest.raises(PtxError, match="static scratch array holds the state of 4096"):
        pt.trace_n_static(cam, 0, 1, 1)
    pt.trace_n(cam, 0, 1, 1)            # the ordinary entry point still takes it
    want = pt.read_accum()
    pt.resize(320, 200)                 # 260 workgroups: fits
    cam = sc.camera(320, 200)
    pt.trace_n(cam, 0, 1, 2)
    small = pt.read_accum()
    pt.clear()
    pt.trace_n_static(cam, 0, 1, 2)
    assert same_bits(pt.read_accum(), small) == 0 and np.isfinite(want).all()
    pt.close()
