"""GPU tier: the lockstep kernel over sorted path pools (long launches of sample jobs on scenes staged in LDS) against the
CPU oracle and against the lane-owned kernel: accumulation, M2, primary ids, rgba8 and counters bit for bit -- with pools
smaller than a workgroup's first fill, pools larger than the whole launch, frames that are no multiple of the 16x16 blocks,
tiles and partitions, bounce limits 0 and 1, continued accumulation."""
import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
from util import fixture, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.reset_options()
    t.close()


def _render(pt, sc, cam, w, h, plan, **opts):
    pt.reset_options()
    pt.set_options(**opts)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    for first_frame, first_sample, n in plan:
        pt.trace_n(cam, first_frame, first_sample, n)
    st = pt.stats()
    out = (pt.read_accum(), pt.read_moments(), pt.read_primary_ids(), pt.read_output(), (st["paths"], st["bounces"], st["shadow_rays"]), pt.kernel_name())
    pt.enable_moments(False)
    pt.reset_options()
    return out


def _same(a, b):
    return same_bits(a[0], b[0]) == 0 and same_bits(a[1], b[1]) == 0 and (a[2] == b[2]).all() and (a[3] == b[3]).all() and a[4] == b[4]


@pytest.mark.parametrize("name", ["cornell_box", "planet", "rings", "torus", "pyramid", "hyperboloid", "earth", "aelita"])
def test_pools_match_the_oracle(pt, name):
    sc = host.Scene.from_fixture(fixture(name)).build()
    sb = hp.prepare_fixture(fixture(name))
    w, h, n = 131, 77, 9
    cam = sc.camera(w, h)
    ref = orc.render(orc.OracleScene(sb), cam, n, want_m2=True, counters=True)
    c = ref["counters"]
    want = (ref["accum"], ref["m2"], ref["primary_id"], ref["rgba8"], (c["paths"], c["bounces"], c["shadow_rays"]))
    for pool in (0, 256, 16384):  # the default, one that is refilled many times, one that never fills
        got = _render(pt, sc, cam, w, h, [(0, 1, n)], sorted_pool=1, pool_paths=pool)
        assert _same(got, want), (name, pool)


@pytest.mark.parametrize("name", ["teapot", "unit_logo", "all"])
def test_pools_over_scenes_read_from_global_memory(pt, name):
    """Opt-in (kernel_family 0 + sorted_pool 1): the lockstep kernel over sorted pools with the scene records in global memory
    and the traversal stack partly there -- meshes, which the wave-scheduled kernels run by default."""
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    sb = hp.prepare_fixture(fixture(name))
    w, h, n = 90, 52, 6
    cam = sc.camera(w, h)
    ref = orc.render(orc.OracleScene(sb), cam, n, want_m2=True, counters=True)
    c = ref["counters"]
    want = (ref["accum"], ref["m2"], ref["primary_id"], ref["rgba8"], (c["paths"], c["bounces"], c["shadow_rays"]))
    got = _render(pt, sc, cam, w, h, [(0, 1, n)], kernel_family=0, sorted_pool=1, split_small=64, pool_paths=1024)
    assert _same(got, want)
    assert got[5] == ("ptx_trace_kernel_sorted" if name != "all" else got[5])  # all.yaml fits LDS: the LDS form
    # by default: a shallow mesh with lights traces launches of more than four samples from pools, one without stays wave-scheduled
    dflt = _render(pt, sc, cam, w, h, [(0, 1, n)])
    assert _same(dflt, want)
    assert dflt[5] == {"teapot": "ptx_sched_kernel_samples", "unit_logo": "ptx_trace_kernel_sorted", "all": "ptx_trace_kernel_lds_sorted"}[name]
    short = _render(pt, sc, cam, w, h, [(0, 1, 3), (3, 4, 3)])
    assert _same(short, want) and ("sched" in short[5]) == (name == "teapot")  # a lit mesh and the Moebius scenes: pools at every length
    # launches of one and two samples back to back: pools on every one of them (a mesh without lights: up to a job count)
    ones = _render(pt, sc, cam, w, h, [(0, 1, 1), (1, 2, 2), (3, 4, 1), (4, 5, 2)], host_waits_per_frame=0)
    assert _same(ones, want) and "sorted" in ones[5]
    single = _render(pt, sc, cam, w, h, [(k, k + 1, 1) for k in range(n)], pool_over=0, kernel_family=0, sorted_pool=1, split_small=64)
    assert _same(single, want) and "sorted" in single[5]


def test_pools_equal_lane_owned_paths_with_tiles_partitions_and_continuation(pt):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h = 203, 117
    cam = sc.camera(w, h)
    plan = [(0, 1, 6), (6, 7, 11), (17, 18, 5)]
    a = _render(pt, sc, cam, w, h, plan, sorted_pool=0)
    b = _render(pt, sc, cam, w, h, plan, sorted_pool=1)
    assert _same(a, b)
    # a tile of the frame, and the four interleaved partitions of it: every rank's blocks from pools, summed
    pt.reset_options()
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.set_tile(16, 32, 150, 70)
    pt.set_option("sorted_pool", 0)
    pt.trace_n(cam, 0, 1, 8)
    whole = pt.read_accum()
    pt.set_option("sorted_pool", 1)
    total = np.zeros_like(whole)
    for r in range(4):
        pt.clear()
        pt.set_partition(r, 4)
        pt.trace_n(cam, 0, 1, 8)
        total += pt.read_accum()
    pt.set_partition(0, 1)
    pt.set_tile(0, 0, w, h)
    pt.reset_options()
    assert same_bits(total, whole) == 0


@pytest.mark.parametrize("max_bounces", [0, 1, 3])
def test_pools_with_short_bounce_limits(pt, max_bounces):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h = 97, 61
    cam = sc.camera(w, h)
    cam[0]["max_bounces"] = max_bounces
    a = _render(pt, sc, cam, w, h, [(0, 1, 7)], sorted_pool=0)
    b = _render(pt, sc, cam, w, h, [(0, 1, 7)], sorted_pool=1, pool_paths=512)
    assert _same(a, b) and a[4][0] == w * h * 7
    assert a[4][1] == (0 if max_bounces == 0 else a[4][1])


@pytest.mark.parametrize("w,h,n,pool", [(8, 16, 8, 256), (8, 16, 5, 256), (24, 8, 8, 512), (40, 24, 16, 256)])
def test_a_reservation_wholly_outside_the_tile_does_not_end_the_workgroup(pt, w, h, n, pool):
    """One-block and few-block frames whose right / bottom quadrants lie outside the tile: 64 * n consecutive jobs that
    job_pixel skips.  With every path ending in its first round (max_bounces 1) the pool is empty when such a run is
    reserved; the workgroup has to go on reserving until the launch's last jobs have been handed out."""
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    sb = hp.prepare_fixture(fixture("cornell_box"))
    cam = sc.camera(w, h)
    cam[0]["max_bounces"] = 1
    ref = orc.render(orc.OracleScene(sb), cam, n, want_m2=True, counters=True)
    c = ref["counters"]
    want = (ref["accum"], ref["m2"], ref["primary_id"], ref["rgba8"], (c["paths"], c["bounces"], c["shadow_rays"]))
    a = _render(pt, sc, cam, w, h, [(0, 1, n)], sorted_pool=0)
    b = _render(pt, sc, cam, w, h, [(0, 1, n)], sorted_pool=1, pool_paths=pool, pool_over=4)
    assert "sorted" in b[5]
    assert _same(a, want)
    assert _same(b, want), (w, h, n, pool)


def test_launch_lengths_pick_the_kernel_form(pt):
    """Every launch length traces from pools on the Cornell box: short launches from small pools with a pool set per flight
    (they overlap with their neighbours), long ones from large pools in turn; lane-owned paths (pool_over 4) give the same frames."""
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h = 64, 48
    cam = sc.camera(w, h)
    a = _render(pt, sc, cam, w, h, [(k, k + 1, 1) for k in range(12)], host_waits_per_frame=0)
    b = _render(pt, sc, cam, w, h, [(0, 1, 12)])
    c = _render(pt, sc, cam, w, h, [(k, k + 1, 1) for k in range(12)], pool_over=4)
    d = _render(pt, sc, cam, w, h, [(0, 1, 1), (1, 2, 2), (3, 4, 4), (7, 8, 3), (10, 11, 2)], host_waits_per_frame=0)
    assert _same(a, b) and _same(c, b) and _same(d, b)
    assert "sorted" in a[5] and "sorted" in b[5] and "sorted" not in c[5] and "sorted" in d[5]


@pytest.mark.parametrize("name,w,h", [("cornell_box", 512, 512), ("rings", 320, 200), ("test", 640, 360)])
def test_short_launches_from_overlapping_pools(pt, name, w, h):
    """One-, two- and four-sample launches back to back, each from the pool set of its flight, with every pool knob moved: the frame
    of lane-owned whole-pixel jobs, and the oracle's on a tile."""
    sc = host.Scene.from_fixture(fixture(name)).build()
    sb = hp.prepare_fixture(fixture(name))
    cam = sc.camera(w, h)
    plan = [(0, 1, 1), (1, 2, 1), (2, 3, 2), (4, 5, 4), (8, 9, 1), (9, 10, 3)]
    want = _render(pt, sc, cam, w, h, plan, split_samples=0)
    tile = (w // 3, h // 3, 32, 16)
    ref = orc.render(orc.OracleScene(sb), cam, 12, tile=tile, want_m2=True)
    sl = (slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2]))
    assert same_bits(want[0][sl], ref["accum"][sl]) == 0 and (want[2][sl] == ref["primary_id"][sl]).all()
    for opts in (dict(), dict(flights=6), dict(flights=1), dict(pool_spread=1), dict(pool_spread=0, pool_guided=4), dict(pool_finish=0),
                 dict(pool_finish=16384), dict(pool_paths=256, flights=3), dict(pool_over=0, pool_paths=4096)):
        got = _render(pt, sc, cam, w, h, plan, host_waits_per_frame=0, **opts)
        assert _same(got, want), (name, opts)
        if name != "rings":
            assert "sorted" in got[5], (name, opts, got[5])
