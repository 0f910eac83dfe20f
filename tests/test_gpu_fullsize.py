"""GPU tier at the benchmark's full sizes, through properties that do not need the oracle to finish:
tiling invariance, determinism, batched == continued accumulation, and a sampled oracle cross-check."""
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
    t.close()


def test_reference_plumbing_configuration_whole_frame(pt):
    """cornell_box.yaml 512x512, 16 samples — the configuration the reference itself would run headless — dispatched as the
    reference dispatches it (one sample per call, frame_index = sample index, sample_count counting from 1) through the
    reference's frame loop restated on the host, the WHOLE frame against the oracle: ids, mean, M2, rgba8, counters; then the
    same 16 frames in one batched launch with per-sample radiance."""
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h, n = 512, 512, 16
    sb = hp.prepare_fixture(fixture("cornell_box"))
    ocam = hp.camera_to_gpu(sb.camera, w, h)
    ref = orc.render(orc.OracleScene(sb), ocam, n, want_radiance=True, want_m2=True, counters=True)
    pt.enable_moments(True)
    loop = host.FrameLoop(sc, pt, w, h)
    for _ in range(n):
        loop.render_frame()
    c = ref["counters"]
    for batched in (False, True):
        if batched:
            pt.clear()
            pt.debug_capture(n)
            pt.trace_n(sc.camera(w, h), 0, 1, n)
            assert same_bits(pt.read_debug_radiance(n), ref["radiance"]) == 0
            pt.debug_capture(0)
        assert (pt.read_primary_ids() == ref["primary_id"]).all()
        assert same_bits(pt.read_accum(), ref["accum"]) == 0
        assert same_bits(pt.read_moments(), ref["m2"]) == 0
        assert (pt.read_output() == ref["rgba8"]).all()
        st = pt.stats()
        assert (st["paths"], st["bounces"], st["shadow_rays"]) == (c["paths"], c["bounces"], c["shadow_rays"])
    pt.enable_moments(False)


def test_cornell_1080p_properties(pt):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h, n = 1920, 1080, 32
    cam = sc.camera(w, h)
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    a = pt.read_accum()
    ids = pt.read_primary_ids()
    st = pt.stats()
    assert st["paths"] == w * h * n and 4.0 < st["bounces"] / st["paths"] < 4.7
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., 3] == 1).all()
    # determinism + split accumulation: 20 then 12 samples == 32 at once
    pt.clear()
    pt.trace_n(cam, 0, 1, 20)
    pt.trace_n(cam, 20, 21, 12)
    assert same_bits(pt.read_accum(), a) == 0
    # interleaved 4-way partition reassembles the frame
    total = np.zeros_like(a)
    for r in range(4):
        pt.clear()
        pt.set_partition(r, 4)
        pt.trace_n(cam, 0, 1, n)
        total += pt.read_accum()
    pt.set_partition(0, 1)
    assert same_bits(total, a) == 0
    # oracle cross-check on three full-resolution tiles (same frame, same seeds)
    sb = hp.prepare_fixture(fixture("cornell_box"))
    ocam = hp.camera_to_gpu(sb.camera, w, h)
    osc = orc.OracleScene(sb)
    for tile in [(0, 0, 48, 32), (936, 524, 48, 32), (1872, 1048, 48, 32)]:
        ref = orc.render(osc, ocam, n, tile=tile)
        x0, y0, tw, th = tile
        assert same_bits(a[y0:y0 + th, x0:x0 + tw], ref["accum"][y0:y0 + th, x0:x0 + tw]) == 0
        assert (ids[y0:y0 + th, x0:x0 + tw] == ref["primary_id"][y0:y0 + th, x0:x0 + tw]).all()


def test_teapot_4k_tiles_and_oracle_sample(pt):
    sc = host.Scene.from_fixture(fixture("teapot")).build(decode_textures=False)
    w, h, n = 3840, 2160, 4
    cam = sc.camera(w, h)
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    a, out = pt.read_accum(), pt.read_output()
    total = np.zeros_like(a)
    for r in range(8):
        pt.clear()
        pt.set_partition(r, 8)
        pt.trace_n(cam, 0, 1, n)
        total += pt.read_accum()
    pt.set_partition(0, 1)
    assert same_bits(total, a) == 0
    sb = hp.prepare_fixture(fixture("teapot"))
    ocam = hp.camera_to_gpu(sb.camera, w, h)
    ref = orc.render(orc.OracleScene(sb), ocam, n, tile=(1800, 1000, 96, 64))
    assert same_bits(a[1000:1064, 1800:1896], ref["accum"][1000:1064, 1800:1896]) == 0
    assert (out[1000:1064, 1800:1896] == ref["rgba8"][1000:1064, 1800:1896]).all()


def test_large_heightfield_bvh_against_oracle_tile(pt):
    sc = host.Scene.from_fixture(fixture("all"))
    sc.append_heightfield(500)  # 500 000 triangles
    sc.build()
    b = sc.buffers()
    w, h, n = 640, 360, 2
    cam = sc.camera(w, h)
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    a = pt.read_accum()
    sb = hp.SceneBuffers(b["figs"], b["mats"], b["nodes"], b["prims"], b["lights"], b["infinite"], b["tex_pixels"], b["tex_infos"],
                         hp.scene_from_fixture(fixture("all"))[1])
    ref = orc.render(orc.OracleScene(sb), cam, n, tile=(200, 200, 128, 64))
    assert same_bits(a[200:264, 200:328], ref["accum"][200:264, 200:328]) == 0


@pytest.mark.parametrize("name,w,h", [("test", 1920, 1080), ("planet", 1920, 1080), ("teapot", 3840, 2160), ("unit_logo", 1920, 1080)])
def test_job_handout_at_full_size_does_not_change_the_frame(pt, name, w, h):
    """At the benchmark's frame sizes the launch-wide job hand-out is what bounds the cheapest scenes, so it has the most
    variants there: sixteen counters or one, reservations the waves size by their pace, by the launch length or not at all,
    launches in turn or overlapping.  Every variant must give the frame of whole-pixel jobs from one counter, reserved 64 at a time."""
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    cam = sc.camera(w, h)

    def frame(launches, **opts):
        pt.reset_options()
        pt.set_options(**opts)
        sc.upload_to(pt)
        pt.resize(w, h)
        done = 0
        for n in launches:
            pt.trace_n(cam, done, done + 1, n)
            done += n
        st = pt.stats()
        return pt.read_accum(), pt.read_primary_ids(), (st["paths"], st["bounces"], st["shadow_rays"]), pt.kernel_name()

    ref = frame((4, 1, 2), split_samples=0, claim_tile=64, claim_paced=0, claim_guided=0, claim_ranges=1)
    default = frame((4, 1, 2))
    # staged in LDS: path pools at every launch length (a pool set per flight); a mesh without lights: wave-scheduled; a mesh with
    # lights: path pools at every length.  Lane-owned paths (pool_over 4) on a scene in LDS without lights: sixteen counters.
    assert {"test": "sorted", "planet": "sorted", "teapot": "sched", "unit_logo": "sorted"}[name] in default[3]
    lane_owned = frame((4, 1, 2), pool_over=4)
    assert {"test": "ranged", "planet": "ranged", "teapot": "sched", "unit_logo": "sched"}[name] in lane_owned[3]
    for got in (default, lane_owned, frame((1, 1, 1, 1, 1, 1, 1)), frame((4, 1, 2), claim_ranges=1, pool_over=4),
                frame((4, 1, 2), claim_ranges=5, claim_cap=1, pool_over=4), frame((7,), sorted_pool=0), frame((4, 1, 2), flights=1, claim_tile=512),
                frame((1, 2, 4), pool_spread=1, pool_guided=3, pool_finish=0), frame((1, 1, 3, 2), pool_spread=0, pool_finish=4096, flights=2)):
        assert same_bits(got[0], ref[0]) == 0 and (got[1] == ref[1]).all() and got[2] == ref[2]
    pt.reset_options()
    sc.upload_to(pt)


def test_fractals_1080p_128_samples_in_one_launch(pt):
    """The benchmark's fractal launch shape: 1920 x 1080 at 128 samples in ONE launch -- 4.25 GB of staged samples, right at the
    4 GiB bound of a chunk -- is deterministic, equals the same frames staged in forced chunks of 48 samples, and equals the
    oracle on a tile across the Mandelbulb's silhouette (ids, mean, rgba8, bit for bit)."""
    sc = host.Scene.from_fixture(fixture("fractals")).build()
    w, h, n = 1920, 1080, 128
    cam = sc.camera(w, h)
    pt.reset_options()
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    info = pt.launch_info()
    assert "ptx_sched_kernel_lds_frac jobs=samples" in info and "chunk=128" in info, info
    a, ids, out = pt.read_accum(), pt.read_primary_ids(), pt.read_output()
    st = pt.stats()
    assert st["paths"] == w * h * n and np.isfinite(a).all()
    pt.clear()
    pt.trace_n(cam, 0, 1, n)
    assert same_bits(pt.read_accum(), a) == 0 and (pt.read_primary_ids() == ids).all()
    pt.set_option("split_chunk", 48)
    pt.clear()
    pt.trace_n(cam, 0, 1, n)
    assert "chunk=48" in pt.launch_info()
    assert same_bits(pt.read_accum(), a) == 0 and (pt.read_primary_ids() == ids).all() and (pt.read_output() == out).all()
    st2 = pt.stats()
    assert (st2["paths"], st2["bounces"], st2["shadow_rays"]) == (st["paths"], st["bounces"], st["shadow_rays"])
    pt.reset_options()
    # a tile that holds fractal and background pixels: found from the ids the GPU reports, checked against the oracle
    sb = hp.prepare_fixture(fixture("fractals"))
    frac_figs = [i for i, t in enumerate(sb.figs["shape_type"]) if int(t) in (9, 10)]
    on = np.isin(ids, frac_figs)
    ys, xs = np.nonzero(on[:, :-1] != on[:, 1:])
    assert len(xs) > 100
    k = len(xs) // 2
    x0, y0 = max(0, int(xs[k]) - 12), max(0, int(ys[k]) - 8)
    tile = (min(x0, w - 24), min(y0, h - 16), 24, 16)
    ref = orc.render(orc.OracleScene(sb), hp.camera_to_gpu(sb.camera, w, h), n, tile=tile)
    sl = (slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2]))
    assert 0 < on[sl].sum() < on[sl].size
    assert (ids[sl] == ref["primary_id"][sl]).all()
    assert same_bits(a[sl], ref["accum"][sl]) == 0 and (out[sl] == ref["rgba8"][sl]).all()


def test_cornell_1080p_64_samples_from_the_large_pools(pt):
    """Launches of 64 samples and more trace from pools of 8192 paths per workgroup: the 1080p frame against three oracle tiles."""
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h, n = 1920, 1080, 64
    cam = sc.camera(w, h)
    pt.reset_options()
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    info = pt.launch_info()
    assert "ptx_trace_kernel_lds_sorted jobs=pools" in info and "pool=8192" in info, info
    a, ids, out = pt.read_accum(), pt.read_primary_ids(), pt.read_output()
    assert pt.stats()["paths"] == w * h * n
    sb = hp.prepare_fixture(fixture("cornell_box"))
    ocam = hp.camera_to_gpu(sb.camera, w, h)
    osc = orc.OracleScene(sb)
    for tile in [(0, 0, 40, 24), (1000, 600, 40, 24), (1880, 1056, 40, 24)]:
        ref = orc.render(osc, ocam, n, tile=tile)
        sl = (slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2]))
        assert same_bits(a[sl], ref["accum"][sl]) == 0 and (ids[sl] == ref["primary_id"][sl]).all() and (out[sl] == ref["rgba8"][sl]).all()
