"""GPU tier: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.
Bar: bit-exact for ids, per-sample radiance, running mean, rgba8 and the work counters (the kernels and
the oracle implement the same binary32 expression trees; no tolerance is needed or used)."""
import os

import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer, PtxError
from util import GOLDEN_CASES, GOLDEN_DIR, fixture, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.close()


def _upload(pt, sb):
    pt.upload_scene(sb.figs, sb.mats, sb.nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)


def _compare(pt, sb, w, h, n, first_frame=0, first_sample=1):
    cam = hp.camera_to_gpu(sb.camera, w, h)
    ref = orc.render(orc.OracleScene(sb), cam, n, first_frame=first_frame, first_sample=first_sample,
                     want_radiance=True, want_m2=True, counters=True)
    _upload(pt, sb)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.debug_capture(n)
    pt.trace_n(cam, first_frame, first_sample, n)
    got = dict(ids=pt.read_primary_ids(), acc=pt.read_accum(), out=pt.read_output(), rad=pt.read_debug_radiance(n),
               m2=pt.read_moments(), st=pt.stats())
    pt.debug_capture(0)
    pt.enable_moments(False)
    assert (got["ids"] == ref["primary_id"]).all()
    assert same_bits(got["rad"], ref["radiance"]) == 0
    assert same_bits(got["acc"], ref["accum"]) == 0
    assert same_bits(got["m2"], ref["m2"]) == 0
    assert (got["out"] == ref["rgba8"]).all()
    c = ref["counters"]
    assert (got["st"]["paths"], got["st"]["bounces"], got["st"]["shadow_rays"]) == (c["paths"], c["bounces"], c["shadow_rays"])
    return got, ref


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_matches_committed_golden(pt, name):
    w, h, n = GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"{name}_{w}x{h}x{n}.npz"))
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.debug_capture(n)
    pt.trace_n(sc.camera(w, h), 0, 1, n)
    assert (pt.read_primary_ids() == g["primary_id"]).all()
    assert same_bits(pt.read_debug_radiance(n), g["radiance"]) == 0
    assert same_bits(pt.read_accum(), g["accum"]) == 0
    assert same_bits(pt.read_moments(), g["m2"]) == 0
    assert (pt.read_output() == g["rgba8"]).all()
    st = pt.stats()
    assert [st["paths"], st["bounces"], st["shadow_rays"]] == list(g["counters"])
    pt.debug_capture(0)
    pt.enable_moments(False)


@pytest.mark.parametrize("name,w,h,n", [("cornell_box", 200, 120, 6), ("teapot", 256, 144, 3), ("fractals", 80, 45, 2),
                                        ("all", 160, 90, 4), ("planet", 160, 90, 4)])
def test_matches_oracle_on_config_scenes(pt, name, w, h, n):
    _compare(pt, hp.prepare_fixture(fixture(name)), w, h, n)


def test_device_math_contract_bitwise(pt):
    rng = np.random.default_rng(11)
    x = rng.uniform(-40, 40, 200000).astype(np.float32)
    y = rng.uniform(-40, 40, 200000).astype(np.float32)
    cases = {"sin": (x, None), "cos": (x, None), "asin": (x / 40, None), "acos": (x / 40, None), "log": (np.abs(x) + 1e-30, None),
             "log2": (np.abs(x), None), "exp2": (x * 4, None), "atan2": (x, y), "pow": (np.abs(x), y / 4), "min": (x, y),
             "max": (x, y), "sqrt": (np.abs(x), None), "div": (x, y), "fma_aba": (x, y), "rcp": (x, None)}
    for op, (a, b) in cases.items():
        assert same_bits(pt.kat_math(op, a, b), orc.math_op(op, a, b)) == 0, op
    # the reciprocal and the square root are correctly rounded; a / b is a * rcp(b)
    r = (1.0 / x.astype(np.float64)).astype(np.float32)
    assert same_bits(pt.kat_math("rcp", x), r) == 0
    assert same_bits(pt.kat_math("div", y, x), y * r) == 0
    assert same_bits(pt.kat_math("sqrt", np.abs(x)), np.sqrt(np.abs(x).astype(np.float64)).astype(np.float32)) == 0
    sp = np.array([0.0, -0.0, np.nan, 1.0, -0.0, np.inf, 1e-42, 1e-39, 3e38, -2e38, 2.0 ** 126, 1e-33, 5e-39], dtype=np.float32)
    sq = np.array([-0.0, 0.0, 1.0, np.nan, -0.0, -np.inf, -1e-42, 2.0, 1e-39, 3.0, -(2.0 ** -126), 1e-36, 1.0], dtype=np.float32)
    for op in ("min", "max", "div", "rcp", "sqrt", "atan2", "pow", "log", "exp2"):
        assert same_bits(pt.kat_math(op, sp, sq), orc.math_op(op, sp, sq)) == 0, op
        assert same_bits(pt.kat_math(op, sq, sp), orc.math_op(op, sq, sp)) == 0, op


def test_reciprocal_and_sqrt_sequences_match_their_definition_on_all_2_32_inputs(pt):
    # the short device sequences against the definition, exhaustively (evaluated on the device)
    assert pt.kat_exhaustive("rcp")[0] == 0
    assert pt.kat_exhaustive("sqrt")[0] == 0
    # and the definition itself against the CPU oracle on the subnormal / huge boundary cases
    b = np.concatenate([np.arange(0, 1 << 12, dtype=np.uint32), np.uint32(0x00800000) + np.arange(-64, 64).astype(np.uint32),
                        np.uint32(0x7E800000) + np.arange(-64, 64).astype(np.uint32), np.uint32(0x7F800000) - np.arange(0, 64).astype(np.uint32),
                        np.uint32(0x0C800000) + np.arange(-64, 64).astype(np.uint32)])
    b = np.concatenate([b, b | np.uint32(0x80000000)])
    x = b.view(np.float32)
    for op in ("rcp", "sqrt"):
        assert same_bits(pt.kat_math(op, x), orc.math_op(op, x)) == 0, op


@pytest.mark.parametrize("name", ["cornell_box", "teapot", "all", "planet"])
def test_closest_hit_and_shadow_queries_on_random_rays(pt, name):
    sb = hp.prepare_fixture(fixture(name))
    cam = hp.camera_to_gpu(sb.camera, 64, 64)
    rng = np.random.default_rng(5)
    n = 20000
    o = rng.uniform(-6, 6, (n, 3)).astype(np.float32) + np.array(sb.camera.position, dtype=np.float32) * 0.5
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    d[:64] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 64)] * rng.choice([-1, 1], (64, 1))  # axis-parallel: 1/0 lanes
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    max_t = rng.uniform(0.1, 30, n).astype(np.float32)
    _upload(pt, sb)
    gid, gt, gs = pt.kat_trace(cam, rays, max_t)
    oid, ot, os_ = orc.trace(orc.OracleScene(sb), cam, rays, max_t)
    assert (gid == oid).all() and (gs == os_).all()
    hit = oid != 0xFFFFFFFF
    assert same_bits(gt[hit], ot[hit]) == 0


def _shape(t, **kw):
    return hp.Shape(t, **kw)


def test_all_17_primitive_types_and_csg(pt):
    mat = hp.Material(base_color=(0.7, 0.6, 0.5), roughness=0.6)
    glass = hp.Material(base_color=(1, 1, 1), transmission=1.0, ior=1.4, roughness=0.0)
    light = hp.Material(emission=(1, 0.9, 0.8), emission_strength=5.0)
    shapes = [
        _shape("plane", position=(0, -1, 0), material=mat), _shape("sphere", position=(0, 6, 0), radius=1.5, material=light),
        _shape("sphere", position=(-4, 0, 6), material=glass), _shape("cube", position=(-2, 0, 6), radius=0.7, material=mat),
        _shape("cylinder", position=(0, 0, 6), normal=(0.2, 1, 0.1), radius=0.5, height=1.6, material=mat),
        _shape("cone", position=(2, -1, 6), radius=0.6, radius2=0.09, height=1.8, material=mat),
        _shape("torus", position=(4, 0, 6), radius=0.7, radius2=0.2, material=mat),
        _shape("disc", position=(-4, 1.5, 5), normal=(0, -1, 0.3), radius=0.6, material=mat),
        _shape("triangle", v0=(-3, 1, 5), v1=(-2, 1, 5.2), v2=(-2.5, 2, 5), material=mat),
        _shape("mandelbulb", position=(-1, 1.8, 5), radius=0.6, material=mat),
        _shape("julia", position=(0.5, 1.8, 5), radius=0.6, rotation=(-0.2, 0.6, -0.2), radius2=0.4, max_iterations=10, material=mat),
        _shape("ellipsoid", position=(2, 1.8, 5), radius=0.4, radius2=0.7, height=0.5, material=glass),
        _shape("paraboloid", position=(3.5, 1.2, 5), radius=0.5, height=0.9, material=mat),
        _shape("hyperboloid", position=(-3, -0.2, 3.5), radius=0.25, height=1.0, material=mat),
        _shape("mebius", position=(0, -0.2, 3.5), radius=0.6, material=mat),
        _shape("pyramid", position=(2, -1, 3.5), radius=0.5, height=1.0, material=mat),
        _shape("tetrahedron", position=(3.5, -0.3, 3.5), radius=0.6, material=mat),
        _shape("skybox", position=(0, 0, 0), radius=100.0, material=mat),
        # subtraction shapes: sphere out of the cube, cylinder out of the cone, and an inert negative disc
        _shape("sphere", position=(-2, 0.3, 5.5), radius=0.5, negative=True),
        _shape("cylinder", position=(2, -0.5, 5.6), normal=(0, 0, 1), radius=0.2, height=2.0, negative=True),
        _shape("cube", position=(4, 0, 5.6), radius=0.3, negative=True),
        _shape("cone", position=(-4, -1.2, 5.3), radius=0.5, radius2=0.2, height=1.5, negative=True),
        _shape("ellipsoid", position=(0, 0.2, 5.6), radius=0.3, radius2=0.3, height=0.3, negative=True),
        _shape("disc", position=(0, 0, 4), normal=(0, 0, 1), radius=3.0, negative=True),
    ]
    cam = hp.CameraConfig(position=(0, 1, -3), fov=70.0, fractal_march_steps=96)
    sb = hp.prepare_shapes(shapes, cam)
    got, ref = _compare(pt, sb, 160, 100, 3)
    present = set(np.unique(ref["primary_id"]).tolist())
    assert set(range(2, 17)) <= present  # every finite primitive type is actually seen by camera rays


def test_textures_uv_paths_and_textured_skybox(pt):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    sky = rng.integers(0, 256, (16, 32, 4), dtype=np.uint8)
    mat = hp.Material(base_color=(0.9, 0.9, 0.9), roughness=0.7)
    shapes = [
        _shape("skybox", texture="sky", material=hp.Material(base_color=(0.8, 0.9, 1.0))),
        _shape("plane", position=(0, -1, 0), texture="img", texture_scale=0.37, material=mat),
        _shape("sphere", position=(-2, 0, 5), texture="img", texture_scale=3.0, material=mat),
        _shape("cube", position=(0, 0, 5), radius=0.8, texture="img", material=mat),
        _shape("cylinder", position=(2, 0, 5), radius=0.5, height=1.5, texture="img", material=mat),
        _shape("cone", position=(3.5, -1, 5), radius=0.6, radius2=0.1, height=1.5, texture="img", material=mat),
        _shape("torus", position=(-3.5, 0, 4), radius=0.6, radius2=0.2, texture="img", material=mat),
        _shape("ellipsoid", position=(0, 2, 5), radius=0.5, radius2=0.8, height=0.4, texture="img", material=mat),
        _shape("triangle", v0=(-1, 1, 4), v1=(1, 1, 4), v2=(0, 2.5, 4.5), uv0=(0.1, 0.2), uv1=(1.7, 0.3), uv2=(0.5, -1.2),
               texture="img", material=mat),
        _shape("disc", position=(2.5, 2, 5), normal=(0, -0.5, -1), radius=0.7, texture="img", material=mat),
        _shape("pyramid", position=(-2, -1, 3), radius=0.5, height=1.0, texture="img", material=mat),
        _shape("paraboloid", position=(1.2, -1, 3), radius=0.4, height=0.8, texture="img", material=mat),
        _shape("hyperboloid", position=(-0.5, -0.4, 3), radius=0.2, height=1.0, texture="img", material=mat),
        _shape("mebius", position=(3, 1.5, 4), radius=0.5, texture="img", material=mat),
        _shape("tetrahedron", position=(-3, 2, 5), radius=0.6, texture="img", material=mat),
    ]
    sb = hp.prepare_shapes(shapes, hp.CameraConfig(position=(0, 0.8, -2)), tex_loader=lambda p: {"img": img, "sky": sky}[p])
    assert sb.mats[0]["texture_id"] == 1 and sb.mats[1]["texture_id"] == 2
    _compare(pt, sb, 144, 96, 2)


def test_no_lights_still_consumes_nee_random_numbers(pt):
    sb = hp.prepare_fixture(fixture("teapot"))
    assert len(sb.lights) == 0
    got, ref = _compare(pt, sb, 96, 54, 5)
    assert got["st"]["shadow_rays"] == 0 and got["st"]["bounces"] > got["st"]["paths"]


def test_empty_scene_and_planes_only(pt):
    sb = hp.prepare_shapes([], hp.CameraConfig(position=(0, 2, -10)))
    assert len(sb.figs) == 0 and len(sb.nodes) == 1
    got, _ = _compare(pt, sb, 33, 17, 2)
    assert (got["ids"] == 0xFFFFFFFF).all()
    sb = hp.prepare_shapes([_shape("plane", position=(0, -1, 0), material=hp.Material(emission=(1, 1, 1), emission_strength=1.0))],
                           hp.CameraConfig(position=(0, 0, 0)))
    _compare(pt, sb, 40, 30, 2)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (17, 33), (129, 65)])
def test_ragged_frame_sizes(pt, w, h):
    _compare(pt, hp.prepare_fixture(fixture("cornell_box")), w, h, 2)


def test_bounce_limits_and_tone_mappers(pt):
    sb = hp.prepare_fixture(fixture("cornell_box"))
    for mb in (0, 1, 2, 5):
        sb.camera.max_bounces = mb
        _compare(pt, sb, 48, 27, 2)
    sb.camera.max_bounces = 16
    for tm in (1, 2):
        sb.camera.tone_mapper = tm
        sb.camera.exposure = 1.7
        _compare(pt, sb, 48, 27, 2)
    sb.camera.firefly_clamp = 0.5
    _compare(pt, sb, 48, 27, 3)


def test_batched_equals_stepwise_and_nonzero_frame_offsets(pt):
    sb = hp.prepare_fixture(fixture("all"))
    w, h, n = 96, 54, 5
    cam = hp.camera_to_gpu(sb.camera, w, h)
    _upload(pt, sb)
    pt.resize(w, h)
    pt.trace_n(cam, 1000, 1, n)
    batched = pt.read_accum()
    pt.clear()
    for k in range(n):
        c = cam.copy()
        c[0]["frame_index"], c[0]["sample_count"] = 1000 + k, 1 + k
        pt.trace(c)
    assert same_bits(pt.read_accum(), batched) == 0
    ref = orc.render(orc.OracleScene(sb), cam, n, first_frame=1000, first_sample=1)
    assert same_bits(batched, ref["accum"]) == 0
    # continuing an accumulation: samples 6..8 on top of 1..5
    pt.trace_n(cam, 1005, 6, 3)
    ref2 = orc.render(orc.OracleScene(sb), cam, 3, first_frame=1005, first_sample=6, accum=ref["accum"].copy())
    assert same_bits(pt.read_accum(), ref2["accum"]) == 0


def test_frame_loop_follows_reference_ordering(pt):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h = 64, 36
    loop = host.FrameLoop(sc, pt, w, h)
    for _ in range(3):
        loop.render_frame()
    a3 = pt.read_accum()
    sb = hp.prepare_fixture(fixture("cornell_box"))
    osc = orc.OracleScene(sb)
    cam = hp.camera_to_gpu(sb.camera, w, h)
    assert same_bits(a3, orc.render(osc, cam, 3, first_frame=0, first_sample=1)["accum"]) == 0
    loop.reset()            # camera moved: sample_count restarts, frame_index keeps running
    loop.render_frames(4)   # batched frames 3..6 as samples 1..4
    assert same_bits(pt.read_accum(), orc.render(osc, cam, 4, first_frame=3, first_sample=1)["accum"]) == 0
    assert loop.acc.sample_count == 4 and loop.acc.frame_index == 7


@pytest.mark.parametrize("name", ["cornell_box", "test", "unit_logo"])  # one job counter; sixteen; path pools over a mesh
def test_tiles_and_interleaved_partitions_reassemble_the_frame(pt, name):
    sb = hp.prepare_fixture(fixture(name))
    w, h, n = 100, 70, 3
    cam = hp.camera_to_gpu(sb.camera, w, h)
    _upload(pt, sb)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    full, full_out = pt.read_accum(), pt.read_output()
    pt.clear()
    for tile in [(0, 0, 37, 70), (37, 0, 63, 29), (37, 29, 63, 41)]:
        pt.set_tile(*tile)
        pt.trace_n(cam, 0, 1, n)
    assert same_bits(pt.read_accum(), full) == 0 and (pt.read_output() == full_out).all()
    pt.set_tile(0, 0, w, h)
    from pathtracer_amd.multigpu import owned_blocks
    total = np.zeros_like(full)
    for world in (2, 3):
        total[:] = 0
        for rank in range(world):
            pt.clear()
            pt.set_partition(rank, world)
            pt.trace_n(cam, 0, 1, n)
            part = pt.read_accum()
            mask = owned_blocks(w, h, rank, world)
            assert (part[~mask] == 0).all() and same_bits(part[mask], full[mask]) == 0
            total += part
        assert same_bits(total, full) == 0
    pt.set_partition(0, 1)


def test_sample_split_chan_merge_matches_single_run_within_tolerance(pt):
    from pathtracer_amd.multigpu import chan_merge, sample_ranges
    sb = hp.prepare_fixture(fixture("cornell_box"))
    w, h, total = 64, 36, 24
    cam = hp.camera_to_gpu(sb.camera, w, h)
    _upload(pt, sb)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, total)
    mean_all, m2_all = pt.read_accum()[..., :3], pt.read_moments()
    parts = []
    for first, cnt in sample_ranges(total, 3):
        pt.clear()
        pt.trace_n(cam, first, 1, cnt)
        parts.append((cnt, pt.read_accum(), pt.read_moments()))
    # device merge: partial 0 is the accumulator; partials 1 and 2 are rendered by a second context (as
    # another GPU's would be) and folded in through ptx_merge_moments from its device buffers
    other = PathTracer(0)
    _upload(other, sb)
    other.enable_moments(True)
    other.resize(w, h)
    pt.clear()
    pt.trace_n(cam, 0, 1, parts[0][0])
    n_a = parts[0][0]
    cn, cm, cm2 = parts[0][0], parts[0][1][..., :3].copy(), parts[0][2].copy()
    for (first, cnt), (_, acc, m2) in zip(sample_ranges(total, 3)[1:], parts[1:]):
        other.clear()
        other.trace_n(cam, first, 1, cnt)
        other.sync()
        pt.merge_moments(0, w * h, n_a, other.device_accum_ptr(), other.device_moments_ptr(), cnt, cam)
        pt.sync()
        n_a += cnt
        cn, cm, cm2 = chan_merge(cn, cm, cm2, cnt, acc[..., :3], m2)
    other.close()
    got_mean, got_m2 = pt.read_accum()[..., :3], pt.read_moments()
    assert same_bits(got_mean, cm) == 0 and same_bits(got_m2, cm2) == 0  # device merge == numpy merge, bitwise
    # merged vs sequential: rounding only.  Tolerance 1e-5 relative to the pixel's scale.
    scale = np.maximum(np.abs(mean_all), 1e-3)
    assert np.max(np.abs(got_mean - mean_all) / scale) < 1e-5
    assert np.max(np.abs(got_m2 - m2_all) / np.maximum(np.abs(m2_all), 1e-2)) < 1e-3
    out_ref = orc.tonemap(cam, got_mean.reshape(-1, 3)).reshape(h, w, 4)
    assert (pt.read_output() == out_ref).all()
    pt.enable_moments(False)


def test_deep_bvh_heightfield_matches_oracle(pt):
    sc = host.Scene.from_fixture(fixture("all"))
    sc.append_heightfield(48, glass_every=16)  # 4608 triangles, BVH depth ~ 12
    sc.build()
    b = sc.buffers()
    sb = hp.SceneBuffers(b["figs"], b["mats"], b["nodes"], b["prims"], b["lights"], b["infinite"], b["tex_pixels"], b["tex_infos"],
                         hp.scene_from_fixture(fixture("all"))[1])
    _compare(pt, sb, 128, 72, 3)


@pytest.mark.parametrize("shares", ["12,2,4,4", "4,2,16,6", "64,2,4,1"])
def test_deep_bvh_heightfield_in_single_visit_mode_matches_oracle(pt, shares):
    """What trees of millions of primitives run (one node visit per WALK step, lanes at leaf primitives as a state of their
    own), forced on a tree the oracle finishes in seconds, with the shares of both deep scene classes and an extreme one."""
    sw, sp, sm, sl = (int(v) for v in shares.split(","))
    pt.set_options(walk_mode=1, share_walk=sw, share_post=sp, share_march=sm, share_leaf=sl)
    sc = host.Scene.from_fixture(fixture("all"))
    sc.append_heightfield(48, glass_every=16)
    sc.build()
    b = sc.buffers()
    sb = hp.SceneBuffers(b["figs"], b["mats"], b["nodes"], b["prims"], b["lights"], b["infinite"], b["tex_pixels"], b["tex_infos"],
                         hp.scene_from_fixture(fixture("all"))[1])
    try:
        _compare(pt, sb, 128, 72, 3)
        assert "sched" in pt.kernel_name()
    finally:
        pt.reset_options()


@pytest.mark.parametrize("name,quads", [("cornell_box", 0), ("teapot", 0), ("fractals", 0), ("all", 0), ("planet", 0), ("all", 40)])
def test_scene_relayout_on_the_device_equals_the_host_relayout(pt, name, quads):
    """ptx_upload_scene builds the kernels' records on the host for small scenes and on the device for large ones:
    forced either way, the same kernel is picked and frames, ids, M2 and counters are identical."""
    sc = host.Scene.from_fixture(fixture(name))
    if quads:
        sc.append_heightfield(quads, glass_every=16)
    sc.build()
    w, h, n = 96, 54, 3
    cam = sc.camera(w, h)
    got = {}
    for where in ("host", "device"):
        pt.set_option("upload_path", 1 if where == "device" else 0)
        sc.upload_to(pt)
        pt.enable_moments(True)
        pt.resize(w, h)
        pt.trace_n(cam, 0, 1, n)
        st = pt.stats()
        got[where] = (pt.kernel_name(), pt.read_accum(), pt.read_moments(), pt.read_primary_ids(), pt.read_output(),
                      (st["paths"], st["bounces"], st["shadow_rays"]))
        pt.enable_moments(False)
    pt.reset_options()
    a, b = got["host"], got["device"]
    assert a[0] == b[0] and a[5] == b[5]
    assert same_bits(a[1], b[1]) == 0 and same_bits(a[2], b[2]) == 0 and (a[3] == b[3]).all() and (a[4] == b[4]).all()


def test_error_reporting_of_the_device_relayout():
    fresh = PathTracer(0)
    fresh.set_option("upload_path", 1)
    sb = hp.prepare_fixture(fixture("teapot"))
    bad = sb.prims.copy()
    bad[5] = len(sb.figs)
    with pytest.raises(PtxError, match="bvh_prims entry out of range"):
        fresh.upload_scene(sb.figs, sb.mats, sb.nodes, bad, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    figs = sb.figs.copy()
    figs[3]["material_idx"] = len(sb.mats)
    with pytest.raises(PtxError, match="material_idx out of range"):
        fresh.upload_scene(figs, sb.mats, sb.nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    nodes = sb.nodes.copy()
    leaf = int(np.nonzero(nodes["prim_count"] > 0)[0][0])
    nodes[leaf]["prim_count"] = 9
    with pytest.raises(PtxError, match="more than 7"):
        fresh.upload_scene(sb.figs, sb.mats, nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    nodes = sb.nodes.copy()
    nodes[leaf]["left_or_prim"] = len(sb.prims)
    with pytest.raises(PtxError, match="leaf range outside"):
        fresh.upload_scene(sb.figs, sb.mats, nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    nodes = sb.nodes.copy()
    inner = int(np.nonzero(nodes["prim_count"] == 0)[0][1])
    nodes[inner]["left_or_prim"] = inner
    with pytest.raises(PtxError, match="child index out of order"):
        fresh.upload_scene(sb.figs, sb.mats, nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    fresh.upload_scene(sb.figs, sb.mats, sb.nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)  # and recovers
    fresh.close()


def test_error_reporting(pt):
    fresh = PathTracer(0)
    cam = hp.camera_to_gpu(hp.CameraConfig(), 8, 8)
    with pytest.raises(PtxError, match="before ptx_upload_scene"):
        fresh.trace(cam)
    sb = hp.prepare_fixture(fixture("fractals"))
    _upload(fresh, sb)
    with pytest.raises(PtxError, match="before ptx_resize"):
        fresh.trace(cam)
    fresh.resize(16, 16)
    with pytest.raises(PtxError, match="differ from the frame size"):
        fresh.trace(cam)
    with pytest.raises(PtxError, match="tile outside"):
        fresh.set_tile(8, 8, 16, 16)
    with pytest.raises(PtxError, match="partition rank"):
        fresh.set_partition(2, 2)
    bad = sb.prims.copy()
    bad[0] = 99
    with pytest.raises(PtxError, match="out of range"):
        fresh.upload_scene(sb.figs, sb.mats, sb.nodes, bad, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    nodes = sb.nodes.copy()
    nodes[0]["prim_count"] = 9
    with pytest.raises(PtxError, match="more than 7"):
        fresh.upload_scene(sb.figs, sb.mats, nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    fresh.close()


@pytest.mark.parametrize("name,w,h,n", [("cornell_box", 160, 90, 5), ("teapot", 200, 112, 3), ("fractals", 64, 36, 3), ("all", 128, 72, 3)])
def test_both_kernel_families_and_claim_modes_give_identical_frames(pt, name, w, h, n):
    """The lockstep and the wave-scheduled kernels, with single-pixel and 64-pixel claims, are interchangeable:
    same accumulation, M2, ids, rgba8 and counters on every scene (the host picks per scene; here each is forced)."""
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    cam = sc.camera(w, h)
    results = {}
    # split: the scheduled fractal kernels' sample-granular jobs + fold ("2": in chunks of two samples); single: one node
    # visit per WALK step with the lanes at leaf primitives served as a state of their own (what deep trees run),
    # with leaf shares that serve them early ("64") and late ("2")
    for sched, tile, split, single in (("0", "1", "0", ""), ("1", "1", "0", ""), ("1", "1", "1", ""), ("1", "1", "2", ""), ("1", "64", "0", ""),
                                       ("0", "64", "0", ""), ("1", "1", "0", "64"), ("1", "64", "1", "2"), ("1", "64", "0", "5")):
        pt.reset_options()
        pt.set_options(kernel_family=int(sched), claim_tile=int(tile), split_samples=0 if split == "0" else 1,
                       split_chunk=2 if split == "2" else 1000)
        if single:
            pt.set_options(walk_mode=1, share_walk=4, share_post=2, share_march=4, share_leaf=int(single))
        sc.upload_to(pt)  # the family is chosen at upload
        assert ("sched" in pt.kernel_name()) == (sched == "1")
        pt.enable_moments(True)
        pt.resize(w, h)
        pt.trace_n(cam, 0, 1, n)
        st = pt.stats()
        results[(sched, tile, split, single)] = (pt.read_accum(), pt.read_moments(), pt.read_primary_ids(), pt.read_output(),
                                                 (st["paths"], st["bounces"], st["shadow_rays"]))
        pt.enable_moments(False)
    pt.reset_options()
    ref = results[("0", "1", "0", "")]
    for key, r in results.items():
        assert same_bits(r[0], ref[0]) == 0 and same_bits(r[1], ref[1]) == 0, key
        assert (r[2] == ref[2]).all() and (r[3] == ref[3]).all() and r[4] == ref[4], key
    sc.upload_to(pt)  # leave the context with the default choice


@pytest.mark.parametrize("name,w,h,n,family", [("teapot", 320, 180, 8, 1), ("all", 192, 108, 6, 1), ("fractals", 64, 36, 4, 1),
                                               ("cornell_box", 320, 180, 4, 0), ("rings", 640, 360, 3, 0), ("test", 640, 360, 4, 0)])
def test_reservation_sizes_do_not_change_the_frame(pt, name, w, h, n, family):
    """How many jobs a wave reserves at a time -- fixed sizes from 16 to 1024, a share of the jobs left that shrinks as the
    launch goes on, or sizes the wave sets by its own pace -- and from how many counters (1 to 16, the lockstep kernel's ranged
    variant) decides only which lane traces which sample: whole-pixel and sample jobs, both kernel families, every setting,
    same frame."""
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    cam = sc.camera(w, h)
    results = {}
    for split in (1, 0):
        for opts in ({}, dict(claim_tile=16), dict(claim_tile=1024), dict(claim_guided=4), dict(claim_guided=64),
                     dict(claim_guided=8, claim_tile=256), dict(claim_guided=0, claim_tile=128), dict(claim_paced=0, claim_guided=0),
                     dict(claim_paced=1, claim_cap=0), dict(claim_paced=64, claim_cap=1), dict(claim_paced=4, claim_tile=512),
                     dict(claim_ranges=1), dict(claim_ranges=16), dict(claim_ranges=3, claim_paced=0), dict(claim_ranges=16, claim_tile=1024),
                     dict(claim_ranges=7, claim_paced=64, claim_cap=0)):
            pt.reset_options()
            pt.set_options(kernel_family=family, split_samples=split, sorted_pool=0, **opts)
            sc.upload_to(pt)
            assert ("sched" in pt.kernel_name()) == (family == 1)
            want_ranged = family == 0 and opts.get("claim_ranges", 16 if name == "test" else 1) > 1
            pt.resize(w, h)
            pt.trace_n(cam, 0, 1, n)
            assert ("ranged" in pt.kernel_name()) == want_ranged, (pt.kernel_name(), opts)
            pt.trace_n(cam, n, n + 1, 2)
            st = pt.stats()
            results[(split, tuple(sorted(opts.items())))] = (pt.read_accum(), pt.read_primary_ids(), (st["paths"], st["bounces"], st["shadow_rays"]))
    pt.reset_options()
    ref = results[(0, ())]
    for key, r in results.items():
        assert same_bits(r[0], ref[0]) == 0 and (r[1] == ref[1]).all() and r[2] == ref[2], key
    sc.upload_to(pt)


def test_every_job_is_traced_once_when_the_counters_run_out(pt):
    """The end of a launch from sixteen counters: waves holding part of a reservation while every counter has run out, lanes
    that ask again after their counter ran out under them.  Timing decides which wave meets which case, so: many launches of
    many sizes (jobs from a few dozen to twenty per lane), whole-pixel jobs (a job traced twice would fold its samples twice)
    and sample jobs, each checked for its path count and against the frame from one counter."""
    sc = host.Scene.from_fixture(fixture("planet")).build(decode_textures=False)
    sizes = [(16, 16), (100, 70), (255, 130), (640, 360), (1001, 563), (1920, 1080)]
    for w, h in sizes:
        cam = sc.camera(w, h)
        pt.reset_options()
        pt.set_options(claim_ranges=1, claim_paced=0, split_samples=0)
        sc.upload_to(pt)
        pt.resize(w, h)
        pt.trace_n(cam, 0, 1, 3)
        ref = pt.read_accum()
        for split in (0, 1):
            for ranges, rep in ((16, 6), (5, 2), (2, 2)):
                pt.reset_options()
                pt.set_options(claim_ranges=ranges, split_samples=split, pool_over=4)  # lane-owned paths: the kernel with the counters
                sc.upload_to(pt)
                pt.resize(w, h)
                for _ in range(rep):
                    pt.clear()
                    pt.trace_n(cam, 0, 1, 3)
                    assert "ranged" in pt.kernel_name()
                    assert pt.stats()["paths"] == w * h * 3, (w, h, split, ranges)
                    assert same_bits(pt.read_accum(), ref) == 0, (w, h, split, ranges)
    pt.reset_options()
    sc.upload_to(pt)


def test_a_host_that_waits_for_every_frame_gets_whole_pixel_launches_and_the_same_frame(pt):
    """One sample per launch: back to back the launches overlap as sample jobs; a host that waits for each frame before it asks for
    the next finds the GPU idle, and from the second such launch on the library folds the sample in the lane (no staging, no
    fold kernel).  Same frame either way, and with the choice switched off."""
    import time
    sc = host.Scene.from_fixture(fixture("cornell_box")).build(decode_textures=False)
    w, h, n = 160, 90, 6
    cam = sc.camera(w, h)
    frames = {}
    for mode in ("burst", "waiting", "waiting, choice off", "slow host, busy GPU"):
        pt.reset_options()
        if mode.endswith("off"):
            pt.set_options(idle_whole=0)
        sc.upload_to(pt)
        pt.resize(w, h)
        if mode.startswith("slow"):  # a long launch first: the next ones come a millisecond apart and find it still running
            pt.trace_n(cam, 0, 1, 1)
            pt.sync()
            big = sc.camera(1920, 1080)
            pt.resize(1920, 1080)
            pt.trace_n(big, 0, 1, 64)
            for k in range(3):
                time.sleep(0.001)
                pt.trace_n(big, 64 + k, 65 + k, 1)
            pt.sync()
            assert pt.stats()["paths"] == 1920 * 1080 * 67
            pt.resize(w, h)
        for k in range(n):
            pt.trace_n(cam, k, k + 1, 1)
            if mode.startswith("waiting"):
                pt.sync()
                time.sleep(0.002)
        frames[mode] = (pt.read_accum(), pt.read_primary_ids(), pt.stats()["paths"])
    ref = frames["burst"]
    for mode, f in frames.items():
        assert same_bits(f[0], ref[0]) == 0 and (f[1] == ref[1]).all() and f[2] == ref[2] == w * h * n, mode
    pt.reset_options()
    sc.upload_to(pt)
