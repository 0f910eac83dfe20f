"""CPU tier: pins the oracle itself (known answers computed independently in pure Python / float64)
and the committed golden renders."""
import os

import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from util import GOLDEN_CASES, GOLDEN_DIR, fixture, same_bits


def _pcg(x):  # random.wgsl:6-10 in exact integer arithmetic
    state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


def test_pcg_hash_known_answers():
    for x in [0, 1, 2, 12345, 0xFFFFFFFF, 0x80000000, 16777259]:
        assert orc.lib().pto_pcg_hash(x) == _pcg(x)
    # literal values, so a change in both implementations is noticed
    assert _pcg(0) == 129708002 and _pcg(1) == 2831084092


def test_rand_sequence_matches_integer_model():
    for (px, py, frame) in [(0, 0, 0), (5, 7, 3), (1919, 1079, 1023), (65535, 65535, 0xFFFFFFFF)]:
        seq = orc.rand_sequence(px, py, frame, 16)
        s = _pcg((px + py * 65536 + frame * 16777259) & 0xFFFFFFFF)
        for v in seq:
            s = _pcg(s)
            assert v == np.float32(np.float32(s) / np.float32(4294967295.0))  # random.wgsl:16-19
        assert (seq >= 0).all() and (seq <= 1.0).all()


def test_rand_can_reach_one():
    # f32(u32) rounds to nearest: any state >= 2^32 - 128 gives exactly 1.0
    assert np.float32(np.float32(0xFFFFFFFF) / np.float32(4294967295.0)) == np.float32(1.0)


def _ulp_err(got, exact):
    got = got.astype(np.float64)
    ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    return np.abs(got - exact) / ulp


@pytest.mark.parametrize("op,lo,hi,bound", [
    ("sin", -30.0, 30.0, 2.5e-7), ("cos", -30.0, 30.0, 2.5e-7),  # absolute error (WGSL: 2^-11 on [-pi, pi])
])
def test_trig_absolute_accuracy(op, lo, hi, bound):
    x = np.linspace(lo, hi, 200001, dtype=np.float32)
    got = orc.math_op(op, x).astype(np.float64)
    exact = getattr(np, op)(x.astype(np.float64))
    assert np.max(np.abs(got - exact)) < bound


@pytest.mark.parametrize("op,gen,max_ulp", [
    ("asin", lambda: np.linspace(-1, 1, 100001), 4), ("acos", lambda: np.linspace(-1, 0.999, 100001), 4),
    ("log", lambda: np.geomspace(1e-30, 1e30, 100001), 3), ("exp2", lambda: np.linspace(-100, 100, 100001), 3),
    ("sqrt", lambda: np.geomspace(1e-30, 1e30, 100001), 0.5000001),
])
def test_unary_ulp_accuracy(op, gen, max_ulp):
    x = gen().astype(np.float32)
    got = orc.math_op(op, x)
    fn = {"asin": np.arcsin, "acos": np.arccos, "log": np.log, "exp2": np.exp2, "sqrt": np.sqrt}[op]
    exact = fn(x.astype(np.float64))
    assert np.max(_ulp_err(got, exact)) <= max_ulp


def test_atan2_and_pow_accuracy():
    rng = np.random.default_rng(7)
    y, x = rng.normal(size=100000).astype(np.float32), rng.normal(size=100000).astype(np.float32)
    got = orc.math_op("atan2", y, x)
    assert np.max(_ulp_err(got, np.arctan2(y.astype(np.float64), x.astype(np.float64)))) <= 6  # WGSL allows 4096 ULP
    assert orc.math_op("atan2", [0.0], [0.0])[0] == 0.0
    b = rng.uniform(1e-3, 300, 100000).astype(np.float32)
    e = rng.uniform(-3, 8, 100000).astype(np.float32)
    got = orc.math_op("pow", b, e).astype(np.float64)
    exact = np.power(b.astype(np.float64), e.astype(np.float64))
    ok = np.isfinite(exact) & (exact < 3e38) & (exact > 1e-37)
    rel = np.abs(got[ok] - exact[ok]) / exact[ok]
    assert rel.max() < 2e-5  # inherited from exp2(y * log2(x)), as WGSL defines pow
    # the tone curve's exponent on its domain
    c = np.linspace(0.0031308, 1.0, 50001, dtype=np.float32)
    got = orc.math_op("pow", c, np.full_like(c, np.float32(1.0 / 2.4))).astype(np.float64)
    assert np.max(np.abs(got - c.astype(np.float64) ** (1 / 2.4))) < 5e-7


def test_min_max_signed_zero_and_nan():
    a = np.array([0.0, -0.0, np.nan, 1.0, -0.0], dtype=np.float32)
    b = np.array([-0.0, 0.0, 1.0, np.nan, -0.0], dtype=np.float32)
    mn, mx = orc.math_op("min", a, b), orc.math_op("max", a, b)
    assert list(np.signbit(mn)) == [True, True, False, False, True] and mn[2] == 1.0 and mn[3] == 1.0
    assert list(np.signbit(mx)) == [False, False, False, False, True]


def test_unpack2x16float_is_exact_and_pack_truncates():
    halves = np.arange(0, 0x10000, dtype=np.uint32)
    packed = halves | (halves[::-1] << 16)
    out = orc.unpack2x16float(packed)
    exp = halves.astype(np.uint16).view(np.float16).astype(np.float32)
    fin = np.isfinite(exp)
    assert (out[fin, 0] == exp[fin]).all() and (out[::-1][fin, 1] == exp[fin]).all()
    # shape.rs:290-322 truncates the mantissa: the packed half never exceeds the input in magnitude
    rng = np.random.default_rng(3)
    v = rng.uniform(-4, 4, 2000).astype(np.float32)
    h = np.array([hp.f32_to_f16_bits(x) for x in v], dtype=np.uint16).view(np.float16).astype(np.float32)
    assert (np.abs(h) <= np.abs(v)).all() and np.max(np.abs(h - v)) < 4e-3
    assert hp.f32_to_f16_bits(np.float32(1.0)) == 0x3C00 and hp.f32_to_f16_bits(np.float32(-2.0)) == 0xC000
    assert hp.f32_to_f16_bits(np.float32(1e-10)) == 0 and hp.f32_to_f16_bits(np.float32(1e10)) == 0x7C00


def _one_fig_scene(shape: hp.Shape):
    return hp.prepare_shapes([shape], hp.CameraConfig())


def test_sphere_plane_cube_known_hits():
    cam = hp.camera_to_gpu(hp.CameraConfig(), 8, 8)
    sc = orc.OracleScene(_one_fig_scene(hp.Shape("sphere", position=(0, 0, 5), radius=1.0)))
    r = orc.intersect(sc, cam, 0, np.array([[0, 0, 0, 0, 0, 1], [0, 2, 0, 0, 0, 1], [0, 0, 5, 0, 0, 1]], dtype=np.float32))
    assert r[0, 0] == 1 and r[0, 1] == 4.0 and tuple(r[0, 5:8]) == (0, 0, -1)
    assert r[1, 0] == 0
    assert r[2, 0] == 1 and r[2, 1] == 1.0  # from inside: far root
    sc = orc.OracleScene(_one_fig_scene(hp.Shape("plane", position=(0, -1, 0), normal=(0, 2, 0))))
    r = orc.intersect(sc, cam, 0, np.array([[0, 0, 0, 0, -1, 0], [0, 0, 0, 1, 0, 0], [0, -2, 0, 0, 1, 0]], dtype=np.float32))
    assert r[0, 0] == 1 and r[0, 1] == 1.0 and tuple(r[0, 5:8]) == (0, 1, 0)
    assert r[1, 0] == 0  # parallel
    assert r[2, 0] == 1 and tuple(r[2, 5:8]) == (0, -1, 0)  # normal faces the ray
    sc = orc.OracleScene(_one_fig_scene(hp.Shape("cube", position=(0, 0, 4), radius=1.0)))
    r = orc.intersect(sc, cam, 0, np.array([[0, 0, 0, 0, 0, 1], [0.25, 0.5, 0, 0, 0, 1]], dtype=np.float32))
    assert r[0, 1] == 3.0 and tuple(r[1, 5:8]) == (0, 0, -1) and tuple(r[1, 8:10]) == (0.625, 0.75)


def test_negative_disc_is_inert_and_negative_sphere_carves():
    cam = hp.camera_to_gpu(hp.CameraConfig(), 8, 8)
    ray = np.array([[0, 0, 0, 0, 0, 1]], dtype=np.float32)
    solid = hp.Shape("sphere", position=(0, 0, 5), radius=1.0)
    carve = hp.Shape("sphere", position=(0, 0, 4.2), radius=0.5, negative=True)
    disc = hp.Shape("disc", position=(0, 0, 4.2), normal=(0, 0, 1), radius=0.5, negative=True)
    hid, t, _ = orc.trace(orc.OracleScene(hp.prepare_shapes([solid, disc], hp.CameraConfig())), cam, ray)
    assert hid[0] == 0 and t[0] == 4.0  # bvh.wgsl:60-62: no disc case
    hid, t, _ = orc.trace(orc.OracleScene(hp.prepare_shapes([solid, carve], hp.CameraConfig())), cam, ray)
    assert hid[0] == 0 and t[0] > 4.5  # front cap removed: the ray re-enters past the carved region


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_oracle_reproduces_golden(name):
    w, h, n = GOLDEN_CASES[name]
    g = np.load(os.path.join(GOLDEN_DIR, f"{name}_{w}x{h}x{n}.npz"))
    sb = hp.prepare_fixture(fixture(name))
    cam = hp.camera_to_gpu(sb.camera, w, h)
    r = orc.render(orc.OracleScene(sb), cam, n, want_radiance=True, want_m2=True, counters=True)
    assert (r["primary_id"] == g["primary_id"]).all()
    assert same_bits(r["radiance"], g["radiance"]) == 0
    assert same_bits(r["accum"], g["accum"]) == 0
    assert (r["rgba8"] == g["rgba8"]).all()
    c = r["counters"]
    assert [c["paths"], c["bounces"], c["shadow_rays"]] == list(g["counters"])


def test_oracle_batched_equals_stepwise_and_tiled():
    sb = hp.prepare_fixture(fixture("cornell_box"))
    sc = orc.OracleScene(sb)
    cam = hp.camera_to_gpu(sb.camera, 40, 24)
    full = orc.render(sc, cam, 3)
    acc = None
    for k in range(3):
        acc = orc.render(sc, cam, 1, first_frame=k, first_sample=k + 1, accum=acc)["accum"]
    assert same_bits(acc, full["accum"]) == 0
    acc = np.zeros((24, 40, 4), dtype=np.float32)
    for tile in [(0, 0, 17, 24), (17, 0, 23, 11), (17, 11, 23, 13)]:
        orc.render(sc, cam, 3, tile=tile, accum=acc)
    assert same_bits(acc, full["accum"]) == 0


def test_division_contract_reciprocal_flushes_subnormals_and_is_correctly_rounded():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1e3, 1e3, 100000), 10.0 ** rng.uniform(-37, 37, 100000)]).astype(np.float32)
    r = orc.math_op("rcp", x)
    assert same_bits(r, (1.0 / x.astype(np.float64)).astype(np.float32)) == 0
    y = rng.uniform(-50, 50, x.size).astype(np.float32)
    with np.errstate(over="ignore"):
        assert same_bits(orc.math_op("div", y, x), y * r) == 0  # a / b = a * rcp(b)
    inf = np.float32(np.inf)
    cases = [(0.0, inf), (-0.0, -inf), (1e-39, inf), (-1e-42, -inf), (1.17549435e-38, 2.0 ** 126), (2.0 ** 126, 2.0 ** -126),
             (2.0 ** 127, 0.0), (-3e38, -0.0), (np.inf, 0.0), (-np.inf, -0.0)]
    got = orc.math_op("rcp", np.array([c[0] for c in cases], dtype=np.float32))
    assert same_bits(got, np.array([c[1] for c in cases], dtype=np.float32)) == 0
    assert np.isnan(orc.math_op("rcp", np.array([np.nan], dtype=np.float32))[0])


def test_oracle_reproduces_the_digests_of_every_reference_scene():
    """All 30 scene files of the reference as fixtures: counters and SHA-256 of ids / mean / M2 / rgba8 (tests/golden/make_golden.py)."""
    import json
    import sys
    sys.path.insert(0, GOLDEN_DIR)
    from make_golden import scene_digest
    want = json.load(open(os.path.join(GOLDEN_DIR, "scene_digests.json")))
    assert len(want) == 30
    for name, d in want.items():
        assert scene_digest(fixture(name)) == d, name
