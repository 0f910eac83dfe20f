"""GPU tier: ptx_bvh_build (Bvh::build, src/accel/bvh.rs:37-213, level by level on the device) gives the node
array and primitive order of the sequential builder bit for bit.  The small sets are checked against the numpy
restatement, the large ones against the host library's builder (itself checked against the restatement on
the CPU tier); a zero bound is compared as +0.0, which is what the device builder returns."""
import numpy as np
import pytest

from oracle import host_prep as hp
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer, PtxError
from util import box_sets, fixture, grid_triangle_boxes, random_boxes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    p = PathTracer(0)
    yield p
    p.close()


def _canon(nodes):
    nodes = nodes.copy()
    nodes["aabb_min"] += np.float32(0.0)
    nodes["aabb_max"] += np.float32(0.0)
    return nodes.tobytes()


def _check(pt, aabbs, want=None):
    nodes, prims = pt.bvh_build(aabbs)
    wn, wp = want if want is not None else host.bvh_build(aabbs)
    assert len(nodes) == len(wn)
    assert (prims == wp).all()
    assert _canon(nodes) == _canon(wn)


@pytest.mark.parametrize("name", sorted(box_sets()))
def test_small_sets_equal_the_restatement(pt, name):
    aabbs = box_sets()[name]
    _check(pt, aabbs, hp.bvh_build([(b[:3].copy(), b[3:].copy()) for b in aabbs]))


@pytest.mark.parametrize("n", [2, 3, 6, 9, 63, 64, 65, 255, 256, 257, 258, 511, 513, 1000, 4097, 8191, 8192, 8193, 8200, 16385, 20000])
def test_random_boxes_around_the_wave_and_range_limits(pt, n):
    _check(pt, random_boxes(n, n))


def test_clustered_and_skewed_distributions(pt):
    rng = np.random.default_rng(11)
    a = random_boxes(30000, 12, spread=1.0, size=0.01)
    a[:, [0, 3]] = np.sign(a[:, [0, 3]]) * a[:, [0, 3]] ** 2 * 50   # dense near x = 0, sparse far out
    _check(pt, a)
    b = np.concatenate([random_boxes(5000, 13, spread=0.01, size=0.001) + np.float32(k * 7.0) for k in range(5)])
    _check(pt, b[rng.permutation(len(b))])
    c = random_boxes(3000, 14)
    c[:, [0, 3]] = np.float32(-1.5), np.float32(-1.25)               # every centroid equal along x
    _check(pt, c)


def test_ranges_that_do_not_split_fall_back_to_the_median(pt):
    one = random_boxes(1, 15)
    _check(pt, np.repeat(one, 5000, axis=0))                          # no extent at all: bvh.rs:88-92 on every level
    two = np.concatenate([np.repeat(one, 700, axis=0), np.repeat(one + np.float32(3.0), 900, axis=0)])
    _check(pt, two[np.random.default_rng(16).permutation(len(two))])


def test_height_field_mesh(pt):
    _check(pt, grid_triangle_boxes(300))                              # 180 000 triangles, many equal coordinates


def test_one_million_boxes(pt):
    _check(pt, random_boxes(1_000_000, 17, spread=100.0, size=0.3))


def test_nan_box_is_refused_and_empty_list_gives_the_placeholder_node(pt):
    a = random_boxes(100, 18)
    a[37, 4] = np.nan
    with pytest.raises(PtxError):
        pt.bvh_build(a)
    nodes, prims = pt.bvh_build(np.zeros((0, 6), dtype=np.float32))
    assert len(nodes) == 1 and len(prims) == 0 and nodes.tobytes() == bytes(32)


@pytest.mark.parametrize("name", ["teapot", "all", "cornell_box"])
def test_scene_built_with_the_device_builder_uploads_the_same_buffers(pt, name):
    host_scene = host.Scene.from_fixture(fixture(name)).build()
    dev_scene = host.Scene.from_fixture(fixture(name)).build(tracer=pt)
    hb, db = host_scene.buffers(), dev_scene.buffers()
    assert (hb["prims"] == db["prims"]).all()
    assert _canon(hb["nodes"]) == _canon(db["nodes"])
    for k in ("figs", "mats", "lights", "infinite", "tex_pixels", "tex_infos"):
        assert hb[k].tobytes() == db[k].tobytes()


@pytest.mark.parametrize("n", [40, 9000, 30000])
def test_zero_bounds_of_either_sign_come_back_as_plus_zero(pt, n):
    """Boxes that touch the coordinate planes with -0.0 as well as +0.0 bounds, below and above the one-wave range limit:
    the device builder returns every zero bound as +0.0 (the documented normalisation; min / max of zeros of both signs
    has no defined sign in the reference's f32::min / f32::max either), all other bits as the host builder's."""
    a = random_boxes(n, 21, spread=1.0, size=0.05)
    k = np.arange(n)
    a[k % 3 == 0, 0] = np.float32(-0.0)    # min x = -0.0
    a[k % 3 == 1, 0] = np.float32(0.0)     # min x = +0.0
    a[k % 3 == 0, 3] = np.abs(a[k % 3 == 0, 3]) + np.float32(0.01)
    a[k % 3 == 1, 3] = np.abs(a[k % 3 == 1, 3]) + np.float32(0.01)
    a[k % 5 == 0, 4] = np.float32(-0.0)    # max y = -0.0
    a[k % 5 == 0, 1] = -np.abs(a[k % 5 == 0, 1]) - np.float32(0.01)
    nodes, prims = pt.bvh_build(a)
    wn, wp = host.bvh_build(a)
    assert (prims == wp).all() and _canon(nodes) == _canon(wn)
    for fld in ("aabb_min", "aabb_max"):
        v = nodes[fld]
        assert not (np.signbit(v) & (v == 0)).any()   # no -0.0 leaves the device builder
