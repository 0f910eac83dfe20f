"""GPU tier: the edges of the path's scope that the scene loaders never produce or that only the benchmark
reaches: a skybox figure inside the BVH (figures/skybox.wgsl:30-62), the 10 M-triangle scene at the benchmark's
own size with default modes, and the option interface's error behaviour."""
import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer, PtxError
from pathtracer_amd.records import NODE_DT
from util import fixture, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.close()


def _with_skybox_leaf(sb: hp.SceneBuffers, sky_fig: int) -> hp.SceneBuffers:
    """The scene's tree under a new root whose right child is a one-primitive leaf holding the skybox figure:
    what the reference's build_bvh (src/app/state.rs:543-567) never makes, but its bind group accepts."""
    f = sb.figs[sky_fig]
    r = np.float32(f["radius"])
    lo, hi = f["position"] - r, f["position"] + r
    old = sb.nodes
    nodes = np.zeros(len(old) + 2, dtype=NODE_DT)
    nodes[1:1 + len(old)] = old
    inner = nodes["prim_count"][1:1 + len(old)] == 0
    nodes["left_or_prim"][1:1 + len(old)][inner] += 1          # right-child indices move with the array
    nodes[0]["aabb_min"], nodes[0]["aabb_max"] = np.minimum(lo, old[0]["aabb_min"]), np.maximum(hi, old[0]["aabb_max"])
    nodes[0]["left_or_prim"], nodes[0]["prim_count"] = len(old) + 1, 0
    leaf = nodes[len(old) + 1]
    leaf["aabb_min"], leaf["aabb_max"], leaf["left_or_prim"], leaf["prim_count"] = lo, hi, len(sb.prims), 1
    prims = np.concatenate([sb.prims, np.array([sky_fig], dtype=np.uint32)])
    return hp.SceneBuffers(sb.figs, sb.mats, nodes, prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos, sb.camera)


@pytest.mark.parametrize("textured", [False, True])
def test_skybox_figure_inside_the_bvh_is_intersected_like_the_reference(pt, textured):
    rng = np.random.default_rng(4)
    sky_img = rng.integers(0, 256, (16, 32, 4), dtype=np.uint8)
    mat = hp.Material(base_color=(0.8, 0.7, 0.6), roughness=0.5)
    sky_mat = hp.Material(base_color=(0.9, 0.8, 1.0), roughness=0.9)
    S = hp.Shape
    shapes = [
        S("skybox", position=(0.5, 0.2, 1.0), radius=9.0, material=sky_mat, **({"texture": "sky", "texture_scale": 2.0} if textured else {})),
        S("plane", position=(0, -1, 0), material=mat),
        S("sphere", position=(0, 4, 3), radius=0.8, material=hp.Material(emission=(1, 0.9, 0.8), emission_strength=6.0)),
        S("sphere", position=(-1.5, 0, 4), material=hp.Material(base_color=(1, 1, 1), transmission=1.0, ior=1.5, roughness=0.0)),
        S("cube", position=(1.5, -0.3, 4), radius=0.7, material=mat),
        S("pyramid", position=(0, -1, 2.5), radius=0.5, height=1.0, material=mat),
    ]
    sb = hp.prepare_shapes(shapes, hp.CameraConfig(position=(0, 0.5, -3), fov=65.0), tex_loader=lambda p: sky_img)
    assert 0 not in sb.prims.tolist()            # the loader's tree leaves the skybox out ...
    sb2 = _with_skybox_leaf(sb, 0)               # ... this one holds it
    cam = hp.camera_to_gpu(sb2.camera, 64, 64)
    # closest-hit and occlusion queries, rays from inside and outside the skybox sphere
    n = 12000
    o = rng.uniform(-14, 14, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    max_t = rng.uniform(0.1, 40, n).astype(np.float32)
    pt.upload_scene(sb2.figs, sb2.mats, sb2.nodes, sb2.prims, sb2.lights, sb2.infinite, sb2.tex_pixels, sb2.tex_infos)
    gid, gt, gs = pt.kat_trace(cam, rays, max_t)
    oid, ot, os_ = orc.trace(orc.OracleScene(sb2), cam, rays, max_t)
    assert (gid == oid).all() and (gs == os_).all()
    hit = oid != 0xFFFFFFFF
    assert same_bits(gt[hit], ot[hit]) == 0
    assert (oid == 0).sum() > n // 10            # the far-root sphere is what many of those rays hit
    # and a render: camera rays that miss everything else end on the skybox figure and are shaded with its material
    from test_gpu_parity import _compare
    for family in (0, 1):
        pt.set_option("kernel_family", family)
        try:
            got, ref = _compare(pt, sb2, 96, 64, 3)
        finally:
            pt.reset_options()
        assert (ref["primary_id"] == 0).any() and not (ref["primary_id"] == 0xFFFFFFFF).any()


def test_option_interface_rejects_what_it_does_not_know(pt):
    with pytest.raises(PtxError, match="unknown key"):
        pt.set_option("no_such_option", 1)
    with pytest.raises(PtxError, match="out of range"):
        pt.set_option("kernel_family", 2)
    with pytest.raises(PtxError, match="out of range"):
        pt.set_option("claim_tile", 8)
    with pytest.raises(PtxError, match="out of range"):
        pt.set_option("walk_mode", 3)
    pt.reset_options()


def test_benchmark_mesh_scene_at_full_size_with_default_modes(pt):
    """all.yaml + the 9 999 392-triangle height field exactly as `bench.py --workload all_mesh` sets it up: tree built
    on the GPU, records laid out on the GPU, deep-walk mode picked by the library; no option is set."""
    sc = host.Scene.from_fixture(fixture("all"))
    sc.append_heightfield(2236)
    sc.build(decode_textures=False, tracer=pt)
    w, h, n = 1920, 1080, 2
    cam = sc.camera(w, h)
    sc.upload_to(pt)
    assert pt.kernel_name() == "ptx_sched_kernel_samples"  # deep tree: sample jobs + fold for launches of every size
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    a, ids, out = pt.read_accum(), pt.read_primary_ids(), pt.read_output()
    st = pt.stats()
    assert st["paths"] == w * h * n and 1.5 < st["bounces"] / st["paths"] < 3.0
    assert np.isfinite(a).all() and (a[..., 3] == 1).all()
    # determinism, and one sample after the other == both at once
    pt.clear()
    pt.trace_n(cam, 0, 1, 1)
    pt.trace_n(cam, 1, 2, 1)
    assert same_bits(pt.read_accum(), a) == 0 and (pt.read_primary_ids() == ids).all()
    # the interleaved 4-way partition reassembles the frame
    total = np.zeros_like(a)
    for r in range(4):
        pt.clear()
        pt.set_partition(r, 4)
        pt.trace_n(cam, 0, 1, n)
        total += pt.read_accum()
    pt.set_partition(0, 1)
    assert same_bits(total, a) == 0
    # the oracle on two full-resolution tiles of the same frames: at the horizon and at the frame centre
    b = sc.buffers()
    sb = hp.SceneBuffers(b["figs"], b["mats"], b["nodes"], b["prims"], b["lights"], b["infinite"], b["tex_pixels"], b["tex_infos"], None)
    osc = orc.OracleScene(sb)
    mesh_hits = 0
    for tile in [(928, 300, 64, 32), (928, 524, 64, 32)]:
        ref = orc.render(osc, cam, n, tile=tile)
        x0, y0, tw, th = tile
        sl = (slice(y0, y0 + th), slice(x0, x0 + tw))
        assert same_bits(a[sl], ref["accum"][sl]) == 0
        assert (ids[sl] == ref["primary_id"][sl]).all() and (out[sl] == ref["rgba8"][sl]).all()
        mesh_hits += int((ref["primary_id"][sl] >= 14).sum())
    assert mesh_hits > 1000  # the tiles look at the mesh, not at the sky
