"""GPU tier: the launch policy is data.  For every scene fixture and launch length, what the policy's table says a launch
would run (ptx_launch_plan) equals what tests/golden/launch_policy.json holds -- recorded on an MI355X from the launches
themselves, by tools/record_launch_policy.py, and re-recorded only when a choice is meant to change -- and equals what the
launch then reports having run (ptx_last_launch_info).  The choice for a host that waits for every frame follows from the call sequence
(read-backs between one-sample launches), not from a clock."""
import json
import os

import pytest

from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
from util import GOLDEN_DIR, fixture

pytestmark = pytest.mark.gpu
GOLDEN = json.load(open(os.path.join(GOLDEN_DIR, "launch_policy.json")))


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.reset_options()
    t.close()


def test_the_golden_covers_every_fixture():
    scenes = sorted(f[:-4] for f in os.listdir(os.path.dirname(fixture("x"))) if f.endswith(".npz"))
    assert sorted(GOLDEN) == scenes and len(scenes) == 30


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_policy_rows_equal_the_recorded_choices(pt, name):
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    pt.reset_options()
    pt.set_option("host_waits_per_frame", 0)
    sc.upload_to(pt)
    want = GOLDEN[name]
    sizes = sorted({k.split(":")[0] for k in want})
    for size in sizes:
        w, h = (int(v) for v in size.split("x"))
        pt.resize(w, h)
        cam = sc.camera(w, h)
        cam[0]["max_bounces"] = 2
        for key, choice in want.items():
            parts = key.split(":")
            if parts[0] != size or len(parts) != 2:
                continue
            n = int(parts[1])
            assert pt.launch_plan(n) == choice, (name, key)
            if w <= 64 and n <= 16:  # ... and the launch runs what the table says
                pt.trace_n(cam, 0, 1, n)
                assert pt.launch_info().split(";")[0] == choice, (name, key)
    # a host that waits for every frame: the third one-sample launch after read-backs runs alone
    pt.set_option("host_waits_per_frame", -1)
    pt.resize(64, 48)
    cam = sc.camera(64, 48)
    cam[0]["max_bounces"] = 2
    plans = []
    for k in range(4):
        plans.append(pt.launch_plan(1))
        pt.trace_n(cam, k, k + 1, 1)
        assert pt.launch_info().split(";")[0] == plans[-1]
        pt.read_output()
    assert plans[0] == plans[1] == want["64x48:1"]
    assert plans[2] == plans[3] == want["64x48:1:host_waits"], name
    # back to back again: the launch after this one follows no synchronising call, and overlaps
    pt.trace_n(cam, 4, 5, 1)
    assert pt.launch_plan(1) == want["64x48:1"]
    # stated outright
    pt.set_option("host_waits_per_frame", 1)
    assert pt.launch_plan(1) == want["64x48:1:host_waits"]
    pt.set_option("host_waits_per_frame", 0)
    pt.read_output(); pt.read_output()
    assert pt.launch_plan(1) == want["64x48:1"]
    pt.reset_options()


def test_forced_pools_on_a_mesh_never_leave_blocks_untraced(pt):
    """kernel_family 0 + sorted_pool 1 on a scene in global memory: launches that cannot trace from pools (whole-pixel jobs:
    split_samples 0, a launch beyond split_small) run the non-claiming lockstep kernel, which must then get one workgroup
    per 16x16 block -- 3072 of them here -- not the resident-workgroup grid of the claiming kernels (1024)."""
    from pathtracer_amd.device import PtxError
    from util import same_bits
    sc = host.Scene.from_fixture(fixture("teapot")).build(decode_textures=False)
    w, h, n = 1024, 768, 1
    blocks = (w // 16) * (h // 16)
    cam = sc.camera(w, h)
    pt.reset_options()
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    want, want_ids = pt.read_accum(), pt.read_primary_ids()
    assert (want_ids != 0xFFFFFFFF).mean() > 0.2
    for opts in (dict(split_samples=0), dict(split_small=0), dict()):
        pt.reset_options()
        pt.set_options(kernel_family=0, sorted_pool=1, pool_over=0, host_waits_per_frame=0, **opts)
        pt.clear()
        pt.trace_n(cam, 0, 1, n)
        info = pt.launch_info()
        assert same_bits(pt.read_accum(), want) == 0 and (pt.read_primary_ids() == want_ids).all(), (opts, info)
        if opts:
            assert info.startswith("ptx_trace_kernel jobs=pixels") and info.endswith(f"grid={blocks}"), info
        else:
            assert info.startswith("ptx_trace_kernel_sorted jobs=pools"), info
    # the static entry point runs the same non-claiming kernel; this frame's 12 288 waves are beyond its scratch array
    with pytest.raises(PtxError, match="waves"):
        pt.trace_n_static(cam, 0, 1, n)
    pt.resize(256, 128)
    cam = sc.camera(256, 128)
    pt.trace_n(cam, 0, 1, 2)
    small = pt.read_accum()
    pt.clear()
    pt.trace_n_static(cam, 0, 1, 2)
    assert same_bits(pt.read_accum(), small) == 0 and pt.launch_info().startswith("ptx_trace_kernel jobs=pixels")
    pt.reset_options()
