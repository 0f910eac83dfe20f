"""A/B of the lockstep LDS kernel with lane-owned paths and over sorted path pools (GPU box): bit-identity on a small frame,
then time at 1080p.  SORTED_SCENES=cornell_box,rings  SORTED_N=16,128  PROBE_OPTS=pool_paths=2048"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

scenes = os.environ.get("SORTED_SCENES", "cornell_box").split(",")
ns = tuple(int(x) for x in os.environ.get("SORTED_N", "16,128").split(","))
extra = [kv.split("=") for kv in filter(None, os.environ.get("PROBE_OPTS", "").split(","))]
pt = PathTracer(0)
for k, v in extra:
    pt.set_option(k, int(v))
for name in scenes:
    sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{name}.npz")).build(decode_textures=False)
    sc.upload_to(pt)
    res = {}
    for mode in (0, 1):
        pt.set_option("sorted_pool", mode)
        pt.enable_moments(True)
        pt.resize(251, 141)
        cam = sc.camera(251, 141)
        pt.clear()
        pt.trace_n(cam, 0, 1, 8)
        pt.trace_n(cam, 8, 9, 5)
        pt.sync()
        res[mode] = (pt.read_accum().copy(), pt.read_output().copy(), pt.read_primary_ids().copy(), pt.read_moments().copy(), pt.stats())
        pt.enable_moments(False)
    same = all(np.array_equal(res[0][i].view(np.uint8), res[1][i].view(np.uint8)) for i in range(4))
    print(f"{name}: pools == lane-owned: {same}; stats {res[0][4]} / {res[1][4]}", flush=True)
    if not same or res[0][4] != res[1][4]:
        print(f"  MISMATCH: {(res[0][0].view(np.uint32) != res[1][0].view(np.uint32)).sum()} accum words, {(res[0][2] != res[1][2]).sum()} ids differ", flush=True)
        sys.exit(1)
    w, h = 1920, 1080
    cam = sc.camera(w, h)
    for mode in (0, 1, 0, 1):
        pt.set_option("sorted_pool", mode)
        pt.resize(w, h)
        pt.trace_n(cam, 0, 1, 8)
        pt.sync()
        pt.kernel_time()
        for n in ns:
            pt.clear()
            pt.sync()
            pt.kernel_time()
            pt.trace_n(cam, 0, 1, n)
            pt.sync()
            ms, _ = pt.kernel_time()
            st = pt.stats()
            print(f"{name} sorted_pool={mode} n={n}: {ms:.2f} ms  {w * h * n / ms / 1e3:.1f} Mpaths/s  {st['bounces'] / ms / 1e6:.3f} Gpath-bounce/s", flush=True)
