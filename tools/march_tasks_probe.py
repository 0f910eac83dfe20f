"""Torus / Moebius marches as tasks of the workgroup (option march_tasks) against lane-owned marches: same frames, and the time of
all.yaml + the height-field mesh.  usage: python tools/march_tasks_probe.py [quads] [samples per launch]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

quads = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pt = PathTracer(0)
sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "all.npz"))
sc.append_heightfield(quads)
sc.build(decode_textures=False, tracer=pt)
frames = {}
for mode in (0, 1):
    pt.reset_options()
    pt.set_option("march_tasks", mode)
    pt.set_option("verbose", 1)
    sc.upload_to(pt)
    # a small frame first: the same bits?
    pt.resize(320, 200)
    c = sc.camera(320, 200)
    pt.trace_n(c, 0, 1, 5)
    pt.trace_n(c, 5, 6, 1)
    frames[mode] = (pt.read_accum().copy(), pt.read_primary_ids().copy(), pt.stats())
    print(f"march_tasks={mode}: small frame done, {pt.kernel_name()}", flush=True)
    pt.resize(1920, 1080)
    c = sc.camera(1920, 1080)
    for i in range(2):
        pt.trace_n(c, i * n, 1 + i * n, n)
    pt.sync(); pt.kernel_time()
    for i in range(3):
        pt.trace_n(c, (2 + i) * n, 1 + (2 + i) * n, n)
    pt.sync()
    ms, launches = pt.kernel_time()
    print(f"march_tasks={mode}: {ms / launches:.2f} ms per {n}-sample launch, {1920 * 1080 * n * launches / ms / 1e3:.1f} Mpaths/s, {pt.launch_info()}", flush=True)
a, b = frames[0], frames[1]
print("words differing:", int((a[0].view(np.uint32) != b[0].view(np.uint32)).sum()), "ids differing:", int((a[1] != b[1]).sum()), "stats", a[2], b[2])
pt.close()
