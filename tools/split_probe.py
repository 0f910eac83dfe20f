"""ms per sample of back-to-back launches of n samples (no host sync in between), whole-pixel jobs against sample jobs + fold
(option split_small); GPU box.  usage: tools/split_probe.py [scene] [w] [h]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
w, h = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 1920), (3, 1080)))
pt = PathTracer(0)
if name == "all_mesh":
    sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "all.npz"))
    sc.append_heightfield(2236)
    sc.build(decode_textures=False, tracer=pt)
else:
    sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{name}.npz")).build(decode_textures=False)
cam = sc.camera(w, h)
sc.upload_to(pt)
pt.resize(w, h)
for n in [int(x) for x in os.environ.get('PROBE_N', '1,2,4,8,16,32,64').split(',')]:
    row = []
    for small in (0, 1 << 20):
        pt.set_option("split_small", small)
        for kv in filter(None, os.environ.get('PROBE_OPTS', '').split(',')):
            pt.set_option(kv.split('=')[0], int(kv.split('=')[1]))
        pt.clear()
        reps = max(4, 256 // n)
        for k in range(2):
            pt.trace_n(cam, k * n, k * n + 1, n)
        pt.sync()
        t0 = time.perf_counter()
        for k in range(reps):
            pt.trace_n(cam, (2 + k) * n, (2 + k) * n + 1, n)
        pt.sync()
        row.append((time.perf_counter() - t0) / (reps * n) * 1e3)
    print(f"{name} {w}x{h} n={n}: whole-pixel jobs {row[0]:.3f} ms per sample, sample jobs {row[1]:.3f}", flush=True)
