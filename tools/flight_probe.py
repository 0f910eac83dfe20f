"""Frames per second of back-to-back ptx_trace calls (one sample per dispatch, no host sync in between) against the
number of launches allowed in flight (option `flights`); GPU box.  usage: tools/flight_probe.py [scene] [w] [h] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fractals"
w, h, frames = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 1920), (3, 1080), (4, 64)))
pt = PathTracer(0)
sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{name}.npz")).build(decode_textures=False)
cam = sc.camera(w, h)
sc.upload_to(pt)
pt.resize(w, h)
for flights in (1, 2, 3, 4, 6, 8):
    pt.set_option("flights", flights)
    pt.clear()
    for k in range(8):
        pt.trace_n(cam, k, k + 1, 1)
    pt.sync()
    t0 = time.perf_counter()
    for k in range(frames):
        pt.trace_n(cam, 8 + k, 9 + k, 1)
    pt.sync()
    dt = time.perf_counter() - t0
    print(f"{name} {w}x{h} flights={flights}: {dt / frames * 1e3:.2f} ms per frame, {w * h * frames / dt / 1e6:.1f} Mpaths/s", flush=True)
