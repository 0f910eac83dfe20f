"""Throughput of the synthetic height-field scene (all.yaml + 2 * quads^2 triangles) at 1080p (GPU box).
usage: [PTX355_LIB=...] python tools/probe_mesh.py [quads] [spp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host, _lib  # noqa: E402
if os.environ.get("PTX355_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PTX355_LIB"])
from pathtracer_amd.device import PathTracer  # noqa: E402

quads = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "all.npz"))
sc.append_heightfield(quads)
sc.build(decode_textures=False)
print(f"library {_lib.LIB_PATH}; host build {time.time() - t0:.1f} s", flush=True)
for lib in [_lib.LIB_PATH] + [os.path.abspath(p) for p in sys.argv[3:]]:
    _lib.LIB_PATH = lib
    _lib._lib = None  # reload: A/B of differently built libraries in one run
    pt = PathTracer(0)
    sc.upload_to(pt)
    w, h = 1920, 1080
    pt.resize(w, h)
    cam = sc.camera(w, h)
    pt.trace_n(cam, 0, 1, 1)
    pt.sync()
    pt.kernel_time()
    pt.clear()
    pt.trace_n(cam, 0, 1, n)
    pt.sync()
    ms, _ = pt.kernel_time()
    st = pt.stats()
    print(f"{os.path.basename(lib)} quads={quads} n={n}: kernel {ms:.1f} ms  {w * h * n / ms / 1e3:.1f} Mpaths/s  {st['bounces'] / ms / 1e6:.3f} Gpath-bounce/s "
          f"bounces/path {st['bounces'] / st['paths']:.2f}", flush=True)
    pt.close()
