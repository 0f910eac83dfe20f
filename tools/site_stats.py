"""Lane statistics per code site of the trace kernel (diagnostic build with -DPTX_SITE_STATS; GPU box).

usage: PTX355_LIB=path/to/diagnostic/libptx355.so python tools/site_stats.py [scene] [w] [h] [n]
Prints, per site, wave-level executions per wave-bounce and the mean number of active lanes."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host, _lib  # noqa: E402
if os.environ.get("PTX355_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PTX355_LIB"])
from pathtracer_amd.device import PathTracer  # noqa: E402

NAMES = {0: "bounce_step", 1: "walk step", 2: "pair test", 3: "serve round", 30: "plane test", 32: "resolve+shadeA", 33: "NEE setup",
         34: "NEE eval+shadow", 35: "sample_brdf", 36: "finalize sample", 37: "glass", 43: "POST step", 44: "mandelbulb iteration", 45: "julia iteration", 48: "evaluation claimed", 46: "stack push beyond LDS (8 B)", 47: "stack pop beyond LDS (8 B)", 40: "torus step", 41: "mebius step", 42: "fractal step"}
TYPES = ["sphere", "plane", "cube", "cylinder", "cone", "torus", "disc", "triangle", "skybox", "mandelbulb", "julia", "ellipsoid",
         "paraboloid", "hyperboloid", "mebius", "pyramid", "tetrahedron", "nop"]
for b in range(8):
    NAMES[50 + b] = f"march round, {8 * b + 1}-{8 * b + 8} lanes"
    NAMES[60 + b] = f"  ... of {8 * b + 1}-{8 * b + 8} live"
for i, t in enumerate(("WALK", "POST", "MANDEL", "JULIA", "IDLE")):
    NAMES[70 + i] = f"full wave marching: in {t}"
for i, t in enumerate(TYPES):
    NAMES[10 + i] = "test " + t

name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 640
h = int(sys.argv[3]) if len(sys.argv) > 3 else 360
n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
quads = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # > 0: add the synthetic height-field mesh (2 * quads^2 triangles)
pt = PathTracer(0)
lib = pt.lib
sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{name}.npz"))
if quads:
    sc.append_heightfield(quads)
sc.build(decode_textures=False)
cam = sc.camera(w, h)
sc.upload_to(pt)
pt.resize(w, h)
buf = np.zeros(96 * 2, dtype=np.uint64)
fn = lib.ptx_debug_sites
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int]
fn(buf.ctypes.data, 96)
pt.trace_n(cam, 0, 1, n)
pt.sync()
fn(buf.ctypes.data, 96)
ms, _ = pt.kernel_time()
st = pt.stats()
wb = float(buf[0]) or st["bounces"] / 64.0  # the scheduled kernels have no common bounce step: normalise per 64 lane-bounces
print(f"{name} {w}x{h} n={n} quads={quads} kernel={pt.kernel_name()} {ms:.1f} ms lane-bounces/wave-bounce={st['bounces'] / wb:.1f} "
      f"paths={st['paths']} bounces={st['bounces']} shadow_rays={st['shadow_rays']}")
pairs, tris = float(buf[2 * 2 + 1]), float(buf[2 * 17 + 1])
print(f"BVH records fetched: {pairs:.0f} node pairs + {tris:.0f} triangles = {(pairs + tris) * 64 / 1e9:.2f} GB (64 B each)")
print(f"{'site':22s} {'execs/wave-bounce':>18s} {'mean lanes':>11s} {'lane-execs/lane-bounce':>23s}")
for i in range(96):
    c, l = float(buf[2 * i]), float(buf[2 * i + 1])
    if c:
        print(f"{NAMES.get(i, str(i)):22s} {c / wb:18.3f} {l / c:11.1f} {l / st['bounces']:23.3f}")
