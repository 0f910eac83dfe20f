"""Quick throughput probe of the trace kernel on the four configuration scenes (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host, _lib  # noqa: E402
if os.environ.get("PTX355_LIB"):  # A/B of differently built libraries
    _lib.LIB_PATH = os.path.abspath(os.environ["PTX355_LIB"])
from pathtracer_amd.device import PathTracer  # noqa: E402

CASES = [("cornell_box", 1920, 1080, (32, 128)), ("teapot", 3840, 2160, (8,)), ("fractals", 1920, 1080, (2,)), ("all", 1920, 1080, (16,)),
         ("all_mesh", 1920, 1080, (8,))]  # all_mesh: all.yaml + the 10 M-triangle height field of bench.py --workload all_mesh
print("library:", _lib.LIB_PATH)
if len(sys.argv) > 1:
    known = {c[0] for c in CASES}
    CASES = [c for c in CASES if c[0] in sys.argv[1:]] + [(a, 1920, 1080, (16,)) for a in sys.argv[1:] if a not in known]  # any other fixture
if os.environ.get("PROBE_N"):  # samples per launch, e.g. PROBE_N=2,8
    CASES = [(c[0], c[1], c[2], tuple(int(x) for x in os.environ["PROBE_N"].split(","))) for c in CASES]
pt = PathTracer(0)
# PROBE_OPTS="key=value,key=value": ptx_set_option settings for A/B runs (the library itself reads no environment)
for kv in filter(None, os.environ.get("PROBE_OPTS", "").split(",")):
    k, v = kv.split("=")
    pt.set_option(k, int(v))
for name, w, h, ns in CASES:
    if name == "all_mesh":
        sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "all.npz"))
        sc.append_heightfield(2236)
        sc.build(decode_textures=False, tracer=pt)
    else:
        sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{name}.npz")).build(decode_textures=False)
    if os.environ.get("PROBE_SIZE"):  # WxH for every case
        w, h = (int(x) for x in os.environ["PROBE_SIZE"].split("x"))
    cam = sc.camera(w, h)
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, 1)
    pt.sync()
    pt.kernel_time()
    for n in ns:
        ms = float("inf")
        for rep in range(4):  # the first launch of a length may grow buffers: one warm-up, then the best of three
            pt.clear()
            pt.sync()
            pt.trace_n(cam, 0, 1, n)
            pt.sync()
            t, _ = pt.kernel_time()
            ms = min(ms, t) if rep else ms
        if os.environ.get("PROBE_REPEAT"):  # K launches back to back (short ones overlap), wall time per launch
            k = int(os.environ["PROBE_REPEAT"])
            pt.clear()
            pt.sync()
            t0 = time.perf_counter()
            for i in range(k):
                pt.trace_n(cam, i * n, i * n + 1, n)
            pt.sync()
            ms = (time.perf_counter() - t0) * 1e3 / k
            pt.kernel_time()
            pt.clear()
            pt.trace_n(cam, 0, 1, n)
            pt.sync()
        st = pt.stats()
        print(f"{name} {w}x{h} n={n}: kernel {ms:.3f} ms  {w * h * n / ms / 1e3:.1f} Mpaths/s  {st['bounces'] / ms / 1e6:.3f} Gpath-bounce/s  "
              f"bounces/path {st['bounces'] / st['paths']:.2f} shadow/path {st['shadow_rays'] / st['paths']:.2f}  {pt.kernel_name()}", flush=True)
