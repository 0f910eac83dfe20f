"""Sweeps the scheduled kernels' shares (walk,post,march,leaf) on the height-field mesh scene (GPU box) and checks
that every setting gives the same frame.  usage: python tools/sweep_leaf.py [quads] [spp] [shares ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

quads = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
settings = sys.argv[3:] or ["8,2,4,64", "8,2,4,8", "8,2,4,4", "8,2,4,3", "8,2,4,2", "8,2,4,1", "64,2,4,4", "64,2,4,2", "16,2,4,3", "8,4,4,3"]
pt = PathTracer(0)
for kv in filter(None, os.environ.get("SWEEP_OPTS", "").split(",")):  # further ptx_set_option settings, e.g. SWEEP_OPTS=claim_tile=128
    k, v = kv.split("=")
    pt.set_option(k, int(v))
sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", os.environ.get("SWEEP_SCENE", "all") + ".npz"))
if quads:
    sc.append_heightfield(quads)
sc.build(decode_textures=False, tracer=pt)
w, h = 1920, 1080
cam = sc.camera(w, h)
first = None
for s in settings:
    for key, v in zip(("share_walk", "share_post", "share_march", "share_leaf"), s.split(",")):
        pt.set_option(key, int(v))
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, 1)
    pt.sync()
    pt.kernel_time()
    pt.clear()
    pt.trace_n(cam, 0, 1, n)
    pt.sync()
    ms, _ = pt.kernel_time()
    acc = pt.read_accum().tobytes()
    first = first or acc
    print(f"shares {s}: kernel {ms:.1f} ms  {w * h * n / ms / 1e3:.1f} Mpaths/s  same frame: {acc == first}", flush=True)
