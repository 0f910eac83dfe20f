"""Diagnostic: Cornell throughput with individual figures removed (cost attribution by ablation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pathtracer_amd import host, _lib
if os.environ.get("PTX355_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PTX355_LIB"])
from pathtracer_amd.device import PathTracer
pt = PathTracer(0)
d = np.load(os.path.join(ROOT, "data", "scenes", "cornell_box.npz"))
names = {5: "disc light", 6: "glass sphere", 7: "gold sphere", 8: "cube", 9: "torus", 10: "pyramid", 11: "ellipsoid", 12: "emissive sphere"}
for drop in [(), (9,), (10,), (9, 10), (9, 10, 11), (6, 7, 8, 9, 10, 11)]:
    keep = [i for i in range(len(d["figs"])) if i not in drop]
    sc = host.Scene.empty()
    sc.append_shapes(d["figs"][keep], d["mats"][keep])
    cam = d["camera"]
    sc.set_camera(cam[0:3], cam[3:6], float(cam[6]))
    sc.build()
    w, h, n = 1920, 1080, 32
    sc.upload_to(pt); pt.resize(w, h)
    c = sc.camera(w, h)
    pt.trace_n(c, 0, 1, 1); pt.sync(); pt.kernel_time(); pt.clear()
    pt.trace_n(c, 0, 1, n); pt.sync()
    ms, _ = pt.kernel_time(); st = pt.stats()
    print(f"drop {[names[i] for i in drop]}: {ms:.1f} ms {w*h*n/ms/1e3:.0f} Mpaths/s {st['bounces']/ms/1e6:.2f} Gpb/s bounces/path {st['bounces']/st['paths']:.2f}", flush=True)
