"""all.yaml + the 10 M-triangle height field (the fourth bench workload) with and without all.yaml's ray-marched figures: what share of
the launch is theirs.  usage: python tools/mesh_marcher_share.py [quads] [samples per launch]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

TORUS, MEBIUS = 5, 14
quads = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = np.load(os.path.join(ROOT, "data", "scenes", "all.npz"), allow_pickle=False)
cam = d["camera"]
pt = PathTracer(0)
for label, drop in (("all figures", ()), ("without the Moebius strip", (MEBIUS,)), ("without torus and Moebius strip", (TORUS, MEBIUS))):
    keep = [i for i, g in enumerate(d["figs"]) if int(g["shape_type"]) not in drop]
    sc = host.Scene.empty()
    sc.append_shapes(d["figs"][keep], d["mats"][keep])
    sc.set_camera(cam[0:3], cam[3:6], float(cam[6]))
    sc.set_render_settings(max_bounces=int(cam[8]), fractal_march_steps=int(cam[15]), tone_mapper=int(cam[14]), firefly_clamp=float(cam[9]), exposure=float(cam[7]))
    sc.set_sky(cam[10:13], float(cam[13]))
    sc.append_heightfield(quads)
    sc.build(decode_textures=False, tracer=pt)
    sc.upload_to(pt)
    pt.resize(1920, 1080)
    c = sc.camera(1920, 1080)
    for i in range(2):
        pt.trace_n(c, i * n, 1 + i * n, n)
    pt.sync(); pt.kernel_time()
    for i in range(3):
        pt.trace_n(c, (2 + i) * n, 1 + (2 + i) * n, n)
    pt.sync()
    ms, launches = pt.kernel_time()
    st = pt.stats()
    print(f"{label}: {ms / launches:.2f} ms per {n}-sample launch, {1920 * 1080 * n * launches / ms / 1e3:.1f} Mpaths/s, {pt.kernel_name()}", flush=True)
pt.close()
