"""Throughput of the post-process pass (GPU box): per effect chain, time per launch from a timed loop of launches
on the context's stream and the algorithmic HBM rate (16 B read + 4 B written per pixel) against the 8 TB/s peak.
usage: python tools/bench_post.py [width height] [launches]   -- prints one JSON line per chain."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 2 else 3840
h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
pt = PathTracer(0)
pt.resize(w, h)
rng = np.random.default_rng(1)
a = rng.uniform(0, 1.2, (h, w, 4)).astype(np.float32)
pt.write_accum(a)
chains = [[], ["negative"], ["sepia"], ["fxaa"], ["casting"], ["comic"], ["oil"], ["fxaa", "sepia", "comic", "negative"],
          ["negative", "sepia", "grayscale", "fxaa", "oil", "bw", "comic", "casting"]]
for chain in chains:
    p = host.build_post_params(w, h, chain)
    for _ in range(5):
        pt.post_process(p)
    pt.sync()
    us = 1e30
    for _ in range(3):  # best of three batches: host-side timing picks up the runtime's occasional stalls
        t0 = time.perf_counter()
        for _ in range(reps):
            pt.post_process(p)
        pt.sync()
        us = min(us, (time.perf_counter() - t0) / reps * 1e6)
    gbs = w * h * 20 / (us * 1e-6) / 1e9
    print(json.dumps({"pass": "post_process", "effects": chain, "width": w, "height": h, "us_per_launch": round(us, 2),
                      "algorithmic_gb_per_s": round(gbs, 1), "hbm_peak_gb_per_s": 8000.0, "frac": round(gbs / 8000.0, 4),
                      "bytes_per_pixel": 20}), flush=True)
pt.close()
