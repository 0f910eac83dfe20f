"""How the distance between the library's frame of the picture's scene and the reference's picture (8x8 block means) changes with the
number of samples accumulated from frame index 0: were the picture's noise independent of ours, the distance would fall steadily
towards the picture's own noise; a minimum at one sample count says the picture was accumulated over the same frame indices with the
same per-pixel random sequences.  usage: python tools/reference_image_scan.py [up to] [step] [first frame index] [samples before the first comparison]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pathtracer_amd import host, _lib  # noqa: E402
if os.environ.get("PTX355_LIB"):  # a differently built library (arithmetic experiments)
    _lib.LIB_PATH = os.path.abspath(os.environ["PTX355_LIB"])
from pathtracer_amd.device import PathTracer  # noqa: E402

upto = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
step = int(sys.argv[2]) if len(sys.argv) > 2 else 128
first_frame = int(sys.argv[3]) if len(sys.argv) > 3 else 0
quiet_until = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # accumulate this many samples before the first comparison
ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_blocks.npy")).astype(np.float64)
d = np.load(os.path.join(ROOT, "data", "scenes", "cornell_box.npz"), allow_pickle=False)
keep = [i for i in range(len(d["figs"])) if i != 12]
mats = d["mats"].copy()
mats["emission_strength"][5] = 3.0
sc = host.Scene.empty()
sc.append_shapes(d["figs"][keep], mats[keep])
cam = d["camera"]
sc.set_camera(cam[0:3], cam[3:6], float(cam[6]))
sc.set_render_settings(max_bounces=int(cam[8]), fractal_march_steps=int(cam[15]), tone_mapper=int(cam[14]), firefly_clamp=float(cam[9]), exposure=float(cam[7]))
sc.set_sky(cam[10:13], float(cam[13]))
sc.build()
w, h = 1920, 1080
pt = PathTracer(0)
sc.upload_to(pt)
pt.resize(w, h)
c = sc.camera(w, h)
done = 0
while done < quiet_until:
    m = min(128, quiet_until - done)
    pt.trace_n(c, first_frame + done, done + 1, m)
    done += m
while done < upto:
    pt.trace_n(c, first_frame + done, done + 1, step)
    done += step
    img = pt.read_output().reshape(h, w, 4)[:, :, :3].astype(np.float64)
    ours = img.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))
    diff = ours - ref
    print(f"{done:6d} samples: mean abs {np.abs(diff).mean():.4f} levels, rms {np.sqrt((diff * diff).mean()):.4f}, within 1 level {(np.abs(diff).max(axis=2) <= 1.0).mean():.4f}", flush=True)
pt.close()
