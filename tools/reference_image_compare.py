"""Compares a converged GPU render of the Cornell box with the block means of the picture the reference repository holds
(tests/golden/reference_cornell_box_blocks.npy).  usage: python tools/reference_image_compare.py [samples] [drop=a,b,...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host, _lib  # noqa: E402
if os.environ.get("PTX355_LIB"):  # a differently built library (arithmetic experiments)
    _lib.LIB_PATH = os.path.abspath(os.environ["PTX355_LIB"])
from pathtracer_amd.device import PathTracer  # noqa: E402

REGIONS = {  # in 8x8 blocks: rows, columns
    "red wall": (slice(40, 95), slice(5, 55)), "green wall": (slice(40, 95), slice(185, 235)), "back wall": (slice(40, 60), slice(90, 150)),
    "floor": (slice(120, 133), slice(70, 170)), "ceiling": (slice(2, 15), slice(70, 170)), "light": (slice(20, 27), slice(108, 132)),
    "glass sphere": (slice(85, 105), slice(85, 108)), "gold sphere": (slice(95, 110), slice(138, 155)), "cube": (slice(94, 104), slice(115, 125)),
    "ellipsoid": (slice(64, 78), slice(116, 123))}


def render(drop, n, w=1920, h=1080, disc_strength=None):
    d = np.load(os.path.join(ROOT, "data", "scenes", "cornell_box.npz"), allow_pickle=False)
    keep = [i for i in range(len(d["figs"])) if i not in drop]
    mats = d["mats"].copy()
    if disc_strength is not None:
        mats["emission_strength"][5] = disc_strength  # figure 5: the ceiling light
    sc = host.Scene.empty()
    sc.append_shapes(d["figs"][keep], mats[keep])
    cam = d["camera"]
    sc.set_camera(cam[0:3], cam[3:6], float(cam[6]))
    sc.set_render_settings(max_bounces=int(cam[8]), fractal_march_steps=int(cam[15]), tone_mapper=int(cam[14]), firefly_clamp=float(cam[9]), exposure=float(cam[7]))
    sc.set_sky(cam[10:13], float(cam[13]))
    sc.build()
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.resize(w, h)
    c = sc.camera(w, h)
    done = 0
    while done < n:
        m = min(128, n - done)
        pt.trace_n(c, done, done + 1, m)
        done += m
    img = pt.read_output().reshape(h, w, 4)
    pt.close()
    return img


def blocks(img, b=8):
    h, w, _ = img.shape
    return img[:, :, :3].astype(np.float64).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


def compare(ours, ref):
    d = ours - ref
    out = {"mean_abs_levels": float(np.abs(d).mean()), "rel_rmse": float(np.sqrt((d * d).mean()) / ref.mean()), "mean_ours": ours.mean(axis=(0, 1)).round(2).tolist(),
           "mean_ref": ref.mean(axis=(0, 1)).round(2).tolist(), "blocks_within_8_levels": float((np.abs(d).max(axis=2) <= 8.0).mean())}
    for k, (r, c) in REGIONS.items():
        out[k] = (ours[r, c].mean(axis=(0, 1)).round(1).tolist(), ref[r, c].mean(axis=(0, 1)).round(1).tolist())
    e = np.abs(d).max(axis=2)
    worst = np.argsort(e.ravel())[::-1][:12]
    out["worst blocks (row, column, levels)"] = [(int(i // e.shape[1]), int(i % e.shape[1]), round(float(e.ravel()[i]), 1)) for i in worst]
    out["blocks within 1 / 2 / 4 levels"] = [round(float((e <= t).mean()), 4) for t in (1.0, 2.0, 4.0)]
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    drops = [tuple(int(v) for v in a[5:].split(",") if v) for a in sys.argv[2:] if a.startswith("drop=")] or [()]
    ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_blocks.npy")).astype(np.float64)
    strength = [float(a[6:]) for a in sys.argv[2:] if a.startswith("light=")]
    for drop in drops:
        ours = blocks(render(set(drop), n, disc_strength=strength[0] if strength else None))
        print("dropped figures", drop, "samples", n, "ceiling light", strength)
        for k, v in compare(ours, ref).items():
            print("  ", k, v)
