"""Regenerates tests/golden/<scene>_<w>x<h>x<n>.npz from data/scenes/*.npz with the CPU oracle:
primary-hit ids, per-sample radiance, running mean, rgba8 and work counters, frames 0..n-1.

Also writes tests/golden/scene_digests.json: for every fixture (all 30 scene files of the reference) the work counters and
SHA-256 digests of the oracle's ids / mean / M2 / rgba8 at 48x27, frames 0..1 — small enough to commit for every scene, enough
to notice any drift of the oracle on primitive types the five full golden renders do not show.

Usage: python tests/golden/make_golden.py
"""
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import host_prep as hp, oracle as orc  # noqa: E402
from util import GOLDEN_CASES, SCENE_DIR, fixture  # noqa: E402

DIGEST_CASE = (48, 27, 2)


def scene_digest(path):
    w, h, n = DIGEST_CASE
    sb = hp.prepare_fixture(path)
    cam = hp.camera_to_gpu(sb.camera, w, h)
    r = orc.render(orc.OracleScene(sb), cam, n, want_m2=True, counters=True, threads=1)
    c = r["counters"]
    d = {k: hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).hexdigest() for k in ("primary_id", "accum", "m2", "rgba8")}
    d.update(paths=int(c["paths"]), bounces=int(c["bounces"]), shadow_rays=int(c["shadow_rays"]))
    return d


def main():
    for name, (w, h, n) in GOLDEN_CASES.items():
        sb = hp.prepare_fixture(fixture(name))
        cam = hp.camera_to_gpu(sb.camera, w, h)
        r = orc.render(orc.OracleScene(sb), cam, n, want_radiance=True, want_m2=True, counters=True, threads=1)
        c = r["counters"]
        np.savez_compressed(os.path.join(HERE, f"{name}_{w}x{h}x{n}.npz"), primary_id=r["primary_id"], radiance=r["radiance"],
                            accum=r["accum"], rgba8=r["rgba8"], m2=r["m2"],
                            counters=np.array([c["paths"], c["bounces"], c["shadow_rays"]], dtype=np.uint64))
        print(name, w, h, n, "bounces/path %.2f" % (c["bounces"] / c["paths"]))
    digests = {os.path.basename(p)[:-4]: scene_digest(p) for p in sorted(glob.glob(os.path.join(SCENE_DIR, "*.npz")))}
    with open(os.path.join(HERE, "scene_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    print(len(digests), "scene digests")


if __name__ == "__main__":
    main()
