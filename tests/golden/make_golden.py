"""Regenerates tests/golden/<scene>_<w>x<h>x<n>.npz from data/scenes/*.npz with the CPU oracle:
primary-hit ids, per-sample radiance, running mean, rgba8 and work counters, frames 0..n-1.

Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import host_prep as hp, oracle as orc  # noqa: E402
from util import GOLDEN_CASES, fixture  # noqa: E402


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


if __name__ == "__main__":
    main()
