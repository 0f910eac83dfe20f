"""Writes tests/golden/post_process_53x31.npz: an accumulation image (a rendered Cornell frame scaled so that it
has both dark and over-bright texels), a list of post-process parameter blocks and the oracle's rgba8 outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import host_prep as hp, oracle as orc  # noqa: E402

w, h = 53, 31
sb = hp.prepare_fixture(os.path.join(ROOT, "data", "scenes", "cornell_box.npz"))
accum = orc.render(orc.OracleScene(sb), hp.camera_to_gpu(sb.camera, w, h), 6)["accum"]
chains = [([], 3, 4), (["negative"], 3, 4), (["sepia"], 3, 4), (["grayscale"], 3, 4), (["fxaa"], 3, 4), (["oil"], 3, 4), (["oil"], 0, 4),
          (["oil"], 9, 4), (["bw"], 3, 4), (["comic"], 3, 4), (["comic"], 3, 0), (["casting"], 3, 4),
          (["fxaa", "sepia", "comic", "negative"], 2, 6), (["oil", "casting", "grayscale", "bw", "negative", "sepia", "fxaa", "comic"], 1, 3),
          ([0, 9, 1], 3, 4)]
params = np.stack([orc.post_params(w, h, e, r, l) for e, r, l in chains])
out = np.stack([orc.post_process(accum, p) for p in params])
np.savez_compressed(os.path.join(HERE, "post_process_53x31.npz"), accum=accum, params=params, out=out)
print("wrote", len(chains), "chains", out.shape)
