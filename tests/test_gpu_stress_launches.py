"""GPU tier: long random sequences of launches of every length -- overlapping flights, a pool set per flight, pools of every size
class, the fold kernels in order behind them, options changed between launches -- end on the frame one launch of whole-pixel jobs
gives, bit for bit.  What a race between flights, a stale pool record or a lost reservation would break."""
import numpy as np
import pytest

from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
from util import fixture, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.reset_options()
    t.close()


@pytest.mark.parametrize("name,w,h,seed", [("cornell_box", 256, 192, 1), ("all", 192, 128, 2), ("teapot", 320, 200, 3), ("unit_logo", 160, 120, 4),
                                           ("rings", 200, 150, 5), ("fractals", 96, 64, 6), ("cornell_box", 1280, 720, 7), ("teapot", 1280, 720, 8)])
def test_random_launch_sequences_end_on_the_same_frame(pt, name, w, h, seed):
    rng = np.random.default_rng(seed)
    sc = host.Scene.from_fixture(fixture(name)).build(decode_textures=False)
    total = 24 if name == "fractals" else (32 if w >= 1024 else 96)  # the large frames: more 16x16 blocks than resident workgroups
    pt.reset_options()
    pt.set_option("split_samples", 0)  # whole-pixel jobs, one launch: the reference frame
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    cam = sc.camera(w, h)
    pt.trace_n(cam, 0, 1, total)
    want = (pt.read_accum().copy(), pt.read_moments().copy(), pt.read_primary_ids().copy(), pt.read_output().copy())
    for rep in range(3):
        pt.reset_options()
        pt.set_option("host_waits_per_frame", 0)
        if rep == 1:
            pt.set_options(flights=int(rng.integers(2, 8)))
        pt.clear()
        done = 0
        while done < total:
            n = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 5, 8, 16]))
            n = min(n, total - done)
            if rep == 2 and rng.random() < 0.15:  # an option changes between launches: the policy table is rebuilt, buffers are kept
                pt.set_option("pool_paths", int(rng.choice([0, 256, 1024, 4096])))
            pt.trace_n(cam, done, done + 1, n)
            if rng.random() < 0.05:
                pt.read_output()  # a synchronising call now and then
            done += n
        got = (pt.read_accum(), pt.read_moments(), pt.read_primary_ids(), pt.read_output())
        assert same_bits(got[0], want[0]) == 0 and same_bits(got[1], want[1]) == 0, (name, rep)
        assert (got[2] == want[2]).all() and (got[3] == want[3]).all(), (name, rep)
    pt.enable_moments(False)
    pt.reset_options()
