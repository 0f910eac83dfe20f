"""GPU tier: the library's OWN multi-GPU exchange (ptx_comm_init, ptx_merge_sample_split, ptx_assemble_tiles) at world
sizes 2, 3 and 8 on the test box's one GPU.  The ranks are threads of one child process, each with a context of its
own on GPU 0; RCCL is the in-process stand-in under tests/loopback/ (sends and receives matched through a table and
carried out as device copies), which the product adopts because it is already loaded when the product looks for RCCL.
What runs is the product's code: slice plan, grouped sends and receives, rank-ordered Chan merge, merged slices to
every rank, block exchange of a tile partition.  Expected values: the numpy Chan merge of the ranks' partials in rank
order (bitwise), the single-context frame (bitwise)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd.multigpu import chan_merge, pixel_slices
from util import ROOT, same_bits

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(ROOT, "tests", "loopback", "driver.py")


def _run(case, world, tmp_path):
    out = tmp_path / f"{case}_{world}.npz"
    env = dict(os.environ, PTX355_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-u", DRIVER, case, str(world), str(out)], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def _check_split(d, world):
    counts = [int(c) for c in d["counts"]]
    # Chan's update in rank order over the ranks that hold samples, as numpy (the merge kernel's arithmetic)
    n_a, mean, m2 = 0, None, None
    for r in range(world):
        if counts[r] == 0:
            continue
        pm, p2 = d[f"part_mean_{r}"][..., :3].copy(), d[f"part_m2_{r}"].copy()
        if mean is None:
            n_a, mean, m2 = counts[r], pm, p2
        else:
            n_a, mean, m2 = chan_merge(n_a, mean, m2, counts[r], pm, p2)
    assert n_a == sum(counts)
    for r in range(world):
        assert same_bits(d[f"mean_{r}"][..., :3], mean) == 0, f"rank {r}: merged mean"
        assert same_bits(d[f"m2_{r}"], m2) == 0, f"rank {r}: merged M2"
        assert (d[f"rgba8_{r}"] == d["rgba8_0"]).all(), f"rank {r}: image"
    return mean, m2


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sample_split_merges_in_rank_order_on_every_rank(world, tmp_path):
    d = _run("split", world, tmp_path)
    mean, _ = _check_split(d, world)
    # the slices really differ in length at 95 x 53 = 5035 pixels
    assert len({c for _, c in pixel_slices(95 * 53, world)}) == (2 if 5035 % world else 1)
    assert np.isfinite(mean).all() and float(mean.mean()) > 0.01


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_without_samples_contributes_nothing(world, tmp_path):
    d = _run("split_rank0_empty", world, tmp_path)
    _check_split(d, world)  # the expected merge starts from rank 1's partial
    assert float(np.abs(d["part_mean_0"] - 7.0).max()) == 0.0  # while rank 0's buffers held something else


def test_more_ranks_than_pixels(tmp_path):
    _check_split(_run("split_tiny_frame", 8, tmp_path), 8)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_partition_assembles_the_single_context_frame(world, tmp_path):
    d = _run("tiles", world, tmp_path)
    for r in range(world):
        assert same_bits(d[f"tiles_mean_{r}"], d["tiles_want_mean"]) == 0, f"rank {r}: block exchange"
        assert (d[f"tiles_rgba8_{r}"] == d["tiles_want_rgba8"]).all()
        assert same_bits(d[f"bands_mean_{r}"], d["bands_want_mean"]) == 0, f"rank {r}: all-reduce of bands"
        assert (d[f"bands_rgba8_{r}"] == d["bands_want_rgba8"]).all()


def test_tile_partition_inside_a_tile(tmp_path):
    d = _run("tiles_in_a_tile", 3, tmp_path)
    for r in range(3):
        assert same_bits(d[f"tiles_mean_{r}"], d["tiles_want_mean"]) == 0


def test_a_failed_send_closes_the_group(tmp_path):
    d = _run("failing_send", 2, tmp_path)
    msgs = [str(m) for m in d["fail_msgs"]]
    assert "injected failure" in msgs[0] and "slice exchange" in msgs[0]
    assert "rendezvous" in msgs[1]
    assert list(d["fail_depths"]) == [0, 0]      # no group left open on either thread
    _check_split(d, 2)                           # and a new world on the same threads' library state works
