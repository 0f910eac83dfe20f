"""CPU tier: the N>1 host logic over torch.distributed/gloo with world sizes 2 and 3 — work split, uneven slice
exchange, Chan merge, all-gather of means and second moments, tile assembly — by the functions of
pathtracer_amd.multigpu themselves on oracle-rendered partials (no GPU involved)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd.multigpu import chan_merge, owned_blocks, pixel_slices, sample_ranges
from util import ROOT, fixture


def test_partitions_cover_exactly_once():
    assert sample_ranges(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sum(c for _, c in sample_ranges(1024, 8)) == 1024
    for (w, h, world) in [(100, 70, 3), (1920, 1080, 8), (15, 15, 2)]:
        cover = sum(owned_blocks(w, h, r, world).astype(int) for r in range(world))
        assert (cover == 1).all()
    assert pixel_slices(7, 2) == [(0, 4), (4, 3)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, w, h, total):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import host_prep as hp, oracle as orc
    from pathtracer_amd import multigpu
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = hp.prepare_fixture(fixture("cornell_box"))
    cam = hp.camera_to_gpu(sb.camera, w, h)
    osc = orc.OracleScene(sb)
    # --- sample split: this rank renders its frame range; the exchange is multigpu.merge_sample_split itself
    first, cnt = sample_ranges(total, world)[rank]
    r = orc.render(osc, cam, cnt, first_frame=first, first_sample=1, want_m2=True, threads=1)
    part = multigpu.HostPartial(r["accum"], r["m2"])
    counts = [c for _, c in sample_ranges(total, world)]
    n_merged = multigpu.merge_sample_split(part, dist, torch.device("cpu"), counts, cam)
    full_mean, full_m2 = part.acc.numpy().reshape(h, w, 4).copy(), part.m2.numpy().reshape(h, w, 3).copy()
    # --- tile partition: disjoint masks, multigpu.assemble_tiles sums them
    mask = owned_blocks(w, h, rank, world)
    one = orc.render(osc, cam, 2, threads=1)["accum"]
    tiles = multigpu.HostPartial(np.where(mask[..., None], one, 0).astype(np.float32))
    multigpu.assemble_tiles(tiles, dist, torch.device("cpu"), cam)
    # every rank must end with the same frame
    mine = torch.from_numpy(full_mean.copy())
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    same_everywhere = torch.tensor([1 if torch.equal(mine, ref) else 0])
    dist.all_reduce(same_everywhere, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put((n_merged, full_mean, full_m2, tiles.acc.numpy().reshape(h, w, 4).copy(), one, int(same_everywhere.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 31, 21), (3, 32, 20)])  # 651 = 2 * 325 + 1 and 640 = 3 * 213 + 1 pixels: uneven slices
def test_sample_split_and_tile_partition_over_gloo(world, w, h):
    sys.path.insert(0, ROOT)
    from oracle import host_prep as hp, oracle as orc
    total = 10
    assert (w * h) % world != 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, w, h, total)) for r in range(world)]
    for p in procs:
        p.start()
    n_merged, full_mean, full_m2, summed, one, same_everywhere = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n_merged == total and same_everywhere == 1
    sb = hp.prepare_fixture(fixture("cornell_box"))
    cam = hp.camera_to_gpu(sb.camera, w, h)
    osc = orc.OracleScene(sb)
    seq = orc.render(osc, cam, total, threads=1, want_m2=True)
    scale = np.maximum(np.abs(seq["accum"][..., :3]), 1e-3)
    assert np.max(np.abs(full_mean[..., :3] - seq["accum"][..., :3]) / scale) < 1e-5  # Chan merge vs sequential running mean: rounding only
    assert (full_mean[..., 3] == 1).all()
    # the distributed merge equals the same merge done in one process, bit for bit: means and second moments of the whole frame
    n_a = mean = m2 = None
    for first, cnt in sample_ranges(total, world):
        r = orc.render(osc, cam, cnt, first_frame=first, first_sample=1, want_m2=True, threads=1)
        if n_a is None:
            n_a, mean, m2 = cnt, r["accum"][..., :3].copy(), r["m2"].copy()
        else:
            n_a, mean, m2 = chan_merge(n_a, mean, m2, cnt, r["accum"][..., :3], r["m2"])
    assert full_mean[..., :3].tobytes() == mean.tobytes() and full_m2.tobytes() == m2.tobytes()
    assert np.max(np.abs(full_m2 - seq["m2"]) / np.maximum(np.abs(seq["m2"]), 1e-2)) < 1e-3
    assert summed.tobytes() == one.tobytes()                # disjoint tiles sum to the frame exactly


def test_chan_merge_is_exact_on_moments_algebra():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 4, (50, 6, 3)).astype(np.float64)
    def mom(v):
        return len(v), v.mean(0), ((v - v.mean(0)) ** 2).sum(0)
    na, ma, sa = mom(x[:20])
    nb, mb, sb_ = mom(x[20:])
    n, m, s = chan_merge(na, ma.astype(np.float32), sa.astype(np.float32), nb, mb.astype(np.float32), sb_.astype(np.float32))
    nt, mt, st = mom(x)
    assert n == nt and np.allclose(m, mt, rtol=1e-5) and np.allclose(s, st, rtol=1e-4)
