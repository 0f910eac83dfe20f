"""CPU tier: the N>1 host logic over torch.distributed/gloo with world_size 2 — work split, slice
exchange and Chan merge — on oracle-rendered partials (no GPU involved)."""
import os
import socket
import sys

import numpy as np
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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import host_prep as hp, oracle as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = hp.prepare_fixture(fixture("cornell_box"))
    w, h, total = 32, 20, 10
    cam = hp.camera_to_gpu(sb.camera, w, h)
    osc = orc.OracleScene(sb)
    # --- sample split: this rank renders its frame range, slices are exchanged and merged in rank order
    first, cnt = sample_ranges(total, world)[rank]
    r = orc.render(osc, cam, cnt, first_frame=first, first_sample=1, want_m2=True, threads=1)
    mean = torch.from_numpy(r["accum"][..., :3].reshape(-1, 3).copy())
    m2 = torch.from_numpy(r["m2"].reshape(-1, 3).copy())
    slices = pixel_slices(w * h, world)
    mine = slices[rank]
    recv_mean = [torch.empty((mine[1], 3)) for _ in range(world)]
    recv_m2 = [torch.empty((mine[1], 3)) for _ in range(world)]
    # gloo has no all_to_all on CPU tensors in every build: use one gather per destination
    for dst in range(world):
        f, c = slices[dst]
        dist.gather(mean[f:f + c].contiguous(), recv_mean if rank == dst else None, dst=dst)
        dist.gather(m2[f:f + c].contiguous(), recv_m2 if rank == dst else None, dst=dst)
    counts = [c for _, c in sample_ranges(total, world)]
    n_acc, a_mean, a_m2 = counts[0], recv_mean[0].numpy(), recv_m2[0].numpy()
    for rr in range(1, world):
        n_acc, a_mean, a_m2 = chan_merge(n_acc, a_mean, a_m2, counts[rr], recv_mean[rr].numpy(), recv_m2[rr].numpy())
    merged = [torch.empty((c, 3)) for _, c in slices]
    for src in range(world):
        buf = torch.from_numpy(np.ascontiguousarray(a_mean)) if src == rank else merged[src]
        dist.broadcast(buf, src=src)
        merged[src] = buf
    full_mean = torch.cat(merged).numpy().reshape(h, w, 3)
    # --- tile partition: disjoint masks, SUM all-reduce assembles the frame
    mask = owned_blocks(w, h, rank, world)
    one = orc.render(osc, cam, 2, threads=1)["accum"]
    part = torch.from_numpy(np.where(mask[..., None], one, 0).astype(np.float32))
    dist.all_reduce(part)
    if rank == 0:
        q.put((full_mean, part.numpy(), one))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_sample_split_and_tile_partition():
    sys.path.insert(0, ROOT)
    from oracle import host_prep as hp, oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full_mean, summed, one = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sb = hp.prepare_fixture(fixture("cornell_box"))
    cam = hp.camera_to_gpu(sb.camera, 32, 20)
    seq = orc.render(orc.OracleScene(sb), cam, 10, threads=1)["accum"][..., :3]
    scale = np.maximum(np.abs(seq), 1e-3)
    assert np.max(np.abs(full_mean - seq) / scale) < 1e-5  # Chan merge vs sequential running mean: rounding only
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
