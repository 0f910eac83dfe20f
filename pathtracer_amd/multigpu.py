"""One process per GPU.  Two ways to split a render, neither with a collective on the per-sample path:

* tile partition: every rank owns an interleaved set of 16x16-pixel blocks of the frame
  (`PathTracer.set_partition`); the ranks' buffers are disjoint, so one SUM all-reduce assembles the
  single-GPU frame bit for bit;
* sample split: every rank renders the full frame for its own frame-index range and keeps per-pixel
  Welford moments (n, mean, M2); the partials are exchanged slice-wise (all-to-all), merged with Chan's
  parallel formula in fixed rank order by `ptx_merge_moments`, and the merged slices all-gathered.

`torch.distributed` provides rendezvous and the collectives ("nccl" is RCCL on ROCm; "gloo" on CPU for
tests); the merge arithmetic for CPU tensors is `chan_merge` below, used by the gloo tests.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def sample_ranges(total: int, world: int) -> List[Tuple[int, int]]:
    """Splits `total` samples into `world` contiguous (first, count) ranges, sizes differing by <= 1."""
    base, extra = divmod(total, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((first, n))
        first += n
    return out


def pixel_slices(n_pixels: int, world: int) -> List[Tuple[int, int]]:
    return sample_ranges(n_pixels, world)


def owned_blocks(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """Boolean (height, width) mask of the pixels `set_partition(rank, world)` makes a rank trace."""
    bx, by = (width + 15) // 16, (height + 15) // 16
    idx = np.arange(bx * by).reshape(by, bx)
    own = (idx % world) == rank
    return np.kron(own, np.ones((16, 16), dtype=bool))[:height, :width]


def chan_merge(n_a, mean_a, m2_a, n_b, mean_b, m2_b):
    """Chan et al. pairwise update of (count, mean, M2); float32 arrays, same expression order as the
    device merge kernel."""
    f = np.float32
    if n_b == 0:
        return n_a, mean_a, m2_a
    if n_a == 0:
        return n_b, mean_b, m2_b
    n = f(n_a) + f(n_b)
    wb, wab = f(n_b) / n, f(n_a) * f(n_b) / n
    d = mean_b - mean_a
    mean = mean_a + d * wb
    m2 = m2_a + m2_b + d * d * wab if m2_a is not None and m2_b is not None else None
    return n_a + n_b, mean, m2


def device_tensor(ptr: int, shape, dtype, device):
    """A torch tensor aliasing raw device memory owned by a ptx context."""
    import torch

    class _Raw:
        pass

    typestr = {torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    raw = _Raw()
    raw.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(raw, device=device)


def assemble_tiles(tracer, dist, device, cam=None):
    """Tile partition epilogue: SUM all-reduce of the disjoint accumulation and output buffers."""
    import torch
    tracer.sync()
    n = tracer.width * tracer.height
    acc = device_tensor(tracer.device_accum_ptr(), (n * 4,), torch.float32, device)
    if _host_staged(dist, device):
        host = acc.cpu()
        dist.all_reduce(host)
        acc.copy_(host)
    else:
        dist.all_reduce(acc)
    torch.cuda.synchronize(device)
    if cam is not None:
        tracer.tonemap(0, n, cam)
        tracer.sync()


def merge_sample_split(tracer, dist, device, counts: List[int], cam):
    """Sample split epilogue.  counts[r] = samples rank r accumulated.  On return every rank holds the
    merged mean / M2 / output for the whole frame."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    tracer.sync()
    n = tracer.width * tracer.height
    acc = device_tensor(tracer.device_accum_ptr(), (n, 4), torch.float32, device)
    m2p = tracer.device_moments_ptr()
    m2 = device_tensor(m2p, (n, 3), torch.float32, device) if m2p else None
    slices = pixel_slices(n, world)
    # slice exchange: rank j receives slice j of every rank's partial
    mine = slices[rank]
    recv_a = _exchange_slices(dist, acc, slices, rank, world, 4, device)
    recv_m = _exchange_slices(dist, m2, slices, rank, world, 3, device) if m2 is not None else None
    # fixed rank order: start from partial 0, fold in 1, 2, ...
    acc[mine[0]:mine[0] + mine[1]].copy_(recv_a[0])
    if m2 is not None:
        m2[mine[0]:mine[0] + mine[1]].copy_(recv_m[0])
    torch.cuda.synchronize(device)
    n_a = counts[0]
    for r in range(1, world):
        tracer.merge_moments(mine[0], mine[1], n_a, recv_a[r].data_ptr(), recv_m[r].data_ptr() if recv_m else 0, counts[r], cam)
        n_a += counts[r]
    tracer.sync()
    # merged slices back to everyone
    xdev = torch.device("cpu") if _host_staged(dist, device) else device
    gathered = [torch.empty((c, 4), dtype=torch.float32, device=xdev) for _, c in slices]
    merged = acc[mine[0]:mine[0] + mine[1]].contiguous().to(xdev)
    if len({c for _, c in slices}) == 1:
        dist.all_gather(gathered, merged)
    else:
        _all_gather_uneven(dist, gathered, merged, slices)
    for (f, c), g in zip(slices, gathered):
        acc[f:f + c].copy_(g)
    torch.cuda.synchronize(device)
    tracer.tonemap(0, n, cam)  # the other ranks' slices arrived as means only
    tracer.sync()
    return n_a


def _host_staged(dist, device) -> bool:
    """gloo (the test backend) moves device buffers through host copies made here: its own handling of device
    tensors stalls now and then when two ranks share one GPU.  RCCL works on the device buffers directly."""
    import torch
    return dist.get_backend() != "nccl" and torch.device(device).type != "cpu"


def _exchange_slices(dist, buf, slices, rank, world, width, device):
    """Every rank sends slice j of `buf` to rank j and receives slice `rank` from everyone.  RCCL: one
    all-to-all (all peer links busy at once); gloo (tests): one gather per destination."""
    import torch
    mine = slices[rank]
    if dist.get_backend() == "nccl":
        recv = [torch.empty((mine[1], width), dtype=torch.float32, device=device) for _ in range(world)]
        send = [buf[f:f + c].contiguous() for f, c in slices]
        dist.all_to_all(recv, send)
        return recv
    xdev = torch.device("cpu") if _host_staged(dist, device) else device
    recv = [torch.empty((mine[1], width), dtype=torch.float32, device=xdev) for _ in range(world)]
    send = [buf[f:f + c].contiguous().to(xdev) for f, c in slices]
    for dst in range(world):
        dist.gather(send[dst], recv if rank == dst else None, dst=dst)
    return [r.to(device) for r in recv]


def _all_gather_uneven(dist, out_list, mine, slices):
    for r, (_, c) in enumerate(slices):
        buf = mine if r == dist.get_rank() else out_list[r]
        dist.broadcast(buf, src=r)
        if r == dist.get_rank():
            out_list[r].copy_(mine)
