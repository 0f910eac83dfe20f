"""One process per GPU.  Two ways to split a render, neither with a collective on the per-sample path:

* tile partition: every rank owns an interleaved set of 16x16-pixel blocks of the frame
  (`PathTracer.set_partition`); the ranks' buffers are disjoint, so one SUM all-reduce assembles the
  single-GPU frame bit for bit;
* sample split: every rank renders the full frame for its own frame-index range and keeps per-pixel
  Welford moments (n, mean, M2); the partials are exchanged slice-wise (all-to-all), merged with Chan's
  parallel formula in fixed rank order by `ptx_merge_moments`, and the merged slices all-gathered.

Two implementations of the exchange.  The product's is inside the library (`ptx_merge_sample_split`,
`ptx_assemble_tiles`: RCCL send/recv groups on the context's stream): `setup_comm` creates its communicator, with
`torch.distributed` (any backend) only carrying the 128-byte id and serving as the launcher's rendezvous.  The
second, `merge_sample_split` / `assemble_tiles` below, runs the same steps over `torch.distributed` collectives on
any "partial" (a GPU context's buffers, or CPU arrays: `HostPartial`), which is how the exchange logic is tested
on CPU with gloo at world sizes a one-GPU box cannot give RCCL.
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


class DevicePartial:
    """A ptx context's accumulation / M2 buffers as torch tensors on its GPU, merged by `ptx_merge_moments`."""

    def __init__(self, tracer, device):
        import torch
        self.tracer, self.device = tracer, device
        self.n = tracer.width * tracer.height
        self.acc = device_tensor(tracer.device_accum_ptr(), (self.n, 4), torch.float32, device)
        m2p = tracer.device_moments_ptr()
        self.m2 = device_tensor(m2p, (self.n, 3), torch.float32, device) if m2p else None

    def sync(self):
        import torch
        self.tracer.sync()
        torch.cuda.synchronize(self.device)

    def merge(self, first, count, n_a, other_mean4, other_m2, n_b, cam):
        self.tracer.merge_moments(first, count, n_a, other_mean4.data_ptr(), other_m2.data_ptr() if other_m2 is not None else 0, n_b, cam)

    def tonemap(self, cam):
        self.tracer.tonemap(0, self.n, cam)
        self.tracer.sync()


class HostPartial:
    """The same interface over CPU arrays (mean (h, w, 4) and M2 (h, w, 3) float32, e.g. rendered by the CPU oracle):
    the exchange logic without a GPU.  Merges with `chan_merge`, the device kernel's arithmetic."""

    def __init__(self, accum: np.ndarray, m2: np.ndarray = None):
        import torch
        self.device = torch.device("cpu")
        self.n = accum.shape[0] * accum.shape[1]
        self.acc = torch.from_numpy(np.ascontiguousarray(accum, dtype=np.float32).reshape(self.n, 4).copy())
        self.m2 = torch.from_numpy(np.ascontiguousarray(m2, dtype=np.float32).reshape(self.n, 3).copy()) if m2 is not None else None

    def sync(self):
        pass

    def merge(self, first, count, n_a, other_mean4, other_m2, n_b, cam):
        if n_b == 0 or count == 0:
            return
        a = self.acc[first:first + count].numpy()
        m = self.m2[first:first + count].numpy() if self.m2 is not None and other_m2 is not None else None
        _, mean, m2 = chan_merge(n_a, a[:, :3].copy(), m.copy() if m is not None else None, n_b, other_mean4.numpy()[:, :3],
                                 other_m2.numpy() if m is not None else None)
        a[:, :3], a[:, 3] = mean, 1.0
        if m is not None:
            m[:] = m2

    def tonemap(self, cam):
        pass


def _as_partial(x, device):
    return x if hasattr(x, "acc") else DevicePartial(x, device)


def setup_comm(tracer, dist):
    """Creates the context's RCCL communicator for the library's own exchange: rank 0 makes the id, the launcher's
    process group (any backend) carries it."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [tracer.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    tracer.comm_init(box[0], rank, world)


def assemble_tiles(tracer, dist, device, cam=None):
    """Tile partition epilogue over torch.distributed: SUM all-reduce of the disjoint accumulation buffers."""
    import torch
    p = _as_partial(tracer, device)
    p.sync()
    acc = p.acc.view(-1)
    if _host_staged(dist, device):
        host = acc.cpu()
        dist.all_reduce(host)
        acc.copy_(host)
    else:
        dist.all_reduce(acc)
    p.sync()
    if cam is not None:
        p.tonemap(cam)


def merge_sample_split(tracer, dist, device, counts: List[int], cam):
    """Sample split epilogue over torch.distributed.  counts[r] = samples rank r accumulated.  On return every rank
    holds the merged mean and M2 of the whole frame (and, for a GPU context, its rgba8 image)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    p = _as_partial(tracer, device)
    p.sync()
    n, acc, m2 = p.n, p.acc, p.m2
    slices = pixel_slices(n, world)
    # slice exchange: rank j receives slice j of every rank's partial
    mine = slices[rank]
    recv_a = _exchange_slices(dist, acc, slices, rank, world, 4, device)
    recv_m = _exchange_slices(dist, m2, slices, rank, world, 3, device) if m2 is not None else None
    # fixed rank order: start from partial 0, fold in 1, 2, ...
    acc[mine[0]:mine[0] + mine[1]].copy_(recv_a[0])
    if m2 is not None:
        m2[mine[0]:mine[0] + mine[1]].copy_(recv_m[0])
    p.sync()
    n_a = counts[0]
    for r in range(1, world):
        p.merge(mine[0], mine[1], n_a, recv_a[r], recv_m[r] if recv_m else None, counts[r], cam)
        n_a += counts[r]
    p.sync()
    # merged slices (means and second moments) back to everyone
    for buf, width in ((acc, 4), (m2, 3)):
        if buf is None:
            continue
        xdev = torch.device("cpu") if _host_staged(dist, device) else device
        gathered = [torch.empty((c, width), dtype=torch.float32, device=xdev) for _, c in slices]
        merged = buf[mine[0]:mine[0] + mine[1]].contiguous().to(xdev)
        if len({c for _, c in slices}) == 1:
            dist.all_gather(gathered, merged)
        else:
            _all_gather_uneven(dist, gathered, merged, slices)
        for (f, c), g in zip(slices, gathered):
            buf[f:f + c].copy_(g)
    p.sync()
    p.tonemap(cam)  # the other ranks' slices arrived as means only
    return n_a


def _host_staged(dist, device) -> bool:
    """gloo (the test backend) is given host copies of device buffers: it is a CPU backend, and its handling of
    device tensors stalled now and then with two ranks on one GPU (cause not established; the product's exchange
    does not go through it)."""
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
    xdev = torch.device("cpu") if _host_staged(dist, device) else torch.device(device)
    recv = [torch.empty((mine[1], width), dtype=torch.float32, device=xdev) for _ in range(world)]
    send = [buf[f:f + c].contiguous().to(xdev) for f, c in slices]
    for dst in range(world):
        # gloo's gather wants equally sized tensors: the destination's slice length is what every rank sends
        dist.gather(send[dst], recv if rank == dst else None, dst=dst)
    return [r.to(device) for r in recv]


def _all_gather_uneven(dist, out_list, mine, slices):
    for r, (_, c) in enumerate(slices):
        buf = mine if r == dist.get_rank() else out_list[r]
        dist.broadcast(buf, src=r)
        if r == dist.get_rank():
            out_list[r].copy_(mine)
