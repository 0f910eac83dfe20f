"""GPU tier: the multi-process host path (torch.distributed rendezvous, device-buffer aliasing, slice
exchange, device Chan merge, tile assembly) with two ranks sharing the one GPU of the test box over gloo.
RCCL itself needs one GPU per rank and is exercised only by `bench.py --gpus N` on a multi-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest

from util import ROOT, fixture, same_bits

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)  # a stuck rank reports where, and does not outlive the test
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from pathtracer_amd import host, multigpu
    from pathtracer_amd.device import PathTracer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda:0")
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h, total = 95, 53, 12   # 5035 pixels: the two ranks' slices differ in length
    cam = sc.camera(w, h)
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    first, cnt = multigpu.sample_ranges(total, world)[rank]
    pt.trace_n(cam, first, 1, cnt)
    counts = [c for _, c in multigpu.sample_ranges(total, world)]
    n = multigpu.merge_sample_split(pt, dist, device, counts, cam)
    merged, merged_out, merged_m2 = pt.read_accum(), pt.read_output(), pt.read_moments()
    # tile partition
    pt.clear()
    pt.set_partition(rank, world)
    pt.trace_n(cam, 0, 1, 3)
    multigpu.assemble_tiles(pt, dist, device, cam)
    tiles, tiles_out = pt.read_accum(), pt.read_output()
    if rank == 0:
        q.put((n, merged, merged_out, merged_m2, tiles, tiles_out))
    dist.barrier()
    pt.close()
    dist.destroy_process_group()


def test_two_ranks_sample_split_and_tiles():
    import torch.multiprocessing as mp
    from pathtracer_amd import host
    from pathtracer_amd.device import PathTracer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        n, merged, merged_out, merged_m2, tiles, tiles_out = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert n == 12
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(95, 53)
    cam = sc.camera(95, 53)
    pt.trace_n(cam, 0, 1, 12)
    seq, seq_m2 = pt.read_accum(), pt.read_moments()
    scale = np.maximum(np.abs(seq[..., :3]), 1e-3)
    assert np.max(np.abs(merged[..., :3] - seq[..., :3]) / scale) < 1e-5
    assert np.mean(merged_out != pt.read_output()) < 0.01  # 8-bit output: at most isolated +-1 LSB flips
    # the merged second moments of the WHOLE frame (every slice, not only the rank's own) against the sequential ones,
    # and bit for bit against the same two partials merged in one process
    assert np.max(np.abs(merged_m2 - seq_m2) / np.maximum(np.abs(seq_m2), 1e-2)) < 1e-3
    from pathtracer_amd.multigpu import chan_merge
    parts = []
    for first, cnt in ((0, 6), (6, 6)):
        pt.clear()
        pt.trace_n(cam, first, 1, cnt)
        parts.append((cnt, pt.read_accum()[..., :3].copy(), pt.read_moments()))
    _, want_mean, want_m2 = chan_merge(*parts[0], *parts[1])
    assert same_bits(merged[..., :3], want_mean) == 0 and same_bits(merged_m2, want_m2) == 0
    pt.enable_moments(False)
    pt.clear()
    pt.trace_n(cam, 0, 1, 3)
    assert same_bits(tiles, pt.read_accum()) == 0 and (tiles_out == pt.read_output()).all()
    pt.close()


def _rccl_worker(port, q):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from pathtracer_amd import host, multigpu
    from pathtracer_amd.device import PathTracer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    device = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h = 96, 54
    cam = sc.camera(w, h)
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, 6)
    before, before_m2, before_out = pt.read_accum(), pt.read_moments(), pt.read_output()
    n = multigpu.merge_sample_split(pt, dist, device, [6], cam)   # all-to-all, all-gather, tonemap on the device buffers
    after, after_m2, after_out = pt.read_accum(), pt.read_moments(), pt.read_output()
    multigpu.assemble_tiles(pt, dist, device, cam)               # all-reduce on the aliased accumulation buffer
    tiles = pt.read_accum()
    # the library's own exchange (what bench.py --gpus N runs): communicator from an id carried by the process group
    multigpu.setup_comm(pt, dist)
    pt.merge_sample_split([6], cam)
    lib_after, lib_m2, lib_out = pt.read_accum(), pt.read_moments(), pt.read_output()
    pt.assemble_tiles(cam)
    lib_tiles = pt.read_accum()
    pt.comm_free()
    q.put((n, before, before_m2, before_out, after, after_m2, after_out, tiles, lib_after, lib_m2, lib_out, lib_tiles))
    pt.close()
    dist.destroy_process_group()


def test_rccl_collectives_on_aliased_device_buffers_single_rank():
    """The calls `bench.py --gpus N` makes over RCCL (all-to-all, all-gather and all-reduce on tensors that alias the
    context's device buffers), with the one rank a one-GPU box allows: a world of one must leave the frame as it was."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    try:
        n, before, before_m2, before_out, after, after_m2, after_out, tiles, lib_after, lib_m2, lib_out, lib_tiles = q.get(timeout=150)
        p.join(timeout=60)
        assert p.exitcode == 0
    finally:
        if p.is_alive():
            p.kill()
    assert n == 6
    assert same_bits(before, after) == 0 and same_bits(before_m2, after_m2) == 0 and (before_out == after_out).all()
    assert same_bits(before, tiles) == 0
    assert same_bits(before, lib_after) == 0 and same_bits(before_m2, lib_m2) == 0 and (before_out == lib_out).all()
    assert same_bits(before, lib_tiles) == 0
