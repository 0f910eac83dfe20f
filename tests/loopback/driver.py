"""Runs the library's own multi-GPU exchange (ptx_comm_init, ptx_merge_sample_split, ptx_assemble_tiles) with several
ranks on ONE GPU: every rank is a thread of this process with its own context, and RCCL is the in-process stand-in
tests/loopback/librccl.so.1, loaded before libptx355.so so that the product adopts it.  Started by
tests/test_gpu_x_loopback.py in a process of its own (PTX355_NO_TORCH=1: torch would bring the real RCCL).

    python tests/loopback/driver.py <case> <world> <out.npz>
"""
import ctypes
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
assert os.environ.get("PTX355_NO_TORCH") == "1", "run with PTX355_NO_TORCH=1"
LOOP = ctypes.CDLL(os.path.join(HERE, "librccl.so.1"), mode=ctypes.RTLD_GLOBAL)

from pathtracer_amd import host, multigpu  # noqa: E402
from pathtracer_amd.device import PathTracer, PtxError  # noqa: E402


def run_ranks(world, body):
    """body(rank) on `world` threads; returns their results in rank order, re-raising the first failure."""
    out, err = [None] * world, [None] * world

    def go(r):
        try:
            out[r] = body(r)
        except BaseException as e:  # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    for t in ts:
        assert not t.is_alive(), "a rank is stuck"
    for e in err:
        if e is not None:
            raise e
    return out


def scene():
    return host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "cornell_box.npz")).build()


def sample_split(world, w, h, counts, out):
    """Every rank renders its own frame range; the exchange merges them.  Saves every rank's partial and merged buffers."""
    sc = scene()
    uid = PathTracer.comm_unique_id()
    firsts = np.concatenate([[0], np.cumsum(counts)[:-1]])

    def body(r):
        pt = PathTracer(0)
        sc.upload_to(pt)
        pt.enable_moments(True)
        pt.resize(w, h)
        cam = sc.camera(w, h)
        if counts[r]:
            pt.trace_n(cam, int(firsts[r]), 1, int(counts[r]))
        elif r == 0:
            # a rank without samples whose buffers hold something else: it must contribute nothing
            pt.write_accum(np.full((h, w, 4), 7.0, np.float32), np.full((h, w, 3), 3.0, np.float32))
        part = (pt.read_accum(), pt.read_moments())
        pt.comm_init(uid, r, world)
        pt.merge_sample_split(counts, cam)
        res = part + (pt.read_accum(), pt.read_moments(), pt.read_output())
        pt.comm_free()
        pt.close()
        return res

    res = run_ranks(world, body)
    out.update({f"part_mean_{r}": res[r][0] for r in range(world)})
    out.update({f"part_m2_{r}": res[r][1] for r in range(world)})
    out.update({f"mean_{r}": res[r][2] for r in range(world)})
    out.update({f"m2_{r}": res[r][3] for r in range(world)})
    out.update({f"rgba8_{r}": res[r][4] for r in range(world)})
    out["counts"] = np.asarray(counts, np.uint32)


def tiles(world, w, h, n, out, tile=None):
    sc = scene()
    uid = PathTracer.comm_unique_id()

    def body(r):
        pt = PathTracer(0)
        sc.upload_to(pt)
        pt.resize(w, h)
        if tile:
            pt.set_tile(*tile)
        cam = sc.camera(w, h)
        pt.set_partition(r, world)
        pt.trace_n(cam, 0, 1, n)
        pt.comm_init(uid, r, world)
        pt.assemble_tiles(cam)
        res = (pt.read_accum(), pt.read_output())
        pt.comm_free()
        pt.close()
        return res

    res = run_ranks(world, body)
    out.update({f"tiles_mean_{r}": res[r][0] for r in range(world)})
    out.update({f"tiles_rgba8_{r}": res[r][1] for r in range(world)})
    # the single-context frame of the same tile
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.resize(w, h)
    if tile:
        pt.set_tile(*tile)
    pt.trace_n(sc.camera(w, h), 0, 1, n)
    out["tiles_want_mean"], out["tiles_want_rgba8"] = pt.read_accum(), pt.read_output()
    pt.close()


def tiles_by_rectangles(world, w, h, n, out):
    """A disjoint split that is not the communicator's partition (horizontal bands set with ptx_set_tile): the all-reduce path."""
    sc = scene()
    uid = PathTracer.comm_unique_id()
    bands = multigpu.sample_ranges(h, world)

    def body(r):
        pt = PathTracer(0)
        sc.upload_to(pt)
        pt.resize(w, h)
        cam = sc.camera(w, h)
        y0, rows = bands[r]
        if rows:
            pt.set_tile(0, y0, w, rows)
            pt.trace_n(cam, 0, 1, n)
        pt.comm_init(uid, r, world)
        pt.assemble_tiles(cam)
        res = (pt.read_accum(), pt.read_output())
        pt.comm_free()
        pt.close()
        return res

    res = run_ranks(world, body)
    out.update({f"bands_mean_{r}": res[r][0] for r in range(world)})
    out.update({f"bands_rgba8_{r}": res[r][1] for r in range(world)})
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.resize(w, h)
    pt.trace_n(sc.camera(w, h), 0, 1, n)
    out["bands_want_mean"], out["bands_want_rgba8"] = pt.read_accum(), pt.read_output()
    pt.close()


def failing_send(out):
    """A send that fails inside the group: the call reports it, the thread's group is closed, and after the failed
    communicators are dropped a new world exchanges correctly."""
    os.environ["LOOPBACK_TIMEOUT_MS"] = "1500"
    world, w, h = 2, 33, 17
    sc = scene()
    uid = PathTracer.comm_unique_id()
    LOOP.loopback_group_depth.restype = ctypes.c_int
    gate = threading.Barrier(world)

    def body(r):
        pt = PathTracer(0)
        sc.upload_to(pt)
        pt.enable_moments(True)
        pt.resize(w, h)
        cam = sc.camera(w, h)
        pt.trace_n(cam, 3 * r, 1, 3)
        pt.comm_init(uid, r, world)
        gate.wait()
        if r == 0:
            LOOP.loopback_fail_next_sends(1)
        gate.wait()
        msg = ""
        try:
            if r == 0:
                pt.merge_sample_split([3, 3], cam)   # its first send fails
            gate.wait()
            if r == 1:
                pt.merge_sample_split([3, 3], cam)   # its peer never comes: the rendezvous gives up
        except PtxError as e:
            msg = str(e)
            if r == 0:
                gate.wait()
        depth = LOOP.loopback_group_depth()
        pt.comm_free()
        pt.close()
        return msg, depth

    res = run_ranks(world, body)
    out["fail_msgs"] = np.array([res[0][0], res[1][0]])
    out["fail_depths"] = np.array([res[0][1], res[1][1]], np.int32)
    os.environ["LOOPBACK_TIMEOUT_MS"] = "60000"
    sample_split(2, w, h, [3, 3], out)


def main():
    case, world, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    out = {}
    if case == "split":
        counts = [c for _, c in multigpu.sample_ranges(12, world)]
        sample_split(world, 95, 53, counts, out)
    elif case == "split_rank0_empty":
        sample_split(world, 40, 24, [0] + [2] * (world - 1), out)
    elif case == "split_tiny_frame":
        sample_split(world, 2, 2, [1] * world, out)   # more ranks than pixels: empty slices
    elif case == "tiles":
        tiles(world, 95, 53, 3, out)
        tiles_by_rectangles(world, 95, 53, 3, out)
    elif case == "tiles_in_a_tile":
        tiles(world, 95, 53, 2, out, tile=(9, 5, 70, 41))
    elif case == "failing_send":
        failing_send(out)
    else:
        raise SystemExit(f"unknown case {case}")
    np.savez(path, **out)
    print("ok", case, world)


if __name__ == "__main__":
    main()
