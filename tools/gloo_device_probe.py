"""One-off probe (GPU box): the exchange over gloo with DEVICE tensors handed to gloo directly, two ranks on one GPU,
repeated; each rank dumps its stack and exits if it is stuck for 60 s.  The tests give gloo host copies instead
(multigpu._host_staged); this shows whether, and where, the direct form stalls.
usage: python tools/gloo_device_probe.py [rounds]"""
import faulthandler
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, rounds):
    faulthandler.dump_traceback_later(60, exit=True)
    import torch
    import torch.distributed as dist
    from pathtracer_amd import host, multigpu
    from pathtracer_amd.device import PathTracer
    multigpu._host_staged = lambda dist, device: False   # the point of the probe
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda:0")
    sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "cornell_box.npz")).build()
    w, h = 95, 53
    cam = sc.camera(w, h)
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    for i in range(rounds):
        pt.clear()
        pt.trace_n(cam, rank * 6, 1, 6)
        multigpu.merge_sample_split(pt, dist, device, [6, 6], cam)
        pt.clear()
        pt.set_partition(rank, world)
        pt.trace_n(cam, 0, 1, 2)
        multigpu.assemble_tiles(pt, dist, device, cam)
        pt.set_partition(0, 1)
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(60, exit=True)
        if rank == 0:
            print(f"round {i} ok", flush=True)
    dist.barrier()
    pt.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, port, rounds)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=200)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    print("exit codes", codes)
    sys.exit(0 if codes == [0, 0] else 1)
