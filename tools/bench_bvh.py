"""Times the BVH build of the height-field mesh scene on the host (pth_bvh_build, one core) and on the GPU
(ptx_bvh_build), checks that both give the same arrays, and prints one JSON line.

python tools/bench_bvh.py [--mesh-quads 2236] [--repeat 3] [--skip-host]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh-quads", type=int, default=2236)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--skip-host", action="store_true")
    args = ap.parse_args()
    from pathtracer_amd import host
    from pathtracer_amd.device import PathTracer

    def scene():
        s = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", "all.npz"))
        s.append_heightfield(args.mesh_quads)
        return s

    pt = PathTracer(0)
    dev = scene()
    dev.build(decode_textures=False, tracer=pt)  # first call: module load, allocator warm-up
    times = []
    for _ in range(args.repeat):
        dev.built = False
        t0 = time.perf_counter()
        dev.build(decode_textures=False, tracer=pt)
        times.append(time.perf_counter() - t0)
    db = dev.buffers()
    out = {"mesh_quads": args.mesh_quads, "primitives": int(len(db["prims"])), "nodes": int(len(db["nodes"])),
           "scene_build_gpu_bvh_s": [round(t, 4) for t in times]}
    # the builder alone, on the boxes of the same primitives (host-resident in, host-resident out)
    nodes = db["nodes"]
    leaves = nodes[nodes["prim_count"] > 0]
    out["leaves"] = int(len(leaves))
    figs = db["figs"]
    tri = figs[figs["shape_type"] == 7]
    v = np.stack([tri["v0"], tri["v1"], tri["v2"]], axis=1)
    mn, mx = v.min(axis=1), v.max(axis=1)
    flat = (mx - mn) < np.float32(1e-4)   # aabb.rs:69-81
    boxes = np.concatenate([np.where(flat, mn - np.float32(1e-4), mn), np.where(flat, mx + np.float32(1e-4), mx)], axis=1).astype(np.float32)
    del v, mn, mx, flat
    alone = []
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        gn, gp = pt.bvh_build(boxes)
        alone.append(time.perf_counter() - t0)
    out["triangle_boxes"] = int(len(boxes))
    out["ptx_bvh_build_s"] = [round(t, 4) for t in alone]
    if not args.skip_host:
        t0 = time.perf_counter()
        hn, hp_ = host.bvh_build(boxes)
        out["pth_bvh_build_s"] = round(time.perf_counter() - t0, 4)
        out["builders_identical_on_triangle_boxes"] = bool(len(hn) == len(gn) and (hp_ == gp).all() and all(
            (hn[k] == gn[k]).all() for k in ("aabb_min", "aabb_max", "left_or_prim", "prim_count")))
    if not args.skip_host:
        hs = scene()
        t0 = time.perf_counter()
        hs.build(decode_textures=False)
        out["scene_build_host_bvh_s"] = round(time.perf_counter() - t0, 4)
        hb = hs.buffers()
        same_nodes = (hb["nodes"]["left_or_prim"] == nodes["left_or_prim"]).all() and (hb["nodes"]["prim_count"] == nodes["prim_count"]).all() \
            and (hb["nodes"]["aabb_min"] == nodes["aabb_min"]).all() and (hb["nodes"]["aabb_max"] == nodes["aabb_max"]).all()
        out["identical_to_host_build"] = bool(same_nodes and len(hb["nodes"]) == len(nodes) and (hb["prims"] == db["prims"]).all())
    print(json.dumps(out), flush=True)
    pt.close()


if __name__ == "__main__":
    main()
