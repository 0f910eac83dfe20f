"""Throughput of the path-tracing compute pass on MI355X.

python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cornell|teapot|fractals|all_mesh] [--spp-per-step S]

A step is one batched launch of the hot path: S samples for every pixel of the workload's frame, inputs
(scene, accumulation buffer) resident in HBM.  N > 1 (one process per GPU under torch.distributed.run; started
without it, this script starts that launcher itself as a child process): sample split -- every rank renders the
full frame for its own frame-index range (weak scaling: per-GPU work is fixed), and the per-GPU Welford moments
are merged once at the end of the timed region with Chan's formula over RCCL; a second, fixed-total-work leg
(the same 1024 samples per pixel whatever N: strong scaling) follows and is reported beside it.  Rank 0 prints
ONE JSON line.  At N = 1 the default run also times the other three workloads briefly (teapot 4K, fractals,
all.yaml + 10 M triangles) and attaches them as `other_workloads`, each with its own roofline and CPU baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (fixture, width, height, default spp per step, label)
    "cornell": ("cornell_box", 1920, 1080, 128, "cornell_box.yaml 1920x1080 (configs[1]: analytic primitives, 1024 spp = 8 steps x 128)"),
    "teapot": ("teapot", 3840, 2160, 16, "teapot.yaml 3840x2160 OBJ BVH"),
    "fractals": ("fractals", 1920, 1080, 128, "fractals.yaml 1920x1080 Mandelbulb/Julia ray-march"),
    "all_mesh": ("all", 1920, 1080, 8, "all.yaml + synthetic height-field mesh 1920x1080"),
}
# bounded CPU samples (whole frames, 5-30 s of work on the box's host cores): frames per workload
CPU_FRAMES = {"cornell": 64, "teapot": 24, "fractals": 8, "all_mesh": 4}
PEAK_FP32_VALU_TFLOPS = 157.3  # 256 CUs x 4 SIMDs x 32 lanes x 2 flop x 2.4 GHz: the vector peak, reached by plain v_fma_f32 (a packed v_pk_fma_f32 issues at half the rate)
PEAK_HBM_GBS = 8000.0
METRIC = "Mpaths/s at 1080p (1/2/4/8 GPUs) + % FP32 VALU peak; image RMSE vs ref"


def cpu_baseline(scene, cam, label: str, spp: int, tile=None):
    """The CPU oracle on a bounded sample of the same workload (same scene buffers, same camera), timed on this
    host's cores; also returns the oracle's work counters (the algorithmic flop figure comes from them)."""
    from oracle import host_prep as hp, oracle as orc
    b = scene.buffers()
    sb = hp.SceneBuffers(b["figs"], b["mats"], b["nodes"], b["prims"], b["lights"], b["infinite"], b["tex_pixels"], b["tex_infos"], None)
    threads = host_threads()
    osc = orc.OracleScene(sb)
    t0 = time.perf_counter()
    r = orc.render(osc, cam, spp, counters=True, threads=threads, tile=tile)
    dt = time.perf_counter() - t0
    c = r["counters"]
    where = "whole frame" if tile is None else f"tile x0={tile[0]} y0={tile[1]} {tile[2]}x{tile[3]}"
    return {"value": round(c["paths"] / dt / 1e6, 4), "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "sample": f"{label}: {where}, frames 0..{spp - 1} ({c['paths']} paths, {dt:.1f} s wall)",
            "cores_note": f"{threads} OpenMP threads = every core this process may use ({host_threads.why})",
            "gpath_bounce_per_s": round(c["bounces"] / dt / 1e9, 6)}, c, orc.counted_flops(c), r


def host_threads() -> int:
    """Every host core this process is given: its affinity mask, or the job's CPU share when the launcher states one
    in OMP_NUM_THREADS (a one-GPU slice of a many-GPU host is allotted 16 of the machine's cores that way)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and 0 < int(share) < aff:
        host_threads.why = f"OMP_NUM_THREADS={share} set by the launcher; affinity mask has {aff}"
        return int(share)
    host_threads.why = f"affinity mask has {aff}"
    return max(1, aff)


def image_rmse(pt, cam, spp: int, ref: dict, tile=None):
    """The metric's 'image RMSE vs ref': the GPU's frames 0..spp-1 of this workload against the oracle's render of the
    same frames (the one timed for cpu_baseline), outside the timed region.  Relative RMSE of the linear mean weighted
    by luminance, RMSE of the rgba8 image in LSB, and the number of f32 words / primary ids that differ at all."""
    pt.clear()
    pt.trace_n(cam, 0, 1, spp)
    a, out, ids = pt.read_accum(), pt.read_output(), pt.read_primary_ids()
    W, H = int(cam[0]["width"]), int(cam[0]["height"])
    x0, y0, tw, th = tile if tile is not None else (0, 0, W, H)
    sl = (slice(y0, y0 + th), slice(x0, x0 + tw))
    g, r = a[sl][..., :3].astype(np.float64), ref["accum"][sl][..., :3].astype(np.float64)
    lw = np.array([0.2126, 0.7152, 0.0722])
    lum_err = ((g - r) @ lw) ** 2
    mean_lum = float((r @ lw).mean())
    d8 = out[sl].astype(np.float64) - ref["rgba8"][sl].astype(np.float64)
    return {"linear_rel_rmse": float(np.sqrt(lum_err.mean()) / max(mean_lum, 1e-30)), "rgba8_rmse_lsb": float(np.sqrt((d8 ** 2).mean())),
            "f32_words_differing": int((a[sl].view(np.uint32) != ref["accum"][sl].view(np.uint32)).sum()),
            "primary_ids_differing": int((ids[sl] != ref["primary_id"][sl]).sum()),
            "frames": spp, "pixels": int(tw * th)}


def relaunch_under_torchrun(args) -> int:
    """N > 1 without a launcher: this process has made no GPU call yet; it starts one process per GPU as a child
    (torch.distributed.run, rendezvous on 127.0.0.1), relays the ranks' output and returns the child's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


class Dist:
    """The launcher's process group (RCCL, or gloo to rehearse with ranks sharing GPUs) and this rank's place in it."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        import torch
        self.torch = torch
        # PTX355_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share GPUs).
        self.backend = os.environ.get("PTX355_DIST_BACKEND", "nccl")
        self.dev_index = self.local_rank % max(torch.cuda.device_count(), 1) if self.backend != "nccl" else self.local_rank
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            torch.cuda.set_device(self.dev_index)
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device(f"cuda:{self.dev_index}"))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.device = torch.device(f"cuda:{self.dev_index}")

    def barrier(self, pt):
        pt.sync()
        if self.dist is not None:
            self.torch.cuda.synchronize(self.device)
            self.dist.barrier()
            self.torch.cuda.synchronize(self.device)

    def max_over_ranks(self, v: float) -> float:
        if self.dist is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v: float) -> float:
        if self.dist is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def run_workload(name: str, args, D: Dist, steps: int, warmup: int, spp_override: int = 0, with_cpu: bool = True, strong_leg: bool = False) -> dict:
    """One workload: scene build + upload, warm-up, the timed region, then (outside it) the CPU baseline, the image RMSE
    and the roofline figures.  Returns the bench line as a dict (rank 0; None on the other ranks)."""
    from pathtracer_amd import host, multigpu
    from pathtracer_amd.device import PathTracer
    rank, world, dist, device = D.rank, D.world, D.dist, D.device
    fx_name, w, h, spp_default, label = WORKLOADS[name]
    spp = spp_override or spp_default
    scene = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{fx_name}.npz"))
    if name == "all_mesh":
        scene.append_heightfield(args.mesh_quads)
    pt = PathTracer(D.dev_index)
    options = {}
    for kv in args.opt:
        k, v = kv.split("=", 1)
        pt.set_option(k, int(v))
        options[k] = int(v)
    # the mesh scene's tree is built on the GPU (ptx_bvh_build: the same arrays as the host builder, 60x sooner)
    gpu_bvh = name == "all_mesh"
    t0 = time.perf_counter()
    scene.build(decode_textures=False, tracer=pt if gpu_bvh else None)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    scene.upload_to(pt)
    upload_s = time.perf_counter() - t0
    partition = args.partition or ("tiles" if name == "teapot" else "samples")
    tiles = world > 1 and partition == "tiles"
    if world > 1 and not tiles:
        pt.enable_moments(True)
    if world > 1 and args.exchange == "library":
        multigpu.setup_comm(pt, dist)  # the context's own RCCL communicator; the process group only carries its id
    pt.resize(w, h)
    if tiles:
        pt.set_partition(rank, world)  # this rank's interleaved 16x16 blocks; every rank traces the same frames
    cam = scene.camera(w, h)

    def epilogue(counts):
        if args.exchange == "library":
            (pt.assemble_tiles(cam) if tiles else pt.merge_sample_split(counts, cam))
        elif tiles:
            multigpu.assemble_tiles(pt, dist, device, cam)
        else:
            multigpu.merge_sample_split(pt, dist, device, counts, cam)

    # sample split: rank r owns frames [r * steps * spp, (r + 1) * steps * spp), samples counted locally
    frame0 = 0 if tiles else rank * (steps + warmup) * spp
    for i in range(warmup):
        pt.trace_n(cam, frame0 + i * spp, 1 + i * spp, spp)
    if dist is not None and warmup > 0:
        # the first collectives set up RCCL's channels: keep that out of the timed region
        epilogue([warmup * spp] * world)
    pt.sync()
    pt.clear()
    pt.kernel_time()
    D.barrier(pt)
    t0 = time.perf_counter()
    for i in range(steps):
        pt.trace_n(cam, frame0 + (warmup + i) * spp, 1 + i * spp, spp)
    if dist is not None:
        epilogue([steps * spp] * world)
    D.barrier(pt)
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    kernel_ms, launches = pt.kernel_time()
    per_launch = pt.kernel_times()[-steps:]
    launch_info = pt.launch_info()
    st = pt.stats()
    total_paths = w * h * spp * steps * (1 if tiles else world)  # tiles: the ranks share one frame
    bounces = st["bounces"]
    total_bounces = D.sum_over_ranks(float(bounces))

    strong = None
    if strong_leg and world > 1 and not tiles:
        # Fixed total work: the SAME 1024 samples per pixel (steps x spp) whatever the number of GPUs, each rank tracing its
        # share of the frame indices in launches of at most `spp`, one merge inside the timed region.
        total_spp = steps * spp
        share = multigpu.sample_ranges(total_spp, world)
        first, cnt = share[rank]
        pt.clear()
        D.barrier(pt)
        t0 = time.perf_counter()
        done = 0
        while done < cnt:
            m = min(spp, cnt - done)
            pt.trace_n(cam, first + done, 1 + done, m)
            done += m
        epilogue([c for _, c in share])
        D.barrier(pt)
        s_elapsed = D.max_over_ranks(time.perf_counter() - t0)
        strong = {"scaling": "strong", "total_spp": total_spp, "spp_per_rank": [c for _, c in share], "value": round(w * h * total_spp / s_elapsed / 1e6, 2),
                  "unit": "Mpaths/s", "ms_total": round(s_elapsed * 1e3, 3),
                  "note": f"whole-job rate of one {total_spp}-spp render split over the ranks by frame index, merge included; at N = 1 this is the headline run"}

    out = None
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": round(total_paths / elapsed / 1e6, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(elapsed * 1e3 / steps, 3), "higher_is_better": True,
            "scaling": "strong" if tiles else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": w, "height": h, "spp_per_step": spp, "frames": steps * spp,
                       "partition": "single GPU" if world == 1 else (f"interleaved 16x16 tiles over {world} GPUs + RCCL block exchange" if tiles
                                                                       else f"sample split over {world} GPUs + RCCL Chan merge"),
                       "exchange": None if world == 1 else args.exchange, "options": options or None,
                       "scene_build_s": round(build_s, 3), "bvh_builder": "gpu" if gpu_bvh else "host", "scene_upload_s": round(upload_s, 3)},
            "gpath_bounce_per_s": round(total_bounces / elapsed / 1e9, 4),
            "bounces_per_path": round(bounces / max(st["paths"], 1), 4),
            # HIP events around every trace call on the context's stream.  Launches of sample-granular jobs overlap (each traces
            # on a stream of its own, the folds follow one another): a call's events then bracket the gap between two folds, not a
            # launch, and only their sum means anything -- no median is reported for such runs.
            "ms_per_step_median": round(float(np.median(per_launch)), 3) if len(per_launch) and float(per_launch.min()) > 0.5 * float(per_launch.mean()) else None,
            "launches_overlap": bool(len(per_launch) and float(per_launch.min()) <= 0.5 * float(per_launch.mean())),
            "image_rmse_vs_ref": None,
        }
        if strong is not None:
            out["strong_scaling"] = strong
        cpu, flops_per_bounce = None, None
        if with_cpu and world == 1:
            cpu_spp = CPU_FRAMES[name]
            cpu, cnt, flops, ref = cpu_baseline(scene, cam, label, cpu_spp, None)
            flops_per_bounce = flops / max(cnt["bounces"], 1)
            # the oracle's frames double as the reference image of the metric's RMSE (after the timed region; uses the tracer)
            out["image_rmse_vs_ref"] = image_rmse(pt, cam, cpu_spp, ref, None)
        if name == "fractals" and world == 1 and not options:
            # The same steps in the tolerance modes (arith_mode 1: marches of rays other than camera rays on the hardware
            # transcendentals; 2: every march), each with its measured distance from the oracle's exact frames.  A fractal's
            # surface is rough at every scale, so a sample whose march took the other arithmetic is in effect another sample of its
            # pixel: at the CPU sample's few frames the distance is sampling noise (it falls as 1 / sqrt(frames); the tests check
            # convergence and the absence of bias).  The headline number above is exact.
            out["tolerance_mode"] = {}
            for mode, label_ in ((1, "camera_rays_exact"), (2, "every_march")):
                pt.set_option("arith_mode", mode)
                pt.clear()
                pt.trace_n(cam, frame0, 1, spp)
                pt.sync()
                pt.clear()
                t0 = time.perf_counter()
                for i in range(steps):
                    pt.trace_n(cam, frame0 + (warmup + i) * spp, 1 + i * spp, spp)
                pt.sync()
                t_elapsed = time.perf_counter() - t0
                out["tolerance_mode"][label_] = {
                    "option": f"arith_mode={mode}", "value": round(w * h * spp * steps / t_elapsed / 1e6, 2), "unit": "Mpaths/s",
                    "ms_per_step": round(t_elapsed * 1e3 / steps, 3),
                    "image_rmse_vs_ref": image_rmse(pt, cam, CPU_FRAMES[name], ref, None) if cpu is not None else None}
            out["tolerance_mode"]["bounds"] = ("primary ids identical (mode 1) / <= 1 in 10 000 differing (mode 2); no bias; <= 1 % relative RMSE of the "
                                               "converged image (2048 spp); per-pixel two-sample test at 4 standard deviations: tests/test_gpu_tolerance.py")
            pt.set_option("arith_mode", 0)
            pt.kernel_time()
        avg_ms = kernel_ms / max(launches, 1)
        roof = {"bound": "valu", "unit": "TFLOP/s", "peak": PEAK_FP32_VALU_TFLOPS, "kernel": launch_info.split(" ")[0],
                "launch": launch_info, "kernel_avg_ms": round(avg_ms, 3), "launches": launches, "traffic": None}
        if flops_per_bounce is not None:
            flops_per_launch = flops_per_bounce * bounces / max(launches, 1)
            ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof.update({"achieved": round(ach, 4), "frac": round(ach / PEAK_FP32_VALU_TFLOPS, 5),
                         "flops_per_path_bounce": round(flops_per_bounce, 1)})
        else:
            roof.update({"achieved": None, "frac": None})
        # streaming floor of the same kernel: accumulation read + write + rgba8 + ids per pixel per launch
        stream_bytes = w * h * (16 + 16 + 4 + 4)
        roof["hbm_streaming"] = {"bytes_per_launch": stream_bytes, "achieved_gbs": round(stream_bytes / (avg_ms * 1e-3) / 1e9, 3),
                                 "peak_gbs": PEAK_HBM_GBS}
        bf = os.path.join(ROOT, "profiles", "bvh_fetch.json")
        if name == "all_mesh" and os.path.exists(bf):
            # The mesh workload is bound by dependent BVH record fetches, not by arithmetic: its roofline is the
            # algorithmic record bytes (64 B per node pair visited and per triangle tested, counted by a
            # diagnostic build on this same scene) over the kernel time, against HBM peak.
            b = json.load(open(bf))
            if b.get("mesh_quads") == args.mesh_quads:
                per_bounce = b["bytes_per_path_bounce"]
                ach = per_bounce * bounces / max(launches, 1) / (avg_ms * 1e-3) / 1e9
                roof = {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "achieved": round(ach, 2), "frac": round(ach / PEAK_HBM_GBS, 5),
                        "traffic": b.get("hbm_bytes_per_launch"),
                        "traffic_source": "profiles/bvh_fetch.json: rocprofv3 PMC passes of this command, committed with the source; not measured in this run",
                        "kernel": launch_info.split(" ")[0], "launch": launch_info, "kernel_avg_ms": round(avg_ms, 3),
                        "launches": launches, "bvh_bytes_per_path_bounce": per_bounce,
                        "valu": {k: roof.get(k) for k in ("achieved", "frac", "flops_per_path_bounce")}}
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                t = json.load(open(tf))
                t = t.get(name, t) if isinstance(t.get(name), dict) else t
                if t.get("workload") == name and t.get("spp_per_step") == spp and roof.get("traffic") is None and not options:
                    roof["traffic"] = t.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = "profiles/traffic.json: rocprofv3 PMC passes of this command, committed with the source; not measured in this run"
                    if t.get("valu_issue"):  # what the VALU counters of the committed profile say about this same launch
                        roof["valu_issue_profiled"] = {k: t["valu_issue"][k] for k in ("valu_busy_frac", "lane_utilisation")}
            except Exception:
                pass
        out["roofline"] = roof
        out["cpu_baseline"] = cpu
    pt.close()
    return out


def reference_picture_check(spp: int = 3272) -> dict:
    """The HIP kernels' 1080p frame of the scene that the reference's own picture shows (resources/cornell_box.png: today's
    cornell_box.yaml without its second light, ceiling light at strength 3; tests/test_reference_image.py has the derivation)
    against the picture's 8x8 block means (tests/golden/reference_cornell_box_blocks.npy), in 8-bit levels.  3272 samples from frame
    index 0: the picture's own count, where the two frames share their noise (tests/test_gpu_reference_image.py).  Not timed."""
    import numpy as np
    from pathtracer_amd import host
    from pathtracer_amd.device import PathTracer
    ref = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_blocks.npy")).astype(np.float64)
    d = np.load(os.path.join(ROOT, "data", "scenes", "cornell_box.npz"), allow_pickle=False)
    second_light, ceiling_light = 12, 5
    keep = [i for i in range(len(d["figs"])) if i != second_light]
    mats = d["mats"].copy()
    mats["emission_strength"][ceiling_light] = 3.0
    sc = host.Scene.empty()
    sc.append_shapes(d["figs"][keep], mats[keep])
    cam = d["camera"]
    sc.set_camera(cam[0:3], cam[3:6], float(cam[6]))
    sc.set_render_settings(max_bounces=int(cam[8]), fractal_march_steps=int(cam[15]), tone_mapper=int(cam[14]), firefly_clamp=float(cam[9]), exposure=float(cam[7]))
    sc.set_sky(cam[10:13], float(cam[13]))
    sc.build()
    w, h = 1920, 1080
    pt = PathTracer(0)
    sc.upload_to(pt)
    pt.resize(w, h)
    c = sc.camera(w, h)
    for first in range(0, spp, 128):
        pt.trace_n(c, first, first + 1, min(128, spp - first))
    img = pt.read_output().reshape(h, w, 4)[:, :, :3].astype(np.float64)
    pt.close()
    ours = img.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))
    diff = ours - ref
    worst = np.abs(diff).max(axis=2)
    return {"picture": "the reference repository's resources/cornell_box.png (its README's render), 8x8 block means", "samples_per_pixel": spp,
            "mean_abs_error_levels": round(float(np.abs(diff).mean()), 4), "bias_levels": round(float(diff.mean()), 4),
            "relative_rmse": round(float(np.sqrt((diff * diff).mean()) / ref.mean()), 5),
            "blocks_within_1_2_4_levels": [round(float((worst <= t).mean()), 4) for t in (1.0, 2.0, 4.0)], "worst_block_levels": round(float(worst.max()), 2),
            "blocks": int(worst.size)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=list(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=0)
    ap.add_argument("--mesh-quads", type=int, default=2236, help="all_mesh: quads per side (2236 -> 10.0 M triangles)")
    ap.add_argument("--partition", default="", choices=["", "samples", "tiles"],
                    help="N > 1: sample split (weak scaling, default) or interleaved 16x16 tiles (strong scaling, default for teapot)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="N = 1, default workload: skip the short legs of the other three workloads")
    ap.add_argument("--other-steps", type=int, default=3, help="timed steps of each of those legs")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1, sample split: skip the fixed-total-work leg")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="ptx_set_option setting for measurement runs (e.g. flights=1: launches one after the other), recorded in config.options")
    ap.add_argument("--exchange", default="library", choices=["library", "torch"],
                    help="N > 1: the multi-GPU epilogue by the library's own RCCL exchange (ptx_merge_sample_split / ptx_assemble_tiles, default) "
                         "or by the same steps over torch.distributed collectives")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))  # before any GPU call: the ranks are children of this process
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={os.environ.get('WORLD_SIZE')}: the two must agree", file=sys.stderr)
        sys.exit(2)

    D = Dist()
    out = run_workload(args.workload, args, D, args.steps, args.warmup, args.spp_per_step, with_cpu=not args.no_cpu_baseline,
                       strong_leg=not args.no_strong_leg)
    if D.world == 1 and args.workload == "cornell" and not args.no_other_workloads and not args.opt and not args.spp_per_step:
        # the other three workloads, briefly, with the same code as --workload <name>: their own roofline, CPU baseline and RMSE
        others = {}
        for name in ("teapot", "fractals", "all_mesh"):
            t0 = time.perf_counter()
            try:
                o = run_workload(name, args, D, args.other_steps, 1, 0, with_cpu=not args.no_cpu_baseline)
                keep = ("value", "unit", "steps", "warmup", "ms_per_step", "config", "gpath_bounce_per_s", "bounces_per_path", "ms_per_step_median",
                        "launches_overlap", "image_rmse_vs_ref", "roofline", "cpu_baseline", "tolerance_mode")
                others[name] = {k: o[k] for k in keep if k in o}
                others[name]["leg_wall_s"] = round(time.perf_counter() - t0, 1)
            except Exception as e:  # noqa: BLE001  a leg that fails must not lose the headline line
                others[name] = {"error": f"{type(e).__name__}: {e}"}
        out["other_workloads"] = others
        try:  # the one output of the reference's own pipeline that exists, against this library's render of the same scene
            out["reference_picture"] = reference_picture_check()
        except Exception as e:  # noqa: BLE001
            out["reference_picture"] = {"error": f"{type(e).__name__}: {e}"}
    if D.rank == 0:
        print(json.dumps(out), flush=True)
    D.close()


if __name__ == "__main__":
    main()
