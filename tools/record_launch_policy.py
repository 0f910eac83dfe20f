"""Records what the launch policy chooses (ptx_last_launch_info) for every scene fixture and launch length, on a GPU:
tests/golden/launch_policy.json.  The policy test compares the library's choices with this file, so that a change of
the policy's code that was not meant to change a choice shows up.  Run on the GPU box; copy the result from gpurun_out/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

LENGTHS = (1, 2, 3, 4, 5, 16, 63, 64, 128)
SIZES = ((64, 48), (1920, 1080), (3840, 2160))  # the largest: short launches whose job count is beyond what pools on a mesh take


def describe(pt, wait_mode=None):
    return pt.launch_info().split(";")[0]


def main(out_path):
    pt = PathTracer(0)
    scenes = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "data", "scenes")) if f.endswith(".npz"))
    rec = {}
    for name in scenes:
        sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", name + ".npz")).build(decode_textures=False)
        pt.reset_options()
        pt.set_option("host_waits_per_frame", 0)
        sc.upload_to(pt)
        per = {}
        for (w, h) in SIZES:
            pt.resize(w, h)
            cam = sc.camera(w, h)
            cam[0]["max_bounces"] = 2  # the policy does not look at it; keeps the recording short
            for n in LENGTHS:
                if (w > 64 and n > 5) or (w > 1920 and n > 2):
                    continue
                pt.trace_n(cam, 0, 1, n)
                pt.sync()
                per[f"{w}x{h}:{n}"] = describe(pt)
        # a host that waits for every frame: one-sample launches with a read-back between them
        pt.set_option("host_waits_per_frame", -1)
        pt.resize(64, 48)
        cam = sc.camera(64, 48)
        cam[0]["max_bounces"] = 2
        import time
        for k in range(4):
            pt.trace_n(cam, k, k + 1, 1)
            pt.read_output()
            time.sleep(0.002)
        per["64x48:1:host_waits"] = describe(pt)
        rec[name] = per
        print(name, per["64x48:1"], "|", per["64x48:16"], flush=True)
    pt.close()
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "launch_policy.json"))
