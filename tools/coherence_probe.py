"""What would perfectly sorted rays buy?  (GPU box; run under `rocprofv3 --kernel-trace --stats` to get the durations.)

Secondary rays of the Cornell box — origins on the surfaces the camera sees and on the surfaces those rays hit, directions
uniform over the hemisphere facing away from the incoming ray — are traced with the library's closest-hit query hook
(`ptx_kat_trace`, the lockstep traversal with the scene in global memory) in four orders:
  0 random                       what a wave holds after the first bounce today
  1 sorted by traversal key      the key of the regrouping experiment: which of the three leaf boxes the ray's box test hits,
                                 and in which order (computed here in numpy float32)
  2 sorted by key, then by the figure the ray starts on
  3 sorted by the figure alone   (a key that costs nothing: it is known at the end of the bounce)
Each order is one dispatch of the same kernel over the same rays; the ratio of the durations bounds what ANY regrouping of
paths can gain in the closest-hit query (45 % of the kernel's time, the occlusion query another 24 %)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pathtracer_amd import host  # noqa: E402
from pathtracer_amd.device import PathTracer  # noqa: E402

f32 = np.float32
rng = np.random.default_rng(7)
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_box"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
sc = host.Scene.from_fixture(os.path.join(ROOT, "data", "scenes", f"{name}.npz")).build(decode_textures=False)
pt = PathTracer(0)
sc.upload_to(pt)
W, H = 1920, 1080
cam = sc.camera(W, H)
c = cam[0]
# primary rays through random pixel positions (camera.wgsl:6-19 in float32, jitter included in the random position)
px, py = rng.random(N, dtype=f32) * f32(W), rng.random(N, dtype=f32) * f32(H)
ndc_x = (f32(2) * px / f32(W) - f32(1)) * f32(c["aspect"])
ndc_y = f32(1) - f32(2) * py / f32(H)
d = (np.outer(ndc_x, c["right"]) + np.outer(ndc_y, c["up"]) + np.asarray(c["forward"], dtype=f32) * f32(c["focal_length"])).astype(f32)
d /= np.linalg.norm(d, axis=1, keepdims=True).astype(f32)
o = np.broadcast_to(np.asarray(c["position"], dtype=f32), d.shape)


def bounce(o, d):
    hid, t, _ = pt.kat_trace(cam, np.concatenate([o, d], axis=1))
    hit = hid != 0xFFFFFFFF
    p = (o + d * t[:, None]).astype(f32)
    nd = rng.normal(size=d.shape).astype(f32)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True).astype(f32)
    back = np.sum(nd * d, axis=1) > 0  # leave on the side the ray came from
    nd[back] = -nd[back]
    return (p + nd * f32(2e-3))[hit], nd[hit], hid[hit]


o1, d1, fig1 = bounce(o, d)
o2, d2, fig2 = bounce(o1, d1)
o_all, d_all, fig = np.concatenate([o1, o2]), np.concatenate([d1, d2]), np.concatenate([fig1, fig2])
n = len(o_all)
# the key boxes: the leaves of the tree (children of the root's inner child, and the root's leaf child)
nodes = sc.buffers()["nodes"]
leaves = [k for k in range(len(nodes)) if nodes[k]["prim_count"] > 0][:4]
with np.errstate(divide="ignore", invalid="ignore"):
    inv = (f32(1) / d_all).astype(f32)
    ts = []
    for k in leaves:
        t1 = (nodes[k]["aabb_min"] - o_all) * inv
        t2 = (nodes[k]["aabb_max"] - o_all) * inv
        tn, tf = np.minimum(t1, t2).max(axis=1), np.maximum(t1, t2).min(axis=1)
        ts.append(np.where((tn > tf) | (tf < 0), f32(np.inf), np.maximum(tn, f32(0))))
ts = np.stack(ts, axis=1)
key = np.zeros(n, dtype=np.int64)
for i in range(ts.shape[1]):
    key |= (ts[:, i] < np.inf).astype(np.int64) << (3 + i)
rank = np.argsort(np.argsort(ts, axis=1), axis=1)  # the order the boxes would be entered in
key = key * 64 + rank[:, 0] * 16 + rank[:, 1] * 4 + (rank[:, 2] if ts.shape[1] > 2 else 0)
octant = (d_all[:, 0] > 0).astype(np.int64) | ((d_all[:, 1] > 0).astype(np.int64) << 1) | ((d_all[:, 2] > 0).astype(np.int64) << 2)
mask_only = key // 64
orders = [rng.permutation(n), np.argsort(key, kind="stable"), np.lexsort((fig, key)), np.argsort(fig, kind="stable"),
          np.argsort(mask_only, kind="stable"), np.lexsort((octant, key)), np.lexsort((octant, fig, key)), np.lexsort((key, fig))]
print("orders: 0 random, 1 key, 2 key+figure, 3 figure, 4 hit mask only, 5 key+octant, 6 key+figure+octant, 7 figure+key", flush=True)
print(f"{name}: {n} secondary rays, {len(np.unique(key))} keys, {len(leaves)} key boxes; figures started from: {np.bincount(fig.astype(np.int64))}", flush=True)
rays = np.concatenate([o_all, d_all], axis=1)
ref = None
for i, order in enumerate(orders):
    hid, t, _ = pt.kat_trace(cam, rays[order])
    back = np.empty_like(hid)
    back[order] = hid
    ref = back if ref is None else ref
    print(f"order {i}: dispatched; same hits as order 0: {bool((back == ref).all())}", flush=True)

# ---- the occlusion query: rays from the same surface points to points on the scene's lights (next-event estimation), closest-hit
# order of the rays that made the points against orders of their own
b = sc.buffers()
lights = b["lights"]
if len(lights):
    figs = b["figs"]
    li = rng.integers(0, len(lights), size=n)
    lp = figs["position"][lights[li]].astype(f32)
    lr = figs["radius"][lights[li]].astype(f32)
    u = rng.normal(size=(n, 3)).astype(f32)
    u /= np.linalg.norm(u, axis=1, keepdims=True).astype(f32)
    target = lp + u * lr[:, None]
    to = target - o_all
    dist = np.linalg.norm(to, axis=1).astype(f32)
    sd = (to / dist[:, None]).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (f32(1) / sd).astype(f32)
        ts2 = []
        for k in leaves:
            t1 = (nodes[k]["aabb_min"] - o_all) * inv
            t2 = (nodes[k]["aabb_max"] - o_all) * inv
            tn, tf = np.minimum(t1, t2).max(axis=1), np.maximum(t1, t2).min(axis=1)
            ts2.append(np.where((tn > tf) | (tf < 0) | (np.maximum(tn, 0) >= dist), f32(np.inf), np.maximum(tn, f32(0))))
    ts2 = np.stack(ts2, axis=1)
    skey = np.zeros(n, dtype=np.int64)
    for i in range(ts2.shape[1]):
        skey |= (ts2[:, i] < np.inf).astype(np.int64) << i
    sorders = [orders[0], orders[1], np.argsort(skey, kind="stable"), np.lexsort((li, skey)), np.lexsort((skey, li)), np.argsort(li, kind="stable")]
    print("occlusion orders: 0 random, 1 the closest-hit key of the ray that starts at the same point, 2 occlusion key (boxes entered before the light), "
          "3 occlusion key + light, 4 light + occlusion key, 5 light", flush=True)
    srays = np.concatenate([o_all, sd], axis=1)
    for i, order in enumerate(sorders):
        pt.kat_trace(cam, srays[order], dist[order])
        print(f"occlusion order {i}: dispatched", flush=True)
