"""Shared helpers for the test-suite."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE_DIR = os.path.join(ROOT, "data", "scenes")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"

# name -> (width, height, samples) of the committed golden renders
GOLDEN_CASES = {
    "cornell_box": (64, 36, 4),
    "teapot": (64, 36, 2),
    "fractals": (48, 27, 2),
    "all": (64, 36, 3),
    "planet": (64, 36, 3),
}


def fixture(name: str) -> str:
    return os.path.join(SCENE_DIR, f"{name}.npz")


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a).view(np.uint32)


def same_bits(a: np.ndarray, b: np.ndarray) -> int:
    """Number of f32 elements whose bit patterns differ (NaN == NaN regardless of payload)."""
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    diff = (bits(a) != bits(b)) & ~(np.isnan(a) & np.isnan(b))
    return int(diff.sum())


def have_gpu() -> bool:
    try:
        from pathtracer_amd.device import PathTracer
        pt = PathTracer(0)
        pt.close()
        return True
    except Exception:
        return False


def grid_triangle_boxes(quads: int, seed: int = 3) -> np.ndarray:
    """Boxes of the 2*quads^2 triangles of a height-field grid over [-1, 1]^2 (flat ones padded like aabb.rs:69-81)."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-1.0, 1.0, quads + 1, dtype=np.float32)
    y = (rng.random((quads + 1, quads + 1)) * 0.1).astype(np.float32)
    x, z = np.meshgrid(g, g, indexing="ij")
    p = np.stack([x, y, z], axis=-1)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[:-1, 1:], p[1:, 1:]
    tris = np.concatenate([np.stack([a, b, c], axis=2).reshape(-1, 3, 3), np.stack([b, d, c], axis=2).reshape(-1, 3, 3)])
    mn, mx = tris.min(axis=1), tris.max(axis=1)
    flat = (mx - mn) < np.float32(1e-4)
    mn = np.where(flat, mn - np.float32(1e-4), mn)
    mx = np.where(flat, mx + np.float32(1e-4), mx)
    return np.concatenate([mn, mx], axis=1).astype(np.float32)


def random_boxes(n: int, seed: int, spread: float = 10.0, size: float = 0.5) -> np.ndarray:
    rng = np.random.default_rng(seed)
    c = (rng.random((n, 3)) * 2 - 1).astype(np.float32) * np.float32(spread)
    e = (rng.random((n, 3)).astype(np.float32) * np.float32(size))
    return np.concatenate([c - e, c + e], axis=1).astype(np.float32)


def box_sets() -> dict:
    """Named box lists (n x {min xyz, max xyz}) for the BVH builders, small enough for the numpy restatement."""
    out = {
        "empty": np.zeros((0, 6), dtype=np.float32),
        "one": random_boxes(1, 1),
        "leaf4": random_boxes(4, 2),
        "five": random_boxes(5, 3),
        "random37": random_boxes(37, 4),
        "random300": random_boxes(300, 5),
        "coincident40": np.repeat(random_boxes(1, 6), 40, axis=0),
        "grid_tris512": grid_triangle_boxes(16),
    }
    flat = random_boxes(200, 7)
    flat[:, 1], flat[:, 4] = np.float32(0.25), np.float32(0.25)   # no extent along y
    out["flat_y200"] = flat
    same = random_boxes(9, 8)
    c = (same[:, :3] + same[:, 3:]) * np.float32(0.5)
    e = (same[:, 3:] - same[:, :3]) * np.float32(0.5)
    out["same_centroid9"] = np.concatenate([-e, e], axis=1).astype(np.float32) + 0 * c[:, :1]  # nested boxes around the origin
    return out
