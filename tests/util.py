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
