"""CPU tier for the BVH builders: the host library's pth_bvh_build against the numpy restatement of
src/accel/bvh.rs:37-213, and the closed form of the in-place partition (bvh.rs:174-186) that the device
builder evaluates with two prefix sums against the sequential loop."""
import numpy as np
import pytest

from oracle import host_prep as hp
from pathtracer_amd import host
from util import box_sets


def _oracle(aabbs):
    return hp.bvh_build([(b[:3].copy(), b[3:].copy()) for b in aabbs])


@pytest.mark.parametrize("name", ["empty", "one", "leaf4", "five", "random37", "random300", "coincident40", "flat_y200",
                                  "same_centroid9", "grid_tris512"])
def test_host_builder_equals_oracle(name):
    aabbs = box_sets()[name]
    nodes, prims = host.bvh_build(aabbs)
    onodes, oprims = _oracle(aabbs)
    assert nodes.tobytes() == onodes.tobytes()
    assert (prims == oprims).all()


def sequential_partition(passing):
    """bvh.rs:174-186 on positions: returns (final order of the original positions, lo)."""
    a = list(range(len(passing)))
    lo, hi = 0, len(a)
    while lo < hi:
        if passing[a[lo]]:
            lo += 1
        else:
            hi -= 1
            a[lo], a[hi] = a[hi], a[lo]
    return a, lo


def closed_form_partition(passing):
    """Destination of every position, from prefix sums only (the device builder's move_kernel)."""
    p = np.asarray(passing, dtype=np.int64)
    n = len(p)
    P = np.concatenate([[0], np.cumsum(p)])
    m = int(P[n])
    fail_left = [i for i in range(m) if not p[i]]                  # ascending
    pass_right = [i for i in range(n - 1, m - 1, -1) if p[i]]      # descending
    K = len(fail_left)
    assert K == len(pass_right)
    dest = np.arange(n)
    for i in range(n):
        if i < m:
            if not p[i]:
                k = i - P[i]
                dest[i] = (n if k == 0 else pass_right[k - 1]) - 1
        elif p[i]:
            dest[i] = fail_left[P[n] - P[i + 1]]
        elif i == m:
            dest[i] = (pass_right[K - 1] if K > 0 else n) - 1
        else:
            dest[i] = i - 1
    out = np.empty(n, dtype=np.int64)
    out[dest] = np.arange(n)
    return list(out), m


def test_partition_closed_form_equals_the_sequential_loop():
    rng = np.random.default_rng(7)
    cases = [[], [0], [1], [0, 0, 0, 0], [1, 1, 1], [0, 1], [1, 0], [0, 0, 1, 1], [1, 1, 0, 0], [0, 1, 0, 1, 0, 1, 1, 0, 0]]
    for n in range(1, 12):   # every pattern of up to 11 elements
        for bits in range(1 << n):
            cases.append([(bits >> k) & 1 for k in range(n)])
    for _ in range(300):
        n = int(rng.integers(1, 400))
        cases.append(list((rng.random(n) < rng.random()).astype(int)))
    for c in cases:
        want, lo = sequential_partition(c)
        got, m = closed_form_partition(c)
        assert lo == m and got == want, c
