"""GPU tier: the tolerance modes of the fractal distance estimators (ptx_set_option "arith_mode").  In mode 1 rays other than
camera rays evaluate the Mandelbulb / Julia estimators with the hardware transcendentals and the camera ray's march stays
exact; in mode 2 every march does.  A fractal's surface is rough at every scale, so a hit point that moves by a rounding error
gets a different normal and its path goes elsewhere: a sample whose march took the other arithmetic is, in effect, another
sample of the same pixel.  The modes are therefore checked as estimators of the same image, against the ORACLE (exact
arithmetic, same seeds, same sample count):
  * primary-hit ids: identical in mode 1, every pixel; mode 2: at most 1 pixel in 10 000 (silhouette pixels);
  * per pixel, the two-sample test: both renders keep second moments (M2: the oracle's and the device's), the difference of two
    means of n independent samples has variance (M2_exact + M2_tol) / ((n - 1) n), and |mean_tol - mean_exact| may exceed four
    standard deviations of it (plus 1e-6 of the mean luminance, for pixels whose samples are all alike in both renders) for at
    most 1 pixel in 1000.  (Radiance is heavy-tailed -- a pixel whose 48 exact samples are all dark while one of the other 48
    found the light -- so a bound from the exact render's variance alone, 4 sigma / sqrt(n), fails for 2-7 pixels in 1000 here.)
  * no bias: the frame's mean luminance within 0.5 %;
  * converged image (2048 samples per pixel): RMSE of the luminance of the linear mean <= 1 % of the mean luminance.
The default mode (arith_mode 0) stays bit-exact: every other test of the suite runs in it."""
import numpy as np
import pytest

from oracle import host_prep as hp, oracle as orc
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
from util import fixture, same_bits

pytestmark = pytest.mark.gpu
LUM = np.array([0.2126, 0.7152, 0.0722])


@pytest.fixture(scope="module")
def pt():
    t = PathTracer(0)
    yield t
    t.reset_options()
    t.close()


def _render(pt, sc, cam, w, h, n, mode):
    pt.reset_options()
    pt.set_option("arith_mode", mode)
    sc.upload_to(pt)
    pt.enable_moments(True)
    pt.resize(w, h)
    pt.trace_n(cam, 0, 1, n)
    out = pt.read_accum(), pt.read_primary_ids(), pt.read_moments()
    pt.enable_moments(False)
    pt.reset_options()
    return out


@pytest.mark.parametrize("name", ["fractals", "mandelbulb", "julia"])
def test_tolerance_modes_are_unbiased_estimators_of_the_exact_image(pt, name):
    sc = host.Scene.from_fixture(fixture(name)).build()
    sb = hp.prepare_fixture(fixture(name))
    w, h, n = 160, 90, 48
    cam = sc.camera(w, h)
    ref = orc.render(orc.OracleScene(sb), cam, n, want_m2=True, threads=16)
    exact, ids0, _ = _render(pt, sc, cam, w, h, n, 0)
    assert same_bits(exact, ref["accum"]) == 0 and (ids0 == ref["primary_id"]).all()  # the default mode
    r = ref["accum"][..., :3].astype(np.float64)
    mean_lum = float((r @ LUM).mean())
    for mode in (1, 2):
        a, ids, m2 = _render(pt, sc, cam, w, h, n, mode)
        bound = 4.0 * np.sqrt((ref["m2"].astype(np.float64) + m2.astype(np.float64)) / ((n - 1) * n)) + 1e-6 * mean_lum
        assert same_bits(a, ref["accum"]) > 0, "the tolerance mode did not run"
        differing = float((ids != ref["primary_id"]).mean())
        assert differing == 0.0 if mode == 1 else differing <= 1e-4, (mode, differing)
        g = a[..., :3].astype(np.float64)
        over = (np.abs(g - r) > bound).any(axis=-1)
        assert over.mean() <= 1e-3, (mode, int(over.sum()))
        assert abs(float((g @ LUM).mean()) / mean_lum - 1.0) <= 5e-3, mode


@pytest.mark.parametrize("name", ["fractals", "mandelbulb", "julia"])
def test_tolerance_modes_converge_to_the_exact_image(pt, name):
    sc = host.Scene.from_fixture(fixture(name)).build()
    sb = hp.prepare_fixture(fixture(name))
    w, h, n = 96, 54, 2048
    cam = sc.camera(w, h)
    ref = orc.render(orc.OracleScene(sb), cam, n, threads=16)
    r = ref["accum"][..., :3].astype(np.float64)
    mean_lum = float((r @ LUM).mean())
    for mode in (1, 2):
        a = _render(pt, sc, cam, w, h, n, mode)[0]
        g = a[..., :3].astype(np.float64)
        rel_rmse = float(np.sqrt((((g - r) @ LUM) ** 2).mean())) / mean_lum
        assert rel_rmse <= 0.01, (mode, rel_rmse)


def test_tolerance_mode_leaves_scenes_without_fractals_alone(pt):
    sc = host.Scene.from_fixture(fixture("cornell_box")).build()
    w, h, n = 64, 48, 6
    cam = sc.camera(w, h)
    a = _render(pt, sc, cam, w, h, n, 0)[0]
    b = _render(pt, sc, cam, w, h, n, 2)[0]
    assert same_bits(a, b) == 0
