import importlib
import os
import sys

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rta():
    """the product package (directory name has a hyphen, so it is imported by name)"""
    return importlib.import_module("ray-tracing_amd")


@pytest.fixture(scope="session")
def scenes(rta):
    return importlib.import_module("ray-tracing_amd.scenes")


@pytest.fixture(scope="session")
def ob():
    """oracle bindings (CPU restatement + real reference where built)"""
    from oracle import bindings
    return bindings


@pytest.fixture(scope="session")
def gpu(rta):
    r = rta.Renderer(0)
    yield r
    r.close()


def _converged_crop_check(gpu, rta, ob, orc, cam, flags, W, rect, spp, label, gpu_flags=0):
    """Independent seeds at a high sample count on a crop of the frame: the GPU image (seed 10) against the oracle's (seed
    77), with the oracle-vs-oracle difference of seeds 77 and 78 as the noise floor. The two images share nothing but the
    scene and the estimator, so what exceeds the floor is bias. Stated bars: |mean signed difference| < 0.25 grey level of
    255, RMSE <= 1.15 x floor."""
    j0, i0, j1, i1 = rect
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, spp, seed=10, flags=flags | gpu_flags, rect=rect))  # gpu_flags: RT_F_FAST for the single-precision kernel
    rows = slice(W - i1, W - i0)  # the tone map flips the rows (raytracer.cpp:1499-1501)
    g = gpu.resolve(W, W, spp)[rows, j0:j1].astype(np.float64)
    o1 = ob.resolve(orc.render(cam, rta.make_params(W, W, spp, seed=77, flags=flags, rect=rect)), spp)[rows, j0:j1].astype(np.float64)
    o2 = ob.resolve(orc.render(cam, rta.make_params(W, W, spp, seed=78, flags=flags, rect=rect)), spp)[rows, j0:j1].astype(np.float64)
    assert g.shape == (i1 - i0, j1 - j0, 3) and g.max() > 0 and o1.max() > 0
    floor = np.sqrt(((o1 - o2) ** 2).mean())
    rmse = np.sqrt(((g - o1) ** 2).mean())
    bias = (g - o1).mean()
    print("converged crop %s @%d spp: RMSE %.3f levels, noise floor %.3f (ratio %.4f), mean signed difference %.4f" % (label, spp, rmse, floor, rmse / floor, bias))
    assert abs(bias) < 0.25, bias
    assert rmse <= 1.15 * floor, (rmse, floor)
    return rmse, floor, bias


@pytest.fixture(scope="session")
def converged_crop_check():
    return _converged_crop_check
