"""pytest wiring: markers, import paths, shared helpers.

`-m "not gpu"` runs on a CPU-only container (oracle vs golden vectors, host logic, C-ABI symbol
checks, gloo data-parallel tests); `-m gpu` needs one MI355X and goes through libvlg_hip.so.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-layout-generation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X) and the built libvlg_hip.so")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")


def assert_close(got, want, rtol=1e-4, atol=1e-5, what=""):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, "%s shape %s vs %s" % (what, tuple(got.shape), tuple(want.shape))
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError("%s: %d/%d outside tol (rtol=%g atol=%g); worst |err|=%.3e at flat index %d "
                             "(got %.8g want %.8g)" % (what, int(bad.sum()), bad.numel(), rtol, atol,
                                                       float(err.flatten()[i]), i, float(got.flatten()[i]),
                                                       float(want.flatten()[i])))
