import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _device_present():
    """Is there a GPU on this box?  Decided WITHOUT the product library, so that a GPU box with a broken or missing
    libldso_hip.so fails the gpu tests instead of skipping them."""
    try:
        import torch
        if torch.cuda.is_available():
            return True
    except Exception:
        pass
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def _gpu_available():
    return _device_present()


@pytest.fixture(scope="session")
def have_gpu():
    return _gpu_available()


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are skipped only on a box without any GPU; on a GPU box a missing / broken HIP library makes
    # them FAIL (binding.lib() raises), never pass or skip silently.
    if _device_present():
        return
    skip = pytest.mark.skip(reason="no GPU on this box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


_WIN_CACHE = {}


def get_window(name, **kw):
    from ldso_amd import synth
    key = (name, tuple(sorted(kw.items())))
    if key not in _WIN_CACHE:
        _WIN_CACHE[key] = synth.make_config(name, **kw)
    return _WIN_CACHE[key]


@pytest.fixture(scope="session")
def tiny():
    return get_window("tiny")


@pytest.fixture(scope="session")
def small():
    return get_window("small")


def rel(a, b, floor=0.0):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor, 1e-300))


def blockrel(A, B, bs=4):
    """max over bs x bs blocks of |dA|_max / |B_block|_max — the Hessian-entry tolerance of DESIGN.md."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    worst = 0.0
    for i in range(0, A.shape[0], bs):
        for j in range(0, A.shape[1], bs):
            m = np.abs(B[i:i + bs, j:j + bs]).max()
            d = np.abs(A[i:i + bs, j:j + bs] - B[i:i + bs, j:j + bs]).max()
            if m > 0:
                worst = max(worst, d / m)
            elif d > 0:
                worst = max(worst, np.inf)
    return worst
