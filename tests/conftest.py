import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from ldso_amd import binding
        return os.path.exists(binding.lib_path()) and binding.lib().ldso_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _gpu_available()


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests must FAIL (not silently pass) on a GPU box without the HIP library; on a box without any
    # GPU they are skipped.
    if _gpu_available():
        return
    try:
        from ldso_amd import binding
        ndev = binding.lib().ldso_device_count() if os.path.exists(binding.lib_path()) else 0
    except Exception:
        ndev = 0
    if ndev == 0:
        skip = pytest.mark.skip(reason="no HIP device visible")
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
