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


def observe(name, value, limit):
    """assert value <= limit, and leave the OBSERVED value behind (gpurun_out/observed_tolerances.jsonl): the limits of the multi-iteration / sharded
    comparisons are set from what the hardware actually produces (x 2), not from a guess"""
    import json
    v = float(value)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "observed_tolerances.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "observed": v, "limit": float(limit)}) + "\n")
    except OSError:
        pass
    assert v <= limit, (name, v, limit)


def gauge_projected_rel(frames, state_a, state_b):
    """max |d - Q Q^T d| / max |state_b| with d = state_a - state_b over the 8 optimised parameters per frame and Q an orthonormal basis of the 7 gauge
    directions (6 pose + scale, FullSystem::getNullspaces, FullSystem.cc:1711-1760): the reduced system is ill-conditioned along the gauge, two
    correct solvers (or two shardings of the fp32 partial sums) drift apart along it and agree off it"""
    F = len(frames)
    N = np.zeros((8 * F, 7))
    for f in range(F):
        P = np.asarray(frames["nullspaces_pose"][f]).reshape(6, 6)
        for i in range(6):
            N[8 * f:8 * f + 6, i] = P[:, i]
        N[8 * f:8 * f + 6, 6] = frames["nullspaces_scale"][f]
        N[8 * f:8 * f + 3, :] *= 2.0                                                     # SCALE_XI_TRANS_INVERSE
    Q, _ = np.linalg.qr(N)
    xa, xb = np.asarray(state_a)[:, :8].reshape(-1), np.asarray(state_b)[:, :8].reshape(-1)
    d = xa - xb
    return float(np.abs(d - Q @ (Q.T @ d)).max() / max(np.abs(xb).max(), 1e-300))
