"""Host logic of the initialiser (no device): the schedule of the in-place optReg sweep, ldso_init_sweep_schedule (include/ldso_hip.h).

CoarseInitializer::optReg (CoarseInitializer.cc:430-459) updates iR of the points in index order from the median of the neighbours' iR, reading neighbours that
may already have been updated.  The device executes passes of points side by side; a schedule reproduces the loop exactly iff every neighbour j < i of a point i
sits in an earlier pass and every neighbour j > i in the same or a later one.  Checked here on the k-d tree graphs of the synthetic frames (the same construction as
makeNN, :717-783) and on random graphs, together with what makes it worth having: never longer than first fit in index order, never shorter than the dependency
depth or than the points would need at the given width, and the sweep emulated pass by pass on the CPU equal to the sequential loop bit for bit."""
import ctypes as C
import numpy as np
import pytest

from ldso_amd import binding, synth


def _lib():
    L = C.CDLL(binding.lib_path())
    L.ldso_init_sweep_schedule.restype = C.c_int
    L.ldso_init_sweep_schedule.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return L


def _schedule(nb, width):
    nb = np.ascontiguousarray(nb, dtype=np.int32)
    out = np.zeros(len(nb), np.int32)
    n_pass = _lib().ldso_init_sweep_schedule(len(nb), nb.ctypes.data, width, out.ctypes.data)
    assert n_pass >= 0, n_pass
    return n_pass, out


def _depth_and_first_fit(nb, width):
    n = len(nb)
    dep = np.zeros(n, int); rd = np.zeros(n, int); ff = np.zeros(n, int); rdf = np.zeros(n, int); fill = {}
    for i in range(n):
        d, e = rd[i], rdf[i]
        for j in nb[i]:
            if 0 <= j < i:
                d = max(d, dep[j] + 1); e = max(e, ff[j] + 1)
        while fill.get(e, 0) >= width:
            e += 1
        fill[e] = fill.get(e, 0) + 1
        dep[i], ff[i] = d, e
        for j in nb[i]:
            if j > i:
                rd[j] = max(rd[j], d); rdf[j] = max(rdf[j], e)
    return (int(dep.max()) + 1 if n else 0), (int(ff.max()) + 1 if n else 0)


def _check(nb, width, n_pass, pas):
    n = len(nb)
    assert pas.min(initial=0) >= 0 and pas.max(initial=-1) < max(n_pass, 1)
    assert np.bincount(pas, minlength=max(n_pass, 1)).max(initial=0) <= width
    if n:
        assert len(np.unique(pas)) == n_pass                # no empty pass
    for i in range(n):
        for j in nb[i]:
            if j < 0 or j == i:
                continue
            assert (pas[j] < pas[i]) if j < i else (pas[j] >= pas[i]), (i, j, pas[i], pas[j])


def _sweep_sequential(nb, ir, idepth, good):
    ir = ir.copy()
    for i in range(len(nb)):
        if not good[i]:
            continue
        v = [ir[j] for j in nb[i] if j >= 0 and good[j]]
        if len(v) > 2:
            ir[i] = np.float32(0.2) * idepth[i] + np.float32(0.8) * np.float32(sorted(v)[len(v) // 2])
    return ir


def _sweep_by_passes(nb, ir, idepth, good, n_pass, pas):
    ir = ir.copy()
    for p in range(n_pass):
        idx = np.nonzero(pas == p)[0]
        new = {}
        for i in idx:                                       # every read of the pass before any write of it
            if not good[i]:
                continue
            v = [ir[j] for j in nb[i] if j >= 0 and good[j]]
            if len(v) > 2:
                new[i] = np.float32(0.2) * idepth[i] + np.float32(0.8) * np.float32(sorted(v)[len(v) // 2])
        for i, x in new.items():
            ir[i] = x
    return ir


def test_schedule_of_the_synthetic_frames():
    seq = synth.make_init_sequence(160, 120, n_frames=1, fx=100.0, seed=11, levels=3)
    pts = synth.select_init_points(synth.make_images(seq["first"], seq["levels"]))
    rng = np.random.default_rng(5)
    for p in pts:
        nb = p["neighbours"]
        n = len(nb)
        n_pass, pas = _schedule(nb, 32)
        _check(nb, 32, n_pass, pas)
        depth, first_fit = _depth_and_first_fit(nb, 32)
        assert max(depth, -(-n // 32)) <= n_pass <= first_fit
        ir = rng.uniform(0.2, 2.0, n).astype(np.float32); idepth = rng.uniform(0.2, 2.0, n).astype(np.float32); good = rng.random(n) < 0.85
        a, b = _sweep_sequential(nb, ir, idepth, good), _sweep_by_passes(nb, ir, idepth, good, n_pass, pas)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("n,width,seed", [(0, 32, 0), (1, 32, 0), (7, 2, 1), (500, 32, 2), (500, 4, 3), (3000, 32, 4)])
def test_schedule_of_random_graphs(n, width, seed):
    rng = np.random.default_rng(seed)
    nb = np.full((n, 10), -1, np.int32)
    for i in range(n):                                      # mostly near in index, some far, some absent, now and then the point itself or a repeated one
        k = int(rng.integers(0, 11))
        c = np.clip(i + rng.integers(-40, 41, k), 0, n - 1)
        far = rng.random(k) < 0.1
        c[far] = rng.integers(0, n, int(far.sum()))
        nb[i, :k] = c
    n_pass, pas = _schedule(nb, width)
    _check(nb, width, n_pass, pas)
    depth, first_fit = _depth_and_first_fit(nb, width)
    assert max(depth, -(-n // width) if n else 0) <= n_pass <= first_fit
    ir = rng.uniform(0.2, 2.0, n).astype(np.float32); idepth = rng.uniform(0.2, 2.0, n).astype(np.float32); good = rng.random(n) < 0.8
    a, b = _sweep_sequential(nb, ir, idepth, good), _sweep_by_passes(nb, ir, idepth, good, n_pass, pas)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_schedule_rejects_bad_arguments():
    nb = np.full((3, 10), -1, np.int32); nb[0, 0] = 3
    out = np.zeros(3, np.int32)
    assert _lib().ldso_init_sweep_schedule(3, nb.ctypes.data, 32, out.ctypes.data) < 0
    nb[0, 0] = 1
    assert _lib().ldso_init_sweep_schedule(3, nb.ctypes.data, 0, out.ctypes.data) < 0
    assert _lib().ldso_init_sweep_schedule(3, None, 32, out.ctypes.data) < 0
