"""Host logic of the hypothesis-batched tracking: ldso_tr_select_hypothesis replays the try loop of
FullSystem::trackNewCoarse (FullSystem.cc:319-356) on the residuals of tries that all ran to the end.  Pure host function of the
C-ABI library (no device work), checked here against a direct Python restatement of that loop on random residual tables."""
import ctypes as C

import numpy as np

from ldso_amd import binding


def _loop(lastres, ok, coarsest, last_rmse0, thr):
    achieved = np.full(5, np.nan)
    have, win, tries = False, -1, 0
    for i in range(len(ok)):
        lr = np.full(5, np.nan)
        good = bool(ok[i])
        for lvl in range(coarsest, -1, -1):              # trackNewestCoarse aborts on the first level that is 1.5x worse (CoarseTracker.cc:193-200)
            lr[lvl] = lastres[i, lvl]
            if lr[lvl] > 1.5 * achieved[lvl]:
                good = False
                break
        tries += 1
        if good and np.isfinite(np.float32(lr[0])) and not (lr[0] >= achieved[0]):
            win, have = i, True
        if have:
            for l in range(5):
                if not np.isfinite(np.float32(achieved[l])) or achieved[l] > lr[l]:
                    achieved[l] = lr[l]
        if have and achieved[0] < last_rmse0 * thr:
            break
    return win, tries, achieved


def test_select_hypothesis_matches_the_reference_loop():
    L = binding.lib()
    rng = np.random.default_rng(5)
    for case in range(300):
        n = int(rng.integers(0, 12))
        coarsest = int(rng.integers(0, 5))
        lr = rng.uniform(0.5, 5.0, (n, 5))
        lr[rng.uniform(size=(n, 5)) < 0.05] = np.nan
        ok = (rng.uniform(size=n) > 0.25).astype(np.int32)
        last = float(rng.choice([np.nan, 0.5, 2.0, 1e9]))
        lr = np.ascontiguousarray(lr); ok = np.ascontiguousarray(ok)
        best, tries = C.c_int(-7), C.c_int(-7)
        ach = np.zeros(5)
        rc = L.ldso_tr_select_hypothesis(C.c_int(n), C.c_int(coarsest), lr.ctypes.data_as(C.c_void_p), ok.ctypes.data_as(C.c_void_p),
                                         C.c_double(last), C.c_double(1.5), C.byref(best), C.byref(tries), ach.ctypes.data_as(C.c_void_p))
        assert rc == 0
        w, t, a = _loop(lr, ok, coarsest, last, 1.5)
        assert (best.value, tries.value) == (w, t), case
        assert np.array_equal(np.isnan(ach), np.isnan(a)) and np.array_equal(ach[~np.isnan(a)], a[~np.isnan(a)]), case


def test_select_hypothesis_rejects_bad_arguments():
    L = binding.lib()
    best = C.c_int()
    assert L.ldso_tr_select_hypothesis(C.c_int(2), C.c_int(7), None, None, C.c_double(1.0), C.c_double(1.5), C.byref(best), None, None) != 0
    assert L.ldso_tr_select_hypothesis(C.c_int(2), C.c_int(3), None, None, C.c_double(1.0), C.c_double(1.5), C.byref(best), None, None) != 0
