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


def test_motion_hypotheses_match_the_oracle():
    """ldso_tr_motion_hypotheses (pure host function of the product library) against the oracle's restatement of FullSystem.cc:189-309,
    which tests/test_ref_pin.py pins to the reference's own trackNewCoarse: 83 tries, 1e-12 (matrix vs quaternion composition)."""
    from oracle import pyoracle as po
    from ldso_amd import synth
    L = binding.lib()
    rng = np.random.default_rng(4)
    P = [np.ascontiguousarray(synth.se3_exp(rng.normal(0, 1, 6) * [0.3, 0.3, 0.3, 0.1, 0.1, 0.1])[:3, :4]) for _ in range(3)]
    out = np.zeros((83, 3, 4)); n = C.c_int()
    assert L.ldso_tr_motion_hypotheses(P[0].ctypes.data_as(C.c_void_p), P[1].ctypes.data_as(C.c_void_p), P[2].ctypes.data_as(C.c_void_p), C.c_int(1),
                                       out.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
    ref = np.zeros((83, 3, 4))
    m = po.lib().orc_tr_motion_hypotheses(P[0].ctypes.data_as(C.c_void_p), P[1].ctypes.data_as(C.c_void_p), P[2].ctypes.data_as(C.c_void_p), C.c_int(1), ref.ctypes.data_as(C.c_void_p))
    assert n.value == m == 83 and np.abs(out - ref).max() < 1e-12
    assert L.ldso_tr_motion_hypotheses(P[0].ctypes.data_as(C.c_void_p), P[1].ctypes.data_as(C.c_void_p), P[2].ctypes.data_as(C.c_void_p), C.c_int(0),
                                       out.ctypes.data_as(C.c_void_p), C.byref(n)) == 0 and n.value == 1 and np.array_equal(out[0], np.eye(4)[:3])
