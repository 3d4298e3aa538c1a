"""The drop-in under the reference's THREADING (FullSystem.h:272-306): LDSO tracks on one thread (trackNewCoarse -> CoarseTracker::trackNewestCoarse,
FullSystem.cc:189-357, under trackMutex) while the mapping thread makes the previous frame a key frame (makeKeyFrame, FullSystem.cc:410-640: trace, activate,
optimize, marginalise, then setCoarseTrackingRef on the OTHER tracker under coarseTrackerSwapMutex, :508-514).  Both go through ONE ldso::GpuBackend: two tracker
handles + the BA handle + the tracer (distinct handles are independent, ldso_hip.h), the frames' device pyramids reference-counted between them
(adapter/ldso_gpu_adapter.cc: pyramidOf / releasePyramids - the mapper releases the pyramids of frames that left its window while a tracker may still hold one
as its reference or new frame).

Thread A replays eight key frames in makeKeyFrame's order through the backend; thread B, concurrently and until A is done, runs trackNewCoarse through the
same backend on two alternating CoarseTracker objects (the double buffer) whose frames are NOT in A's window - so every releasePyramids() of A meets pyramids
that only a tracker holds.  Expected: B's results are bit-for-bit those of the same calls made serially (same inputs, one wavefront order); A's key frames equal
the serial run's within the reproducibility of the fused BA path (fp64 atomics: 1e-12 per iteration)."""
import threading

import numpy as np
import pytest

from conftest import observe
from ldso_amd import synth
from oracle import pyref as pr

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (pr.available() and pr.adapter_available()), reason="oracle/_ref/libldso_ref.so / adapter/_build/libldso_adapter_test.so not built")]

K = 8


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _tracker_pair(sc):
    w = sc["win"]
    rts = [pr.RefTracker(w.w, w.h, sc["levels"], w.settings, w.calib) for _ in range(2)]
    for t in rts:
        t.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"]); t.set_new_frame(sc["new_pyr"], 1.0)
    return rts


def _track(A, rt, sc):
    w2c = sc["win"].truth["w2c"]; F = sc["win"].F
    r = A.track_new_coarse(rt, w2c[F - 2], w2c[F - 1], w2c[F - 1], sc["new_aff"], np.array([100.0] * 5))
    return np.concatenate([r["result"].ravel(), r["w2c"].ravel(), np.asarray(r["aff"], np.float64).ravel(), r["lastCoarseRMSE"].ravel()])


def test_tracking_thread_and_mapping_thread_share_one_backend():
    from adapter_sequence_common import run_sequence
    from tracker_common import tracker_scenario
    sc = tracker_scenario("small")
    win = synth.make_config("small", extra_frames=K)

    keep = pr.RefWindow(win)          # the reference's process-wide image size / calibration (internal/GlobalCalib.h) are set by the first object graph: the backend reads them
    # ---- serial: the key frames, then the tracks, one after the other on one backend ----
    A = pr.GpuAdapter(max_frames=8, max_points=4000)
    pr.set_device_marginalisation(True)
    try:
        r0, log_serial = run_sequence(win, K, adapter=A)
    finally:
        pr.set_device_marginalisation(False)
    # (a tracker object carries state from call to call - its FullSystem's lastCoarseRMSE, the new frame's pose - so the yardstick is the same SEQUENCE of calls on
    # a fresh pair of tracker objects, call by call)
    N_TR = 48
    rts = _tracker_pair(sc)
    serial_tracks = [_track(A, rts[i % 2], sc) for i in range(N_TR)]
    for t in rts:
        t.close()
    rts = _tracker_pair(sc)
    again = [_track(A, rts[i % 2], sc) for i in range(8)]
    for i, v in enumerate(again):
        assert np.array_equal(v, serial_tracks[i], equal_nan=True), "the same sequence of tracks on a fresh pair of tracker objects is the same numbers"
    for t in rts:
        t.close()
    A.close()

    # ---- concurrent: thread A = mapper (key frames), thread B = tracker (double-buffered trackNewCoarse), ONE backend ----
    A = pr.GpuAdapter(max_frames=8, max_points=4000)
    rts = _tracker_pair(sc)
    out = {"log": None, "tracks": [], "err": []}
    done = threading.Event()

    def mapper():
        try:
            pr.set_device_marginalisation(True)
            try:
                _, out["log"] = run_sequence(win, K, adapter=A)
            finally:
                pr.set_device_marginalisation(False)
        except BaseException as e:                      # noqa: B036 - reported by the main thread
            out["err"].append(("mapper", repr(e)))
        finally:
            done.set()

    def tracker():
        try:
            i = 0
            while (not done.is_set() or i < 8) and i < N_TR:
                out["tracks"].append(_track(A, rts[i % 2], sc))          # the double buffer: the two CoarseTracker objects alternate (FullSystem.cc:105-111)
                i += 1
            while not done.is_set():                                     # keep tracking beside the mapper (results beyond the serial sequence are not compared)
                _track(A, rts[i % 2], sc); i += 1
            out["calls"] = i
        except BaseException as e:                      # noqa: B036
            out["err"].append(("tracker", repr(e)))

    ta, tb = threading.Thread(target=mapper), threading.Thread(target=tracker)
    ta.start(); tb.start(); ta.join(600); tb.join(600)
    assert not ta.is_alive() and not tb.is_alive(), "a thread hangs"
    assert not out["err"], out["err"]
    log = out["log"]
    assert log is not None and len(log) == K
    n_tr = len(out["tracks"])
    print("two threads on one backend:", K, "key frames beside", out.get("calls", n_tr), "trackNewCoarse calls (", n_tr, "compared ); device pyramids built:", A.pyramids_built())
    assert n_tr >= 8, "the tracker thread ran beside the mapper"
    # the tracker side: bit for bit the serial sequence, call by call
    for i, v in enumerate(out["tracks"]):
        assert np.array_equal(v, serial_tracks[i], equal_nan=True), ("track", i, np.nanmax(np.abs(v - serial_tracks[i])))
    # the mapper side: the serial sequence within the fused path's run-to-run reproducibility
    worst = 0.0
    for a, b in zip(log, log_serial):
        sa, sb = a["summary"], b["summary"]
        assert not a["lost"] and not b["lost"]
        for k in ("candidates", "activated", "new_residuals", "points"):
            assert a[k] == b[k], (a["k"], k, a[k], b[k])
        assert sa["F"] == sb["F"] and np.array_equal(sa["ids"], sb["ids"])
        for k in ("points", "residuals", "immature", "host"):
            assert np.array_equal(sa[k], sb[k]), (a["k"], k)
        assert np.array_equal(sa["uv"], sb["uv"])
        worst = max(worst, abs(a["rmse"] - b["rmse"]) / b["rmse"], *[_rel(sa[k], sb[k]) for k in ("c2w", "aff", "idepth", "HM", "bM")])
    observe("two_threads_vs_serial_sequence_floats", worst, 1e-9)          # two runs of the fused fast path (fp64 atomics) differ by ~1e-12 per iteration
    for t in rts:
        t.close()
    A.close()
    keep.close()
