"""The key-frame driver of tests/test_adapter_sequence_gpu.py (adapter/adapter_capi.cc: adp_make_keyframe, FullSystem::makeKeyFrame's order,
FullSystem.cc:410-640) with the reference's OWN members at every stage - no GPU involved: the reference leg of the end-to-end comparison must itself behave
like a sliding-window system (the window grows to `max_frames` and slides, points get activated and marginalised, the prior builds up, the poses stay near
the truth).  Needs the libraries built where /root/reference exists (oracle/_ref/libldso_ref.so, adapter/_build/*.so)."""
import numpy as np
import pytest

from ldso_amd import synth
from oracle import pyref as pr

pytestmark = pytest.mark.skipif(not (pr.available() and pr.adapter_available()), reason="oracle/_ref/libldso_ref.so / adapter/_build/libldso_adapter_test.so not built")


def test_reference_leg_of_the_key_frame_sequence_slides_a_window():
    from adapter_sequence_common import run_sequence
    K = 4
    win = synth.make_config("small", extra_frames=K)
    r, log = run_sequence(win, K, adapter=None, max_frames=6)
    assert len(log) == K and not any(rec["lost"] for rec in log)
    F0 = win.F
    first_new = int(win.frames["frameID"].max()) + 1
    for k, rec in enumerate(log):
        s = rec["summary"]
        assert s["F"] == min(F0 + k + 1, 6)
        assert s["ids"][-1] == first_new + k, "the new key frame is the newest of the window"
        assert rec["candidates"] > 0 and rec["activated"] > 0.5 * rec["candidates"]
        assert np.isfinite(rec["rmse"]) and 0 < rec["rmse"] < 10
        assert s["immature"][-1] > 0, "fresh immature points on the new key frame"
        # the estimated pose of the new key frame stays near the scene's ground truth (the tracker's hand-over error is ~2e-3)
        c2w = np.eye(4); c2w[:3] = s["c2w"][-1]
        err = np.abs(c2w @ win.truth["w2c"][F0 + k] - np.eye(4)).max()
        assert err < 2e-2, err
    # once the window is full a frame is marginalised per key frame: the prior is there and symmetric
    last = log[-1]["summary"]
    assert np.abs(last["HM"]).max() > 0 and np.allclose(last["HM"], last["HM"].T, rtol=1e-9, atol=1e-6 * np.abs(last["HM"]).max())
    assert list(last["ids"]) == sorted(last["ids"]) and last["ids"][0] > int(win.frames["frameID"].min()), "the oldest frames are gone"
    r.close()


def test_reference_against_itself_is_the_yardstick_of_the_sequence_test():
    """tests/test_adapter_sequence_gpu.py bounds the drop-in's distance to the reference by 3 x the distance between two runs of the REFERENCE ITSELF over the
    same eight key frames: the single-threaded -O2 pin build against (i) its own 6-worker IndexThreadReduce (IndexThreadReduce.h:126-139: chunks go to whichever
    worker asks first, the per-thread float accumulators sum in another order every run) and (ii) the -O3 build of the same translation units.  This is the CPU
    half: the yardstick exists, both kinds of run keep the same key frames, and they really are different arithmetic (non-zero distances) - a sliding-window
    system amplifies rounding differences through threshold decisions (an immature point traced to one side of `interval < 8`, a residual to one side of its
    outlier energy), which is why the end-to-end distances are 1e-4 .. 1e-2 and not the 1e-6 of a single stage."""
    from adapter_sequence_common import reference_yardstick, QUANTITIES
    yard, per = reference_yardstick("small", 8, mt_runs=2)
    print("reference vs reference over 8 key frames:", {k: {q: float("%.3g" % v) for q, v in d.items()} for k, d in per.items()})
    assert set(yard) == set(QUANTITIES)
    assert "O3_build" in per, "this container builds oracle/_ref/fast/libldso_ref.so and adapter/_build_fast (make -C oracle ref_fast; make -C adapter fast)"
    for name, d in per.items():
        assert all(np.isfinite(v) for v in d.values()), name
        assert d["counts"] <= 8 and d["unmatched_points"] <= 40, (name, d)          # two runs of the reference stay the same system
    assert per["O3_build"]["pose"] > 0 and per["O3_build"]["HM"] > 0, "the -O3 build rounds differently from the pin build"
    assert max(per[k]["bM"] for k in per if k.startswith("six_threads")) > 0, "six workers sum in another order than one"
