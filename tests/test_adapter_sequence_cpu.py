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
