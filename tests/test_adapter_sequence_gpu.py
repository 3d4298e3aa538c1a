"""The drop-in END TO END: eight consecutive key frames in FullSystem::makeKeyFrame's order (FullSystem.cc:410-640) on two reference object graphs of
the same synthetic scene - the reference's own members on one (traceNewCoarse, optimizeImmaturePoint loop, optimize), ldso::GpuBackend on the other -
with everything around them (insertFrame, new residuals, removeOutliers, flagPointsForRemoval, dropPointsF, marginalizePointsF, marginalizeFrame) the
reference's host code on both.  The window grows from 5 to 6 key frames, then slides: image slots are recycled, the prior H_M / b_M is re-uploaded
after every marginalisation, lastResiduals / maxRelBaseline / numGoodResiduals travel through six optimize() calls.  Compared after EVERY key frame:
the key-frame set (ids) and per-frame point / residual / immature counts, the trajectory (camToWorld), the affine parameters, the prior, the
inverse depths - maxima and medians."""
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


@pytest.mark.parametrize("device_marg", [False, True])
def test_eight_key_frames_through_the_adapter_follow_the_reference(device_marg):
    """device_marg: the point marginalisation after optimize() through GpuBackend::flagPointsForRemoval + marginalizePoints (the policy on the host, the
    re-linearise / fix / accumulate pass of FullSystem.cc:1241-1250 + EnergyFunctional.cc:165-222 as ldso_ba_marginalize_points on the resident window)
    instead of the reference's host members on what the adapter wrote back"""
    from adapter_sequence_common import run_sequence
    win = synth.make_config("small", extra_frames=K)
    r_ref, log_ref = run_sequence(win, K)
    A = pr.GpuAdapter(max_frames=8, max_points=4000)
    pr.set_device_marginalisation(device_marg)
    try:
        r_adp, log_adp = run_sequence(win, K, adapter=A)
    finally:
        pr.set_device_marginalisation(False)
    assert len(log_ref) == len(log_adp) == K
    # FrameHessian::dIp on the device: ONE pyramid per frame seen (the 5 frames of the initial window + the 8 new key frames), shared by the tracer, the
    # BA image slot and - in a full system - the coarse trackers; built from 4 bytes per pixel
    assert A.pyramids_built() == win.F + K, A.pyramids_built()
    worst = dict(pose=0.0, aff=0.0, HM=0.0, bM=0.0, idepth_max=0.0, idepth_med=0.0, rmse=0.0, counts=0, unmatched_points=0)
    for a, b in zip(log_ref, log_adp):
        sa, sb = a["summary"], b["summary"]
        assert not a["lost"] and not b["lost"]
        assert sa["F"] == sb["F"] and np.array_equal(sa["ids"], sb["ids"]), "same key frames in the window"
        # The two graphs run DIFFERENT arithmetic for three stages (the device's summation orders): their states agree to ~1e-6, so an immature
        # point or a residual sitting exactly on a threshold (trace interval < 8, outlier energy, inlier count) may go the other way - a handful
        # per key frame at most, and the difference must not grow
        dc = max(abs(a["candidates"] - b["candidates"]), abs(a["new_residuals"] - b["new_residuals"]), abs(a["activated"] - b["activated"]), abs(a["points"] - b["points"]),
                 int(np.abs(sa["points"] - sb["points"]).max()), int(np.abs(sa["immature"] - sb["immature"]).max()))
        assert dc <= 8, (a["k"], {k: (a[k], b[k]) for k in ("candidates", "activated", "new_residuals", "points")}, sa["points"], sb["points"], sa["immature"], sb["immature"])
        assert int(np.abs(sa["residuals"] - sb["residuals"]).max()) <= 40, (sa["residuals"], sb["residuals"])
        worst["counts"] = max(worst["counts"], dc)
        worst["rmse"] = max(worst["rmse"], abs(a["rmse"] - b["rmse"]) / a["rmse"])
        scale = np.abs(sa["c2w"][:, :, 3]).max()
        worst["pose"] = max(worst["pose"], float(np.abs(sa["c2w"] - sb["c2w"]).max() / max(scale, 1.0)))
        worst["aff"] = max(worst["aff"], float(np.abs(sa["aff"] - sb["aff"]).max()))
        if np.abs(sa["HM"]).max() > 0:
            worst["HM"] = max(worst["HM"], _rel(sb["HM"], sa["HM"])); worst["bM"] = max(worst["bM"], _rel(sb["bM"], sa["bM"]))
        # inverse depths of the points both graphs hold, matched by (host key frame, pixel)
        ka = {(int(h), float(u), float(v)): float(d) for h, (u, v), d in zip(sa["host"], sa["uv"], sa["idepth"])}
        kb = {(int(h), float(u), float(v)): float(d) for h, (u, v), d in zip(sb["host"], sb["uv"], sb["idepth"])}
        both = sorted(set(ka) & set(kb))
        worst["unmatched_points"] = max(worst["unmatched_points"], len(set(ka) ^ set(kb)))
        assert len(both) > 0.97 * max(len(ka), len(kb))
        e = np.array([abs(ka[q] - kb[q]) / max(abs(ka[q]), 1e-3) for q in both])
        worst["idepth_max"] = max(worst["idepth_max"], float(e.max())); worst["idepth_med"] = max(worst["idepth_med"], float(np.median(e)))
    print("adapter sequence (device marginalisation: %s), worst over" % device_marg, K, "key frames:", {k: (round(v, 7) if isinstance(v, float) else v) for k, v in worst.items()})
    # observed on MI355X (round 4): rmse 1.9e-3, pose 2.0e-4 of the scene scale, affine 0.02 (b is in intensity units, 0..255), H_M 1.3e-3, b_M 2.3e-2,
    # inverse depths 1e-4 median / 8e-4 maximum, <= 1 object per count, 5 points held by one graph only - limits = 2..3 x observed
    observe(("sequence_dm_" if device_marg else "sequence_") + "rmse", worst["rmse"], 5e-3)
    observe(("sequence_dm_" if device_marg else "sequence_") + "pose", worst["pose"], 6e-4); observe(("sequence_dm_" if device_marg else "sequence_") + "affine", worst["aff"], 6e-2)
    observe(("sequence_dm_" if device_marg else "sequence_") + "HM", worst["HM"], 4e-3); observe(("sequence_dm_" if device_marg else "sequence_") + "bM", worst["bM"], 6e-2)
    observe(("sequence_dm_" if device_marg else "sequence_") + "idepth_median", worst["idepth_med"], 3e-4); observe(("sequence_dm_" if device_marg else "sequence_") + "idepth_max", worst["idepth_max"], 3e-3)
    observe(("sequence_dm_" if device_marg else "sequence_") + "unmatched_points", worst["unmatched_points"], 20)
    A.close()


def test_resident_window_sequence_equals_full_uploads():
    """The same eight key frames through two GpuBackends: one keeps the window resident between optimize() calls and sends deltas (ldso_ba_update_window: frames
    that left / arrived, surviving points, one bit per residual, the records of the activated points), the other flattens and uploads the whole window every
    time.  Both describe the same window to the same kernels: key-frame sets, point / residual counts and ids identical after every key frame, every float the
    drop-in writes back within the run-to-run reproducibility of the fused fast path (fp64 atomics, 1e-12 per iteration)."""
    from adapter_sequence_common import run_sequence
    win = synth.make_config("small", extra_frames=K)
    out = []
    for resident in (True, False):
        A = pr.GpuAdapter(max_frames=8, max_points=4000)
        A.set_resident_window(resident)
        pr.set_device_marginalisation(True)
        try:
            r, log = run_sequence(win, K, adapter=A)
        finally:
            pr.set_device_marginalisation(False)
        d, f = A.upload_counts()
        out.append((log, d, f))
        A.close()
    (la, da, fa), (lb, db, fb) = out
    assert db == 0 and fb == 2 * K, "without the resident window every upload is a full one (activatePoints + optimize per key frame)"
    assert fa == 1 and da == 2 * K - 1, (da, fa)          # only the very first window of the handle is flattened
    worst = 0.0
    for a, b in zip(la, lb):
        sa, sb = a["summary"], b["summary"]
        assert not a["lost"] and not b["lost"]
        for k in ("candidates", "activated", "new_residuals", "points"):
            assert a[k] == b[k], (a["k"], k, a[k], b[k])
        assert sa["F"] == sb["F"] and np.array_equal(sa["ids"], sb["ids"])
        for k in ("points", "residuals", "immature", "host"):
            assert np.array_equal(sa[k], sb[k]), (a["k"], k)
        assert np.array_equal(sa["uv"], sb["uv"])
        worst = max(worst, abs(a["rmse"] - b["rmse"]) / b["rmse"], *[_rel(sa[k], sb[k]) for k in ("c2w", "aff", "idepth", "HM", "bM")])
    observe("resident_vs_full_upload_sequence_floats", worst, 1e-9)          # observed 1.6e-13: two runs of the fused fast path (fp64 atomics) differ by ~1e-12 per iteration
