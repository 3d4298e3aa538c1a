"""A key-frame SEQUENCE through the drop-in (tests/test_adapter_sequence_gpu.py, scripts/time_adapter.py): the synthetic scene of a window plus K
further frames; every further frame becomes a key frame in FullSystem::makeKeyFrame's order (adapter/adapter_capi.cc: adp_make_keyframe,
FullSystem.cc:410-640) - trace the immature points into it, insert it, add the new residuals of the old points, activate, optimize, remove outliers,
flag / drop / marginalise points, marginalise the oldest frame once the window holds `max_frames`, hand new immature points to the new frame."""
import numpy as np

from ldso_amd import synth
from oracle import pyref as pr


def noisy_pose(T_w2c, k, t_sigma=2e-3, r_sigma=2e-4):
    """what the coarse tracker would hand over: the true pose with a small deterministic error"""
    rng = np.random.default_rng(1000 + k)
    xi = np.concatenate([rng.normal(0, t_sigma, 3), rng.normal(0, r_sigma, 3)])
    return synth.se3_exp(xi) @ T_w2c


def run_sequence(win, K, adapter=None, max_frames=6, per_frame=120, iterations=6, on_keyframe=None):
    """-> (RefWindow, list of per-key-frame records).  win = synth.make_config(name, extra_frames=K)."""
    r = pr.RefWindow(win)
    r.fs_attach()
    F0 = win.F
    imm, _ = synth.make_immature_points(win, per_frame, seed=7)
    r.fs_add_immature(imm)
    log = []
    next_id = int(win.frames["frameID"].max()) + 1
    for k in range(K):
        img = win.images[F0 + k][0]
        T = noisy_pose(win.truth["w2c"][F0 + k], k)
        fh = r.fs_new_frame(img, T, float(win.truth["aff_a"][F0 + k]), float(win.truth["aff_b"][F0 + k]), 1.0)
        r.fs_set_frame_id(fh, next_id + k)
        nF = r.num_frames()
        marg = 0 if nF + 1 > max_frames else -1                     # flagFramesForMarginalization stand-in: the oldest frame once the window is full
        rmse, st = pr.make_keyframe(r, adapter, fh, marg, next_id + k, iterations)
        if not st["lost"]:
            # makeNewTraces stand-in (pixel selection is upstream of the hot path): fresh immature points on the new key frame
            hostIdx = r.num_frames() - 1
            imm, _ = synth.make_immature_points(win, per_frame, seed=100 + k, frames=[F0 + k], hosts=[hostIdx])
            r.fs_add_immature(imm)
        rec = dict(k=k, rmse=rmse, **st, summary=pr.graph_summary(r))
        if on_keyframe is not None:
            on_keyframe(rec)
        log.append(rec)
        if st["lost"]:
            break
    return r, log
