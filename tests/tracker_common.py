"""Shared builder of the CoarseTracker test scenario (BASELINE config C2): reference keyframe = newest frame of
a window, points = its active points projected with their (noisy) idepths, new frame = one more pose step."""
import numpy as np

from ldso_amd import synth


def tracker_scenario(name="small", levels=None, seed=20260925, **kw):
    win = synth.make_config(name, extra_frames=1, seed=seed, **kw)
    F = win.F
    lv = win.levels if levels is None else levels
    if levels is not None and levels != win.levels:
        win = synth.make_config(name, extra_frames=1, seed=seed, levels=levels, **kw)
    K = win.K
    ref_T = win.truth["w2c"][F - 1]
    # points of all frames projected into the reference keyframe (what lastResiduals[0] IN provides)
    pts = []
    rng = np.random.default_rng(5)
    for i in range(win.P):
        h = int(win.points["host"][i])
        Th = win.truth["w2c"][h]
        u, v, idp = float(win.points["u"][i]), float(win.points["v"][i]), float(win.points["idepth"][i])
        pc = np.array([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], 1.0]) / idp
        pw = np.linalg.inv(Th) @ np.append(pc, 1.0)
        pr = ref_T @ pw
        if pr[2] <= 0.1:
            continue
        Ku, Kv = K[0, 0] * pr[0] / pr[2] + K[0, 2], K[1, 1] * pr[1] / pr[2] + K[1, 2]
        if not (3 < Ku < win.w - 4 and 3 < Kv < win.h - 4):
            continue
        pts.append((Ku, Kv, 1.0 / pr[2], rng.uniform(1e-4, 1e-2)))
    pts = np.asarray(pts, np.float32)
    ref_pyr = win.images[F - 1]
    new_pyr = win.images[F]
    T_true = win.truth["w2c"][F] @ np.linalg.inv(ref_T)
    ref_aff = (float(win.truth["aff_a"][F - 1]), float(win.truth["aff_b"][F - 1]))
    new_aff = (float(win.truth["aff_a"][F]), float(win.truth["aff_b"][F]))
    return dict(win=win, levels=lv, pts=pts, ref_pyr=ref_pyr, new_pyr=new_pyr, T_true=T_true, ref_aff=ref_aff, new_aff=new_aff)
