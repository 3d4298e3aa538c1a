"""The window kept RESIDENT across key frames (ldso_ba_update_window, include/ldso_hip.h) against a fresh ldso_ba_set_window of the same objects.

Between two optimize() calls the reference edits its window in place - EnergyFunctional::marginalizeFrame (EF.cc:72), dropPointsF (:224), removePoint (:153),
dropResidual (:63), insertFrame (:32), insertResidual (:26), makeIDX (:380).  Here the host state after an optimize() is taken from the device (what the
drop-in's write-back stores into the reference objects), the same edits are applied to it - the oldest frame leaves with its points, some points are removed,
residuals are dropped, a new frame arrives with one residual per surviving point, fresh points are activated in the middle of every host frame's list - and
the resulting window is loaded twice: as a DELTA onto the resident window of the handle that ran the optimize(), and from scratch into a second handle.  The
two must be the same window: identical residual / point indices, bitwise identical linearisation (energies, states, JpJdF, per-point Schur scalars, the stitched
system of the step-wise path), and the same optimize() (the fused fast path is reproducible to 1e-12)."""
import copy

import numpy as np
import pytest

from ldso_amd import synth, binding

pytestmark = pytest.mark.gpu


def _sub(big, frame_ids, point_ids):
    """frames / points of `big` (given as index lists), residuals between the kept frames: a synth.Window numbered from zero"""
    fmap = -np.ones(big.F, np.int64); fmap[frame_ids] = np.arange(len(frame_ids))
    pmap = -np.ones(big.P, np.int64); pmap[point_ids] = np.arange(len(point_ids))
    pts = big.points[point_ids].copy()
    assert (fmap[pts["host"]] >= 0).all()
    pts["host"] = fmap[pts["host"]]
    r = big.residuals
    keep = (pmap[r["point"]] >= 0) & (fmap[r["target"]] >= 0)
    res = r[keep].copy()
    res["point"] = pmap[res["point"]]; res["host"] = fmap[res["host"]]; res["target"] = fmap[res["target"]]
    order = np.lexsort((res["target"], res["point"]))                                    # point-major, target-ascending: the flat order of the update path
    res = res[order]
    cnt = np.bincount(res["point"], minlength=len(pts))
    pts["res_count"] = cnt; pts["res_begin"] = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    n = 8 * len(frame_ids) + 4
    w = synth.Window(w=big.w, h=big.h, levels=big.levels, K=big.K, settings=big.settings, calib=big.calib.copy(), frames=big.frames[frame_ids].copy(), points=pts, residuals=res,
                     images=[big.images[f] for f in frame_ids], HM=np.zeros((n, n)), bM=np.zeros(n))
    return w, keep, order


def _scenario(F_big, old_frames, new_frames, P=600, seed=5):
    rng = np.random.default_rng(seed)
    big = synth.make_config("small", F=F_big, P=P)
    old_pts = np.nonzero(np.isin(big.points["host"], old_frames) & (rng.random(big.P) < 0.85))[0]
    w0, _, _ = _sub(big, old_frames, old_pts)
    synth.add_synthetic_prior(w0, seed=3)
    A = binding.BA(big.w, big.h, max_frames=big.F, max_points=big.P)
    A.set_settings(big.settings)
    for f in range(big.F):
        A.set_image(f, big.images[f][0])                                                  # image slot = frame index in `big`
    A.set_window(old_frames, w0.points, w0.residuals)
    A.set_frames(w0.frames, w0.calib)
    A.set_prior(w0.HM, w0.bM)
    A.optimize(3, force_all=True)
    pa, ra, fa = A.get_points(), A.get_residuals(), A.get_frames()
    # ---- the host objects after the write-back, then the reference's edits ---------------------------------------------------------------
    big2 = copy.deepcopy(big)
    big2.points["idepth"][old_pts] = pa["idepth"]; big2.points["idepth_zero"][old_pts] = pa["idepth"]
    big2.frames[old_frames] = fa["frames"]
    big2.calib["value"] = fa["calib_value"]
    rb = big.residuals
    in_old = np.isin(rb["point"], old_pts) & np.isin(rb["target"], old_frames)
    idx_old = np.nonzero(in_old)[0]
    idx_old = idx_old[np.lexsort((rb["target"][idx_old], rb["point"][idx_old]))]          # = the flat order of w0 (big is point-major with ascending targets already)
    assert len(idx_old) == w0.R
    big2.residuals["state_state"][idx_old] = ra["state_state"]; big2.residuals["is_active"][idx_old] = ra["is_active"]
    big2.residuals["state_energy"][idx_old] = ra["out"]["state_NewEnergy"]; big2.residuals["is_new"][idx_old] = 1
    dropped_res = np.zeros(big.R, bool)
    dropped_res[idx_old] = (ra["to_remove"] != 0) | (rng.random(w0.R) < 0.03)                # linearizeAll(true)'s toRemove + a few more dropResidual calls
    gone_frames = [f for f in old_frames if f not in new_frames]
    added_frames = [f for f in new_frames if f not in old_frames]
    survive = old_pts[~np.isin(big.points["host"][old_pts], gone_frames) & (rng.random(len(old_pts)) > 0.1)]
    fresh = np.nonzero(~np.isin(np.arange(big.P), old_pts) & np.isin(big.points["host"], [f for f in new_frames if f not in added_frames]))[0]
    # residuals of a surviving point towards an added frame: insertResidual (IN, energy 0, not active, isNew)
    new_res = np.isin(rb["point"], survive) & np.isin(rb["target"], added_frames)
    big2.residuals["state_state"][new_res] = 0; big2.residuals["is_active"][new_res] = 0; big2.residuals["state_energy"][new_res] = 0; big2.residuals["is_new"][new_res] = 1
    big2.residuals["is_new"][np.isin(rb["point"], fresh)] = 1
    new_pts = np.sort(np.concatenate([survive, fresh]))                                   # host-major order of `big` = makeIDX order
    big3 = copy.deepcopy(big2)
    big3.residuals = big2.residuals[~dropped_res]
    w1, _, _ = _sub(big3, new_frames, new_pts)
    synth.add_synthetic_prior(w1, seed=4)
    mrb = np.zeros(big.P, np.float32); ngr = np.zeros(big.P, np.int32)
    mrb[old_pts] = pa["maxRelBaseline"]; ngr[old_pts] = pa["numGoodResiduals"]
    # ---- the delta -------------------------------------------------------------------------------------------------------------------------
    frame_from = np.array([old_frames.index(f) if f in old_frames else -1 for f in new_frames], np.int32)
    old_row = -np.ones(big.P, np.int64); old_row[old_pts] = np.arange(len(old_pts))
    point_from = old_row[new_pts].astype(np.int32)
    is_fresh = point_from < 0
    point_from[is_fresh] = -1 - np.arange(is_fresh.sum())
    res_mask = np.zeros(len(new_pts), np.uint32)
    np.bitwise_or.at(res_mask, w1.residuals["point"], (1 << w1.residuals["target"]).astype(np.uint32))
    fresh_rows = np.nonzero(is_fresh)[0]
    fresh_pts = w1.points[fresh_rows]
    fr_sel = np.isin(w1.residuals["point"], fresh_rows)
    fresh_res = w1.residuals[fr_sel].copy()
    remap = -np.ones(len(new_pts), np.int64); remap[fresh_rows] = np.arange(len(fresh_rows))
    fresh_res["point"] = remap[fresh_res["point"]]
    return dict(old_frames=old_frames, old_pts=old_pts, survive=survive, fresh=fresh, dropped_res=dropped_res, big2=big2, added_frames=added_frames, gone_frames=gone_frames,
                A=A, big=big, w1=w1, new_frames=new_frames, frame_from=frame_from, point_from=point_from, res_mask=res_mask, fresh_pts=fresh_pts, fresh_res=fresh_res,
                mrb=mrb[new_pts], ngr=ngr[new_pts], fresh_rows=fresh_rows, n_survive=len(survive), n_fresh=len(fresh), n_dropped=int(dropped_res.sum()))


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.tobytes() == b.tobytes(), f"{what}: {int((a != b).sum()) if a.shape == b.shape else 'shape'} entries differ"


@pytest.mark.parametrize("F_big,old_frames,new_frames", [
    (7, [0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 6]),                    # makeKeyFrame's steady state: the oldest frame leaves, a new one arrives
    (7, [0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 5]),                    # nothing but point / residual edits
    (7, [0, 1, 2, 3, 4], [0, 1, 3, 4, 5, 6]),                       # a middle frame leaves, two arrive
    (10, [0, 1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7, 8, 9]),    # 8 -> 9 frames: the second slot group appears (FS 8 -> 16)
    (10, [0, 1, 2, 3, 4, 5, 6, 7, 8], [2, 3, 4, 5, 6, 7, 8, 9]),    # ... and disappears again
], ids=["steady", "points-only", "middle", "grow-to-9", "shrink-to-8"])
def test_resident_window_equals_fresh_upload(F_big, old_frames, new_frames):
    sc = _scenario(F_big, old_frames, new_frames)
    A, big, w1 = sc["A"], sc["big"], sc["w1"]
    assert sc["n_survive"] > 100 and sc["n_fresh"] > 20 and sc["n_dropped"] > 5
    # resident: the delta onto the handle that ran optimize()
    A.update_window(sc["new_frames"], sc["frame_from"], sc["point_from"], sc["res_mask"], sc["fresh_pts"], sc["fresh_res"], sc["mrb"][sc["fresh_rows"]], sc["ngr"][sc["fresh_rows"]])
    A.set_frames(w1.frames, w1.calib); A.set_prior(w1.HM, w1.bM)
    # fresh: the same objects from scratch
    Bh = binding.BA(big.w, big.h, max_frames=big.F, max_points=big.P)
    Bh.set_settings(big.settings)
    for f in range(big.F):
        Bh.set_image(f, big.images[f][0])
    Bh.set_window(sc["new_frames"], w1.points, w1.residuals)
    Bh.set_point_stats(sc["mrb"], sc["ngr"])
    Bh.set_frames(w1.frames, w1.calib); Bh.set_prior(w1.HM, w1.bM)
    assert (A.F, A.P, A.R) == (Bh.F, Bh.P, Bh.R) == (w1.F, w1.P, w1.R)
    # the window as loaded: what the handles say about it before anything runs
    ra, rb = A.get_residuals(), Bh.get_residuals()
    _same(ra["state_state"], rb["state_state"], "state_state as loaded"); _same(ra["is_active"], rb["is_active"], "is_active as loaded")
    _same(ra["out"]["state_NewEnergy"], rb["out"]["state_NewEnergy"], "state_energy as loaded")
    pa, pb = A.get_points(), Bh.get_points()
    for k in ("idepth", "maxRelBaseline", "numGoodResiduals"):
        _same(pa[k], pb[k], "point." + k + " as loaded")
    _same(ra["state_state"], w1.residuals["state_state"], "state_state against the host objects")
    # one stage-wise pass: bitwise
    for h in (A, Bh):
        h.collect_active()
    Ea, Eb = A.linearize_all(False), Bh.linearize_all(False)
    assert Ea == Eb and np.isfinite(Ea)
    ra, rb = A.get_residuals(), Bh.get_residuals()
    for k in ra["out"].dtype.names:
        _same(ra["out"][k], rb["out"][k], "linearize " + k)
    for h in (A, Bh):
        h.apply_res(); h.backup_state(); h.solve_system(0)
    sa, sb = A.get_system(), Bh.get_system()
    for k in ("HA", "Hsc", "HFinal", "bFinal", "x"):
        _same(sa[k], sb[k], "system " + k)
    pa, pb = A.get_points(), Bh.get_points()
    for k in pa.dtype.names:
        _same(pa[k], pb[k], "point." + k)
    # ... and the optimize() that follows (fast path)
    rma, ia = A.optimize(4, force_all=True); rmb, ib = Bh.optimize(4, force_all=True)
    ea, eb = A.get_energy_log(), Bh.get_energy_log()
    assert ia == ib and np.abs(ea - eb).max() <= 1e-9 * np.abs(eb).max() and abs(rma - rmb) <= 1e-9 * rmb
    _same(A.get_residuals()["state_state"], Bh.get_residuals()["state_state"], "state_state after optimize")
    A.close(); Bh.close()


def _load_fresh(sc):
    big, w1 = sc["big"], sc["w1"]
    Bh = binding.BA(big.w, big.h, max_frames=big.F, max_points=big.P)
    Bh.set_settings(big.settings)
    for f in range(big.F):
        Bh.set_image(f, big.images[f][0])
    Bh.set_window(sc["new_frames"], w1.points, w1.residuals)
    Bh.set_point_stats(sc["mrb"], sc["ngr"])
    Bh.set_frames(w1.frames, w1.calib); Bh.set_prior(w1.HM, w1.bM)
    return Bh


@pytest.mark.parametrize("F_big,old_frames,new_frames", [(7, [0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 6]), (7, [0, 1, 2, 3, 4], [0, 1, 3, 4, 5, 6])], ids=["steady", "middle"])
def test_call_by_call_edit_equals_fresh_upload(F_big, old_frames, new_frames):
    """The same window edit recorded as the reference's own maintenance calls (ldso_ba_window_begin, _remove_frame ≙ marginalizeFrame, _insert_frame ≙ insertFrame,
    _remove_points ≙ removePoint / dropPointsF, _drop_residuals ≙ dropResidual, _add_residuals ≙ insertResidual, _add_points ≙ insertPoint, _window_commit):
    frames / targets named by their index in the RESIDENT window, points by their resident row."""
    sc = _scenario(F_big, old_frames, new_frames)
    A, big, w1, big2 = sc["A"], sc["big"], sc["w1"], sc["big2"]
    old_pts, survive, fresh = sc["old_pts"], sc["survive"], sc["fresh"]
    A.window_begin()
    for f in sc["gone_frames"]:
        A.remove_frame(old_frames.index(f))
    fid = {f: old_frames.index(f) for f in old_frames}
    for f in sc["added_frames"]:
        fid[f] = A.insert_frame(f)                                                         # image slot = frame index in `big`
    old_row = -np.ones(big.P, np.int64); old_row[old_pts] = np.arange(len(old_pts))
    gone_pts = old_pts[~np.isin(old_pts, survive) & ~np.isin(big.points["host"][old_pts], sc["gone_frames"])]
    A.remove_points(old_row[gone_pts])
    rb = big.residuals
    is_surv = np.isin(rb["point"], survive)
    d = sc["dropped_res"] & is_surv & np.isin(rb["target"], [f for f in old_frames if f in new_frames])
    A.drop_residuals(old_row[rb["point"][d]], [fid[int(t)] for t in rb["target"][d]])
    a = is_surv & np.isin(rb["target"], sc["added_frames"]) & ~sc["dropped_res"]
    A.add_residuals(old_row[rb["point"][a]], [fid[int(t)] for t in rb["target"][a]])
    fr = np.isin(rb["point"], fresh) & np.isin(rb["target"], new_frames) & ~sc["dropped_res"]
    fres = big2.residuals[fr].copy()
    remap = -np.ones(big.P, np.int64); remap[fresh] = np.arange(len(fresh))
    fres["point"] = remap[fres["point"]]; fres["target"] = [fid[int(t)] for t in fres["target"]]
    fpts = big2.points[fresh].copy(); fpts["host"] = [fid[int(h)] for h in fpts["host"]]
    A.add_points(fpts, np.searchsorted(old_pts, fresh), fres)                              # in front of the first resident row that follows it in makeIDX order
    A.window_commit(w1.F, w1.P, w1.R)
    A.set_frames(w1.frames, w1.calib); A.set_prior(w1.HM, w1.bM)
    Bh = _load_fresh(sc)
    ra, rb_ = A.get_residuals(), Bh.get_residuals()
    _same(ra["state_state"], rb_["state_state"], "state_state as loaded"); _same(ra["out"]["state_NewEnergy"], rb_["out"]["state_NewEnergy"], "state_energy as loaded")
    for h in (A, Bh):
        h.collect_active()
    Ea, Eb = A.linearize_all(False), Bh.linearize_all(False)
    assert Ea == Eb and np.isfinite(Ea)
    ra, rb_ = A.get_residuals(), Bh.get_residuals()
    for k in ra["out"].dtype.names:
        _same(ra["out"][k], rb_["out"][k], "linearize " + k)
    pa, pb = A.get_points(), Bh.get_points()
    for k in pa.dtype.names:
        _same(pa[k], pb[k], "point." + k)
    # an edit that cannot be committed leaves the window as it is
    A.apply_res()
    A.window_begin()
    with pytest.raises(binding.LdsoError):
        A.add_residuals([0], [int(w1.points["host"][0])])                                  # a residual towards the point's own host
    A.remove_frame(0)
    with pytest.raises(binding.LdsoError):
        A.add_points(w1.points[:1], [0], w1.residuals[:0])                                  # ... hosted by the frame just removed: rejected at the commit
        A.window_commit(w1.F, w1.P, w1.R)
    Bh.apply_res()
    for h in (A, Bh):
        h.collect_active()
    Ea2, Eb2 = A.linearize_all(False), Bh.linearize_all(False)
    assert Ea2 == Eb2 and np.isfinite(Ea2) and (A.F, A.P, A.R) == (w1.F, w1.P, w1.R)
    A.close(); Bh.close()


def test_update_window_rejects_bad_deltas():
    sc = _scenario(7, [0, 1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 6])
    A = sc["A"]
    args = [sc["new_frames"], sc["frame_from"], sc["point_from"], sc["res_mask"], sc["fresh_pts"], sc["fresh_res"], sc["mrb"][sc["fresh_rows"]], sc["ngr"][sc["fresh_rows"]]]

    def bad(i, v):
        a = list(args); a[i] = v
        with pytest.raises(binding.LdsoError):
            A.update_window(*a)

    ff = sc["frame_from"].copy(); ff[0], ff[1] = ff[1], ff[0]; bad(2 - 1, ff)                              # surviving frames out of order
    pf = sc["point_from"].copy(); keep = np.nonzero(pf >= 0)[0]; pf[keep[0]], pf[keep[1]] = pf[keep[1]], pf[keep[0]]; bad(2, pf)      # surviving points out of order
    mk = sc["res_mask"].copy(); mk[0] |= np.uint32(1 << 20); bad(3, mk)                                       # a residual to a frame outside the window
    bad(5, sc["fresh_res"][:-1])                                                                              # a fresh residual short
    # a delta that fails its validation touches nothing: the resident window is still there and takes the good delta
    A.update_window(*args)
    w1 = sc["w1"]
    A.set_frames(w1.frames, w1.calib)
    A.collect_active()
    assert np.isfinite(A.linearize_all(False)) and (A.F, A.P, A.R) == (w1.F, w1.P, w1.R)
    A.close()
