"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_build/liboracle.so (the CPU restatement of the LDSO hot path).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker / reported baseline.  The product path (ldso_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from ldso_amd import synth

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(fast: bool = False) -> str:
    tgt = "_build/liboracle_fast.so" if fast else "_build/liboracle.so"
    subprocess.run(["make", "-C", _HERE, tgt], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, tgt)


def lib(fast: bool = False):
    key = "fast" if fast else "det"
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "_build", "liboracle_fast.so" if fast else "liboracle.so")
    if not os.path.exists(path):
        path = build(fast)
    L = C.CDLL(path)
    L.orc_create.restype = C.c_void_p
    L.orc_tr_create.restype = C.c_void_p
    L.orc_linearize_all.restype = C.c_double
    L.orc_optimize.restype = C.c_float
    L.orc_time_optimize.restype = C.c_double
    L.orc_tr_time_track.restype = C.c_double
    _LIBS[key] = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _img_ptrs(images, levels):
    """images: list (per frame) of list (per level) of float32 [h,w,3] -> (void*[F*levels], keepalive)."""
    keep = []
    arr = (C.c_void_p * (len(images) * levels))()
    for f, pyr in enumerate(images):
        for l in range(levels):
            a = np.ascontiguousarray(pyr[l], dtype=np.float32)
            keep.append(a)
            arr[f * levels + l] = a.ctypes.data
    return arr, keep


class OracleWindow:
    """FullSystem/EnergyFunctional slice of the reference, driven on one flattened window."""

    def __init__(self, win: synth.Window, multithreading: bool = False, fast: bool = False):
        self.L = lib(fast)
        self.win = win
        self.F, self.P, self.R = win.F, win.P, win.R
        imgs, self._keep = _img_ptrs(win.images[: win.F], win.levels)
        self.frames = np.ascontiguousarray(win.frames)
        self.points = np.ascontiguousarray(win.points)
        self.residuals = np.ascontiguousarray(win.residuals)
        linJ = np.ascontiguousarray(win.lin_J) if win.lin_J is not None else None
        rtz = np.ascontiguousarray(win.lin_res_toZeroF, dtype=np.float32) if win.lin_res_toZeroF is not None else None
        self._keep += [linJ, rtz]
        HM = np.ascontiguousarray(win.HM, dtype=np.float64)
        bM = np.ascontiguousarray(win.bM, dtype=np.float64)
        s = np.ascontiguousarray(win.settings)
        c = np.ascontiguousarray(win.calib)
        self.h = C.c_void_p(self.L.orc_create(
            C.c_int(win.w), C.c_int(win.h), C.c_int(win.levels), _p(s), _p(c), C.c_int(self.F), _p(self.frames), imgs,
            C.c_int(self.P), _p(self.points), C.c_int(self.R), _p(self.residuals), _p(linJ), _p(rtz), _p(HM), _p(bM),
            C.c_int(1 if multithreading else 0)))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- stage calls -----------------------------------------------------------------------------
    def collect_active(self, reset_oob=True):
        if reset_oob:
            self.L.orc_collect_active(self.h)
        else:
            self.L.orc_collect_active_keep_states(self.h)

    def linearize_all(self, fix=False) -> float:
        return float(self.L.orc_linearize_all(self.h, C.c_int(1 if fix else 0)))

    def apply_res(self):
        self.L.orc_apply_res(self.h)

    def set_precalc(self):
        self.L.orc_set_precalc(self.h)

    def backup_state(self):
        self.L.orc_backup_state(self.h)

    def do_step(self) -> bool:
        return bool(self.L.orc_do_step(self.h))

    def solve_system(self, iteration: int, lam: float = 1e-1):
        self.L.orc_solve_system(self.h, C.c_int(iteration), C.c_double(lam))

    def calc_lm_energies(self):
        m, l = C.c_double(), C.c_double()
        self.L.orc_calc_lm_energies(self.h, C.byref(m), C.byref(l))
        return m.value, l.value

    def set_force_all_iterations(self, v=True):
        self.L.orc_set_force_all_iterations(self.h, C.c_int(1 if v else 0))

    def optimize(self, niters: int) -> float:
        return float(self.L.orc_optimize(self.h, C.c_int(niters)))

    def is_lost(self) -> bool:
        return bool(self.L.orc_is_lost(self.h))

    def time_optimize(self, niters: int) -> float:
        return float(self.L.orc_time_optimize(self.h, C.c_int(niters)))

    def energy_log(self):
        buf = np.zeros(64, np.float64)
        n = self.L.orc_energy_log(self.h, _p(buf), C.c_int(64))
        return buf[:n].copy()

    def counts(self):
        a, l, m = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_counts(self.h, C.byref(a), C.byref(l), C.byref(m))
        return a.value, l.value, m.value

    def num_frames(self):
        return int(self.L.orc_num_frames(self.h))

    # --- fetchers ---------------------------------------------------------------------------------
    def get_residuals(self, with_J=True):
        out = np.zeros(self.R, synth.RES_OUT_DTYPE)
        J = np.zeros(self.R, synth.RAWJAC_DTYPE) if with_J else None
        st = np.zeros(self.R, np.int32)
        act = np.zeros(self.R, np.int32)
        rtz = np.zeros((self.R, 8), np.float32)
        lin = np.zeros(self.R, np.int32)
        alive = np.zeros(self.R, np.int32)
        self.L.orc_get_residuals(self.h, _p(out), _p(J), _p(st), _p(act), _p(rtz), _p(lin), _p(alive))
        return dict(out=out, J=J, state_state=st, is_active=act, res_toZeroF=rtz, is_linearized=lin, alive=alive)

    def get_points(self):
        out = np.zeros(self.P, synth.POINT_OUT_DTYPE)
        status = np.zeros(self.P, np.int32)
        self.L.orc_get_points(self.h, _p(out), _p(status))
        return out, status

    def get_frames(self):
        F = self.num_frames()
        fr = np.zeros(F, synth.FRAME_DTYPE)
        step = np.zeros((F, 10))
        cv = np.zeros(4)
        cs = np.zeros(4)
        pre = np.zeros((F, 12))
        self.L.orc_get_frames(self.h, _p(fr), _p(step), _p(cv), _p(cs), _p(pre))
        return dict(frames=fr, step=step, calib_value=cv, calib_step=cs, pre_worldToCam=pre)

    def get_precalc(self):
        F = self.num_frames()
        out = np.zeros((F, F, 27), np.float32)
        self.L.orc_get_precalc(self.h, _p(out))
        return out

    def get_adjoints(self):
        F = self.num_frames()
        ah = np.zeros((F * F, 8, 8))
        at = np.zeros((F * F, 8, 8))
        d = np.zeros((F * F, 8), np.float32)
        self.L.orc_get_adjoints(self.h, _p(ah), _p(at), _p(d))
        return ah, at, d

    def get_accumulators(self):
        F = self.num_frames()
        d = dict(topA=np.zeros((F * F, 13, 13), np.float32), topL=np.zeros((F * F, 13, 13), np.float32),
                 accD=np.zeros((F * F * F, 8, 8), np.float32), accE=np.zeros((F * F, 8, 4), np.float32),
                 accEB=np.zeros((F * F, 8), np.float32), accHcc=np.zeros((4, 4), np.float32), accbc=np.zeros(4, np.float32))
        self.L.orc_get_accumulators(self.h, _p(d["topA"]), _p(d["topL"]), _p(d["accD"]), _p(d["accE"]), _p(d["accEB"]),
                                    _p(d["accHcc"]), _p(d["accbc"]))
        return d

    def get_system(self):
        n = 8 * self.num_frames() + 4
        names_m = ["HA", "HL", "Hsc", "HFinal", "lastHS"]
        names_v = ["bA", "bL", "bsc", "bFinal", "x", "lastbS"]
        d = {k: np.zeros((n, n)) for k in names_m}
        d.update({k: np.zeros(n) for k in names_v})
        self.L.orc_get_system(self.h, _p(d["HA"]), _p(d["bA"]), _p(d["HL"]), _p(d["bL"]), _p(d["Hsc"]), _p(d["bsc"]),
                              _p(d["HFinal"]), _p(d["bFinal"]), _p(d["x"]), _p(d["lastHS"]), _p(d["lastbS"]))
        return d

    def get_prior(self):
        n = 8 * self.num_frames() + 4
        HM = np.zeros((n, n))
        bM = np.zeros(n)
        self.L.orc_get_prior(self.h, _p(HM), _p(bM))
        return HM, bM

    def fix_linearization(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self.L.orc_fix_linearization(self.h, _p(ids), C.c_int(len(ids)))

    # --- marginalisation ---------------------------------------------------------------------------
    def flag_frame(self, idx):
        self.L.orc_flag_frame(self.h, C.c_int(idx))

    def flag_points_for_removal(self):
        self.L.orc_flag_points_for_removal(self.h)

    def drop_points(self):
        self.L.orc_drop_points(self.h)

    def marginalize_points(self):
        self.L.orc_marginalize_points(self.h)

    def marginalize_frame(self, idx):
        self.L.orc_marginalize_frame(self.h, C.c_int(idx))

    def export_window(self):
        pts = np.zeros(self.P, synth.POINT_DTYPE)
        res = np.zeros(self.R, synth.RESIDUAL_DTYPE)
        J = np.zeros(self.R, synth.RAWJAC_DTYPE)
        rtz = np.zeros((self.R, 8), np.float32)
        op = np.zeros(self.P, np.int32)
        orr = np.zeros(self.R, np.int32)
        F, P, R = C.c_int(), C.c_int(), C.c_int()
        self.L.orc_export_window(self.h, C.byref(F), C.byref(P), C.byref(R), _p(pts), _p(res), _p(J), _p(rtz), _p(op), _p(orr))
        P, R = P.value, R.value
        return dict(F=F.value, points=pts[:P].copy(), residuals=res[:R].copy(), lin_J=J[:R].copy(), lin_res_toZeroF=rtz[:R].copy(),
                    orig_point=op[:P].copy(), orig_res=orr[:R].copy())


def make_mixed_window(win: synth.Window, oldest=2, perturb_seed=7) -> synth.Window:
    """SURVEY.md §8(d) "mixed" variant: residuals whose target is among the `oldest` frames are pre-linearised
    (isLinearized, res_toZeroF, J from a prior linearize), then the frame states / idepths are perturbed so that
    the delta terms of mode 1 are non-zero."""
    import copy
    o = OracleWindow(win)
    o.collect_active()
    o.linearize_all(False)
    o.apply_res()
    ids = np.nonzero(win.residuals["target"] < oldest)[0]
    o.fix_linearization(ids)
    ex = o.export_window()
    w2 = copy.deepcopy(win)
    assert len(ex["points"]) == win.P and len(ex["residuals"]) == win.R
    w2.residuals = ex["residuals"]
    w2.lin_J = ex["lin_J"]
    w2.lin_res_toZeroF = ex["lin_res_toZeroF"]
    rng = np.random.default_rng(perturb_seed)
    for k in range(1, win.F):
        w2.frames["state"][k][0:3] += rng.normal(0, 5e-4, 3)
        w2.frames["state"][k][3:6] += rng.normal(0, 5e-5, 3)
        w2.frames["state"][k][6] += rng.normal(0, 1e-4)
        w2.frames["state"][k][7] += rng.normal(0, 5e-5)
    w2.points["idepth"] = (w2.points["idepth"] * (1 + rng.normal(0, 2e-3, win.P))).astype(np.float32)
    w2.calib["value"] = w2.calib["value"] * (1 + rng.normal(0, 1e-6, 4))
    o.close()
    return w2


def make_convergent_mixed_window(win: synth.Window, oldest=2, perturb_seed=11) -> synth.Window:
    """A window with linearised residuals that a forced-accept GN sequence converges on: the window is first optimised to convergence, the
    active residuals into the `oldest` frames are fixed THERE (fixLinearizationF at a consistent state, as FullSystem::flagPointsForRemoval
    does for points about to be marginalised), and only then the state is perturbed a little.  (make_mixed_window fixes at an unconverged,
    perturbed state: its sequence overshoots from the second step on.)"""
    import copy
    o = OracleWindow(win)
    o.optimize(8)
    r = o.get_residuals(False)
    ids = np.nonzero((win.residuals["target"] < oldest) & (r["alive"] != 0) & (r["is_active"] != 0))[0]
    o.fix_linearization(ids)
    ex, fo = o.export_window(), o.get_frames()
    w2 = copy.deepcopy(win)
    w2.points, w2.residuals, w2.lin_J, w2.lin_res_toZeroF = ex["points"], ex["residuals"], ex["lin_J"], ex["lin_res_toZeroF"]
    w2.frames = fo["frames"].copy()
    w2.calib = w2.calib.copy(); w2.calib["value"] = fo["calib_value"]
    rng = np.random.default_rng(perturb_seed)
    for k in range(1, win.F):
        w2.frames["state"][k][0:3] += rng.normal(0, 1e-4, 3)
        w2.frames["state"][k][3:6] += rng.normal(0, 1e-5, 3)
    w2.points["idepth"] = (w2.points["idepth"] * (1 + rng.normal(0, 5e-4, w2.P))).astype(np.float32)
    o.close()
    return w2


def _track_new_coarse(fn, h, sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold, poses_valid, with_tries):
    P = [np.ascontiguousarray(np.asarray(m, np.float64)[:3, :4]) for m in (sprelast, slast, lastF)]
    aff = np.asarray(aff_last, np.float32); rmse = np.ascontiguousarray(last_rmse, np.float64).copy()
    res4 = np.zeros(4); w2c = np.zeros((3, 4)); aff_out = np.zeros(2, np.float32); tries = C.c_int(-1)
    args = [h, _p(P[0]), _p(P[1]), _p(P[2]), C.c_int(1 if poses_valid else 0), _p(aff), _p(rmse), C.c_double(retrack_threshold), _p(res4), _p(w2c), _p(aff_out)]
    if with_tries:
        args.append(C.byref(tries))
    good = fn(*args)
    return dict(result=res4, w2c=w2c, aff=aff_out, lastCoarseRMSE=rmse, tries=tries.value, good=good)


class OracleTracker:
    def __init__(self, w, h, levels, settings, calib, fast=False):
        self.L = lib(fast)
        self.w, self.h_, self.levels = w, h, levels
        s = np.ascontiguousarray(settings)
        c = np.ascontiguousarray(calib)
        self.h = C.c_void_p(self.L.orc_tr_create(C.c_int(w), C.c_int(h), C.c_int(levels), _p(s), _p(c)))
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_tr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_ref(self, pyr, a, b, exposure, pts):
        arr, keep = _img_ptrs([pyr], self.levels)
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        self.L.orc_tr_set_ref(self.h, arr, C.c_float(a), C.c_float(b), C.c_float(exposure), _p(pts), C.c_int(len(pts)))

    def set_new_frame(self, pyr, exposure=1.0):
        arr, keep = _img_ptrs([pyr], self.levels)
        self.L.orc_tr_set_new_frame(self.h, arr, C.c_float(exposure))

    def pc(self, lvl):
        n = self.L.orc_tr_pc_n(self.h, C.c_int(lvl))
        u, v, d, c = (np.zeros(n, np.float32) for _ in range(4))
        self.L.orc_tr_get_pc(self.h, C.c_int(lvl), _p(u), _p(v), _p(d), _p(c))
        return u, v, d, c

    def K(self):
        fx, fy, cx, cy = (np.zeros(self.levels, np.float32) for _ in range(4))
        self.L.orc_tr_get_K(self.h, _p(fx), _p(fy), _p(cx), _p(cy))
        return fx, fy, cx, cy

    def calc_res(self, lvl, T, a, b, cutoff):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64)
        rs = np.zeros(6)
        n = self.L.orc_tr_calc_res(self.h, C.c_int(lvl), _p(T), C.c_float(a), C.c_float(b), C.c_float(cutoff), _p(rs))
        return rs, n

    def warped(self, n):
        bufs = [np.zeros(n, np.float32) for _ in range(8)]
        self.L.orc_tr_get_warped(self.h, *[_p(b) for b in bufs])
        return dict(zip(["idepth", "u", "v", "dx", "dy", "residual", "weight", "refColor"], bufs))

    def calc_gs(self, lvl, T, a, b):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64)
        H = np.zeros((8, 8))
        bb = np.zeros(8)
        self.L.orc_tr_calc_gs(self.h, C.c_int(lvl), _p(T), C.c_float(a), C.c_float(b), _p(H), _p(bb))
        return H, bb

    def track_new_coarse(self, sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold=1.5, poses_valid=True):
        """FullSystem::trackNewCoarse restated (tracker_capi.inc): poses are worldToCam 4x4 / 3x4."""
        return _track_new_coarse(self.L.orc_tr_track_new_coarse, self.h, sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold, poses_valid, True)

    def motion_hypotheses(self, sprelast, slast, lastF, poses_valid=True):
        P = [np.ascontiguousarray(np.asarray(m, np.float64)[:3, :4]) for m in (sprelast, slast, lastF)]
        out = np.zeros((83, 3, 4))
        n = self.L.orc_tr_motion_hypotheses(_p(P[0]), _p(P[1]), _p(P[2]), C.c_int(1 if poses_valid else 0), _p(out))
        return out[:n]

    def track(self, T, a, b, coarsest, min_res=None):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64).copy()
        ab = np.array([a, b], np.float32)
        mr = np.full(5, np.nan) if min_res is None else np.asarray(min_res, np.float64)
        lr = np.zeros(5)
        fl = np.zeros(3)
        its = C.c_int()
        ok = self.L.orc_tr_track(self.h, _p(T), _p(ab), C.c_int(coarsest), _p(mr), _p(lr), _p(fl), C.byref(its))
        return dict(ok=bool(ok), T=T, a=float(ab[0]), b=float(ab[1]), lastResiduals=lr, flow=fl, iterations=its.value)

    def time_track(self, T, a, b, coarsest, reps):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64)
        ab = np.array([a, b], np.float32)
        return float(self.L.orc_tr_time_track(self.h, _p(T), _p(ab), C.c_int(coarsest), C.c_int(reps)))


def trace_on(points, dI_level0, KRKi, Kt, aff, settings=None):
    """FullSystem::traceNewCoarse over immature-point records (ldso_amd.synth.IMMATURE_DTYPE, modified in place) -> counts[6]"""
    L = lib()
    s = np.ascontiguousarray(synth.default_trace_settings() if settings is None else settings)
    img = np.ascontiguousarray(dI_level0, np.float32)
    h, w = img.shape[:2]
    K1 = np.ascontiguousarray(KRKi, np.float32); K2 = np.ascontiguousarray(Kt, np.float32); A = np.ascontiguousarray(aff, np.float32)
    counts = np.zeros(6, np.int32)
    assert points.flags["C_CONTIGUOUS"] and points.dtype == synth.IMMATURE_DTYPE
    L.orc_trace_on(C.c_int(len(points)), _p(points), _p(img), C.c_int(w), C.c_int(h), C.c_int(len(K1)), _p(K1), _p(K2), _p(A), _p(s), _p(counts))
    return counts


def activate_points(points, images_level0, K4, pairs, w, h, huberTH=9.0, min_idepth_hessian=100.0, gn_iterations=3, min_obs=1):
    """FullSystem::optimizeImmaturePoint over immature-point records.  images_level0: list of F arrays [h,w,3]; K4 = (fx,fy,cx,cy);
    pairs [F*F,14] = R (9), t (3), aff (2) of the current state at [host*F + target]."""
    L = lib()
    pts = np.ascontiguousarray(points)
    F = len(images_level0)
    imgs = [np.ascontiguousarray(im, np.float32) for im in images_level0]
    arr = (C.c_void_p * F)(*[im.ctypes.data for im in imgs])
    K = np.ascontiguousarray(K4, np.float32); pr = np.ascontiguousarray(pairs, np.float32)
    out = np.zeros(len(pts), synth.ACTIVATION_DTYPE)
    L.orc_activate_points(C.c_int(len(pts)), _p(pts), C.c_int(F), arr, C.c_int(w), C.c_int(h), _p(K), _p(pr), C.c_float(huberTH),
                          C.c_float(min_idepth_hessian), C.c_int(gn_iterations), C.c_int(min_obs), _p(out))
    return out


def make_images(color, levels):
    L = lib()
    h, w = color.shape
    outs = [np.zeros(((h >> l), (w >> l), 3), np.float32) for l in range(levels)]
    arr = (C.c_void_p * levels)(*[o.ctypes.data for o in outs])
    c = np.ascontiguousarray(color, np.float32)
    L.orc_make_images(_p(c), C.c_int(w), C.c_int(h), C.c_int(levels), arr)
    return outs


def nullspaces(T_w2c, aff_a0, exposure):
    L = lib()
    T = np.ascontiguousarray(T_w2c[:3, :4], np.float64)
    p, s, a = np.zeros(36), np.zeros(6), np.zeros(8)
    L.orc_nullspaces(_p(T), C.c_float(aff_a0), C.c_float(exposure), _p(p), _p(s), _p(a))
    return p.reshape(6, 6), s, a.reshape(4, 2)


class OracleInitializer:
    """CoarseInitializer restatement (oracle/initializer.cc)."""

    def __init__(self, w, h, levels):
        self.L = lib()
        L = self.L
        L.orc_init_create.restype = C.c_void_p
        L.orc_init_track_frame.restype = C.c_int
        self.w, self.h_, self.levels = w, h, levels
        self.h = C.c_void_p(L.orc_init_create(C.c_int(w), C.c_int(h), C.c_int(levels)))
        self.n = [0] * levels

    def close(self):
        if self.h:
            self.L.orc_init_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_first(self, K4, pyr, exposure, points, huberTH=9.0, fixAffine=True):
        arr, keep = _img_ptrs([pyr], self.levels)
        pts = [np.ascontiguousarray(p, dtype=synth.INIT_POINT_DTYPE) for p in points]
        pp = (C.c_void_p * self.levels)(*[p.ctypes.data for p in pts])
        n = np.array([len(p) for p in pts], dtype=np.int32)
        self.n = [int(x) for x in n]
        k = np.ascontiguousarray(K4, dtype=np.float32)
        self.L.orc_init_set_first(self.h, _p(k), arr, C.c_float(exposure), pp, _p(n), C.c_float(huberTH), C.c_int(1 if fixAffine else 0))

    def set_new_frame(self, pyr, exposure=1.0):
        arr, keep = _img_ptrs([pyr], self.levels)
        self.L.orc_init_set_new_frame(self.h, arr, C.c_float(exposure))

    def track_frame(self):
        st = np.zeros((), synth.INIT_STATE_DTYPE)
        self.L.orc_init_track_frame(self.h, _p(st))
        return st

    def state(self):
        st = np.zeros((), synth.INIT_STATE_DTYPE)
        self.L.orc_init_get_state(self.h, _p(st))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=synth.INIT_STATE_DTYPE)
        self.L.orc_init_set_state(self.h, _p(st))

    def points(self, lvl):
        out = np.zeros(self.n[lvl], synth.INIT_POINT_DTYPE)
        self.L.orc_init_get_points(self.h, C.c_int(lvl), _p(out))
        return out

    def set_points(self, lvl, pts):
        pts = np.ascontiguousarray(pts, dtype=synth.INIT_POINT_DTYPE)
        assert len(pts) == self.n[lvl]
        self.L.orc_init_set_points(self.h, C.c_int(lvl), _p(pts))

    def calc_res_and_gs(self, lvl, T, a, b):
        T = np.ascontiguousarray(np.asarray(T, dtype=np.float64)[:3, :4])
        H = np.zeros((8, 8), np.float32); bo = np.zeros(8, np.float32); Hsc = np.zeros((8, 8), np.float32); bsc = np.zeros(8, np.float32)
        res = np.zeros(3, np.float32); ec = np.zeros(3, np.float32)
        self.L.orc_init_calc_res_and_gs(self.h, C.c_int(lvl), _p(T), C.c_double(a), C.c_double(b), _p(H), _p(bo), _p(Hsc), _p(bsc), _p(res), _p(ec))
        return H, bo, Hsc, bsc, res, ec

    def jb(self, lvl):
        out = np.zeros((self.n[lvl], 10), np.float32)
        self.L.orc_init_get_jb(self.h, C.c_int(lvl), _p(out))
        return out
