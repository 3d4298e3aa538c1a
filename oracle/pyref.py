"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/_ref/libldso_ref.so: the reference's own hot-path translation
units compiled unmodified from /root/reference against the header shim oracle/ref_shim (see oracle/ref_driver.cc).  Built by
`make -C oracle ref` where /root/reference exists (this container); the .so then travels to the GPU box with the repo snapshot.
Only tests/ use it: it pins the oracle (oracle/*.cc, the CPU restatement) to reference-compiled code."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from ldso_amd import synth
from .pyoracle import _p, _img_ptrs, _track_new_coarse

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path() -> str:
    # LDSO_REF_LIB: another build of the reference-compiled library (e.g. an AddressSanitizer build)
    return os.environ.get("LDSO_REF_LIB") or os.path.join(_HERE, "_ref", "libldso_ref.so")


def available() -> bool:
    """True when the reference-compiled library exists or can be built here (needs /root/reference)."""
    if os.path.exists(lib_path()):
        return True
    if not os.path.isdir("/root/reference"):
        return False
    try:
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)
    except Exception:
        return False
    return os.path.exists(lib_path())


_LIB_FAST = None


def lib_fast():
    """The same reference translation units at the reference's own optimisation level (-O3, x86-64-v3): bench.py's cpu_baseline only.
    Returns None where the file is missing or the host lacks AVX2 / FMA (then the pin build is timed and the JSON says so)."""
    global _LIB_FAST
    if _LIB_FAST is None:
        path = os.path.join(_HERE, "_ref", "libldso_ref_fast.so")
        if not os.path.exists(path) and os.path.isdir("/root/reference"):
            try:
                subprocess.run(["make", "-C", _HERE, "ref_fast"], check=True, capture_output=True)
            except Exception:
                return None
        try:
            flags = open("/proc/cpuinfo").read()
        except OSError:
            flags = ""
        if not os.path.exists(path) or " avx2" not in flags or " fma" not in flags:
            return None
        L = C.CDLL(path)
        L.ref_create.restype = C.c_void_p
        L.ref_fs_time_optimize.restype = C.c_double
        L.ref_fs_optimize.restype = C.c_float
        L.ref_linearize_all.restype = C.c_double
        _LIB_FAST = L
    return _LIB_FAST


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise RuntimeError("oracle/_ref/libldso_ref.so is missing and /root/reference is not here to build it")
        L = C.CDLL(lib_path())
        L.ref_create.restype = C.c_void_p
        L.ref_linearize_all.restype = C.c_double
        L.ref_fs_optimize.restype = C.c_float
        L.ref_fs_time_optimize.restype = C.c_double
        _LIB = L
    return _LIB


class RefWindow:
    """The reference's EnergyFunctional / PointFrameResidual / accumulators on one flattened window (same inputs as OracleWindow)."""

    def __init__(self, win: synth.Window, fast: bool = False):
        self.L = lib_fast() if fast else lib()
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
        self.h = C.c_void_p(self.L.ref_create(
            C.c_int(win.w), C.c_int(win.h), C.c_int(win.levels), _p(s), _p(c), C.c_int(self.F), _p(self.frames), imgs,
            C.c_int(self.P), _p(self.points), C.c_int(self.R), _p(self.residuals), _p(linJ), _p(rtz), _p(HM), _p(bM), C.c_int(0)))

    def close(self):
        if self.h:
            self.L.ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- the reference's own FullSystem members on this window (FullSystem.cc compiled unmodified; ref_driver.cc ref_fs_*) ---------
    def fs_attach(self, multithreading=None):
        # the choice sticks to the window: later helpers re-attach without an argument (fs_handle) and must not switch the reference's
        # process-wide `multiThreading` back
        if multithreading is not None:
            self._mt = bool(multithreading)
        self.L.ref_fs_attach(self.h, C.c_int(1 if getattr(self, "_mt", False) else 0))

    def fs_log(self) -> str:
        n = self.L.ref_fs_log(self.h, None, C.c_int(0))
        buf = C.create_string_buffer(n + 1)
        self.L.ref_fs_log(self.h, buf, C.c_int(n + 1))
        return buf.value.decode(errors="replace")

    def fs_optimize(self, iterations: int):
        """FullSystem::optimize (FullSystem.cc:725-864).  Returns (return value, energies printed by printOptRes: one before the loop
        and one per executed iteration - the only place the reference reports them)."""
        import re
        rv = float(self.L.ref_fs_optimize(self.h, C.c_int(iterations)))
        log = self.fs_log()
        energies = [float(m) for m in re.findall(r"A\(([-0-9.eE+naif]+)\)=\(AV", log)]
        return rv, np.array(energies)

    def fs_time_optimize(self, iterations: int) -> float:
        return float(self.L.ref_fs_time_optimize(self.h, C.c_int(iterations)))

    def fs_is_lost(self) -> bool:
        return bool(self.L.ref_fs_is_lost(self.h))

    def fs_collect_active(self, reset_oob=True):
        self.L.ref_fs_collect_active(self.h, C.c_int(1 if reset_oob else 0))

    def fs_linearize_all(self, fix=False):
        out = np.zeros(3)
        self.L.ref_fs_linearize_all(self.h, C.c_int(1 if fix else 0), _p(out))
        return out

    def fs_apply_res(self):
        self.L.ref_fs_apply_res(self.h)

    def fs_set_new_frame_energy_th(self):
        self.L.ref_fs_set_new_frame_energy_th(self.h)

    def fs_backup_state(self, backup_last_step=False):
        self.L.ref_fs_backup_state(self.h, C.c_int(1 if backup_last_step else 0))

    def fs_do_step(self, stepfac=1.0) -> bool:
        return bool(self.L.ref_fs_do_step(self.h, C.c_float(stepfac)))

    def fs_load_state_backup(self):
        self.L.ref_fs_load_state_backup(self.h)

    def fs_solve_system(self, iteration: int, lam: float = 1e-1):
        self.L.ref_fs_solve_system(self.h, C.c_int(iteration), C.c_double(lam))

    def fs_calc_energies(self):
        el, em = C.c_double(), C.c_double()
        self.L.ref_fs_calc_energies(self.h, C.byref(el), C.byref(em))
        return el.value, em.value

    def fs_flag_frame(self, idx):
        self.L.ref_fs_flag_frame(self.h, C.c_int(idx))

    def fs_flag_points_for_removal(self):
        self.L.ref_fs_flag_points_for_removal(self.h)

    def fs_marginalize_frame(self, idx):
        self.L.ref_fs_marginalize_frame(self.h, C.c_int(idx))

    # --- FullSystem::traceNewCoarse on this window: immature points on the key frames, a new frame with a pose ---
    def fs_add_immature(self, points):
        pts = np.ascontiguousarray(points)
        self.L.ref_fs_add_immature(self.h, C.c_int(len(pts)), _p(pts))

    def fs_new_frame(self, dI_level0, w2c, aff_a, aff_b, exposure=1.0):
        self.L.ref_fs_new_frame.restype = C.c_void_p
        img = np.ascontiguousarray(dI_level0, np.float32); T = np.ascontiguousarray(np.asarray(w2c, np.float64)[:3, :4])
        return C.c_void_p(self.L.ref_fs_new_frame(self.h, _p(img), _p(T), C.c_float(aff_a), C.c_float(aff_b), C.c_float(exposure)))

    def fs_set_frame_id(self, fh, frame_id):
        self.L.ref_fs_set_frame_id(fh, C.c_long(frame_id))

    def fs_handle(self):
        self.fs_attach()
        self.L.ref_fs_handle.restype = C.c_void_p
        return C.c_void_p(self.L.ref_fs_handle(self.h))

    def fs_trace_new_coarse(self, fh):
        self.L.ref_fs_trace_new_coarse(self.h, fh)

    def fs_get_immature(self, cap=100000):
        out = np.zeros(cap, synth.IMMATURE_DTYPE)
        n = self.L.ref_fs_get_immature(self.h, _p(out), C.c_int(cap))
        return out[:n].copy()

    def get_pair_rt(self):
        out = np.zeros((self.num_frames() ** 2, 14), np.float32)
        self.L.ref_get_pair_rt(self.h, _p(out))
        return out

    def fs_activate_points(self, points, min_obs=1, min_idepth_hessian=100.0, gn_iterations=3):
        """FullSystem::optimizeImmaturePoint (the member) per record: ok, idepth (NaN when rejected), res_state, numGoodRes; the function's
        locals (energy, Hdd, bd, iterations) are not observable from outside and stay NaN / -1."""
        pts = np.ascontiguousarray(points)
        out = np.zeros(len(pts), synth.ACTIVATION_DTYPE)
        self.L.ref_fs_activate_points(self.h, C.c_int(len(pts)), _p(pts), C.c_int(min_obs), C.c_float(min_idepth_hessian), C.c_int(gn_iterations), _p(out))
        return out

    def collect_active(self, reset_oob=True):
        self.L.ref_collect_active(self.h, C.c_int(1 if reset_oob else 0))

    def linearize_all(self) -> float:
        return float(self.L.ref_linearize_all(self.h))

    def apply_res(self):
        self.L.ref_apply_res(self.h)

    def set_frame_energy_th(self, f, th):
        self.L.ref_set_frame_energy_th(self.h, C.c_int(f), C.c_float(th))

    def solve_system(self, iteration: int, lam: float = 1e-1):
        self.L.ref_solve_system(self.h, C.c_int(iteration), C.c_double(lam))

    def calc_lm_energies(self):
        m, l = C.c_double(), C.c_double()
        self.L.ref_calc_lm_energies(self.h, C.byref(m), C.byref(l))
        return m.value, l.value

    def num_frames(self):
        return int(self.L.ref_num_frames(self.h))

    def counts(self):
        a, l, m = C.c_int(), C.c_int(), C.c_int()
        self.L.ref_counts(self.h, C.byref(a), C.byref(l), C.byref(m))
        return a.value, l.value, m.value

    def get_residuals(self, with_J=True):
        out = np.zeros(self.R, synth.RES_OUT_DTYPE)
        J = np.zeros(self.R, synth.RAWJAC_DTYPE) if with_J else None
        st = np.zeros(self.R, np.int32); act = np.zeros(self.R, np.int32)
        rtz = np.zeros((self.R, 8), np.float32); lin = np.zeros(self.R, np.int32); alive = np.zeros(self.R, np.int32)
        self.L.ref_get_residuals(self.h, _p(out), _p(J), _p(st), _p(act), _p(rtz), _p(lin), _p(alive))
        return dict(out=out, J=J, state_state=st, is_active=act, res_toZeroF=rtz, is_linearized=lin, alive=alive)

    def get_points(self):
        out = np.zeros(self.P, synth.POINT_OUT_DTYPE)
        status = np.zeros(self.P, np.int32)
        self.L.ref_get_points(self.h, _p(out), _p(status))
        return out, status

    def get_frames(self):
        F = self.num_frames()
        fr = np.zeros(F, synth.FRAME_DTYPE); step = np.zeros((F, 10)); cv = np.zeros(4); cs = np.zeros(4); pre = np.zeros((F, 12))
        self.L.ref_get_frames(self.h, _p(fr), _p(step), _p(cv), _p(cs), _p(pre))
        return dict(frames=fr, step=step, calib_value=cv, calib_step=cs, pre_worldToCam=pre)

    def get_precalc(self):
        F = self.num_frames()
        out = np.zeros((F, F, 27), np.float32)
        self.L.ref_get_precalc(self.h, _p(out))
        return out

    def get_adjoints(self):
        F = self.num_frames()
        ah = np.zeros((F * F, 8, 8)); at = np.zeros((F * F, 8, 8)); d = np.zeros((F * F, 8), np.float32)
        self.L.ref_get_adjoints(self.h, _p(ah), _p(at), _p(d))
        return ah, at, d

    def get_accumulators(self):
        F = self.num_frames()
        d = dict(topA=np.zeros((F * F, 13, 13), np.float32), topL=np.zeros((F * F, 13, 13), np.float32),
                 accD=np.zeros((F * F * F, 8, 8), np.float32), accE=np.zeros((F * F, 8, 4), np.float32),
                 accEB=np.zeros((F * F, 8), np.float32), accHcc=np.zeros((4, 4), np.float32), accbc=np.zeros(4, np.float32))
        self.L.ref_get_accumulators(self.h, _p(d["topA"]), _p(d["topL"]), _p(d["accD"]), _p(d["accE"]), _p(d["accEB"]), _p(d["accHcc"]), _p(d["accbc"]))
        return d

    def get_system(self):
        n = 8 * self.num_frames() + 4
        d = dict(lastHS=np.zeros((n, n)), lastbS=np.zeros(n), x=np.zeros(n))
        self.L.ref_get_system(self.h, _p(d["lastHS"]), _p(d["lastbS"]), _p(d["x"]))
        return d

    def get_prior(self):
        n = 8 * self.num_frames() + 4
        HM = np.zeros((n, n)); bM = np.zeros(n)
        self.L.ref_get_prior(self.h, _p(HM), _p(bM))
        return HM, bM

    def flag_points(self, status):
        st = np.ascontiguousarray(status, np.int32)
        self.L.ref_flag_points(self.h, _p(st))

    def drop_points(self):
        self.L.ref_drop_points(self.h)

    def marginalize_points(self):
        self.L.ref_marginalize_points(self.h)

    def marginalize_frame(self, idx):
        self.L.ref_marginalize_frame(self.h, C.c_int(idx))


# ---- the compiled drop-in adapter (adapter/ldso_gpu_adapter.cc -> adapter/_build/libldso_gpu_adapter.so) on reference object graphs, driven through the
# ---- test-only C face adapter/adapter_capi.cc -> adapter/_build/libldso_adapter_test.so -------------------------------------------------------
_ADP = None


def adapter_path() -> str:
    # LDSO_ADAPTER_LIB: another build of the harness library (e.g. an AddressSanitizer build, run with LD_PRELOAD=libasan.so)
    return os.environ.get("LDSO_ADAPTER_LIB") or os.path.join(_HERE, "..", "adapter", "_build", "libldso_adapter_test.so")


def adapter_available() -> bool:
    if os.path.exists(adapter_path()):
        return True
    if not os.path.isdir("/root/reference") or not available():
        return False
    try:
        subprocess.run(["make", "-C", os.path.join(_HERE, "..", "adapter")], check=True, capture_output=True)
    except Exception:
        return False
    return os.path.exists(adapter_path())


def adapter_lib():
    global _ADP
    if _ADP is None:
        from ldso_amd import binding
        binding.lib()                      # the product library first (it decides how the HIP runtime is brought in)
        lib()
        A = C.CDLL(adapter_path())
        A.adp_create.restype = C.c_void_p
        A.adp_last_error.restype = C.c_char_p
        _ADP = A
        L = lib()
        for f in ("ref_fs_handle", "ref_tr_prepare", "ref_tr_coarse_tracker", "ref_tr_frame_hessians", "ref_tr_new_frame_hessian", "ref_tr_calib_hessian"):
            getattr(L, f).restype = C.c_void_p
    return _ADP


class GpuAdapter:
    """ldso::GpuBackend (adapter/ldso_gpu_adapter.h) driven on the reference objects of a RefWindow / RefTracker."""

    def __init__(self, max_frames=16, max_points=20000, device=0):
        self.A = adapter_lib()
        self.h = C.c_void_p(self.A.adp_create(C.c_int(device), C.c_int(max_frames), C.c_int(max_points)))
        if not self.h:
            raise RuntimeError(self.A.adp_last_error().decode())

    def close(self):
        if self.h:
            self.A.adp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.A.adp_last_error().decode())

    def optimize(self, ref_window: "RefWindow", iterations: int):
        """GpuBackend::optimize(fs, n) in place of FullSystem::optimize on the window's reference objects -> (rmse, iterations, lost)"""
        ref_window.fs_attach()
        fs = C.c_void_p(ref_window.L.ref_fs_handle(ref_window.h))
        rmse, its, lost = C.c_float(), C.c_int(), C.c_int()
        self._chk(self.A.adp_optimize(self.h, fs, C.c_int(iterations), C.byref(rmse), C.byref(its), C.byref(lost)))
        ref_window.L.ref_fs_sync_back(ref_window.h)
        return rmse.value, its.value, bool(lost.value)

    def activate_points(self, ref_window: "RefWindow", points, min_idepth_hessian=100.0, gn_iterations=3):
        """GpuBackend::activatePoints on ImmaturePoint objects built from the records on the window's reference frames -> ok, idepth, per-frame
        residual state of the created points (-1: no residual), states of lastResiduals[0..1]"""
        ref_window.fs_attach()
        L = ref_window.L
        L.ref_fs_build_immature.restype = C.c_void_p
        pts = np.ascontiguousarray(points); n = len(pts); F = ref_window.num_frames()
        vec = C.c_void_p(L.ref_fs_build_immature(ref_window.h, C.c_int(n), _p(pts), C.c_int(1), C.c_float(min_idepth_hessian), C.c_int(gn_iterations)))
        fs = C.c_void_p(L.ref_fs_handle(ref_window.h))
        ok = np.zeros(n, np.int32); idepth = np.zeros(n, np.float32); tgt = np.zeros((n, F), np.int32); last = np.zeros((n, 2), np.int32)
        try:
            self._chk(self.A.adp_activate_points(self.h, fs, vec, C.c_int(n), C.c_int(F), _p(ok), _p(idepth), _p(tgt), _p(last)))
        finally:
            L.ref_fs_free_immature(vec)
        return dict(ok=ok, idepth=idepth, res_target=tgt, last=last)

    def set_device_pyramids(self, on: bool):
        self.A.adp_set_device_pyramids(self.h, C.c_int(1 if on else 0))

    def pyramids_built(self) -> int:
        return int(self.A.adp_pyramids_built(self.h))

    def set_resident_window(self, on: bool):
        self.A.adp_set_resident_window(self.h, C.c_int(1 if on else 0))

    def upload_counts(self):
        d, f = C.c_int(), C.c_int()
        self.A.adp_upload_counts(self.h, C.byref(d), C.byref(f))
        return d.value, f.value

    def set_write_back_jacobians(self, on: bool):
        self.A.adp_set_write_back_jacobians(self.h, C.c_int(1 if on else 0))

    def last_optimize_times(self):
        """wall-clock split of the last GpuBackend::optimize (s): flatten + upload, device, fetch, write-back"""
        t = np.zeros(4, np.float64)
        self.A.adp_last_optimize_times(self.h, _p(t))
        return t

    def last_upload_times(self):
        """split of the flatten + upload part (s): settings + image slots, host walk, set_window, set_point_stats, set_frames, set_prior"""
        t = np.zeros(6, np.float64)
        self.A.adp_last_upload_times(self.h, _p(t))
        return t

    def trace_new_coarse(self, ref_window: "RefWindow", fh):
        """GpuBackend::traceNewCoarse(fs, fh) in place of FullSystem::traceNewCoarse -> the six status counters"""
        ref_window.fs_attach()
        fs = C.c_void_p(ref_window.L.ref_fs_handle(ref_window.h))
        counts = np.zeros(6, np.int32)
        self._chk(self.A.adp_trace_new_coarse(self.h, fs, fh, _p(counts)))
        return counts

    def track_new_coarse(self, ref_tracker: "RefTracker", sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold=1.5, poses_valid=True):
        """GpuBackend::makeK + setCoarseTrackingRef + trackNewCoarse in place of FullSystem::trackNewCoarse"""
        L, h = ref_tracker.L, ref_tracker.h
        P = [np.ascontiguousarray(np.asarray(m, np.float64)[:3, :4]) for m in (sprelast, slast, lastF)]
        aff = np.asarray(aff_last, np.float32); rmse = np.ascontiguousarray(last_rmse, np.float64).copy()
        fs = C.c_void_p(L.ref_tr_prepare(h, _p(P[0]), _p(P[1]), _p(P[2]), C.c_int(1 if poses_valid else 0), _p(aff), _p(rmse), C.c_double(retrack_threshold)))
        res4 = np.zeros(4)
        self._chk(self.A.adp_track_new_coarse(self.h, fs, C.c_void_p(L.ref_tr_coarse_tracker(h)), C.c_void_p(L.ref_tr_frame_hessians(h)),
                                              C.c_void_p(L.ref_tr_new_frame_hessian(h)), C.c_void_p(L.ref_tr_calib_hessian(h)), _p(res4)))
        w2c = np.zeros((3, 4)); aff_out = np.zeros(2, np.float32)
        L.ref_tr_read_result(h, _p(rmse), _p(w2c), _p(aff_out))
        return dict(result=res4, w2c=w2c, aff=aff_out, lastCoarseRMSE=rmse)

    def track_newest_coarse(self, ref_tracker: "RefTracker", T, a, b, coarsest, min_res=None):
        L, h = ref_tracker.L, ref_tracker.h
        Tm = np.ascontiguousarray(np.asarray(T, np.float64)[:3, :4]).copy(); ab = np.array([a, b], np.float32)
        mr = np.full(5, np.nan) if min_res is None else np.ascontiguousarray(min_res, np.float64)
        lr = np.zeros(5); ok = C.c_int()
        self._chk(self.A.adp_track_newest_coarse(self.h, C.c_void_p(L.ref_tr_coarse_tracker(h)), C.c_void_p(L.ref_tr_frame_hessians(h)), C.c_void_p(L.ref_tr_new_frame_hessian(h)),
                                                 C.c_void_p(L.ref_tr_calib_hessian(h)), _p(Tm), _p(ab), C.c_int(coarsest), _p(mr), _p(lr), C.byref(ok)))
        return dict(T=Tm, a=float(ab[0]), b=float(ab[1]), ok=bool(ok.value), lastResiduals=lr)


def make_images(color, levels):
    """FrameHessian::makeImages of the reference on a raw irradiance image: list of [h_l, w_l, 3] float32 levels."""
    L = lib()
    h, w = color.shape
    col = np.ascontiguousarray(color, np.float32)
    outs = [np.zeros(((h >> l), (w >> l), 3), np.float32) for l in range(levels)]
    arr = (C.c_void_p * levels)(*[o.ctypes.data for o in outs])
    L.ref_make_images(C.c_int(w), C.c_int(h), C.c_int(levels), _p(col), arr)
    return outs


class RefTracker:
    """The reference's CoarseTracker (src/frontend/CoarseTracker.cc compiled unmodified), same inputs as pyoracle.OracleTracker."""

    def __init__(self, w, h, levels, settings, calib, fast=False):
        # fast: the -O3 build of the same translation units (bench.py's tracker cpu_baseline: the reference's own CoarseTracker at its own optimisation level)
        self.L = lib_fast() if fast else lib()
        if self.L is None:
            raise RuntimeError("oracle/_ref/libldso_ref_fast.so is not available on this host")
        self.L.ref_tr_create.restype = C.c_void_p
        self.levels = levels
        s = np.ascontiguousarray(settings); c = np.ascontiguousarray(calib)
        self.h = C.c_void_p(self.L.ref_tr_create(C.c_int(w), C.c_int(h), C.c_int(levels), _p(s), _p(c)))

    def close(self):
        if self.h:
            self.L.ref_tr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_ref(self, pyr, a, b, exposure, pts):
        arr, keep = _img_ptrs([pyr], self.levels)
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        self.L.ref_tr_set_ref(self.h, arr, C.c_float(a), C.c_float(b), C.c_float(exposure), _p(pts), C.c_int(len(pts)))

    def set_new_frame(self, pyr, exposure=1.0):
        arr, keep = _img_ptrs([pyr], self.levels)
        self.L.ref_tr_set_new_frame(self.h, arr, C.c_float(exposure))

    def pc(self, lvl):
        n = self.L.ref_tr_pc_n(self.h, C.c_int(lvl))
        u, v, d, c = (np.zeros(n, np.float32) for _ in range(4))
        self.L.ref_tr_get_pc(self.h, C.c_int(lvl), _p(u), _p(v), _p(d), _p(c))
        return u, v, d, c

    def K(self):
        fx, fy, cx, cy = (np.zeros(self.levels, np.float32) for _ in range(4))
        self.L.ref_tr_get_K(self.h, _p(fx), _p(fy), _p(cx), _p(cy))
        return fx, fy, cx, cy

    def calc_res(self, lvl, T, a, b, cutoff):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64)
        rs = np.zeros(6)
        n = self.L.ref_tr_calc_res(self.h, C.c_int(lvl), _p(T), C.c_float(a), C.c_float(b), C.c_float(cutoff), _p(rs))
        return rs, n

    def warped(self, n):
        bufs = [np.zeros(n, np.float32) for _ in range(8)]
        self.L.ref_tr_get_warped(self.h, *[_p(b) for b in bufs])
        return dict(zip(["idepth", "u", "v", "dx", "dy", "residual", "weight", "refColor"], bufs))

    def calc_gs(self, lvl, T, a, b):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64)
        H = np.zeros((8, 8)); bb = np.zeros(8)
        self.L.ref_tr_calc_gs(self.h, C.c_int(lvl), _p(T), C.c_float(a), C.c_float(b), _p(H), _p(bb))
        return H, bb

    def track_new_coarse(self, sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold=1.5, poses_valid=True):
        """Vec4 FullSystem::trackNewCoarse (the member) with this tracker: poses are worldToCam 4x4 / 3x4."""
        return _track_new_coarse(self.L.ref_tr_track_new_coarse, self.h, sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold, poses_valid, False)

    def track(self, T, a, b, coarsest, min_res=None):
        T = np.ascontiguousarray(T[:3, :4], dtype=np.float64).copy()
        ab = np.array([a, b], np.float32)
        mr = np.full(5, np.nan) if min_res is None else np.asarray(min_res, np.float64)
        lr = np.zeros(5); fl = np.zeros(3)
        ok = self.L.ref_tr_track(self.h, _p(T), _p(ab), C.c_int(coarsest), _p(mr), _p(lr), _p(fl))
        return dict(ok=bool(ok), T=T, a=float(ab[0]), b=float(ab[1]), lastResiduals=lr, flow=fl)


def trace_on(points, dI_level0, KRKi, Kt, aff, settings=None):
    """ImmaturePoint::traceOn of the reference over immature-point records (modified in place) -> counts[6]; as pyoracle.trace_on."""
    L = lib()
    s = np.ascontiguousarray(synth.default_trace_settings() if settings is None else settings)
    img = np.ascontiguousarray(dI_level0, np.float32)
    h, w = img.shape[:2]
    K1 = np.ascontiguousarray(KRKi, np.float32); K2 = np.ascontiguousarray(Kt, np.float32); A = np.ascontiguousarray(aff, np.float32)
    counts = np.zeros(6, np.int32)
    assert points.flags["C_CONTIGUOUS"] and points.dtype == synth.IMMATURE_DTYPE
    L.ref_trace_on(C.c_int(len(points)), _p(points), _p(img), C.c_int(w), C.c_int(h), C.c_int(len(K1)), _p(K1), _p(K2), _p(A), _p(s), _p(counts))
    return counts


class RefInitializer:
    """The reference's CoarseInitializer::trackFrame (CoarseInitializer.cc compiled unmodified); points arrive as records."""

    def __init__(self, w, h, levels):
        self.L = lib()
        self.L.ref_init_create.restype = C.c_void_p
        self.levels = levels
        self.h = C.c_void_p(self.L.ref_init_create(C.c_int(w), C.c_int(h), C.c_int(levels)))
        self.n = [0] * levels

    def close(self):
        if self.h:
            self.L.ref_init_destroy(self.h)
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
        self.L.ref_init_set_first(self.h, _p(k), arr, C.c_float(exposure), pp, _p(n), C.c_float(huberTH), C.c_int(1 if fixAffine else 0))

    def track_frame(self, pyr, exposure=1.0):
        arr, keep = _img_ptrs([pyr], self.levels)
        st = np.zeros((), synth.INIT_STATE_DTYPE)
        self.L.ref_init_track_frame(self.h, arr, C.c_float(exposure), _p(st))
        return st

    def points(self, lvl):
        out = np.zeros(self.n[lvl], synth.INIT_POINT_DTYPE)
        self.L.ref_init_get_points(self.h, C.c_int(lvl), _p(out))
        return out


def set_device_marginalisation(on: bool):
    """adp_make_keyframe with a GpuAdapter: GpuBackend::flagPointsForRemoval + marginalizePoints (ldso_ba_marginalize_points on the window optimize() left
    resident) instead of the reference's flagPointsForRemoval / ef->marginalizePointsF on the host"""
    adapter_lib().adp_set_device_marginalisation(C.c_int(1 if on else 0))


def make_keyframe(ref_window: "RefWindow", adapter, fh, marg_idx: int, kf_id: int, iterations: int = 6):
    """one key frame in FullSystem::makeKeyFrame's order (adapter/adapter_capi.cc: adp_make_keyframe) on the window's reference objects: adapter = None
    runs the reference's own members, a GpuAdapter puts GpuBackend::traceNewCoarse / activatePoints / optimize in their place
    -> (rmse, dict(candidates, activated, new_residuals, points, lost))"""
    A = adapter_lib()
    fs = ref_window.fs_handle()
    rmse = C.c_float()
    st = np.zeros(8, np.int32)
    rc = A.adp_make_keyframe(adapter.h if adapter is not None else None, fs, fh, C.c_int(marg_idx), C.c_int(kf_id), C.c_int(iterations), C.byref(rmse), _p(st))
    if rc != 0:
        raise RuntimeError(A.adp_last_error().decode())
    ref_window.L.ref_fs_sync_back(ref_window.h)
    return rmse.value, dict(candidates=int(st[0]), activated=int(st[1]), new_residuals=int(st[2]), points=int(st[3]), lost=bool(st[4]))


def graph_summary(ref_window: "RefWindow", cap_frames=16):
    """frames: [F] rows of (id, camToWorld 3x4, a, b, active points, residuals, immature points), the prior HM / bM, the calibration, inverse depths + hosts"""
    A = adapter_lib()
    fs = ref_window.fs_handle()
    fr = np.zeros((cap_frames, 18), np.float64)
    n = 8 * cap_frames + 4
    HM = np.zeros(n * n, np.float64); bM = np.zeros(n, np.float64); cal = np.zeros(4, np.float64)
    F = A.adp_graph_summary(fs, C.c_int(cap_frames), _p(fr), _p(HM), _p(bM), _p(cal))
    n = 8 * F + 4
    idp = np.zeros(200000, np.float32); host = np.zeros(200000, np.int32); uv = np.zeros((200000, 2), np.float32)
    P = A.adp_graph_idepths(fs, C.c_int(len(idp)), _p(idp), _p(host), _p(uv))
    return dict(F=F, ids=fr[:F, 0].astype(int), c2w=fr[:F, 1:13].reshape(F, 3, 4), aff=fr[:F, 13:15], points=fr[:F, 15].astype(int), residuals=fr[:F, 16].astype(int),
                immature=fr[:F, 17].astype(int), HM=HM[:n * n].reshape(n, n).copy(), bM=bM[:n].copy(), calib=cal, idepth=idp[:P].copy(), host=host[:P].copy(), uv=uv[:P].copy())
