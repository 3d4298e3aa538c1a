"""ctypes binding of libldso_hip.so — the product path.  Every call goes through the C-ABI declared in
include/ldso_hip.h; there is NO CPU fallback: if the HIP library is missing or no GPU is visible the
constructors raise."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import synth

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LdsoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ldso_hip error {code}: {msg}")
        self.code = code


def lib_path() -> str:
    # LDSO_HIP_LIB: an alternative build of the same library (kernel experiments: scripts/build_variant.sh), never a CPU substitute
    return os.environ.get("LDSO_HIP_LIB") or os.path.join(_HERE, "libldso_hip.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: build it with `python -m ldso_amd.build` (hipcc, gfx950). There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64 (same SONAMEs as /opt/rocm).  Two HIP
        # runtimes in one process cannot both own the GPU, so when torch is part of the process (streams, RCCL) it
        # must be loaded first; libldso_hip.so then binds to the runtime that is already resident.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(p)
        L.ldso_last_error.restype = C.c_char_p
        L.ldso_ba_reduce_doubles.restype = C.c_size_t
        L.ldso_ba_gn_reduce_doubles.restype = C.c_size_t
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(code):
    if code != 0:
        raise LdsoError(code, lib().ldso_last_error().decode())


def default_settings() -> np.ndarray:
    s = np.zeros((), synth.SETTINGS_DTYPE)
    _chk(lib().ldso_settings_default(_p(s)))
    return s


class Pyramid:
    """FrameHessian::dIp of one frame resident in HBM (ldso_pyramid_t), shared zero-copy by Tracker / Tracer / BA."""

    def __init__(self, w, h, levels, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.ldso_pyr_create(C.c_int(device), C.c_int(w), C.c_int(h), C.c_int(levels), C.byref(self.h)))
        self.w, self.hh, self.levels = w, h, levels

    def close(self):
        if self.h:
            self.L.ldso_pyr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def make_images(self, irradiance, stream=None):
        a = np.ascontiguousarray(irradiance, np.float32)
        assert a.size == self.w * self.hh
        _chk(self.L.ldso_pyr_make_images(self.h, _p(a), C.c_void_p(stream)))
        return self

    def level_ptr(self, lvl):
        ptr, wl, hl = C.c_void_p(), C.c_int(), C.c_int()
        _chk(self.L.ldso_pyr_level(self.h, C.c_int(lvl), C.byref(ptr), C.byref(wl), C.byref(hl)))
        return ptr.value, wl.value, hl.value

    def get_level(self, lvl):
        out = np.zeros((self.hh >> lvl, self.w >> lvl, 3), np.float32)
        _chk(self.L.ldso_pyr_get_level(self.h, C.c_int(lvl), _p(out)))
        return out


class BA:
    """Windowed bundle adjustment handle (EnergyFunctional + FullSystem::optimize slice) on one GPU."""

    def __init__(self, w, h, max_frames, max_points, device=0, stream=None):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.ldso_ba_create(C.c_int(device), C.c_int(w), C.c_int(h), C.c_int(max_frames), C.c_int(max_points), C.byref(self.h)))
        self.w, self.hh = w, h
        self.F = self.P = self.R = 0
        if stream is not None:
            self.set_stream(stream)

    @classmethod
    def from_window(cls, win: synth.Window, device=0, stream=None, max_frames=None, max_points=None):
        ba = cls(win.w, win.h, max_frames or max(win.F, 2), max_points or win.P, device=device, stream=stream)
        ba.load_window(win)
        return ba

    def close(self):
        if self.h:
            self.L.ldso_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        """Run the handle on a caller's HIP stream.  The legacy default stream (handle 0, what torch.cuda.current_stream() is
        unless a torch.cuda.Stream was made current) cannot be passed through the C-ABI (NULL = "own stream") and would not
        order against the handle's own non-blocking stream: refuse it instead of racing silently."""
        if not stream_ptr:
            raise ValueError("pass a non-default HIP stream (e.g. s = torch.cuda.Stream(); torch.cuda.set_stream(s); s.cuda_stream)")
        _chk(self.L.ldso_ba_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_settings(self, s):
        s = np.ascontiguousarray(s)
        _chk(self.L.ldso_ba_set_settings(self.h, _p(s)))

    def set_image(self, slot, dI0):
        a = np.ascontiguousarray(dI0, np.float32)
        _chk(self.L.ldso_ba_set_image(self.h, C.c_int(slot), _p(a)))

    def set_image_raw(self, slot, irradiance):
        a = np.ascontiguousarray(irradiance, np.float32)
        _chk(self.L.ldso_ba_set_image_raw(self.h, C.c_int(slot), _p(a)))

    def set_image_pyramid(self, slot, pyr: "Pyramid"):
        _chk(self.L.ldso_ba_set_image_pyramid(self.h, C.c_int(slot), pyr.h))
        self._pyramids = getattr(self, "_pyramids", {}); self._pyramids[slot] = pyr          # keep it alive while the slot refers to it

    def get_image(self, slot):
        out = np.zeros((self.hh, self.w, 3), np.float32)
        _chk(self.L.ldso_ba_get_image(self.h, C.c_int(slot), _p(out)))
        return out

    def set_window(self, image_slots, points, residuals, lin_J=None, lin_rtz=None):
        sl = np.ascontiguousarray(image_slots, np.int32)
        pts = np.ascontiguousarray(points)
        res = np.ascontiguousarray(residuals)
        J = np.ascontiguousarray(lin_J) if lin_J is not None else None
        rt = np.ascontiguousarray(lin_rtz, np.float32) if lin_rtz is not None else None
        self.F, self.P, self.R = len(sl), len(pts), len(res)
        _chk(self.L.ldso_ba_set_window(self.h, C.c_int(self.F), _p(sl), C.c_int(self.P), _p(pts), C.c_int(self.R), _p(res), _p(J), _p(rt)))

    def set_point_stats(self, max_rel_baseline, num_good_residuals):
        a = np.ascontiguousarray(max_rel_baseline, np.float32); b = np.ascontiguousarray(num_good_residuals, np.int32)
        assert len(a) == len(b) == self.P
        _chk(self.L.ldso_ba_set_point_stats(self.h, _p(a), _p(b)))

    def update_window(self, image_slots, frame_from, point_from, res_mask, fresh=None, fresh_res=None, fresh_mrb=None, fresh_ngr=None):
        """ldso_ba_update_window: the next window as a delta against the resident one (include/ldso_hip.h)."""
        sl = np.ascontiguousarray(image_slots, np.int32); ff = np.ascontiguousarray(frame_from, np.int32)
        pf = np.ascontiguousarray(point_from, np.int32); mk = np.ascontiguousarray(res_mask, np.uint32)
        assert len(sl) == len(ff) and len(pf) == len(mk)
        nf = 0 if fresh is None else len(fresh); nr = 0 if fresh_res is None else len(fresh_res)
        fp = np.ascontiguousarray(fresh) if nf else None; fr = np.ascontiguousarray(fresh_res) if nr else None
        fm = np.ascontiguousarray(fresh_mrb, np.float32) if nf else None; fg = np.ascontiguousarray(fresh_ngr, np.int32) if nf else None
        _chk(self.L.ldso_ba_update_window(self.h, C.c_int(len(sl)), _p(sl), _p(ff), C.c_int(len(pf)), _p(pf), _p(mk), C.c_int(nf), _p(fp), C.c_int(nr), _p(fr), _p(fm), _p(fg)))
        self._sync_dims()

    # ---- the same delta call by call (ldso_ba_window_begin .. ldso_ba_window_commit) ----
    def window_begin(self):
        _chk(self.L.ldso_ba_window_begin(self.h))

    def remove_frame(self, frame_idx):
        _chk(self.L.ldso_ba_remove_frame(self.h, C.c_int(frame_idx)))

    def insert_frame(self, image_slot) -> int:
        fid = C.c_int()
        _chk(self.L.ldso_ba_insert_frame(self.h, C.c_int(image_slot), C.byref(fid)))
        return fid.value

    def remove_points(self, rows):
        a = np.ascontiguousarray(rows, np.int32)
        _chk(self.L.ldso_ba_remove_points(self.h, C.c_int(len(a)), _p(a)))

    def drop_residuals(self, rows, targets):
        a = np.ascontiguousarray(rows, np.int32); b = np.ascontiguousarray(targets, np.int32)
        _chk(self.L.ldso_ba_drop_residuals(self.h, C.c_int(len(a)), _p(a), _p(b)))

    def add_residuals(self, rows, targets):
        a = np.ascontiguousarray(rows, np.int32); b = np.ascontiguousarray(targets, np.int32)
        _chk(self.L.ldso_ba_add_residuals(self.h, C.c_int(len(a)), _p(a), _p(b)))

    def add_points(self, points, before_row, residuals, mrb=None, ngr=None):
        pts = np.ascontiguousarray(points); br = np.ascontiguousarray(before_row, np.int32); res = np.ascontiguousarray(residuals)
        m = np.ascontiguousarray(mrb, np.float32) if mrb is not None else None; g = np.ascontiguousarray(ngr, np.int32) if ngr is not None else None
        _chk(self.L.ldso_ba_add_points(self.h, C.c_int(len(pts)), _p(pts), _p(br), C.c_int(len(res)), _p(res), _p(m), _p(g)))

    def _sync_dims(self):
        """F / P / R as the HANDLE holds them (the getters size their numpy buffers from these: never from what a caller says)"""
        F, P, R = C.c_int(), C.c_int(), C.c_int()
        _chk(self.L.ldso_ba_get_dims(self.h, C.byref(F), C.byref(P), C.byref(R)))
        self.F, self.P, self.R = F.value, P.value, R.value

    def window_commit(self, F=None, P=None, R=None):
        _chk(self.L.ldso_ba_window_commit(self.h))
        self._sync_dims()
        assert (F is None or F == self.F) and (P is None or P == self.P) and (R is None or R == self.R), ((F, P, R), (self.F, self.P, self.R))

    def set_frames(self, frames, calib):
        fr = np.ascontiguousarray(frames)
        c = np.ascontiguousarray(calib)
        _chk(self.L.ldso_ba_set_frames(self.h, _p(fr), _p(c)))

    def set_prior(self, HM, bM):
        HM = np.ascontiguousarray(HM, np.float64) if HM is not None else None
        bM = np.ascontiguousarray(bM, np.float64) if bM is not None else None
        _chk(self.L.ldso_ba_set_prior(self.h, _p(HM), _p(bM)))

    def load_window(self, win: synth.Window):
        self.set_settings(win.settings)
        for f in range(win.F):
            self.set_image(f, win.images[f][0])
        self.set_window(np.arange(win.F), win.points, win.residuals, win.lin_J, win.lin_res_toZeroF)
        self.set_frames(win.frames, win.calib)
        if np.any(win.HM) or np.any(win.bM):
            self.set_prior(win.HM, win.bM)

    def set_shard(self, begin, end):
        _chk(self.L.ldso_ba_set_shard(self.h, C.c_int(begin), C.c_int(end)))

    # ---- one-shot peer-write all-reduce (ldso_ba_enqueue_gn_p2p) ----
    def p2p_window_alloc(self, n_ranks, with_ipc_handle=False):
        w = C.c_void_p(); hnd = C.create_string_buffer(64) if with_ipc_handle else None
        _chk(self.L.ldso_ba_p2p_window_alloc(self.h, C.c_int(n_ranks), C.byref(w), hnd))
        return (w.value, hnd.raw) if with_ipc_handle else w.value

    def p2p_window_open(self, ipc_handle: bytes):
        w = C.c_void_p()
        _chk(self.L.ldso_ba_p2p_window_open(self.h, C.c_char_p(ipc_handle), C.byref(w)))
        return w.value

    def p2p_window_close(self, window, opened_from_handle=False):
        _chk(self.L.ldso_ba_p2p_window_close(self.h, C.c_void_p(window), C.c_int(1 if opened_from_handle else 0)))

    def enqueue_gn_p2p(self, rank, n_ranks, windows, first_iteration, iters):
        arr = (C.c_void_p * n_ranks)(*windows)
        _chk(self.L.ldso_ba_enqueue_gn_p2p(self.h, C.c_int(rank), C.c_int(n_ranks), arr, C.c_int(first_iteration), C.c_int(iters)))

    def p2p_check(self):
        _chk(self.L.ldso_ba_p2p_check(self.h))

    def set_chunk_points(self, n):
        _chk(self.L.ldso_ba_set_chunk_points(self.h, C.c_int(n)))

    def set_reduce_splits(self, splits):
        """K-splits per Schur tile of this handle's GN fast path (8; a batch of >= 4 windows uses 4: BABatch.reduce_splits())"""
        _chk(self.L.ldso_ba_set_reduce_splits(self.h, C.c_int(int(splits))))

    def get_chunk_cuts(self):
        """ends of the chunks in force (one past the last point of every chunk)"""
        n = C.c_int()
        _chk(self.L.ldso_ba_get_chunk_cuts(self.h, None, C.c_int(0), C.byref(n)))
        ends = np.zeros(n.value, np.int32)
        _chk(self.L.ldso_ba_get_chunk_cuts(self.h, _p(ends), C.c_int(len(ends)), C.byref(n)))
        return ends

    def set_chunk_cuts(self, ends):
        ends = np.ascontiguousarray(ends, np.int32)
        _chk(self.L.ldso_ba_set_chunk_cuts(self.h, _p(ends) if len(ends) else None, C.c_int(len(ends))))

    def get_chunk_points(self):
        n, w = C.c_int(), C.c_int()
        _chk(self.L.ldso_ba_get_chunk_points(self.h, C.byref(n), C.byref(w)))
        return n.value, w.value

    def reduce_doubles(self) -> int:
        return int(self.L.ldso_ba_reduce_doubles(self.h))

    # --- optimisation slice --------------------------------------------------------------------------
    def collect_active(self):
        _chk(self.L.ldso_ba_collect_active(self.h))

    def linearize_all(self, fix=False) -> float:
        e = C.c_double()
        _chk(self.L.ldso_ba_linearize_all(self.h, C.c_int(1 if fix else 0), C.byref(e)))
        return e.value

    def apply_res(self):
        _chk(self.L.ldso_ba_apply_res(self.h))

    def backup_state(self):
        _chk(self.L.ldso_ba_backup_state(self.h))

    def solve_system(self, iteration, lam=1e-1):
        _chk(self.L.ldso_ba_solve_system(self.h, C.c_int(iteration), C.c_double(lam)))

    def do_step(self) -> bool:
        cb = C.c_int()
        _chk(self.L.ldso_ba_do_step(self.h, C.byref(cb)))
        return bool(cb.value)

    def load_state_backup(self):
        _chk(self.L.ldso_ba_load_state_backup(self.h))

    def optimize(self, niters, force_all=False):
        rm = C.c_float()
        it = C.c_int()
        _chk(self.L.ldso_ba_optimize(self.h, C.c_int(niters), C.c_int(1 if force_all else 0), C.byref(rm), C.byref(it)))
        return rm.value, it.value

    def calc_lm_energies(self):
        m, l = C.c_double(), C.c_double()
        _chk(self.L.ldso_ba_calc_lm_energies(self.h, C.byref(m), C.byref(l)))
        return m.value, l.value

    def marginalize_points(self, flags):
        """EnergyFunctional::marginalizePointsF for the flagged points -> (HM, bM)"""
        flags = np.ascontiguousarray(flags, np.int32)
        assert len(flags) == self.P
        n = 8 * self.F + 4
        HM = np.zeros((n, n)); bM = np.zeros(n)
        _chk(self.L.ldso_ba_marginalize_points(self.h, _p(flags), _p(HM), _p(bM)))
        return HM, bM

    def marginalize_frame(self, idx):
        """EnergyFunctional::marginalizeFrame on the device prior -> (HM, bM) of the window without frame idx"""
        n = 8 * (self.F - 1) + 4
        HM = np.zeros((n, n)); bM = np.zeros(n)
        _chk(self.L.ldso_ba_marginalize_frame(self.h, C.c_int(idx), _p(HM), _p(bM)))
        return HM, bM

    def time_linearize(self, reps=50):
        us = C.c_double()
        _chk(self.L.ldso_ba_time_linearize(self.h, C.c_int(reps), C.byref(us)))
        return us.value

    def get_pair_rt(self):
        out = np.zeros((self.F * self.F, 14), np.float32)
        _chk(self.L.ldso_ba_get_pair_rt(self.h, _p(out)))
        return out

    def activate_points(self, pts, min_obs=1, min_idepth_hessian=100.0, gn_iterations=3):
        """FullSystem::optimizeImmaturePoint for a batch of immature points against the resident window"""
        a = np.ascontiguousarray(pts)
        assert a.dtype == synth.IMMATURE_DTYPE
        out = np.zeros(len(a), synth.ACTIVATION_DTYPE)
        _chk(self.L.ldso_ba_activate_points(self.h, C.c_int(len(a)), _p(a), C.c_int(min_obs), C.c_float(min_idepth_hessian), C.c_int(gn_iterations), _p(out)))
        return out

    def enqueue_gn(self, first_iteration, iters):
        _chk(self.L.ldso_ba_enqueue_gn(self.h, C.c_int(first_iteration), C.c_int(iters)))

    def sync(self):
        _chk(self.L.ldso_ba_sync(self.h))

    def gn_reduce_doubles(self):
        return int(self.L.ldso_ba_gn_reduce_doubles(self.h))

    def gn_reduce_local(self, buf_ptr: int, lam=1e-1):
        _chk(self.L.ldso_ba_gn_reduce_local(self.h, C.c_void_p(buf_ptr), C.c_double(lam)))

    def gn_solve_reduced(self, buf_ptr: int, iteration, lam=1e-1):
        _chk(self.L.ldso_ba_gn_solve_reduced(self.h, C.c_void_p(buf_ptr), C.c_int(iteration), C.c_double(lam)))

    def enqueue_gn_rccl(self, comm_ptr: int, first_iteration, iters):
        _chk(self.L.ldso_ba_enqueue_gn_rccl(self.h, C.c_void_p(comm_ptr), C.c_int(first_iteration), C.c_int(iters)))

    def reduce_local(self, buf_ptr: int):
        _chk(self.L.ldso_ba_reduce_local(self.h, C.c_void_p(buf_ptr)))

    def solve_reduced(self, buf_ptr: int, iteration, lam=1e-1, do_step=True):
        _chk(self.L.ldso_ba_solve_reduced(self.h, C.c_void_p(buf_ptr), C.c_int(iteration), C.c_double(lam), C.c_int(1 if do_step else 0)))

    # --- results ----------------------------------------------------------------------------------------
    def get_residuals(self):
        out = np.zeros(self.R, synth.RES_OUT_DTYPE)
        st = np.zeros(self.R, np.int32)
        act = np.zeros(self.R, np.int32)
        rem = np.zeros(self.R, np.int32)
        _chk(self.L.ldso_ba_get_residuals(self.h, _p(out), _p(st), _p(act), _p(rem)))
        return dict(out=out, state_state=st, is_active=act, to_remove=rem)

    def get_points(self):
        out = np.zeros(self.P, synth.POINT_OUT_DTYPE)
        _chk(self.L.ldso_ba_get_points(self.h, _p(out)))
        return out

    def get_frames(self):
        fr = np.zeros(self.F, synth.FRAME_DTYPE)
        step = np.zeros((self.F, 10))
        cv, cs = np.zeros(4), np.zeros(4)
        pre = np.zeros((self.F, 12))
        _chk(self.L.ldso_ba_get_frames(self.h, _p(fr), _p(step), _p(cv), _p(cs), _p(pre)))
        return dict(frames=fr, step=step, calib_value=cv, calib_step=cs, pre_worldToCam=pre)

    def get_results(self):
        """ldso_ba_get_results: residuals + points + frames behind one synchronisation; the same records as the three getters."""
        out = np.zeros(self.R, synth.RES_OUT_DTYPE); st = np.zeros(self.R, np.int32); act = np.zeros(self.R, np.int32); rem = np.zeros(self.R, np.int32)
        pts = np.zeros(self.P, synth.POINT_OUT_DTYPE)
        fr = np.zeros(self.F, synth.FRAME_DTYPE); step = np.zeros((self.F, 10)); cv, cs = np.zeros(4), np.zeros(4)
        _chk(self.L.ldso_ba_get_results(self.h, _p(out), _p(st), _p(act), _p(rem), _p(pts), _p(fr), _p(step), _p(cv), _p(cs)))
        return dict(residuals=dict(out=out, state_state=st, is_active=act, to_remove=rem), points=pts, frames=dict(frames=fr, step=step, calib_value=cv, calib_step=cs))

    def get_system(self):
        n = 8 * self.F + 4
        d = {k: np.zeros((n, n)) for k in ("HA", "HL", "Hsc", "HFinal")}
        d.update({k: np.zeros(n) for k in ("bA", "bL", "bsc", "bFinal", "x")})
        _chk(self.L.ldso_ba_get_system(self.h, _p(d["HA"]), _p(d["bA"]), _p(d["HL"]), _p(d["bL"]), _p(d["Hsc"]), _p(d["bsc"]),
                                       _p(d["HFinal"]), _p(d["bFinal"]), _p(d["x"])))
        return d

    def get_jacobians(self, ids=None):
        ids = np.arange(self.R, dtype=np.int32) if ids is None else np.ascontiguousarray(ids, np.int32)
        out = np.zeros(len(ids), synth.RAWJAC_DTYPE)
        _chk(self.L.ldso_ba_get_jacobians(self.h, _p(ids), C.c_int(len(ids)), _p(out)))
        return out

    def set_debug_dump(self, enable=True):
        _chk(self.L.ldso_ba_set_debug_dump(self.h, C.c_int(1 if enable else 0)))

    def set_debug_split_launch(self, enable=True):
        _chk(self.L.ldso_ba_set_debug_split_launch(self.h, C.c_int(1 if enable else 0)))

    def get_precalc(self):
        out = np.zeros((self.F, self.F, 27), np.float32)
        _chk(self.L.ldso_ba_get_precalc(self.h, _p(out)))
        return out

    def get_counts(self):
        a, l = C.c_int(), C.c_int()
        _chk(self.L.ldso_ba_get_counts(self.h, C.byref(a), C.byref(l)))
        return a.value, l.value

    def get_energy_log(self):
        buf = np.zeros(64)
        n = self.L.ldso_ba_get_energy_log(self.h, _p(buf), C.c_int(64))
        self._dbg = buf[40:56].copy()
        return buf[:n].copy()

    def profile(self, enable=True):
        _chk(self.L.ldso_ba_profile(self.h, C.c_int(1 if enable else 0)))

    def kernel_time_ms(self, which):
        ms = C.c_double()
        n = C.c_int()
        _chk(self.L.ldso_ba_kernel_time_ms(self.h, C.c_int(which), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class BABatch:
    """ldso_ba_batch_*: the GN iteration of several independent windows per launch."""

    def __init__(self, handles):
        self.L = lib()
        self.handles = list(handles)
        arr = (C.c_void_p * len(self.handles))(*[h.h for h in self.handles])
        self.h = C.c_void_p()
        _chk(self.L.ldso_ba_batch_create(arr, C.c_int(len(self.handles)), C.byref(self.h)))

    def enqueue_gn(self, first_iteration, iters):
        _chk(self.L.ldso_ba_batch_enqueue_gn(self.h, C.c_int(first_iteration), C.c_int(iters)))

    def sync(self):
        self.handles[0].sync()

    def time_linearize(self, reps=50):
        us = C.c_double()
        _chk(self.L.ldso_ba_batch_time_linearize(self.h, C.c_int(reps), C.byref(us)))
        return us.value

    def reduce_splits(self):
        """K-splits per Schur tile the batch reduces its windows with (BA.set_reduce_splits gives a lone handle the same arithmetic)"""
        n = C.c_int(0)
        _chk(self.L.ldso_ba_batch_reduce_splits(self.h, C.byref(n)))
        return int(n.value)

    def chunk_points(self):
        n = C.c_int()
        _chk(self.L.ldso_ba_batch_chunk_points(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            self.L.ldso_ba_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tracker:
    """CoarseTracker handle (src/frontend/CoarseTracker.cc) on one GPU."""

    def __init__(self, w, h, levels, settings, calib, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.ldso_tr_create(C.c_int(device), C.c_int(w), C.c_int(h), C.c_int(levels), C.byref(self.h)))
        self.w, self.hh, self.levels = w, h, levels
        s = np.ascontiguousarray(settings)
        _chk(self.L.ldso_tr_set_settings(self.h, _p(s)))
        c = np.ascontiguousarray(calib)
        _chk(self.L.ldso_tr_make_k(self.h, _p(c)))

    def close(self):
        if self.h:
            self.L.ldso_tr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pyr(self, pyr):
        keep = [np.ascontiguousarray(pyr[l], np.float32) for l in range(self.levels)]
        arr = (C.c_void_p * self.levels)(*[a.ctypes.data for a in keep])
        return arr, keep

    def set_ref(self, pyr, a, b, exposure, pts):
        arr, keep = self._pyr(pyr)
        pts = np.ascontiguousarray(pts, np.float32)
        _chk(self.L.ldso_tr_set_ref(self.h, arr, C.c_float(a), C.c_float(b), C.c_float(exposure), _p(pts), C.c_int(len(pts))))

    def set_new_frame(self, pyr, exposure=1.0):
        arr, keep = self._pyr(pyr)
        _chk(self.L.ldso_tr_set_new_frame(self.h, arr, C.c_float(exposure)))

    def set_new_frame_image(self, irradiance, exposure=1.0):
        a = np.ascontiguousarray(irradiance, np.float32)
        _chk(self.L.ldso_tr_set_new_frame_image(self.h, _p(a), C.c_float(exposure)))

    def set_new_frame_pyramid(self, pyr: "Pyramid", exposure=1.0):
        _chk(self.L.ldso_tr_set_new_frame_pyramid(self.h, pyr.h, C.c_float(exposure)))
        self._new_pyr = pyr

    def set_ref_pyramid(self, pyr: "Pyramid", a, b, exposure, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        _chk(self.L.ldso_tr_set_ref_pyramid(self.h, pyr.h, C.c_float(a), C.c_float(b), C.c_float(exposure), _p(pts), C.c_int(len(pts))))
        self._ref_pyr = pyr

    def get_new_frame_level(self, lvl):
        out = np.zeros((self.hh >> lvl, self.w >> lvl, 3), np.float32)
        _chk(self.L.ldso_tr_get_new_frame_level(self.h, C.c_int(lvl), _p(out)))
        return out

    def motion_hypotheses(self, sprelast, slast, lastF, poses_valid=True):
        P = [np.ascontiguousarray(np.asarray(m, np.float64)[:3, :4]) for m in (sprelast, slast, lastF)]
        out = np.zeros((83, 3, 4)); n = C.c_int()
        _chk(self.L.ldso_tr_motion_hypotheses(_p(P[0]), _p(P[1]), _p(P[2]), C.c_int(1 if poses_valid else 0), _p(out), C.byref(n)))
        return out[: n.value]

    def track_new_coarse(self, sprelast, slast, lastF, aff_last, last_rmse, retrack_threshold=1.5, poses_valid=True):
        """ldso_tr_track_new_coarse: FullSystem::trackNewCoarse (FullSystem.cc:179-386); poses are worldToCam 4x4 / 3x4"""
        P = [np.ascontiguousarray(np.asarray(m, np.float64)[:3, :4]) for m in (sprelast, slast, lastF)]
        aff = np.asarray(aff_last, np.float32); rmse = np.ascontiguousarray(last_rmse, np.float64).copy()
        res4 = np.zeros(4); w2c = np.zeros((3, 4)); aff_out = np.zeros(2, np.float32); tries, good = C.c_int(-1), C.c_int(-1)
        _chk(self.L.ldso_tr_track_new_coarse(self.h, _p(P[0]), _p(P[1]), _p(P[2]), C.c_int(1 if poses_valid else 0), _p(aff), _p(rmse), C.c_double(retrack_threshold),
                                             _p(res4), _p(w2c), _p(aff_out), C.byref(tries), C.byref(good)))
        return dict(result=res4, w2c=w2c, aff=aff_out, lastCoarseRMSE=rmse, tries=tries.value, good=good.value)

    def select_hypothesis(self, batch, coarsest, last_coarse_rmse0=float("nan"), retrack_threshold=1.5):
        lr = np.ascontiguousarray(batch["lastResiduals"], np.float64)
        ok = np.ascontiguousarray(batch["ok"], np.int32)
        best, tries = C.c_int(), C.c_int()
        ach = np.zeros(5)
        _chk(self.L.ldso_tr_select_hypothesis(C.c_int(len(ok)), C.c_int(coarsest), _p(lr), _p(ok), C.c_double(last_coarse_rmse0), C.c_double(retrack_threshold),
                                              C.byref(best), C.byref(tries), _p(ach)))
        return best.value, tries.value, ach

    def pc(self, lvl):
        n = C.c_int()
        _chk(self.L.ldso_tr_get_pc(self.h, C.c_int(lvl), None, None, None, None, C.byref(n)))
        u, v, d, c = (np.zeros(n.value, np.float32) for _ in range(4))
        _chk(self.L.ldso_tr_get_pc(self.h, C.c_int(lvl), _p(u), _p(v), _p(d), _p(c), C.byref(n)))
        return u, v, d, c

    def calc_res(self, lvl, T, a, b, cutoff):
        T = np.ascontiguousarray(T[:3, :4], np.float64)
        rs = np.zeros(6)
        n = C.c_int()
        _chk(self.L.ldso_tr_calc_res(self.h, C.c_int(lvl), _p(T), C.c_float(a), C.c_float(b), C.c_float(cutoff), _p(rs), C.byref(n)))
        return rs, n.value

    def calc_gs(self, lvl, T, a, b):
        T = np.ascontiguousarray(T[:3, :4], np.float64)
        H = np.zeros((8, 8))
        bb = np.zeros(8)
        _chk(self.L.ldso_tr_calc_gs(self.h, C.c_int(lvl), _p(T), C.c_float(a), C.c_float(b), _p(H), _p(bb)))
        return H, bb

    def track(self, T, a, b, coarsest, min_res=None):
        r = self.track_batch([T], [(a, b)], coarsest, min_res)
        return {k: (v[0] if isinstance(v, (list, np.ndarray)) else v) for k, v in r.items()}

    def last_track_pivoted_solves(self):
        """LM solves of the last track call that fell back to the pivoted LDL^T (ill-conditioned 8 x 8 system)"""
        n = C.c_int(0)
        _chk(self.L.ldso_tr_last_track_pivoted_solves(self.h, C.byref(n)))
        return int(n.value)

    def last_track_evals(self):
        """(evals[5], pc_n[5]) of the last track: calcRes evaluations and reference points per pyramid level"""
        ev = np.zeros(5, np.int32); pc = np.zeros(5, np.int32)
        _chk(self.L.ldso_tr_last_track_evals(self.h, _p(ev), _p(pc)))
        return ev, pc

    def track_batch(self, Ts, affs, coarsest, min_res=None):
        n = len(Ts)
        T = np.ascontiguousarray(np.stack([np.asarray(t)[:3, :4] for t in Ts]), np.float64).copy()
        ab = np.ascontiguousarray(np.asarray(affs, np.float32).reshape(n, 2)).copy()
        mr = np.full(5, np.nan) if min_res is None else np.asarray(min_res, np.float64)
        lr = np.zeros((n, 5))
        fl = np.zeros((n, 3))
        ok = np.zeros(n, np.int32)
        it = np.zeros(n, np.int32)
        _chk(self.L.ldso_tr_track_batch(self.h, C.c_int(n), _p(T), _p(ab), C.c_int(coarsest), _p(mr), _p(lr), _p(fl), _p(ok), _p(it)))
        return dict(ok=ok.astype(bool), T=T, a=ab[:, 0].copy(), b=ab[:, 1].copy(), lastResiduals=lr, flow=fl, iterations=it)


class Tracer:
    """Immature-point tracing handle: FullSystem::traceNewCoarse / ImmaturePoint::traceOn on one GPU."""

    def __init__(self, w, h, max_points, settings=None, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.ldso_trace_create(C.c_int(device), C.c_int(w), C.c_int(h), C.c_int(max_points), C.byref(self.h)))
        self.w, self.hh, self.n = w, h, 0
        if settings is not None:
            s = np.ascontiguousarray(settings)
            _chk(self.L.ldso_trace_set_settings(self.h, _p(s)))

    def close(self):
        if self.h:
            self.L.ldso_trace_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_points(self, pts):
        a = np.ascontiguousarray(pts)
        assert a.dtype == synth.IMMATURE_DTYPE
        _chk(self.L.ldso_trace_set_points(self.h, C.c_int(len(a)), _p(a)))
        self.n = len(a)

    def get_points(self):
        out = np.zeros(self.n, synth.IMMATURE_DTYPE)
        _chk(self.L.ldso_trace_get_points(self.h, _p(out)))
        return out

    def set_frame(self, dI_level0):
        a = np.ascontiguousarray(dI_level0, np.float32)
        _chk(self.L.ldso_trace_set_frame(self.h, _p(a)))

    def set_frame_raw(self, irradiance):
        a = np.ascontiguousarray(irradiance, np.float32)
        _chk(self.L.ldso_trace_set_frame_raw(self.h, _p(a)))

    def set_frame_pyramid(self, pyr: "Pyramid"):
        _chk(self.L.ldso_trace_set_frame_pyramid(self.h, pyr.h))
        self._pyr = pyr

    def trace_on(self, KRKi, Kt, aff):
        K1 = np.ascontiguousarray(KRKi, np.float32); K2 = np.ascontiguousarray(Kt, np.float32); A = np.ascontiguousarray(aff, np.float32)
        counts = np.zeros(6, np.int32)
        _chk(self.L.ldso_trace_on(self.h, C.c_int(len(K1)), _p(K1), _p(K2), _p(A), _p(counts)))
        return counts


class Initializer:
    """Monocular initialiser handle: CoarseInitializer::setFirst / trackFrame on one GPU (include/ldso_hip.h)."""

    def __init__(self, w, h, levels, device=0, stream=None):
        self.L = lib()
        self.h = C.c_void_p()
        _chk(self.L.ldso_init_create(C.c_int(device), C.c_int(w), C.c_int(h), C.c_int(levels), C.byref(self.h)))
        self.w, self.hh, self.levels = w, h, levels
        self.n = [0] * levels
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self.h:
            self.L.ldso_init_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        if not stream_handle:
            raise ValueError("pass a non-default HIP stream handle (0 is the legacy default stream)")
        _chk(self.L.ldso_init_set_stream(self.h, C.c_void_p(stream_handle)))

    def set_first(self, K4, irradiance, points, exposure=1.0, huberTH=9.0, fixAffine=True):
        """K4 = (fxl, fyl, cxl, cyl) of level 0; irradiance float32 [h,w]; points: one INIT_POINT_DTYPE array per level."""
        k = np.ascontiguousarray(K4, np.float32)
        img = np.ascontiguousarray(irradiance, np.float32)
        assert img.shape == (self.hh, self.w) and len(points) == self.levels
        pts = [np.ascontiguousarray(p, dtype=synth.INIT_POINT_DTYPE) for p in points]
        pp = (C.c_void_p * self.levels)(*[p.ctypes.data for p in pts])
        n = np.array([len(p) for p in pts], dtype=np.int32)
        _chk(self.L.ldso_init_set_first(self.h, _p(k), _p(img), C.c_float(exposure), pp, _p(n), C.c_float(huberTH), C.c_int(1 if fixAffine else 0)))
        self.n = [int(x) for x in n]

    def set_schedule(self, first_steps=0, prepare_on_grid=True):
        """Launch schedule of track_frame (ldso_init_set_schedule): the results do not depend on it."""
        _chk(self.L.ldso_init_set_schedule(self.h, C.c_int(first_steps), C.c_int(1 if prepare_on_grid else 0)))

    def set_new_frame(self, irradiance, exposure=1.0):
        img = np.ascontiguousarray(irradiance, np.float32)
        assert img.shape == (self.hh, self.w)
        _chk(self.L.ldso_init_set_new_frame(self.h, _p(img), C.c_float(exposure)))

    def track_frame(self, irradiance=None, exposure=1.0):
        """trackFrame; irradiance None: the frame given to set_new_frame.  Returns the INIT_STATE_DTYPE record."""
        st = np.zeros((), synth.INIT_STATE_DTYPE)
        img = None
        if irradiance is not None:
            img = np.ascontiguousarray(irradiance, np.float32)
            assert img.shape == (self.hh, self.w)
        _chk(self.L.ldso_init_track_frame(self.h, _p(img), C.c_float(exposure), _p(st)))
        return st

    def state(self):
        st = np.zeros((), synth.INIT_STATE_DTYPE)
        _chk(self.L.ldso_init_get_state(self.h, _p(st)))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, dtype=synth.INIT_STATE_DTYPE)
        _chk(self.L.ldso_init_set_state(self.h, _p(st)))

    def points(self, lvl):
        out = np.zeros(self.n[lvl], synth.INIT_POINT_DTYPE)
        _chk(self.L.ldso_init_get_points(self.h, C.c_int(lvl), _p(out)))
        return out

    def set_points(self, lvl, pts):
        pts = np.ascontiguousarray(pts, dtype=synth.INIT_POINT_DTYPE)
        assert len(pts) == self.n[lvl]
        _chk(self.L.ldso_init_set_points(self.h, C.c_int(lvl), _p(pts)))

    def calc_res_and_gs(self, lvl, T, a, b):
        T = np.ascontiguousarray(np.asarray(T, dtype=np.float64)[:3, :4])
        H = np.zeros((8, 8), np.float32); bo = np.zeros(8, np.float32); Hsc = np.zeros((8, 8), np.float32); bsc = np.zeros(8, np.float32)
        res = np.zeros(3, np.float32); ec = np.zeros(3, np.float32)
        _chk(self.L.ldso_init_calc_res_and_gs(self.h, C.c_int(lvl), _p(T), C.c_double(a), C.c_double(b), _p(H), _p(bo), _p(Hsc), _p(bsc), _p(res), _p(ec)))
        return H, bo, Hsc, bsc, res, ec
