"""Wall time of the per-key-frame device calls a FullSystem::makeKeyFrame adapter issues (C3 window, one MI355X):
window upload, optimize(), marginalizePointsF, marginalizeFrame, point activation, immature-point tracing.  Run on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
win = synth.add_synthetic_prior(synth.make_config(cfg, extra_frames=1))
rng = np.random.default_rng(0)


def med(f, n=12, setup=None):
    ts = []
    for _ in range(n):
        a = setup() if setup else None
        t0 = time.perf_counter(); f(a); ts.append(time.perf_counter() - t0)
    return float(np.median(ts[2:])) * 1e6


g0 = binding.BA.from_window(win)
t_window = med(lambda _: binding.BA.from_window(win), n=6)                 # handle creation + images + tables (allocation included)
g = binding.BA.from_window(win)
t_tables = med(lambda _: (g.set_window(np.arange(win.F), win.points, win.residuals, win.lin_J, win.lin_res_toZeroF), g.set_frames(win.frames, win.calib), g.set_image(win.F - 1, win.images[win.F - 1][0]), g.sync()))


def fresh():
    h = binding.BA.from_window(win); h.sync(); return h


t_opt = med(lambda h: h.optimize(6, force_all=False), setup=fresh)
flags = (win.points["host"] == 0).astype(np.int32)


def optimised():
    h = fresh(); h.optimize(6, force_all=False); return h


t_margp = med(lambda h: h.marginalize_points(flags), setup=optimised)


def margp():
    h = optimised(); h.marginalize_points(flags); return h


t_margf = med(lambda h: h.marginalize_frame(0), setup=margp)
pts, _ = synth.make_immature_points(win, 300)
KRKi, Kt, aff = synth.trace_poses(win, win.F)
tr = binding.Tracer(win.w, win.h, len(pts)); tr.set_frame(win.images[win.F][0])


def traced():
    tr.set_points(pts); return None


t_trace = med(lambda _: tr.trace_on(KRKi, Kt, aff), setup=traced)
tr.set_points(pts); tr.trace_on(KRKi, Kt, aff); cand = tr.get_points()
cand = cand[np.isfinite(cand["idepth_max"]) & (cand["lastTraceStatus"] != 1)]
h = optimised()
t_act = med(lambda _: h.activate_points(cand))
print(f"{cfg}: tables+frames upload {t_tables:.0f} us | optimize (un-forced) {t_opt:.0f} us | marginalizePointsF ({int(flags.sum())} pts) {t_margp:.0f} us | "
      f"marginalizeFrame {t_margf:.0f} us | activate {len(cand)} candidates {t_act:.0f} us | traceOn {len(pts)} points {t_trace:.0f} us | "
      f"(new handle incl. allocation and {win.F} images: {t_window / 1e3:.1f} ms)")
