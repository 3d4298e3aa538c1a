"""FullSystem::traceNewCoarse timing: ldso_trace_on on the GPU (points resident; one call = pose upload + launch + counts
read-back) vs the oracle on one host core, 640x480, fresh immature points (full-length epipolar searches). Run on the GPU box."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding
from oracle import pyoracle as po

def main(per_frame=1000):
    win = synth.make_config("C3", extra_frames=1)
    pts, _ = synth.make_immature_points(win, per_frame)
    F = win.F
    KRKi, Kt, aff = synth.trace_poses(win, F)
    img = win.images[F][0]
    g = binding.Tracer(win.w, win.h, len(pts))
    g.set_frame(img)
    tg = []
    for rep in range(12):
        g.set_points(pts)                      # fresh (untraced) points each repetition
        t0 = time.perf_counter(); c = g.trace_on(KRKi, Kt, aff); tg.append(time.perf_counter() - t0)
    to = []
    for rep in range(3):
        ref = pts.copy()
        t0 = time.perf_counter(); co = po.trace_on(ref, img, KRKi, Kt, aff); to.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); g.set_frame(img); t_set = time.perf_counter() - t0
    t0 = time.perf_counter(); g.set_frame_raw(np.ascontiguousarray(img[:, :, 0])); t_raw = time.perf_counter() - t0
    print(json.dumps({"workload": "%d fresh immature points on %d key frames, %dx%d" % (len(pts), F, win.w, win.h), "status_counts": c.tolist(),
                      "gpu_trace_on_ms": round(float(np.median(tg[2:])) * 1e3, 4), "cpu_oracle_ms": round(float(np.median(to)) * 1e3, 3), "cpu_cores": 1,
                      "points_per_s_gpu": round(len(pts) / float(np.median(tg[2:])), 1), "set_frame_ms": round(t_set * 1e3, 3), "set_frame_raw_ms": round(t_raw * 1e3, 3)}))
main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)


def activation(per_frame=400):
    """FullSystem::activatePointsMT core: optimizeImmaturePoint for a batch (GPU: one call incl. upload / download) vs oracle, 1 core"""
    win = synth.make_config("C3", extra_frames=2)
    pts, _ = synth.make_immature_points(win, per_frame)
    F = win.F
    for fidx in (F, F + 1):
        KRKi, Kt, aff = synth.trace_poses(win, fidx)
        po.trace_on(pts, win.images[fidx][0], KRKi, Kt, aff)
    pts = pts[np.isfinite(pts["idepth_max"]) & (pts["lastTraceStatus"] != 1)].copy()
    g = binding.BA.from_window(win)
    pairs = g.get_pair_rt()
    K4 = np.asarray([np.float32(50.0 * v) for v in win.calib["value"]], np.float32)
    tg = []
    for _ in range(12):
        t0 = time.perf_counter(); out = g.activate_points(pts); tg.append(time.perf_counter() - t0)
    imgs = [win.images[f][0] for f in range(F)]
    t0 = time.perf_counter(); ref = po.activate_points(pts, imgs, K4, pairs, win.w, win.h); to = time.perf_counter() - t0
    print(json.dumps({"workload": "activation of %d traced immature points against %d key frames, %dx%d" % (len(pts), F, win.w, win.h),
                      "gpu_activate_ms": round(float(np.median(tg[2:])) * 1e3, 4), "cpu_oracle_ms": round(to * 1e3, 3), "cpu_cores": 1,
                      "activated": int((out["ok"] == 1).sum())}))
activation()
