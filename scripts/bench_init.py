"""Initialiser (CoarseInitializer::trackFrame) on the GPU against the CPU oracle: agreement and time per frame."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from ldso_amd import synth, binding
from oracle import pyoracle

w, h, n = int(sys.argv[1]) if len(sys.argv) > 1 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 480, int(sys.argv[3]) if len(sys.argv) > 3 else 8
seq = synth.make_init_sequence(w, h, n_frames=n, fx=400.0 * w / 640)
L = seq["levels"]
pyr0 = synth.make_images(seq["first"], L)
pts = synth.select_init_points(pyr0)
print("levels", L, "points", [len(p) for p in pts])
o = pyoracle.OracleInitializer(w, h, L); o.set_first(seq["K4"], pyr0, 1.0, pts)
g = binding.Initializer(w, h, L); g.set_first(seq["K4"], seq["first"], pts)
for k in range(n):
    img = seq["frames"][k]
    pyr = synth.make_images(img, L)
    o.set_new_frame(pyr, 1.0)
    t0 = time.perf_counter(); so = o.track_frame(); tc = time.perf_counter() - t0
    g.set_new_frame(img, 1.0)
    t0 = time.perf_counter(); sg = g.track_frame(); tg = time.perf_counter() - t0
    import ctypes as C
    dbg = (C.c_longlong * 8)(); g.L.ldso_init_debug_counters(g.h, dbg)
    if dbg[3]:       # -DLDSO_STAMPS builds; the counters accumulate over the frames
        print("   sweeps", dbg[3], "passes", dbg[1], "sweep us %.0f (%.3f us/pass)" % (dbg[0] / 100.0, dbg[0] / 100.0 / max(dbg[1], 1)), "ctl kernels us %.0f" % (dbg[2] / 100.0),
              "of it: fill+prepare %.0f, up to the decision %.0f, next increment %.0f" % (dbg[4] / 100.0, dbg[5] / 100.0, dbg[6] / 100.0))
    To, Tg = so["thisToNext"].reshape(3, 4), sg["thisToNext"].reshape(3, 4)
    p0o, p0g = o.points(0), g.points(0)
    gd = (p0o["isGood"] != 0) & (p0g["isGood"] != 0)
    print(k, "snapped", so["snapped"], sg["snapped"], "evals", so["evals"], sg["evals"], "cpu %.1f ms gpu %.2f ms" % (tc * 1e3, tg * 1e3),
          "dR %.1e dt %.1e" % (np.abs(To[:, :3] - Tg[:, :3]).max(), np.abs(To[:, 3] - Tg[:, 3]).max()),
          "good diff", int((p0o["isGood"] != p0g["isGood"]).sum()), "iR med %.1e max %.1e" % (np.median(np.abs(p0o["iR"] - p0g["iR"])[gd]), np.abs(p0o["iR"] - p0g["iR"])[gd].max()))
