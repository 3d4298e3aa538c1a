"""Wall time of the pieces of the per-key-frame window upload (C3, one MI355X).  Run on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
win = synth.add_synthetic_prior(synth.make_config(cfg, extra_frames=1))
g = binding.BA.from_window(win); g.sync()


def med(f, n=14):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); g.sync(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts[3:])) * 1e6


img = win.images[win.F - 1][0]
raw = np.ascontiguousarray(img[:, :, 0])
pyr = binding.Pyramid(win.w, win.h, 1)
print(cfg, "set_window %.0f us | set_frames %.0f us | set_prior %.0f us | set_image (12 B/px) %.0f us | set_image_raw (4 B/px + device gradients) %.0f us | "
      "pyramid make_images + set_image_pyramid %.0f us | set_image_pyramid alone %.0f us" % (
          med(lambda: g.set_window(np.arange(win.F), win.points, win.residuals, win.lin_J, win.lin_res_toZeroF)),
          med(lambda: g.set_frames(win.frames, win.calib)),
          med(lambda: g.set_prior(win.HM, win.bM)) if getattr(win, "HM", None) is not None else float("nan"),
          med(lambda: g.set_image(win.F - 1, img)),
          med(lambda: g.set_image_raw(win.F - 1, raw)),
          med(lambda: (pyr.make_images(raw), g.set_image_pyramid(win.F - 1, pyr))),
          med(lambda: g.set_image_pyramid(win.F - 1, pyr))))
