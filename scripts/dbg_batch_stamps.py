"""Device wall-clock stamps (10 ns ticks since the workgroup started) of one mid-launch workgroup of the batched k_linearize (window 20 of 32,
its chunk 0, wave 0): 1 operands staged, 2 first point starts, 3 its taps about to be issued, 4 taps back, 5 its slot loop done, 6 all points
of the wave done, 7 block reduction barrier passed.  Needs a library built with -DLDSO_STAMPS for ba_linearize.hip (scripts/build_variant.sh)."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from ldso_amd import synth, binding
B = 32
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
hs = []
for i in range(B):
    w = synth.add_synthetic_prior(synth.make_config("C3", seed=20260925 + i))
    g = binding.BA.from_window(w, stream=ts.cuda_stream); g.collect_active(); g.linearize_all(False); g.apply_res(); hs.append(g)
bt = binding.BABatch(hs)
print("chunk points", bt.chunk_points())
bt.enqueue_gn(0, 6); bt.sync(); torch.cuda.synchronize()
for rep in range(3):
    bt.time_linearize(3)
    for wdx in (0, 10, 20, 31):
        buf = np.zeros(64)
        hs[wdx].L.ldso_ba_get_energy_log(hs[wdx].h, buf.ctypes.data_as(C.c_void_p), C.c_int(64))
        st = buf[9:16] / 100.0
        print("window", wdx, "stamps us:", " ".join("%d:%.2f" % (i + 1, v) for i, v in enumerate(st)), " per point (t6-t2)/4 = %.2f" % ((st[5] - st[1]) / 4))
