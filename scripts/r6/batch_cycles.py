"""Cycle accounting of the pipelined batched k_linearize (library built with -DLDSO_STAMPS for ba_linearize.hip): per point of wave 0 / chunk 0 of a few windows."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from ldso_amd import synth, binding
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
hs = []
for i in range(B):
    w = synth.add_synthetic_prior(synth.make_config("C3", seed=20260925 + i))
    g = binding.BA.from_window(w, stream=ts.cuda_stream); g.collect_active(); g.linearize_all(False); g.apply_res(); hs.append(g)
bt = binding.BABatch(hs)
print("chunk points", bt.chunk_points())
bt.enqueue_gn(0, 6); bt.sync(); torch.cuda.synchronize()
for rep in range(2):
    us = bt.time_linearize(3)
    for wdx in (0, B // 3, B - 1):
        buf = np.zeros(64)
        hs[wdx].L.ldso_ba_get_energy_log(hs[wdx].h, buf.ctypes.data_as(C.c_void_p), C.c_int(64))
        c = buf[24:30]; n = max(c[5], 1)
        print("launch %.1f us, window %d: points %d | cycles per point: step+front %.0f, record loads %.0f, back %.0f, in flight at the end %.0f | total %.0f" % (us, wdx, n, c[1] / n, c[2] / n, c[3] / n, c[4] / n, c[1:5].sum() / n))
