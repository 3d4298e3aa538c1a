"""Would more than two pipelined parts help the batched iteration?  K independent batches of 32 / K windows each, every batch on its own stream (each batch runs as two halves on
two streams internally): aggregate window-iterations/s against ONE batch of 32."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np, torch
from ldso_amd import synth, binding
B = 32
wins = [synth.add_synthetic_prior(synth.make_config("C3", seed=20260925 + i)) for i in range(B)]
for K in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(K)]
    batches, handles = [], []
    for j in range(K):
        hs = []
        for w in wins[j * (B // K):(j + 1) * (B // K)]:
            g = binding.BA.from_window(w, stream=streams[j].cuda_stream); g.collect_active(); g.linearize_all(False); g.apply_res(); hs.append(g)
        handles.append(hs); batches.append(binding.BABatch(hs))
    for b in batches: b.enqueue_gn(0, 10)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for b in batches: b.enqueue_gn(2, 50)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print("K = %d batches of %d windows: %.1f k window-iterations/s (%.4f ms per iteration of all 32)" % (K, B // K, B * 50 / dt / 1e3, dt / 50 * 1e3))
    for b in batches: b.close()
    for hs in handles:
        for g in hs: g.close()
