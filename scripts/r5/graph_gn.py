"""GN iterations enqueued launch by launch against the same launches captured once into a HIP graph and replayed (run on the GPU box).
The iteration is two dependent kernels (k_reduce_solve -> k_linearize_one) the host enqueues far ahead of the device: what a graph can remove is
per-packet dispatch work on the device side, if there is any to remove.  N must be even (the residual sets ping-pong: an even count leaves the handle's
host-side state where the capture found it)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from ldso_amd import synth, binding

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
win = synth.add_synthetic_prior(synth.make_config(cfg))
g = binding.BA.from_window(win)
s = torch.cuda.Stream()
g.set_stream(s.cuda_stream)
g.collect_active(); g.linearize_all(False); g.apply_res()
g.enqueue_gn(0, 20); g.sync()

def timed(fn, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / N * 1e6)
    return sorted(out)

plain = timed(lambda: g.enqueue_gn(2, N))
print("launch by launch: us per iteration (sorted)", [round(x, 2) for x in plain])
try:
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=s):
        g.enqueue_gn(2, N)
    def replay():
        with torch.cuda.stream(s):          # replay() launches on torch's CURRENT stream (the first version of this script timed an unsynchronised launch)
            cg.replay()
    graph = timed(replay)
    print("graph replay:     us per iteration (sorted)", [round(x, 2) for x in graph])
    st = g.get_frames()["frames"]["state"]
    print("state finite after the replays:", bool(np.isfinite(st).all()))
except Exception as e:          # capture refuses an API the enqueue path calls: say which
    print("capture failed:", repr(e))
