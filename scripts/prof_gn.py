"""Enqueue GN iterations on a config (for rocprofv3 --kernel-trace --stats)."""
import sys, time
sys.path.insert(0, '.')
from ldso_amd import synth, binding
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
win = synth.make_config(cfg)
g = binding.BA.from_window(win)
g.collect_active(); g.linearize_all(False); g.apply_res()
g.enqueue_gn(0, 10); g.sync()
t = time.time(); g.enqueue_gn(0, iters); g.sync(); dt = time.time() - t
print('%s: %.1f us/iter' % (cfg, dt / iters * 1e6))
