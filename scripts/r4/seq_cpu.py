import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from ldso_amd import synth
from adapter_sequence_common import run_sequence
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
win = synth.make_config("small", extra_frames=K)
t0 = time.time()
r, log = run_sequence(win, K)
for rec in log:
    s = rec["summary"]
    print(rec["k"], "rmse %.4f" % rec["rmse"], {k: rec[k] for k in ("candidates", "activated", "new_residuals", "points", "lost")}, "F", s["F"], "ids", s["ids"], "pts", s["points"], "imm", s["immature"], "|HM|", float(np.abs(s["HM"]).max()))
print("s", time.time() - t0)
