"""C5 after the marginalisation (tests/test_fullsize_gpu.py::test_c5_end_to_end, F = 11, 10 iterations): which residual states differ between the device and the
oracle, and how close to their outlier threshold those residuals are in the oracle's own run.  Run on the GPU box."""
import copy, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ldso_amd import synth, binding
from oracle import pyoracle as po

win = synth.make_config("C5")
o = po.OracleWindow(win); o.set_force_all_iterations(True); o.optimize(4)
ex, fo = o.export_window(), o.get_frames()
w2 = copy.deepcopy(win)
w2.points, w2.residuals, w2.lin_J, w2.lin_res_toZeroF = ex["points"], ex["residuals"], ex["lin_J"], ex["lin_res_toZeroF"]
w2.frames = fo["frames"]; w2.calib = w2.calib.copy(); w2.calib["value"] = fo["calib_value"]
g = binding.BA.from_window(w2)
o.flag_frame(0); o.flag_points_for_removal()
_, status = o.get_points()
flags = (status == 3).astype(np.int32)
o.drop_points(); o.marginalize_points()
HMo, bMo = o.get_prior()
HMg, bMg = g.marginalize_points(flags)
rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max()
print("marginalize_points HM rel", rel(HMg, HMo), "bM rel", rel(bMg, bMo))
o.marginalize_frame(0)
HM2o, bM2o = o.get_prior()
HM2g, bM2g = g.marginalize_frame(0)
print("marginalize_frame chained: HM rel", rel(HM2g, HM2o), "bM rel", rel(bM2g, bM2o))
g.set_prior(HMo, bMo)
HM2i, bM2i = g.marginalize_frame(0)
print("marginalize_frame on the oracle's prior: HM rel", rel(HM2i, HM2o), "bM rel", rel(bM2i, bM2o))
# the oracle's own arithmetic on the device's prior
w2b = copy.deepcopy(w2); w2b.HM, w2b.bM = HMg, bMg
ob = po.OracleWindow(w2b); ob.marginalize_frame(0)
HM2b, bM2b = ob.get_prior()
print("oracle marginalize_frame on the device's prior vs device: HM rel", rel(HM2g, HM2b), "bM rel", rel(bM2g, bM2b), "| oracle's sensitivity to the prior difference: HM", rel(HM2b, HM2o), "bM", rel(bM2b, bM2o))
ex = o.export_window(); fo = o.get_frames()
w3 = copy.deepcopy(win)
w3.points, w3.residuals, w3.lin_J, w3.lin_res_toZeroF = ex["points"], ex["residuals"], ex["lin_J"], ex["lin_res_toZeroF"]
w3.frames = fo["frames"]; w3.calib = w3.calib.copy(); w3.calib["value"] = fo["calib_value"]; w3.images = win.images[1:]
w3.HM, w3.bM = HM2g, bM2g
o3 = po.OracleWindow(w3); o3.set_force_all_iterations(True)
g3 = binding.BA.from_window(w3)
o3.optimize(10); g3.optimize(10, force_all=True)
ro, rg = o3.get_residuals(False), g3.get_residuals()
fl = np.nonzero(ro["state_state"] != rg["state_state"])[0]
print("flipped residual states", len(fl), "of", w3.R)
fr = o3.get_frames()["frames"]
th = np.maximum(fr["frameEnergyTH"][w3.residuals["host"]], fr["frameEnergyTH"][w3.residuals["target"]])
ewo = ro["out"]["state_NewEnergyWithOutlier"]
d = np.abs(ewo - th) / th
print("oracle |E_withOutlier - th| / th of the flipped ones:", np.sort(d[fl])[:40], "...", "max", d[fl].max() if len(fl) else None)
print("states of flipped (oracle, gpu):", list(zip(ro["state_state"][fl][:20], rg["state_state"][fl][:20])))
for m in (1e-4, 1e-3, 1e-2):
    print("residuals within", m, "of their threshold in the oracle:", int((d < m).sum()), "| flipped ones among them:", int((d[fl] < m).sum()))
