"""GPU parity of the immature-point tracer (C-ABI ldso_trace_*) against the oracle restatement of ImmaturePoint::traceOn:
the whole 128-byte record of every point - status, idepth interval, quality, lastTraceUV - bit for bit, over two consecutive
frames (the second starts from the intervals of the first) and for the raw-irradiance frame entry point."""
import numpy as np
import pytest

from ldso_amd import synth, binding
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,per_frame", [("small", 150), ("C3", 300)])
def test_trace_matches_oracle_bit_exact(name, per_frame):
    win = synth.make_config(name, extra_frames=2)
    pts, _ = synth.make_immature_points(win, per_frame)
    F = win.F
    ref = pts.copy()
    g = binding.Tracer(win.w, win.h, len(pts))
    g.set_points(pts)
    for step, fidx in enumerate((F, F + 1)):
        KRKi, Kt, aff = synth.trace_poses(win, fidx)
        co = po.trace_on(ref, win.images[fidx][0], KRKi, Kt, aff)
        if step == 0:
            g.set_frame(win.images[fidx][0])
        else:
            g.set_frame_raw(np.ascontiguousarray(win.images[fidx][0][:, :, 0]))      # makeImages on the device
        cg = g.trace_on(KRKi, Kt, aff)
        assert np.array_equal(co, cg), (co, cg)
        out = g.get_points()
        assert np.array_equal(out["lastTraceStatus"], ref["lastTraceStatus"])
        for k in ("idepth_min", "idepth_max", "quality", "lastTraceUV", "lastTracePixelInterval"):
            assert np.array_equal(out[k].view(np.uint32), ref[k].view(np.uint32)), (step, k)
        assert out.tobytes() == ref.tobytes()
    assert co[0] > 0


def test_trace_edge_cases():
    """points projecting outside the new frame (OOB), already-OOB points (untouched), NaN energy threshold (never GOOD)."""
    win = synth.make_config("small", extra_frames=1)
    pts, _ = synth.make_immature_points(win, 40)
    pts["u"][0] = 5.0; pts["v"][0] = 5.0                       # near the border: projects out of bounds for some hosts
    pts["lastTraceStatus"][1] = 1                              # already OOB
    pts["energyTH"][2] = np.nan
    ref = pts.copy()
    KRKi, Kt, aff = synth.trace_poses(win, win.F)
    po.trace_on(ref, win.images[win.F][0], KRKi, Kt, aff)
    g = binding.Tracer(win.w, win.h, len(pts)); g.set_points(pts); g.set_frame(win.images[win.F][0])
    g.trace_on(KRKi, Kt, aff)
    out = g.get_points()
    assert out.tobytes() == ref.tobytes()
    assert out["lastTraceStatus"][1] == 1 and out["lastTraceStatus"][2] != 0
    with pytest.raises(binding.LdsoError):
        binding.Tracer(win.w, win.h, 4).trace_on(KRKi, Kt, aff)          # no frame set: loud failure
