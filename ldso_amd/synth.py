"""Deterministic synthetic keyframe windows for the windowed photometric-BA / coarse-tracker hot path.

This is INPUT generation only (numpy); it is neither the product path nor the oracle.  It builds the
flattened window images declared in include/ldso_window.h:

  * a textured height-field scene rendered by ray casting into F keyframes (+ optional extra frames
    for the tracker), with per-frame affine brightness  I_k = exp(a_k) * T + b_k;
  * the image pyramid + central-difference gradients exactly as FrameHessian::makeImages produces
    them (reference src/internal/FrameHessian.cc:44-113; 12-byte AoS (I,dx,dy) per pixel);
  * points with colour/weight patterns as the ImmaturePoint constructor computes them
    (reference src/internal/ImmaturePoint.cc:21-36 with getInterpolatedElement33BiLin,
    include/internal/GlobalFuncs.h:186-207);
  * one residual from every point to every other keyframe, in p->residuals order.

Workloads follow BASELINE.md / SURVEY.md §8(d): C3 = 7 KF x 2000 pt @640x480, C4 = 7 x 3000 @1232x368,
C5 = 12 x 8000 @640x480.
"""
from __future__ import annotations

import dataclasses
import numpy as np

PATTERN = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], dtype=np.int32)  # Setting.cc:221

SCALE_XI_TRANS = 0.5
SCALE_XI_ROT = 1.0
SCALE_A = 10.0
SCALE_B = 1000.0
SCALE_F = 50.0
SCALE_C = 50.0

# ----------------------------------------------------------------------------------------------
# C struct images (must match include/ldso_window.h)
# ----------------------------------------------------------------------------------------------
SETTINGS_DTYPE = np.dtype([
    ("huberTH", "f4"), ("outlierTHSumComponent", "f4"), ("affineOptModeA", "f4"), ("affineOptModeB", "f4"),
    ("frameEnergyTHN", "f4"), ("frameEnergyTHFacMedian", "f4"), ("frameEnergyTHConstWeight", "f4"),
    ("overallEnergyTHWeight", "f4"), ("initialCalibHessian", "f4"), ("margWeightFac", "f4"),
    ("idepthFixPriorMargFac", "f4"), ("thOptIterations", "f4"), ("coarseCutoffTH", "f4"),
    ("minOptIterations", "i4"), ("solverMode", "i4"), ("forceAcceptStep", "i4"), ("solverModeDelta", "f8"),
], align=True)

FRAME_DTYPE = np.dtype([
    ("worldToCam_evalPT", "f8", (12,)), ("state", "f8", (10,)), ("state_zero", "f8", (10,)), ("prior", "f8", (8,)),
    ("nullspaces_pose", "f8", (36,)), ("nullspaces_scale", "f8", (6,)), ("nullspaces_affine", "f8", (8,)),
    ("ab_exposure", "f4"), ("frameEnergyTH", "f4"), ("frameID", "i4"), ("pad_", "i4"),
], align=True)

CALIB_DTYPE = np.dtype([("value", "f8", (4,)), ("value_zero", "f8", (4,))], align=True)

POINT_DTYPE = np.dtype([
    ("u", "f4"), ("v", "f4"), ("idepth", "f4"), ("idepth_zero", "f4"), ("color", "f4", (8,)), ("weights", "f4", (8,)),
    ("priorF", "f4"), ("host", "i4"), ("res_begin", "i4"), ("res_count", "i4"),
], align=True)

RESIDUAL_DTYPE = np.dtype([
    ("point", "i4"), ("host", "i4"), ("target", "i4"), ("state_state", "i4"), ("is_linearized", "i4"),
    ("is_active", "i4"), ("is_new", "i4"), ("state_energy", "f4"),
], align=True)

RAWJAC_DTYPE = np.dtype([
    ("resF", "f4", (8,)), ("Jpdxi", "f4", (2, 6)), ("Jpdc", "f4", (2, 4)), ("Jpdd", "f4", (2,)),
    ("JIdx", "f4", (2, 8)), ("JabF", "f4", (2, 8)), ("JIdx2", "f4", (4,)), ("JabJIdx", "f4", (4,)), ("Jab2", "f4", (4,)),
], align=True)

RES_OUT_DTYPE = np.dtype([
    ("state_NewEnergy", "f4"), ("state_NewEnergyWithOutlier", "f4"), ("state_NewState", "i4"),
    ("centerProjectedTo", "f4", (3,)), ("JpJdF", "f4", (8,)),
], align=True)

POINT_OUT_DTYPE = np.dtype([
    ("step", "f4"), ("HdiF", "f4"), ("bdSumF", "f4"), ("idepth_hessian", "f4"),
    ("Hdd_accAF", "f4"), ("bd_accAF", "f4"), ("Hcd_accAF", "f4", (4,)),
    ("Hdd_accLF", "f4"), ("bd_accLF", "f4"), ("Hcd_accLF", "f4", (4,)),
    ("idepth", "f4"), ("maxRelBaseline", "f4"), ("numGoodResiduals", "i4"),
], align=True)

assert FRAME_DTYPE.itemsize == 736 and POINT_DTYPE.itemsize == 96 and RESIDUAL_DTYPE.itemsize == 32
assert RAWJAC_DTYPE.itemsize == 74 * 4 and RES_OUT_DTYPE.itemsize == 56 and POINT_OUT_DTYPE.itemsize == 76


def default_settings() -> np.ndarray:
    """Defaults of src/Setting.cc for the knobs the hot path reads."""
    s = np.zeros((), dtype=SETTINGS_DTYPE)
    s["huberTH"] = 9
    s["outlierTHSumComponent"] = 50 * 50
    s["affineOptModeA"] = 1e12
    s["affineOptModeB"] = 1e8
    s["frameEnergyTHN"] = 0.7
    s["frameEnergyTHFacMedian"] = 1.5
    s["frameEnergyTHConstWeight"] = 0.5
    s["overallEnergyTHWeight"] = 1
    s["initialCalibHessian"] = 5e9
    s["margWeightFac"] = 0.25
    s["idepthFixPriorMargFac"] = 600 * 600
    s["thOptIterations"] = 1.2
    s["coarseCutoffTH"] = 20
    s["minOptIterations"] = 1
    s["solverMode"] = 128 | 2048
    s["forceAcceptStep"] = 1
    s["solverModeDelta"] = 1e-5
    return s


# ----------------------------------------------------------------------------------------------
# SE3 helpers (double), tangent order (translation, rotation) as Sophus (thirdparty/sophus/se3.hpp)
# ----------------------------------------------------------------------------------------------
def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def so3_exp(w):
    th = np.linalg.norm(w)
    W = hat(w)
    if th < 1e-10:
        return np.eye(3) + W + 0.5 * W @ W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th**2 * (W @ W)


def se3_exp(xi):
    """xi = (upsilon, omega) -> 4x4."""
    xi = np.asarray(xi, dtype=np.float64)
    ups, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    R = so3_exp(w)
    W = hat(w)
    if th < 1e-10:
        V = R
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th**2 * W + (th - np.sin(th)) / th**3 * (W @ W)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def so3_log(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-10:
        return 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])


def se3_log(T):
    w = so3_log(T[:3, :3])
    th = np.linalg.norm(w)
    W = hat(w)
    if th < 1e-10:
        Vinv = np.eye(3) - 0.5 * W + (1.0 / 12.0) * (W @ W)
    else:
        Vinv = np.eye(3) - 0.5 * W + (1 - th / (2 * np.tan(th / 2))) / th**2 * (W @ W)
    return np.concatenate([Vinv @ T[:3, 3], w])


def se3_inv(T):
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


def nullspaces_for_evalpt(T_w2c, aff_a0, ab_exposure):
    """FrameHessian::setStateZero numeric nullspaces (reference src/internal/FrameHessian.cc:12-42)."""
    Ti = se3_inv(T_w2c)
    ns_pose = np.zeros((6, 6))
    for i in range(6):
        eps = np.zeros(6)
        eps[i] = 1e-3
        P = (T_w2c @ se3_exp(eps)) @ Ti
        M = (T_w2c @ se3_exp(-eps)) @ Ti
        ns_pose[:, i] = (se3_log(P) - se3_log(M)) / 2e-3
    Tp = T_w2c.copy()
    Tp[:3, 3] *= 1.00001
    Tm = T_w2c.copy()
    Tm[:3, 3] /= 1.00001
    ns_scale = (se3_log(Tp @ Ti) - se3_log(Tm @ Ti)) / 2e-3
    ns_aff = np.zeros((4, 2))
    ns_aff[0, 0] = 1
    ns_aff[1, 1] = np.float32(np.exp(np.float32(aff_a0))) * ab_exposure
    return ns_pose, ns_scale, ns_aff


# ----------------------------------------------------------------------------------------------
# scene
# ----------------------------------------------------------------------------------------------
class Scene:
    """Height field z = S(X,Y) with an analytic texture T(X,Y)."""

    def __init__(self, rng: np.random.Generator, z0=3.0, px=3.0 / 400.0):
        n = 24
        wl = rng.uniform(6, 80, n) * px                    # wavelength in world units (6..80 px at depth z0)
        ang = rng.uniform(0, 2 * np.pi, n)
        self.kx = 2 * np.pi * np.cos(ang) / wl
        self.ky = 2 * np.pi * np.sin(ang) / wl
        self.amp = rng.uniform(4, 24, n)
        self.ph = rng.uniform(0, 2 * np.pi, n)
        m = 5
        wl2 = rng.uniform(0.8, 3.0, m)
        ang2 = rng.uniform(0, 2 * np.pi, m)
        self.rkx = 2 * np.pi * np.cos(ang2) / wl2
        self.rky = 2 * np.pi * np.sin(ang2) / wl2
        self.ramp = rng.uniform(0.03, 0.10, m)
        self.rph = rng.uniform(0, 2 * np.pi, m)
        self.z0 = z0

    def surface(self, X, Y):
        z = np.full(X.shape, self.z0, dtype=np.float64)
        for i in range(len(self.ramp)):
            z += self.ramp[i] * np.sin(self.rkx[i] * X + self.rky[i] * Y + self.rph[i])
        return z

    def texture(self, X, Y):
        t = np.full(X.shape, 128.0, dtype=np.float64)
        for i in range(len(self.amp)):
            t += self.amp[i] * np.sin(self.kx[i] * X + self.ky[i] * Y + self.ph[i])
        return t

    def render(self, T_w2c, K, w, h, a, b, noise_rng=None, noise_sigma=0.0):
        """Ray-cast the scene into a camera. Returns (irradiance float32 [h,w], depth float64 [h,w])."""
        T_c2w = se3_inv(T_w2c)
        R, c = T_c2w[:3, :3], T_c2w[:3, 3]
        xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        dc = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], axis=-1)
        dw = dc @ R.T
        lam = np.full(xs.shape, self.z0 - c[2], dtype=np.float64) / dw[..., 2]
        for _ in range(12):
            X = c[0] + lam * dw[..., 0]
            Y = c[1] + lam * dw[..., 1]
            lam = (self.surface(X, Y) - c[2]) / dw[..., 2]
        X = c[0] + lam * dw[..., 0]
        Y = c[1] + lam * dw[..., 1]
        img = np.exp(a) * self.texture(X, Y) + b
        if noise_rng is not None and noise_sigma > 0:
            img = img + noise_rng.normal(0, noise_sigma, img.shape)
        img = np.clip(img, 1, 254).astype(np.float32)
        return img, lam


# ----------------------------------------------------------------------------------------------
# FrameHessian::makeImages restated for input generation (FrameHessian.cc:44-113)
# ----------------------------------------------------------------------------------------------
def pyr_levels_used(w, h):
    """setGlobalCalib rule (reference src/internal/GlobalCalib.cc:20-29)."""
    lv, wl, hl = 1, w, h
    while wl % 2 == 0 and hl % 2 == 0 and wl * hl > 5000 and lv < 6:
        wl //= 2
        hl //= 2
        lv += 1
    return lv


def make_images(color: np.ndarray, levels: int):
    """Returns list of float32 [h_l, w_l, 3] AoS (I, dx, dy) per level."""
    out = []
    I = color.astype(np.float32)
    for lvl in range(levels):
        if lvl > 0:
            p = out[-1][..., 0]
            hl, wl = p.shape[0] // 2, p.shape[1] // 2
            I = (np.float32(0.25) * (((p[0:2 * hl:2, 0:2 * wl:2] + p[0:2 * hl:2, 1:2 * wl:2]) + p[1:2 * hl:2, 0:2 * wl:2])
                                     + p[1:2 * hl:2, 1:2 * wl:2])).astype(np.float32)
        hl, wl = I.shape
        d = np.zeros((hl * wl, 3), dtype=np.float32)
        flat = np.ascontiguousarray(I).ravel()
        d[:, 0] = flat
        lo, hi = wl, wl * (hl - 1)
        dx = np.float32(0.5) * (flat[lo + 1:hi + 1] - flat[lo - 1:hi - 1])
        dy = np.float32(0.5) * (flat[lo + wl:hi + wl] - flat[lo - wl:hi - wl])
        dx[np.isnan(dx) | (np.abs(dx) > 255.0)] = 0
        dy[np.isnan(dy) | (np.abs(dy) > 255.0)] = 0
        d[lo:hi, 1] = dx
        d[lo:hi, 2] = dy
        out.append(d.reshape(hl, wl, 3))
    return out


def interp_bilin33(dI, x, y):
    """getInterpolatedElement33BiLin (GlobalFuncs.h:186-207), vectorised; dI float32 [h,w,3]."""
    x = x.astype(np.float32)
    y = y.astype(np.float32)
    ix = x.astype(np.int32)
    iy = y.astype(np.int32)
    tl = dI[iy, ix, 0]
    tr = dI[iy, ix + 1, 0]
    bl = dI[iy + 1, ix, 0]
    br = dI[iy + 1, ix + 1, 0]
    dx = x - ix.astype(np.float32)
    dy = y - iy.astype(np.float32)
    one = np.float32(1)
    top = dx * tr + (one - dx) * tl
    bot = dx * br + (one - dx) * bl
    left = dy * bl + (one - dy) * tl
    right = dy * br + (one - dy) * tr
    return dx * right + (one - dx) * left, right - left, bot - top


# ----------------------------------------------------------------------------------------------
# window
# ----------------------------------------------------------------------------------------------
@dataclasses.dataclass
class Window:
    w: int
    h: int
    levels: int
    K: np.ndarray                      # 3x3 float64 pixel-unit intrinsics (level 0)
    settings: np.ndarray               # () SETTINGS_DTYPE
    calib: np.ndarray                  # () CALIB_DTYPE
    frames: np.ndarray                 # [F] FRAME_DTYPE
    points: np.ndarray                 # [P] POINT_DTYPE
    residuals: np.ndarray              # [R] RESIDUAL_DTYPE
    images: list                       # F lists of per-level float32 [h_l,w_l,3]
    HM: np.ndarray                     # (8F+4)^2 float64 marginalisation prior
    bM: np.ndarray
    lin_J: np.ndarray | None = None    # [R] RAWJAC_DTYPE, valid where is_linearized
    lin_res_toZeroF: np.ndarray | None = None   # [R,8] float32
    truth: dict | None = None

    @property
    def F(self):
        return len(self.frames)

    @property
    def P(self):
        return len(self.points)

    @property
    def R(self):
        return len(self.residuals)


def frame_prior(frame_id, s):
    """FrameHessian::getPrior (FrameHessian.h:129-154) with default priors (Setting.cc:18-21).  The reference's settings are
    `float` globals, so the values that reach the double prior vector are float32(1e10), float32(1e11) = 99999997952 and
    float32(1e14) = 100000000376832 (pinned against the reference-compiled getPrior, tests/test_ref_pin.py)."""
    f32 = lambda v: float(np.float32(v))
    p = np.zeros(8)
    if frame_id == 0:
        p[0:3] = f32(1e10)
        p[3:6] = f32(1e11)
        p[6] = f32(1e14)
        p[7] = f32(1e14)
    else:
        p[6] = f32(1e14) if s["affineOptModeA"] < 0 else float(s["affineOptModeA"])
        p[7] = f32(1e14) if s["affineOptModeB"] < 0 else float(s["affineOptModeB"])
    return p


def make_window(F=7, P=2000, w=640, h=480, seed=20260925, fx=None, fy=None, cx=None, cy=None,
                first_frame_id=0, idepth_noise=0.02, pose_noise_t=2e-3, pose_noise_r=2e-4,
                state_noise=True, noise_sigma=1.0, extra_frames=0, levels=None) -> Window:
    rng = np.random.default_rng(seed)
    fx = 400.0 if fx is None else fx
    fy = fx if fy is None else fy
    cx = (w - 1) / 2.0 if cx is None else cx
    cy = (h - 1) / 2.0 if cy is None else cy
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)
    levels = pyr_levels_used(w, h) if levels is None else levels
    scene = Scene(rng, px=3.0 / fx)
    s = default_settings()

    nF = F + extra_frames
    # true poses on a smooth arc (camera-to-world), then world-to-camera
    true_w2c = []
    pos = np.zeros(3)
    rot = np.zeros(3)
    for k in range(nF):
        if k > 0:
            pos = pos + np.array([rng.uniform(0.03, 0.06), rng.uniform(-0.01, 0.01), rng.uniform(-0.01, 0.01)])
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            rot = rot + ax * np.deg2rad(rng.uniform(0.1, 0.5))
        T_c2w = np.eye(4)
        T_c2w[:3, :3] = so3_exp(rot)
        T_c2w[:3, 3] = pos
        true_w2c.append(se3_inv(T_c2w))
    aff_a = rng.uniform(-0.05, 0.05, nF)
    aff_b = rng.uniform(-3, 3, nF)
    aff_a[0] = 0
    aff_b[0] = 0

    images, depths = [], []
    for k in range(nF):
        img, dep = scene.render(true_w2c[k], K, w, h, aff_a[k], aff_b[k], rng, noise_sigma)
        images.append(make_images(img, levels))
        depths.append(dep)

    # frames
    frames = np.zeros(F, dtype=FRAME_DTYPE)
    for k in range(F):
        if k == 0:
            T_eval = true_w2c[k]
        else:
            xi = np.concatenate([rng.normal(0, pose_noise_t, 3), rng.normal(0, pose_noise_r, 3)])
            T_eval = se3_exp(xi) @ true_w2c[k]
        a0 = aff_a[k] + (rng.normal(0, 0.005) if k > 0 else 0)
        b0 = aff_b[k] + (rng.normal(0, 0.3) if k > 0 else 0)
        st0 = np.zeros(10)
        st0[6] = a0 / SCALE_A
        st0[7] = b0 / SCALE_B
        st = st0.copy()
        if state_noise and 0 < k < F - 1:
            st[0:3] += rng.normal(0, 1e-3, 3)
            st[3:6] += rng.normal(0, 1e-4, 3)
            st[6] += rng.normal(0, 2e-4)
            st[7] += rng.normal(0, 1e-4)
        fr = frames[k]
        fr["worldToCam_evalPT"] = T_eval[:3, :4].ravel()
        fr["state"] = st
        fr["state_zero"] = st0
        fid = first_frame_id + k
        fr["frameID"] = fid
        fr["prior"] = frame_prior(fid, s)
        fr["ab_exposure"] = 1.0
        fr["frameEnergyTH"] = 8 * 8 * 8
        nsP, nsS, nsA = nullspaces_for_evalpt(T_eval, st0[6] * SCALE_A, 1.0)
        fr["nullspaces_pose"] = nsP.ravel()
        fr["nullspaces_scale"] = nsS
        fr["nullspaces_affine"] = nsA.ravel()

    calib = np.zeros((), dtype=CALIB_DTYPE)
    calib["value"] = [fx / SCALE_F, fy / SCALE_F, cx / SCALE_C, cy / SCALE_C]
    calib["value_zero"] = calib["value"]

    # points
    per = [P // F + (1 if k < P % F else 0) for k in range(F)]
    pts = np.zeros(P, dtype=POINT_DTYPE)
    res_list = []
    n = 0
    for k in range(F):
        dI = images[k][0]
        got = 0
        us = np.zeros(per[k], np.float32)
        vs = np.zeros(per[k], np.float32)
        while got < per[k]:
            m = (per[k] - got) * 2 + 16
            cu = rng.integers(16, w - 16, m)
            cv = rng.integers(16, h - 16, m)
            g2 = dI[cv, cu, 1] ** 2 + dI[cv, cu, 2] ** 2
            ok = np.nonzero(g2 >= 50)[0][: per[k] - got]
            us[got:got + len(ok)] = cu[ok]
            vs[got:got + len(ok)] = cv[ok]
            got += len(ok)
        true_id = 1.0 / depths[k][vs.astype(int), us.astype(int)]
        ide = (true_id * (1 + rng.normal(0, idepth_noise, per[k]))).astype(np.float32)
        sl = slice(n, n + per[k])
        pts["u"][sl] = us
        pts["v"][sl] = vs
        pts["idepth"][sl] = ide
        pts["idepth_zero"][sl] = ide
        pts["host"][sl] = k
        pts["priorF"][sl] = 0.0
        for j in range(8):
            c, gx, gy = interp_bilin33(dI, us + PATTERN[j, 0], vs + PATTERN[j, 1])
            pts["color"][sl, j] = c
            pts["weights"][sl, j] = np.sqrt(np.float32(2500.0) / (np.float32(2500.0) + (gx * gx + gy * gy))).astype(np.float32)
        n += per[k]

    # residuals: point -> every other frame, in frame order
    res = np.zeros(P * (F - 1), dtype=RESIDUAL_DTYPE)
    r = 0
    for i in range(P):
        pts["res_begin"][i] = r
        hst = int(pts["host"][i])
        for t in range(F):
            if t == hst:
                continue
            res[r] = (i, hst, t, 0, 0, 0, 1, 0.0)
            r += 1
        pts["res_count"][i] = r - pts["res_begin"][i]

    n8 = 8 * F + 4
    win = Window(w=w, h=h, levels=levels, K=K, settings=s, calib=calib, frames=frames, points=pts, residuals=res,
                 images=[im for im in images], HM=np.zeros((n8, n8)), bM=np.zeros(n8),
                 truth=dict(w2c=np.stack(true_w2c), aff_a=aff_a, aff_b=aff_b, depths=depths if extra_frames else None))
    return win


CONFIGS = {
    "C3": dict(F=7, P=2000, w=640, h=480),
    "C4": dict(F=7, P=3000, w=1232, h=368, fx=718.856, fy=718.856, cx=607.1928 - 4.5, cy=185.2157 - 4.0),
    "C5": dict(F=12, P=8000, w=640, h=480),
    "tiny": dict(F=4, P=64, w=160, h=128, fx=120.0),
    "small": dict(F=5, P=400, w=320, h=240, fx=200.0),
}


# ----------------------------------------------------------------------------------------------
# immature points (ImmaturePoint::traceOn)
# ----------------------------------------------------------------------------------------------
IMMATURE_DTYPE = np.dtype([
    ("u", "f4"), ("v", "f4"), ("color", "f4", (8,)), ("weights", "f4", (8,)), ("gradH", "f4", (4,)), ("energyTH", "f4"),
    ("idepth_min", "f4"), ("idepth_max", "f4"), ("quality", "f4"), ("lastTraceStatus", "i4"), ("lastTraceUV", "f4", (2,)),
    ("lastTracePixelInterval", "f4"), ("host", "i4"), ("pad_", "i4"),
], align=True)
assert IMMATURE_DTYPE.itemsize == 128

TRACE_SETTINGS_DTYPE = np.dtype([
    ("maxPixSearch", "f4"), ("trace_stepsize", "f4"), ("trace_GNThreshold", "f4"), ("trace_extraSlackOnTH", "f4"),
    ("trace_slackInterval", "f4"), ("trace_minImprovementFactor", "f4"), ("huberTH", "f4"), ("trace_GNIterations", "i4"),
    ("minTraceTestRadius", "i4"), ("pad_", "i4"),
], align=True)
assert TRACE_SETTINGS_DTYPE.itemsize == 40

ACTIVATION_DTYPE = np.dtype([
    ("idepth", "f4"), ("ok", "i4"), ("numGoodRes", "i4"), ("iterations", "i4"), ("energy", "f4"), ("Hdd", "f4"), ("bd", "f4"), ("pad_", "i4"),
    ("res_state", "i4", (16,)),
], align=True)
assert ACTIVATION_DTYPE.itemsize == 96


def default_trace_settings():
    s = np.zeros((), TRACE_SETTINGS_DTYPE)
    s["maxPixSearch"] = 0.027; s["trace_stepsize"] = 1.0; s["trace_GNThreshold"] = 0.1; s["trace_extraSlackOnTH"] = 1.2
    s["trace_slackInterval"] = 1.5; s["trace_minImprovementFactor"] = 2; s["huberTH"] = 9; s["trace_GNIterations"] = 3
    s["minTraceTestRadius"] = 2
    return s


def make_immature_points(win: "Window", per_frame: int, seed: int = 7, frames=None, hosts=None):
    """Fresh immature points on the key frames of a window, as the ImmaturePoint constructor makes them
    (ImmaturePoint.cc:14-38): pattern colours / weights / gradH at gradient-rich pixels; idepth interval [0, NaN).
    frames: indices into win.images (default: the window's F key frames; the extra frames of make_window(extra_frames=...) are F, F+1, ...),
    hosts: the `host` index each of them gets (default: the image index)."""
    rng = np.random.default_rng(seed)
    frames = list(range(win.F)) if frames is None else list(frames)
    hosts = frames if hosts is None else list(hosts)
    F = len(frames)
    out = np.zeros(F * per_frame, IMMATURE_DTYPE)
    true_id = np.zeros(F * per_frame, np.float32)
    n = 0
    for k, hostIdx in zip(frames, hosts):
        dI = win.images[k][0]
        us, vs = [], []
        while len(us) < per_frame:
            cu = rng.integers(16, win.w - 16, per_frame * 2 + 16)
            cv = rng.integers(16, win.h - 16, per_frame * 2 + 16)
            g2 = dI[cv, cu, 1] ** 2 + dI[cv, cu, 2] ** 2
            ok = np.nonzero(g2 >= 50)[0]
            us += list(cu[ok]); vs += list(cv[ok])
        us = np.asarray(us[:per_frame], np.float32); vs = np.asarray(vs[:per_frame], np.float32)
        sl = slice(n, n + per_frame)
        out["u"][sl] = us; out["v"][sl] = vs; out["host"][sl] = hostIdx
        gh = np.zeros((per_frame, 4), np.float32)
        for j in range(8):
            c, gx, gy = interp_bilin33(dI, us + PATTERN[j, 0], vs + PATTERN[j, 1])
            out["color"][sl, j] = c
            out["weights"][sl, j] = np.sqrt(np.float32(2500.0) / (np.float32(2500.0) + (gx * gx + gy * gy))).astype(np.float32)
            gh[:, 0] += gx * gx; gh[:, 1] += gx * gy; gh[:, 2] += gx * gy; gh[:, 3] += gy * gy
        out["gradH"][sl] = gh
        if win.truth is not None and win.truth.get("depths") is not None:
            true_id[sl] = 1.0 / win.truth["depths"][k][vs.astype(int), us.astype(int)]
        n += per_frame
    out["energyTH"] = 8 * 12 * 12
    out["idepth_min"] = 0.0
    out["idepth_max"] = np.nan
    out["quality"] = 10000.0
    out["lastTraceStatus"] = 5          # IPS_UNINITIALIZED
    out["lastTraceUV"] = -1.0
    return out, true_id


def trace_poses(win: "Window", new_index: int):
    """Per host key frame of the window: K R K^-1, K t and the affine brightness pair towards frame `new_index`
    (FullSystem::traceNewCoarse, FullSystem.cc:1018-1032), from the ground-truth poses of the synthetic scene."""
    K = win.K.astype(np.float32)
    Ki = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    F = win.F
    KRKi = np.zeros((F, 9), np.float32); Kt = np.zeros((F, 3), np.float32); aff = np.zeros((F, 2), np.float32)
    Tn = win.truth["w2c"][new_index]
    for k in range(F):
        T = Tn @ np.linalg.inv(win.truth["w2c"][k])
        R = T[:3, :3].astype(np.float32); t = T[:3, 3].astype(np.float32)
        KRKi[k] = (K @ R @ Ki).ravel(); Kt[k] = K @ t
        a = np.float32(np.exp(np.float32(win.truth["aff_a"][new_index] - win.truth["aff_a"][k])))
        aff[k] = (a, np.float32(win.truth["aff_b"][new_index]) - a * np.float32(win.truth["aff_b"][k]))
    return KRKi, Kt, aff


def add_synthetic_prior(win: Window, seed: int = 11, rank: int = 6, scale: float = 2e3, b_scale: float = 5.0) -> Window:
    """Give the window a marginalisation prior (H_M symmetric PSD of the given rank, b_M), as every steady-state LDSO window
    has one (EnergyFunctional::marginalizeFrame).  Synthetic: it only has to be the same on both sides of a comparison."""
    rng = np.random.default_rng(seed)
    n = win.HM.shape[0]
    A = rng.standard_normal((n, rank))
    win.HM = scale * (A @ A.T)
    win.bM = b_scale * rng.standard_normal(n)
    return win


def make_config(name: str, **over) -> Window:
    kw = dict(CONFIGS[name])
    kw.update(over)
    return make_window(**kw)


# ----------------------------------------------------------------------------------------------
# monocular initialiser (CoarseInitializer)
# ----------------------------------------------------------------------------------------------
INIT_POINT_DTYPE = np.dtype([
    ("u", "f4"), ("v", "f4"), ("idepth", "f4"), ("isGood", "i4"), ("energy", "f4", (2,)), ("isGood_new", "i4"), ("idepth_new", "f4"),
    ("energy_new", "f4", (2,)), ("iR", "f4"), ("iRSumNum", "f4"), ("lastHessian", "f4"), ("lastHessian_new", "f4"), ("maxstep", "f4"),
    ("parent", "i4"), ("parentDist", "f4"), ("neighbours", "i4", (10,)), ("neighboursDist", "f4", (10,)), ("my_type", "f4"),
    ("outlierTH", "f4"), ("pad_", "f4"),
], align=True)
assert INIT_POINT_DTYPE.itemsize == 160

INIT_STATE_DTYPE = np.dtype([
    ("thisToNext", "f8", (12,)), ("aff_a", "f8"), ("aff_b", "f8"), ("snapped", "i4"), ("snappedAt", "i4"), ("frameID", "i4"),
    ("ready", "i4"), ("evals", "i4"), ("pad_", "i4"),
], align=True)
assert INIT_STATE_DTYPE.itemsize == 136

INIT_DENSITIES = (0.03, 0.05, 0.15, 0.5, 1.0)     # CoarseInitializer.cc:557


def select_init_points(pyr, densities=INIT_DENSITIES, grad_th=8.0, outlier_th=12.0 * 12.0):
    """Stand-in for the pixel selection of CoarseInitializer::setFirst (PixelSelector::makeMaps / makePixelStatus, not on the
    path) plus an exact restatement of the point records (:567-603) and of makeNN (:717-783; k-d tree queries by scipy).
    pyr: list of float32 [h,w,3] (I,dx,dy).  Returns one INIT_POINT_DTYPE array per level, points in raster order."""
    from scipy.spatial import cKDTree
    levels = len(pyr)
    h0, w0 = pyr[0].shape[:2]
    out = []
    for lvl in range(levels):
        hl, wl = pyr[lvl].shape[:2]
        g2 = pyr[lvl][..., 1] ** 2 + pyr[lvl][..., 2] ** 2
        want = densities[lvl] * w0 * h0
        x0, x1, y0, y1 = 3, wl - 4, 3, hl - 4              # patternPadding + 1 .. wl - patternPadding - 2 (exclusive)
        area = max(1, (x1 - x0) * (y1 - y0))
        bs = max(1, int(round(np.sqrt(area / want))))
        sel = np.zeros((hl, wl), dtype=bool)
        sub = g2[y0:y1, x0:x1]
        nby, nbx = (sub.shape[0] + bs - 1) // bs, (sub.shape[1] + bs - 1) // bs
        pad = np.full((nby * bs, nbx * bs), -1.0, dtype=np.float32)
        pad[:sub.shape[0], :sub.shape[1]] = sub
        blk = pad.reshape(nby, bs, nbx, bs).transpose(0, 2, 1, 3).reshape(nby, nbx, bs * bs)
        am = blk.argmax(axis=2)
        mx = blk.max(axis=2)
        by, bx = np.nonzero(mx > grad_th)
        yy = y0 + by * bs + am[by, bx] // bs
        xx = x0 + bx * bs + am[by, bx] % bs
        sel[yy, xx] = True
        ys, xs = np.nonzero(sel)                            # raster order
        pts = np.zeros(len(xs), INIT_POINT_DTYPE)
        pts["u"] = (xs + 0.1).astype(np.float32)
        pts["v"] = (ys + 0.1).astype(np.float32)
        pts["idepth"] = 1; pts["idepth_new"] = 1; pts["iR"] = 1; pts["isGood"] = 1
        pts["my_type"] = 1
        pts["outlierTH"] = 8 * outlier_th
        pts["parent"] = -1; pts["parentDist"] = -1
        pts["neighbours"] = -1
        out.append(pts)
    trees = [cKDTree(np.stack([p["u"], p["v"]], axis=1).astype(np.float32)) if len(p) else None for p in out]
    for lvl in range(levels):
        p = out[lvl]
        if len(p) == 0:
            continue
        xy = np.stack([p["u"], p["v"]], axis=1).astype(np.float32)
        k = min(10, len(p))
        d, idx = trees[lvl].query(xy, k=k)
        d = d.reshape(len(p), k); idx = idx.reshape(len(p), k)
        df = np.exp(-(d.astype(np.float32) ** 2) * np.float32(0.05)).astype(np.float32)
        p["neighbours"][:, :k] = idx
        p["neighboursDist"][:, :k] = df * (np.float32(10) / df.sum(axis=1, keepdims=True))
        if lvl < levels - 1 and trees[lvl + 1] is not None:
            q = xy * np.float32(0.5) - np.float32(0.25)
            d1, i1 = trees[lvl + 1].query(q, k=1)
            p["parent"] = i1
            p["parentDist"] = np.exp(-(d1.astype(np.float32) ** 2) * np.float32(0.05))
    return out


def make_init_sequence(w=640, h=480, n_frames=10, seed=20260925, fx=None, step=0.035, rot_deg=0.15, noise_sigma=0.5, levels=None):
    """First frame + n_frames followers of the synthetic scene: camera translating (mostly sideways) by `step` scene units per
    frame at depth ~3.  Returns dict(K4, levels, first (irradiance), frames [irradiance], T_first_to_k [4x4], depth0 [h,w])."""
    rng = np.random.default_rng(seed)
    fx = 400.0 if fx is None else fx
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    K = np.array([[fx, 0, cx], [0, fx, cy], [0, 0, 1]], dtype=np.float64)
    levels = pyr_levels_used(w, h) if levels is None else levels
    scene = Scene(rng, px=3.0 / fx)
    nrng = np.random.default_rng(seed + 1)
    first, depth0 = scene.render(np.eye(4), K, w, h, 0.0, 0.0, nrng, noise_sigma)
    frames, poses = [], []
    pos, rot = np.zeros(3), np.zeros(3)
    for k in range(n_frames):
        pos = pos + np.array([step * rng.uniform(0.8, 1.2), step * rng.uniform(-0.2, 0.2), step * rng.uniform(-0.2, 0.2)])
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        rot = rot + ax * np.deg2rad(rng.uniform(0.5, 1.0) * rot_deg)
        T_c2w = np.eye(4); T_c2w[:3, :3] = so3_exp(rot); T_c2w[:3, 3] = pos
        T_w2c = se3_inv(T_c2w)
        img, _ = scene.render(T_w2c, K, w, h, 0.0, 0.0, nrng, noise_sigma)
        frames.append(img); poses.append(T_w2c)          # first frame is the world frame: T_first_to_k = T_w2c
    return dict(K4=np.array([fx, fx, cx, cy], dtype=np.float32), levels=levels, w=w, h=h, first=first, frames=frames, poses=poses, depth0=depth0)
