// lie_dev.h — double-precision SE(3) helpers usable from host and device code.
// Poses are stored as row-major [R|t] 3x4 (12 doubles).  Tangent order is (translation, rotation),
// matching the reference's Sophus convention (thirdparty/sophus/se3.hpp:407-428, :131-139, :560-585).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LD_HD __host__ __device__ __forceinline__
#else
#define LD_HD inline
#endif

#ifndef LD_EXP_SERIES_TH2
#define LD_EXP_SERIES_TH2 0.25      // se3_exp: |omega|^2 below which the coefficient series replace the closed forms (0: always the closed forms)
#endif

namespace ld {

LD_HD void mat3_mul(const double *A, const double *B, double *C) {   // C = A*B (3x3 row-major)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}

LD_HD void hat3(const double *w, double *W) {
    W[0] = 0; W[1] = -w[2]; W[2] = w[1];
    W[3] = w[2]; W[4] = 0; W[5] = -w[0];
    W[6] = -w[1]; W[7] = w[0]; W[8] = 0;
}

// T = exp(xi), xi = (upsilon, omega); T is [R|t] 3x4
LD_HD void se3_exp(const double *xi, double *T) {
    const double *ups = xi, *om = xi + 3;
    double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    // W = hat(omega) and W^2 written out: the terms the 3 x 3 product forms minus its products with the zeros of W (bitwise the same values for finite
    // input - a compiler may not drop x * 0 - at 9 instead of 45 instructions; this runs on one lane of a latency-bound control step)
    const double w0 = om[0], w1 = om[1], w2 = om[2];
    const double W[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
    const double q01 = w1 * w0, q02 = w2 * w0, q12 = w2 * w1;
    const double W2[9] = {-(w2 * w2 + w1 * w1), q01, q02, q01, -(w2 * w2 + w0 * w0), q12, q02, q12, -(w1 * w1 + w0 * w0)};
    double a, b, c;   // R = I + a W + b W^2 ; V = I + b W + c W^2
    if (th2 < LD_EXP_SERIES_TH2) {
        // a = sin(th)/th, b = (1 - cos th)/th^2, c = (th - sin th)/th^3 are power series in th^2: for |th| < 0.5 nine terms reach 1e-19, with
        // no square root, no division, no trigonometric call (every one of them a dependent chain of 30..140 ns on one lane of the control
        // step / the tracker's leader, which call this on their critical path) and without the cancellation of the closed forms at small th
        const double x = th2;
        a = 1.0 + x * (-1.0 / 6 + x * (1.0 / 120 + x * (-1.0 / 5040 + x * (1.0 / 362880 + x * (-1.0 / 39916800 + x * (1.0 / 6227020800.0 + x * (-1.0 / 1307674368000.0 + x * (1.0 / 355687428096000.0))))))));
        b = 0.5 + x * (-1.0 / 24 + x * (1.0 / 720 + x * (-1.0 / 40320 + x * (1.0 / 3628800 + x * (-1.0 / 479001600 + x * (1.0 / 87178291200.0 + x * (-1.0 / 20922789888000.0 + x * (1.0 / 6402373705728000.0))))))));
        c = 1.0 / 6 + x * (-1.0 / 120 + x * (1.0 / 5040 + x * (-1.0 / 362880 + x * (1.0 / 39916800 + x * (-1.0 / 6227020800.0 + x * (1.0 / 1307674368000.0 + x * (-1.0 / 355687428096000.0 + x * (1.0 / 121645100408832000.0))))))));
    } else {
        const double th = sqrt(th2), sn = sin(th), cs = cos(th);
        a = sn / th; b = (1.0 - cs) / th2; c = (th - sn) / (th2 * th);
    }
    double V[9];
    for (int i = 0; i < 9; i++) {
        const bool diag = (i == 0 || i == 4 || i == 8);          // W is zero there: (1 + a 0) + b W2 = 1 + b W2; elsewhere (0 + a W) + b W2 = a W + b W2
        T[(i / 3) * 4 + (i % 3)] = diag ? 1.0 + b * W2[i] : a * W[i] + b * W2[i];
        V[i] = diag ? 1.0 + c * W2[i] : b * W[i] + c * W2[i];
    }
    for (int i = 0; i < 3; i++) T[i * 4 + 3] = V[i * 3 + 0] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
}

// C = A * B for [R|t]
LD_HD void se3_mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) C[i * 4 + j] = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j] + A[i * 4 + 2] * B[2 * 4 + j];
        C[i * 4 + 3] = A[i * 4 + 0] * B[0 * 4 + 3] + A[i * 4 + 1] * B[1 * 4 + 3] + A[i * 4 + 2] * B[2 * 4 + 3] + A[i * 4 + 3];
    }
}

LD_HD void se3_inv(const double *A, double *C) {
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) C[i * 4 + j] = A[j * 4 + i];
    }
    for (int i = 0; i < 3; i++) C[i * 4 + 3] = -(A[0 * 4 + i] * A[0 * 4 + 3] + A[1 * 4 + i] * A[1 * 4 + 3] + A[2 * 4 + i] * A[2 * 4 + 3]);
}

// rotation log: omega from R (row-major 3x4 pose)
LD_HD void so3_log(const double *T, double *om) {
    double tr = T[0] + T[5] + T[10];
    double c = 0.5 * (tr - 1.0);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    double vx = T[2 * 4 + 1] - T[1 * 4 + 2], vy = T[0 * 4 + 2] - T[2 * 4 + 0], vz = T[1 * 4 + 0] - T[0 * 4 + 1];
    double s = 0.5 * sqrt(vx * vx + vy * vy + vz * vz);   // sin(theta)
    double th = atan2(s, c);
    double f = (th < 1e-10) ? 0.5 : th / (2.0 * s);
    om[0] = f * vx; om[1] = f * vy; om[2] = f * vz;
}

LD_HD void se3_log(const double *T, double *xi) {
    double om[3];
    so3_log(T, om);
    double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    double th = sqrt(th2);
    double W[9], W2[9];
    hat3(om, W);
    mat3_mul(W, W, W2);
    double k = (th < 1e-10) ? (1.0 / 12.0) : (1.0 - th / (2.0 * tan(0.5 * th))) / th2;
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int j = 0; j < 3; j++) {
            double I = (i == j) ? 1.0 : 0.0;
            s += (I - 0.5 * W[i * 3 + j] + k * W2[i * 3 + j]) * T[j * 4 + 3];
        }
        xi[i] = s;
    }
    xi[3] = om[0]; xi[4] = om[1]; xi[5] = om[2];
}

// Adj(T) 6x6 row-major: [[R, t^ R],[0, R]]
LD_HD void se3_adj(const double *T, double *A) {
    double tx[9], R[9], tR[9];
    double t[3] = {T[3], T[7], T[11]};
    hat3(t, tx);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = T[i * 4 + j];
    mat3_mul(tx, R, tR);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[i * 6 + j] = R[i * 3 + j];
            A[(3 + i) * 6 + 3 + j] = R[i * 3 + j];
            A[i * 6 + 3 + j] = tR[i * 3 + j];
            A[(3 + i) * 6 + j] = 0.0;
        }
}

// FrameHessian::setStateZero numeric nullspaces (reference src/internal/FrameHessian.cc:12-42).
// ns_pose: 6x6 row-major, column i = i-th pose nullspace; ns_scale: 6; ns_affine: 4x2 row-major.
LD_HD void frame_nullspaces(const double *evalPT, double state_zero_a, float ab_exposure, double *ns_pose, double *ns_scale, double *ns_affine) {
    double Ti[12], A[12], Bm[12], lp[6], lm[6];
    se3_inv(evalPT, Ti);
    for (int i = 0; i < 6; i++) {
        double eps[6] = {0, 0, 0, 0, 0, 0};
        eps[i] = 1e-3;
        double E[12];
        se3_exp(eps, E); se3_mul(evalPT, E, A); se3_mul(A, Ti, Bm); se3_log(Bm, lp);
        eps[i] = -1e-3;
        se3_exp(eps, E); se3_mul(evalPT, E, A); se3_mul(A, Ti, Bm); se3_log(Bm, lm);
        for (int r = 0; r < 6; r++) ns_pose[r * 6 + i] = (lp[r] - lm[r]) / 2e-3;
    }
    double Tp[12], Tm[12];
    for (int i = 0; i < 12; i++) { Tp[i] = evalPT[i]; Tm[i] = evalPT[i]; }
    for (int i = 0; i < 3; i++) { Tp[i * 4 + 3] *= 1.00001; Tm[i * 4 + 3] /= 1.00001; }
    se3_mul(Tp, Ti, A); se3_log(A, lp);
    se3_mul(Tm, Ti, A); se3_log(A, lm);
    for (int r = 0; r < 6; r++) ns_scale[r] = (lp[r] - lm[r]) / 2e-3;
    for (int i = 0; i < 8; i++) ns_affine[i] = 0;
    ns_affine[0] = 1;
    ns_affine[3] = (double) (expf((float) (state_zero_a * 10.0)) * ab_exposure);
}

}  // namespace ld
