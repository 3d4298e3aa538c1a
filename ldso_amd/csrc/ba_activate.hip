// ba_activate.hip — point activation on the device (gfx950): FullSystem::optimizeImmaturePoint (reference
// src/frontend/FullSystem.cc:892-1010) with ImmaturePoint::linearizeResidual (src/internal/ImmaturePoint.cc:312-381), projectPoint
// (include/internal/ResidualProjections.h:57-84) and derive_idepth (:12-18) for a batch of immature points against the key frames
// of the window that is resident in the bundle-adjustment handle (images, calibration, pair transforms of the current state).
//
// Mapping: one wavefront per immature point, lane = slot*8 + k like the linearize kernel (slot = target frame, k = pattern pixel;
// two slot groups for F > 8).  All sums follow the reference's order exactly - the pattern energy of a residual lane by lane,
// Hdd / bd as ONE running sum over (residual, pattern pixel) including the partial contributions of a residual that goes out of
// bounds half-way through its pattern - so idepth, energies and residual states are bit-identical to the CPU path.
#include <hip/hip_runtime.h>
#include "ba_dev.h"

struct ActArgs {
    const ldso_immature_t *pts;
    ldso_activation_t *out;
    int n, minObs, GNIts;
    float minIdepthH_act;
};

template <int CTRL> static __device__ __forceinline__ int act_dpp(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <int CTRL> static __device__ __forceinline__ float act_dppf(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true)); }
static __device__ __forceinline__ int min8(int v) {      // minimum over the 8 lanes of a slot, in all 8 lanes
    v = min(v, act_dpp<0xB1>(v)); v = min(v, act_dpp<0x4E>(v)); v = min(v, act_dpp<0x141>(v));
    return v;
}
static __device__ __forceinline__ float rl(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }

template <int NSG>
__global__ __launch_bounds__(256) void k_activate(BaPtrs B, BaDims D, ldso_settings_t S, ActArgs A) {
    const int lane = threadIdx.x & 63, s = lane >> 3, k = lane & 7;
    const int i = __builtin_amdgcn_readfirstlane((int) ((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (i >= A.n) return;
    const ldso_immature_t &P = A.pts[i];
    const int F = D.F, host = P.host;
    ldso_activation_t &O = A.out[i];
    if (host < 0 || host >= F) { if (lane == 0) { O.ok = 0; O.idepth = 0; O.numGoodRes = 0; O.iterations = 0; O.energy = 0; O.Hdd = 0; O.bd = 0; } if (lane < LDSO_MAX_FRAMES) O.res_state[lane] = -1; return; }
    const float fx = B.calib->sf[0], fy = B.calib->sf[1], cx = B.calib->sf[2], cy = B.calib->sf[3], fxi = B.calib->si[0], fyi = B.calib->si[1];
    const int ox = (k == 1 || k == 6) ? -1 : (k == 2) ? 1 : (k == 3) ? -2 : (k == 5) ? 2 : 0;     // staticPattern[8], Setting.cc:221
    const int oy = (k == 0) ? -2 : (k == 1 || k == 2) ? -1 : (k == 6) ? 1 : (k == 7) ? 2 : 0;
    const float pu = P.u, pv = P.v, col = P.color[k], wk = P.weights[k], energyTH = P.energyTH;
    const int W = D.w;
    // per slot (uniform over its 8 lanes): target validity, pair transform, residual state
    bool valid[NSG];
    float R[NSG][9], t[NSG][3], aff0[NSG], aff1[NSG];
    int st[NSG], nst[NSG];
    double sten[NSG], nen[NSG];
#pragma unroll
    for (int g = 0; g < NSG; g++) {
        const int tg = g * 8 + s;
        valid[g] = tg < F && tg != host;
        const int pi = host * F + (valid[g] ? tg : host);
#pragma unroll
        for (int q = 0; q < 9; q++) R[g][q] = B.pairRt[(size_t) pi * 12 + q];
#pragma unroll
        for (int q = 0; q < 3; q++) t[g][q] = B.pairRt[(size_t) pi * 12 + 9 + q];
        aff0[g] = B.pairs[pi].aff[0]; aff1[g] = B.pairs[pi].aff[1];
        st[g] = 0 /*IN*/; nst[g] = 2 /*OUTLIER*/; sten[g] = 0; nen[g] = 0;                      // FullSystem.cc:899-907
    }

    // one pass of linearizeResidual over all residuals at inverse depth `idp`; returns the sum of the return values (as the
    // reference accumulates it: float += double), Hdd / bd by reference
    auto pass = [&](float idp, float slack, float &Hdd, float &bd) -> float {
        float termH[NSG], termB[NSG];
        double ret[NSG];
#pragma unroll
        for (int g = 0; g < NSG; g++) {
            termH[g] = 0; termB[g] = 0; ret[g] = 0;
            const bool live = valid[g] && st[g] != 1;
            // projectPoint (ResidualProjections.h:57-84)
            const float Kl0 = (pu + ox - cx) * fxi, Kl1 = (pv + oy - cy) * fyi;
            const float q0 = (R[g][0] * Kl0 + R[g][1] * Kl1) + R[g][2] * 1.0f + t[g][0] * idp;
            const float q1 = (R[g][3] * Kl0 + R[g][4] * Kl1) + R[g][5] * 1.0f + t[g][1] * idp;
            const float q2 = (R[g][6] * Kl0 + R[g][7] * Kl1) + R[g][8] * 1.0f + t[g][2] * idp;
            const float drescale = 1.0f / q2;
            const float u = q0 * drescale, v = q1 * drescale;
            const float Ku = u * fx + cx, Kv = v * fy + cy;
            const bool okp = (drescale > 0) && Ku > 1.1f && Kv > 1.1f && Ku < D.wM3G && Kv < D.hM3G;
            float h0 = 0, h1 = 0, h2 = 0;
            if (live && okp) {
                const int ix = (int) Ku, iy = (int) Kv;
                const float dx = Ku - ix, dy = Kv - iy, dxdy = dx * dy;
                const float *bp = B.img[g * 8 + s] + 3 * (ix + iy * W);
                const float *bq = bp + 3 * W;
                const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
                h0 = ((w11 * bq[3] + w01 * bq[0]) + w10 * bp[3]) + w00 * bp[0];
                h1 = ((w11 * bq[4] + w01 * bq[1]) + w10 * bp[4]) + w00 * bp[1];
                h2 = ((w11 * bq[5] + w01 * bq[2]) + w10 * bp[5]) + w00 * bp[2];
            }
            const bool bad = !okp || !isfinite(h0);
            const int firstBad = min8(bad ? k : 8);                     // the reference returns at the first bad pattern pixel
            const float residual = h0 - (aff0[g] * col + aff1[g]);
            float hw = fabsf(residual) < S.huberTH ? 1 : S.huberTH / fabsf(residual);
            float eTerm = wk * wk * hw * residual * residual * (2 - hw);
            const float dxInterp = h1 * fx, dyInterp = h2 * fy;
            const float d_idepth = (dxInterp * drescale * (t[g][0] - t[g][2] * u) + dyInterp * drescale * (t[g][1] - t[g][2] * v)) * 1.0f;
            hw *= wk * wk;
            const bool contrib = live && k < firstBad;
            termH[g] = contrib ? (hw * d_idepth) * d_idepth : 0.0f;
            termB[g] = contrib ? (hw * residual) * d_idepth : 0.0f;
            if (!contrib) eTerm = 0.0f;
            // energy of the residual in pattern order (only meaningful when no pixel was bad)
            float e = act_dppf<0x117>(eTerm);
            e = e + act_dppf<0x116>(eTerm); e = e + act_dppf<0x115>(eTerm); e = e + act_dppf<0x114>(eTerm);
            e = e + act_dppf<0x113>(eTerm); e = e + act_dppf<0x112>(eTerm); e = e + act_dppf<0x111>(eTerm); e = e + eTerm;
            float energyLeft = __shfl(e, lane | 7, 64);
            if (!valid[g]) { ret[g] = 0; }
            else if (st[g] == 1) { nst[g] = 1; ret[g] = sten[g]; }                                       // :317-320
            else if (firstBad < 8) { nst[g] = 1; ret[g] = sten[g]; }                                   // OOB mid-pattern: partial Hdd / bd stay
            else {
                if (energyLeft > energyTH * slack) { energyLeft = energyTH * slack; nst[g] = 2; } else nst[g] = 0;
                nen[g] = (double) energyLeft;
                ret[g] = (double) energyLeft;
            }
        }
        // the running sums in residual order (targets ascending, host skipped), pattern pixels in order
        float E = 0;
        for (int tgt = 0; tgt < F; tgt++) {
            if (tgt == host) continue;
            const int g = tgt >> 3, l0 = (tgt & 7) * 8;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const float hH = (NSG == 1 || g == 0) ? rl(termH[0], l0 + kk) : rl(termH[NSG - 1], l0 + kk);
                const float hB = (NSG == 1 || g == 0) ? rl(termB[0], l0 + kk) : rl(termB[NSG - 1], l0 + kk);
                Hdd += hH; bd += hB;
            }
            const double r0 = (NSG == 1 || g == 0) ? ret[0] : ret[NSG - 1];
            unsigned long long ub = __builtin_bit_cast(unsigned long long, r0);
            const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (ub & 0xFFFFFFFFu), l0), hi = (unsigned) __builtin_amdgcn_readlane((int) (ub >> 32), l0);
            const double rr = __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
            E = (float) ((double) E + rr);                                                             // lastEnergy += linearizeResidual(...)
        }
        return E;
    };

    float lastHdd = 0, lastbd = 0;
    float currentIdepth = (P.idepth_max + P.idepth_min) * 0.5f;
    float lastEnergy = pass(currentIdepth, 1000.0f, lastHdd, lastbd);
#pragma unroll
    for (int g = 0; g < NSG; g++) { st[g] = nst[g]; sten[g] = nen[g]; }
    bool failed = !isfinite(lastEnergy) || lastHdd < A.minIdepthH_act;
    float lambda = 0.1f;
    int its = 0;
    for (int iteration = 0; !failed && iteration < A.GNIts; iteration++) {
        its++;
        float H = lastHdd;
        H *= 1 + lambda;
        const float step = (float) ((1.0 / (double) H) * (double) lastbd);
        const float newIdepth = currentIdepth - step;
        float newHdd = 0, newbd = 0;
        const float newEnergy = pass(newIdepth, 1.0f, newHdd, newbd);
        if (!isfinite(lastEnergy) || newHdd < A.minIdepthH_act) { failed = true; break; }
        if (newEnergy < lastEnergy) {
            currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
#pragma unroll
            for (int g = 0; g < NSG; g++) { st[g] = nst[g]; sten[g] = nen[g]; }
            lambda *= 0.5f;
        } else lambda *= 5;
        if ((double) fabsf(step) < 0.0001 * (double) currentIdepth) break;
    }
    // outputs
    int good = 0;
#pragma unroll
    for (int g = 0; g < NSG; g++) {
        const bool in = valid[g] && st[g] == 0 && k == 0;
        good += __popcll(__ballot(in));
        if (k == 0 && g * 8 + s < LDSO_MAX_FRAMES) O.res_state[g * 8 + s] = valid[g] ? st[g] : -1;
    }
    if (NSG == 1 && lane < 8) O.res_state[8 + lane] = -1;
    if (lane == 0) {
        O.idepth = currentIdepth; O.energy = lastEnergy; O.Hdd = lastHdd; O.bd = lastbd; O.iterations = its; O.numGoodRes = good; O.pad_ = 0;
        O.ok = (!failed && isfinite(currentIdepth) && good >= A.minObs) ? 1 : 0;
    }
}

hipError_t ba_launch_activate(const BaPtrs &B, const BaDims &D, const ldso_settings_t &S, const ldso_immature_t *d_pts, ldso_activation_t *d_out, int n, int minObs,
                              float minIdepthH_act, int GNIts, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    ActArgs A; A.pts = d_pts; A.out = d_out; A.n = n; A.minObs = minObs; A.GNIts = GNIts; A.minIdepthH_act = minIdepthH_act;
    const int blocks = (n + 3) / 4;
    if (D.nsg == 1) hipLaunchKernelGGL(k_activate<1>, dim3(blocks), dim3(256), 0, st, B, D, S, A);
    else hipLaunchKernelGGL(k_activate<2>, dim3(blocks), dim3(256), 0, st, B, D, S, A);
    return hipGetLastError();
}
