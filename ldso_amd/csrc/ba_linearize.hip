// ba_linearize.hip — fused per-residual / per-point kernel of the windowed photometric BA (gfx950, wave64).
//
// Replaces, for every active point of the window, the reference's
//   PointFrameResidual::linearize        src/internal/Residuals.cc:13-214
//   PointFrameResidual::applyRes/takeData include/internal/Residuals.h:70-87,123-128
//   AccumulatedTopHessianSSE::addPoint<0|1|2>  src/internal/OptimizationBackend/AccumulatedTopHessian.cc:8-118
//   AccumulatedSCHessianSSE::addPoint (per-point part)  .../AccumulatedSCHessian.cc:9-31
//
// Mapping (round 2): ONE LANE PER RESIDUAL.  A wavefront holds 64 / SL points (SL = slots per point = F rounded up to 8 or 16);
// lane = point_in_wave * SL + slot, slot = target frame.  Each lane walks the 8 pattern pixels of its residual sequentially - the
// reference's own loop, so energies / states / Jacobians keep the reference's operation order without any cross-lane traffic -
// and owns the full 13x13 relative Hessian block of ITS (host, target) pair in 91 registers: all residuals of a point share the
// host, points arrive host-major (EnergyFunctional::allPoints order), a lane's slot never changes, so the block is accumulated over
// all points the lane visits and reduced across the wave once, at the end (no atomics, one partial per slot and workgroup).
// The sums over the residuals of a point (Hdd, bd, Hcd, host block of the Schur row) are DPP sums over the SL lanes of the point.
// (Round 1 used one wavefront per point with lane = slot*8 + pixel: every geometric quantity was recomputed by 8 lanes and every
// sum over pixels was a DPP chain - 8x the instructions per residual, issue-bound at 12 % of the HBM roofline on large windows.)
//
// The Schur complement is kept in lifted form: per point the row g_p = [Ad^T JpJdF | Hcd] is stored (G matrix) and reduced later
// as G diag(HdiF) G^T by ba_reduce.hip.
//
// Roofline: HBM/gather bound - per residual 8 px x 4 taps x 12 B of the target image (436 B per residual + 112 B per point of
// algorithmic traffic).  Arithmetic is IEEE (compile with -ffp-contract=off): every elementwise expression follows the
// reference's operation order so that energies and residual states are bit-identical to the CPU path; fused multiply-adds are
// used only (explicitly) in the accumulators.
#include <hip/hip_runtime.h>
#include "ba_dev.h"

#define RES_IN 0
#define RES_OOB 1
#define RES_OUTLIER 2

template <int CTRL> __device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over the SL lanes of a point, result in all of them (tree order): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_ror:8
template <int SL> __device__ __forceinline__ float sum_point(float x) {
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    x += dpp_f<0x141>(x);
    if (SL == 16) x += dpp_f<0x128>(x);
    return x;
}
template <int SL> __device__ __forceinline__ float max_point(float x) {
    x = fmaxf(x, dpp_f<0xB1>(x));
    x = fmaxf(x, dpp_f<0x4E>(x));
    x = fmaxf(x, dpp_f<0x141>(x));
    if (SL == 16) x = fmaxf(x, dpp_f<0x128>(x));
    return x;
}
// sum over the points of a wave (lanes with equal slot), result valid in the lanes of point 0
template <int SL> __device__ __forceinline__ float sum_wave_points(float x) {
    if (SL == 8) x += dpp_f<0x128>(x);      // row_ror:8 : points (0,1), (2,3), ... share a 16-lane row
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// flat index of entry (r,c), r<=c, in the packed upper triangle of a 13x13 matrix
__host__ __device__ constexpr int tri13(int r, int c) { return r * 13 - (r * (r - 1)) / 2 + (c - r); }

// element i of a device array with the BYTE offset computed in 32 bits: base pointer (kernel argument, SGPR pair) + zero-extended
// lane offset is the addressing mode the hardware has (saddr + voffset).  Every table of a window is far below 4 GB.
#define AT(ptr, i) (*(decltype(ptr)) ((char *) (ptr) + (size_t) ((unsigned) (i) * (unsigned) sizeof(*(ptr)))))

struct __attribute__((aligned(4))) Tap2 { float v[6]; };      // two horizontally adjacent texels (I,dx,dy | I,dx,dy)

// AccumulatorApprox::update / updateTopRight / updateBotRight (MatrixAccumulators.h:893-1045) on a register-resident block:
// x,y = [Jpdc | Jpdxi] rows (10), (a,b,c) = JIdx2, top-right from JabJIdx / JI_r, bottom-right from Jab2 / Jab_r / rr.
__device__ __forceinline__ void acc13_update(float (&A)[LD_TOPN], const float (&x)[10], const float (&y)[10], float a, float b, float c,
                                             float TR00, float TR10, float TR01, float TR11, float TR02, float TR12,
                                             float BR00, float BR01, float BR02, float BR11, float BR12, float BR22) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const float ta = __builtin_fmaf(a, x[r], b * y[r]), tb = __builtin_fmaf(b, x[r], c * y[r]);
#pragma unroll
        for (int cc = r; cc < 10; cc++) A[tri13(r, cc)] = __builtin_fmaf(ta, x[cc], __builtin_fmaf(tb, y[cc], A[tri13(r, cc)]));
        A[tri13(r, 10)] = __builtin_fmaf(x[r], TR00, __builtin_fmaf(y[r], TR10, A[tri13(r, 10)]));
        A[tri13(r, 11)] = __builtin_fmaf(x[r], TR01, __builtin_fmaf(y[r], TR11, A[tri13(r, 11)]));
        A[tri13(r, 12)] = __builtin_fmaf(x[r], TR02, __builtin_fmaf(y[r], TR12, A[tri13(r, 12)]));
    }
    A[tri13(10, 10)] += BR00; A[tri13(10, 11)] += BR01; A[tri13(10, 12)] += BR02;
    A[tri13(11, 11)] += BR11; A[tri13(11, 12)] += BR12; A[tri13(12, 12)] += BR22;
}

// stepMode != 0 fuses the point part of resubstituteF_MT + backupState + doStepFromBackup (EnergyFunctional.cc:518-547,
// FullSystem.cc:1585-1602, 1625-1673) in front of the linearisation of each point (what k_point_step does with
// PS_RESUB|PS_BACKUP|PS_STEP), so a forced-accept GN iteration needs no separate point pass.
// MARG = true: the accumulate of EnergyFunctional::marginalizePointsF (EnergyFunctional.cc:165-222) for the points flagged in
// margFlags, fused with what FullSystem::flagPointsForRemoval did to them just before (FullSystem.cc:1241-1250): every residual
// of a flagged point is reset (resetOOB), re-linearised at the current state, applied, and - if active - fixed
// (fixLinearizationF, Residuals.cc:216-242: res_toZeroF = resF - [JIdx (Jp delta) + JabF delta_ab]); the top / Schur accumulators
// then take addPoint<2> (resApprox = res_toZeroF) with priorF * idepthFixPriorMargFac and no prior shift.  Unflagged points
// contribute nothing.  The output set is scratch (never applied).
template <int SL, int NW, bool HAS_L, bool FIX, bool MARG>
__global__ __launch_bounds__(64 * NW) void k_linearize(BaPtrs B, BaDims D, ResSet cur, ResSet nxt, ldso_settings_t S, int stepMode, GnInit gi,
                                                       const int32_t *__restrict__ margFlags) {
    if (LD_ITER_SKIPPED(B, gi.itCheck)) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 64 * NW, PPW = 64 / SL;
    const int F = D.F;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, s = lane & (SL - 1), pw = lane / SL;
    const int chunk = blockIdx.x;
    const long long t0_ = wall_clock64();
#define LSTAMP(i) do { if (LD_STAMP_ON && blockIdx.x == 0 && tid == 0) B.energyLog[8 + (i)] = (double) (wall_clock64() - t0_); } while (0)
    const int p0 = B.chunk_p0[chunk], np = B.chunk_n[chunk], h = B.chunk_host[chunk];

    // ---- LDS carve: pair structs of this host, float adjoints of this host, image bases, reduction scratch --------
    constexpr int PW = (int) (sizeof(DevPair) / 4);
    float *sPair = smem;                                      // [SL][PW]
    float *sAdH = sPair + SL * PW;                            // [SL][64]
    float *sAdT = sAdH + SL * 64;                             // [SL][64]
    float *sXa = sAdT + SL * 64;                              // [SL][8]  xAd of this host (stepMode)
    unsigned long long *sImg = (unsigned long long *) (sXa + SL * 8);      // [SL]
    float *sRed = (float *) (sImg + SL);                      // [NW][SL][91]
    float *sRedL = sRed + NW * SL * LD_TOPN;                  // [NW][SL][91] when HAS_L
    double *sE = (double *) (sRedL + (HAS_L ? NW * SL * LD_TOPN : 0));      // [NW]
    int *sC = (int *) (sE + NW);                              // [NW][4]
    float *sN = (float *) (sC + 4 * NW);                      // [NW]

    if (gi.enable) {
        // initialise the HFinal / bFinal accumulator of the k_reduce that follows (lower triangle) with H_M and the diagonal
        // priors (EnergyFunctional.cc:257-291; the lambda scaling of these terms is added by k_reduce); b starts at zero.
        const int n = D.n, N1 = n * n + n, per = (N1 + (int) gridDim.x - 1) / (int) gridDim.x;
        const int z0 = blockIdx.x * per, z1 = min(N1, z0 + per);
        for (int e = z0 + tid; e < z1; e += NT) {
            double v = 0.0;
            if (e < n * n) {
                const int i = e / n, j = e % n;
                if (j <= i && gi.enable == 1) {      // enable == 2 (ranks > 0 of a sharded window): zeros only
                    if (gi.hasPrior) v = B.HM[e];
                    if (i == j) { v += (i < 4) ? (double) gi.calibPrior : B.frames[(i - 4) >> 3].prior[(i - 4) & 7]; }
                }
            }
            B.acc[e] = v;
        }
    }
    const float fx = B.calib->sf[0], fy = B.calib->sf[1], cx = B.calib->sf[2], cy = B.calib->sf[3];
    const float fxi = B.calib->si[0], fyi = B.calib->si[1];
    const float cD0 = B.calib->cDeltaF[0], cD1 = B.calib->cDeltaF[1], cD2 = B.calib->cDeltaF[2], cD3 = B.calib->cDeltaF[3];
    float xc0 = 0, xc1 = 0, xc2 = 0, xc3 = 0;
    if (stepMode & 1) { xc0 = B.xc[0]; xc1 = B.xc[1]; xc2 = B.xc[2]; xc3 = B.xc[3]; }
    // the energy threshold of a pair is max(host, target) of the frames' CURRENT thresholds (Residuals.cc:191)
    const float thMax = fmaxf(B.frames[h].frameEnergyTH, B.frames[min(s, F - 1)].frameEnergyTH);

    // ---- staging: all global loads first (one latency level, 16-byte loads), then the LDS stores -------------------
    {
        // items (float4): pairs of this host F*9 (contiguous), then per target 16 of adHostF and 16 of adTargetF, then xAd F*2
        constexpr int P4 = PW / 4, U = (SL * (P4 + 32 + 2) + NT - 1) / NT;
        const int nP = F * P4, nA = F * 16, nX = (stepMode & 1) ? F * 2 : 0, total = nP + 2 * nA + nX;
        const float4 *gP = (const float4 *) (B.pairs + (size_t) h * F), *gX = (const float4 *) (B.xAd + (size_t) h * F * 8);
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = tid + u * NT;
            float4 q = make_float4(0, 0, 0, 0);
            if (i < nP) q = gP[i];
            else if (i < nP + nA) { const int j = i - nP; q = ((const float4 *) (B.adHostF + (size_t) (h + (j >> 4) * F) * 64))[j & 15]; }
            else if (i < nP + 2 * nA) { const int j = i - nP - nA; q = ((const float4 *) (B.adTargetF + (size_t) (h + (j >> 4) * F) * 64))[j & 15]; }
            else if (i < total) q = gX[i - nP - 2 * nA];
            v[u] = q;
        }
        // unused slots (t >= F) read as zeros
        for (int i = tid; i < (SL - F) * PW; i += NT) sPair[F * PW + i] = 0.0f;
        for (int i = tid; i < (SL - F) * 64; i += NT) { sAdH[F * 64 + i] = 0.0f; sAdT[F * 64 + i] = 0.0f; }
        for (int i = tid; i < SL * 8; i += NT) if (i >= F * 8 || !(stepMode & 1)) sXa[i] = 0.0f;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = tid + u * NT;
            if (i < nP) ((float4 *) sPair)[i] = v[u];
            else if (i < nP + nA) ((float4 *) sAdH)[i - nP] = v[u];
            else if (i < nP + 2 * nA) ((float4 *) sAdT)[i - nP - nA] = v[u];
            else if (i < total) ((float4 *) sXa)[i - nP - 2 * nA] = v[u];
        }
#pragma unroll
        for (int t = 0; t < SL; t++) if (tid == t) sImg[t] = (unsigned long long) B.img[t < LD_MAXF ? t : 0];
    }
    __syncthreads();
    LSTAMP(1);

    // ---- this lane's pair (fixed for the whole kernel) --------------------------------------------------------------
    const float *pr = sPair + s * PW;
    const float KRKi0 = pr[0], KRKi1 = pr[1], KRKi2 = pr[2], KRKi3 = pr[3], KRKi4 = pr[4], KRKi5 = pr[5], KRKi6 = pr[6], KRKi7 = pr[7], KRKi8 = pr[8];
    const float Kt0 = pr[9], Kt1 = pr[10], Kt2 = pr[11];
    const float R00 = pr[12], R01 = pr[13], R02 = pr[14], R03 = pr[15], R04 = pr[16], R05 = pr[17], R06 = pr[18], R07 = pr[19], R08 = pr[20];
    const float t00 = pr[21], t01 = pr[22], t02 = pr[23];
    const float aff0 = pr[24], aff1 = pr[25], b0 = pr[26];
    const float *dp = pr + 28;                                 // adHTdeltaF[host + target*F] (modes 1 and 2 only)
    const char *img = (const char *) sImg[s];
    const unsigned rowB = (unsigned) D.w * 12u;

    float accA[LD_TOPN];
#pragma unroll
    for (int i = 0; i < LD_TOPN; i++) accA[i] = 0.0f;
    float accL[LD_TOPN];      // linearised residuals (H_L); dead code unless HAS_L
#pragma unroll
    for (int i = 0; i < LD_TOPN; i++) accL[i] = 0.0f;
    double energySum = 0.0;     // sum of linearize() return values
    int nresA = 0, nresL = 0;
    float nidSum = 0.0f;
    int nidCnt = 0;

    for (int grp = wave; grp * PPW < np; grp += NW) {
        if (grp == wave) LSTAMP(2);
        const int pl = grp * PPW + pw;
        const bool pvalid = pl < np;
        const unsigned p = (unsigned) (p0 + (pvalid ? pl : 0));
        const unsigned slot = p * (unsigned) SL + (unsigned) s;
        // ---- everything this lane reads besides the image taps (one latency level) ----------------------------------
        const float pu = AT(B.pu, p), pv = AT(B.pv, p), priorF0 = AT(B.ppriorF, p);
        float idp = AT(B.pidepth, p), idz = AT(B.pidepth_zero, p);
        const float4 col0 = ((const float4 *) B.pcolor)[p * 2], col1 = ((const float4 *) B.pcolor)[p * 2 + 1];
        const float4 wg0 = ((const float4 *) B.pweights)[p * 2], wg1 = ((const float4 *) B.pweights)[p * 2 + 1];
        const int rflat = AT(B.rflat, slot), rlin = AT(B.rlin, slot), rnew = FIX ? AT(B.rnew, slot) : 0, rlidx = HAS_L ? AT(B.rlidx, slot) : 0;
        const int oldState = AT(cur.state, slot), oldActive = AT(cur.active, slot);
        const float oldEnergy = AT(cur.energy, slot);
        const float4 jpo0 = ((const float4 *) cur.JpJdF)[slot * 2], jpo1 = ((const float4 *) cur.JpJdF)[slot * 2 + 1];
        const float cen0 = AT(cur.center, slot * 3), cen1 = AT(cur.center, slot * 3 + 1), cen2 = AT(cur.center, slot * 3 + 2);
        float maxRelBS = AT(cur.maxRelBS, p);
        int numGood = AT(cur.numGood, p);
        const bool flagged = MARG ? (margFlags[p] != 0) : true;
        const float priorF = MARG ? priorF0 * S.idepthFixPriorMargFac : priorF0;
        const float color[8] = {col0.x, col0.y, col0.z, col0.w, col1.x, col1.y, col1.z, col1.w};
        const float wgt[8] = {wg0.x, wg0.y, wg0.z, wg0.w, wg1.x, wg1.y, wg1.z, wg1.w};
        float jp[8] = {jpo0.x, jpo0.y, jpo0.z, jpo0.w, jpo1.x, jpo1.y, jpo1.z, jpo1.w};      // JpJdF of this residual (applied set -> new set)

        const bool exists = pvalid && (s < F) && (rflat >= 0) && flagged;
        if (stepMode & 1) {
            // ---- resubstituteFPt for this point, then backupState + doStepFromBackup (stepfacD = 1) ------------
            const float pstep = AT(B.pstep, p), bdS = AT(cur.bdSumF, p), HdiFo = AT(cur.HdiF, p);
            const int nAct = AT(cur.nActive, p);
            const float4 hA = ((const float4 *) cur.HcdA)[p], hL = ((const float4 *) cur.HcdL)[p];
            float step = 0.0f;
            float sres = sXa[s * 8] * jp[0];
#pragma unroll
            for (int k = 1; k < 8; k++) sres = sres + sXa[s * 8 + k] * jp[k];
            sres = (exists && oldActive != 0) ? sres : 0.0f;
            const float ssum = sum_point<SL>(sres);
            if (nAct > 0) {
                float b = bdS;
                float dot = 0;
                dot += xc0 * (hA.x + hL.x); dot += xc1 * (hA.y + hL.y); dot += xc2 * (hA.z + hL.z); dot += xc3 * (hA.w + hL.w);
                b -= dot;
                b -= ssum;
                if (isfinite(b)) step = -b * HdiFo; else { step = pstep; if (s == 0 && pvalid) B.scalars[4] = 1.0; }
            }
            const float ni = idp + 1.0f * step;
            if (s == 0 && pvalid) { AT(B.pstep, p) = step; AT(B.pidepth_backup, p) = idp; AT(B.pidepth, p) = ni; AT(B.pidepth_zero, p) = ni; }
            idp = ni; idz = ni;
        }
        const float deltaF = idp - idz;

        const int t = s;
        const bool isLin = MARG ? false : (exists && (rlin != 0));
        // resetOOB (Residuals.h): MARG always; stepMode bit 1 = the optimize() preamble on every non-linearised residual (FullSystem.cc:744-748)
        const bool reset = MARG || ((stepMode & 2) && !isLin);
        const int st = exists ? (reset ? RES_IN : oldState) : RES_OOB;

        int newState = st;
        float newEnergy = (exists && !reset) ? oldEnergy : 0.0f;
        float newEnergyWO = -1.0f;
        int activeNew = exists ? oldActive : 0;
        if (!exists) {
#pragma unroll
            for (int k = 0; k < 8; k++) jp[k] = 0.0f;
        }
        float c0 = 0, c1 = 0, c2 = 0;
        int toRemove = 0;
        double ret = 0.0;                                      // linearize() return value
        const bool doLin = exists && !isLin;

        // ================= active-set residual: PointFrameResidual::linearize ==================
        if (doLin && st == RES_OOB) { newState = RES_OOB; ret = (double) newEnergy; if (FIX) toRemove = 1; }
        bool compute = doLin && st != RES_OOB;
        // ---- centre projection at the linearisation point (ResidualProjections.h:57-84) --------
        const float KliP0 = (pu + 0 - cx) * fxi, KliP1 = (pv + 0 - cy) * fyi;
        const float ptp0 = ((R00 * KliP0 + R01 * KliP1) + R02 * 1.0f) + t00 * idz;
        const float ptp1 = ((R03 * KliP0 + R04 * KliP1) + R05 * 1.0f) + t01 * idz;
        const float ptp2 = ((R06 * KliP0 + R07 * KliP1) + R08 * 1.0f) + t02 * idz;
        const float drescale = 1.0f / ptp2;
        const float new_idepth = idz * drescale;
        const float uu = ptp0 * drescale, vv = ptp1 * drescale;
        const float cKu = uu * fx + cx, cKv = vv * fy + cy;
        const bool centerOK = (drescale > 0) && cKu > 1.1f && cKv > 1.1f && cKu < D.wM3G && cKv < D.hM3G;
        if (compute && !centerOK) { newState = RES_OOB; ret = (double) newEnergy; compute = false; }

        // ---- geometric Jacobians at the linearisation point (Residuals.cc:67-104) ----------------
        const float Jpdd0 = drescale * (t00 - t02 * uu) * 1.0f * fx;
        const float Jpdd1 = drescale * (t01 - t02 * vv) * 1.0f * fy;
        float x[10], y[10];
        {
            const float dCx2 = drescale * (R06 * uu - R00);
            const float dCx3 = fx * drescale * (R07 * uu - R01) * fyi;
            const float dCx0 = KliP0 * dCx2, dCx1 = KliP1 * dCx3;
            const float dCy2 = fy * drescale * (R06 * vv - R03) * fxi;
            const float dCy3 = drescale * (R07 * vv - R04);
            const float dCy0 = KliP0 * dCy2, dCy1 = KliP1 * dCy3;
            x[0] = (dCx0 + uu) * 50.0f; x[1] = dCx1 * 50.0f; x[2] = (dCx2 + 1) * 50.0f; x[3] = dCx3 * 50.0f;
            y[0] = dCy0 * 50.0f; y[1] = (dCy1 + vv) * 50.0f; y[2] = dCy2 * 50.0f; y[3] = (dCy3 + 1) * 50.0f;
            x[4] = new_idepth * fx; x[5] = 0; x[6] = -new_idepth * uu * fx; x[7] = -uu * vv * fx; x[8] = (1 + uu * uu) * fx; x[9] = -vv * fx;
            y[4] = 0; y[5] = new_idepth * fy; y[6] = -new_idepth * vv * fy; y[7] = -(1 + vv * vv) * fy; y[8] = uu * vv * fy; y[9] = uu * fy;
        }
        // fixLinearizationF's J delta (MARG): Jp_delta = Jpdxi * adHTdelta[0:6] + Jpdc * cDelta + Jpdd * deltaF
        float Jp_delta_x = 0, Jp_delta_y = 0, dp6 = 0, dp7 = 0;
        if (MARG) {
            float dpx = 0, dpy = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) { dpx += x[4 + i] * dp[i]; dpy += y[4 + i] * dp[i]; }
            const float dcx = ((x[0] * cD0 + x[1] * cD1) + x[2] * cD2) + x[3] * cD3;
            const float dcy = ((y[0] * cD0 + y[1] * cD1) + y[2] * cD2) + y[3] * cD3;
            Jp_delta_x = dpx + dcx + Jpdd0 * deltaF; Jp_delta_y = dpy + dcy + Jpdd1 * deltaF;
            dp6 = dp[6]; dp7 = dp[7];
        }

        // ---- the 8 pattern pixels (Residuals.cc:126-188), in the reference's order -------------------------------------
        // pass 1: projections (ResidualProjections.h:24-33) and tap addresses; pass 2: all taps in flight; pass 3: photometric terms
        unsigned off[8];
        float fdx[8], fdy[8];
        bool bad = false;
        const float z10 = (S.affineOptModeA < 0) ? 0.0f : 1.0f, z11 = (S.affineOptModeB < 0) ? 0.0f : 1.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            constexpr int OX[8] = {0, -1, 1, -2, 0, 2, -1, 0}, OY[8] = {-2, -1, -1, 0, 0, 0, 1, 2};      // staticPattern[8], Setting.cc:221
            const float px_ = pu + (float) OX[k], py_ = pv + (float) OY[k];
            const float q0 = ((KRKi0 * px_ + KRKi1 * py_) + KRKi2 * 1.0f) + Kt0 * idp;
            const float q1 = ((KRKi3 * px_ + KRKi4 * py_) + KRKi5 * 1.0f) + Kt1 * idp;
            const float q2 = ((KRKi6 * px_ + KRKi7 * py_) + KRKi8 * 1.0f) + Kt2 * idp;
            const float Ku = q0 / q2, Kv = q1 / q2;
            const bool pixOK = Ku > 1.1f && Kv > 1.1f && Ku < D.wM3G && Kv < D.hM3G;
            bad = bad || !pixOK;
            const int ix = pixOK ? (int) Ku : 0, iy = pixOK ? (int) Kv : 0;
            fdx[k] = Ku - (float) ix; fdy[k] = Kv - (float) iy;
            off[k] = (unsigned) (ix + iy * D.w) * 12u;
        }
        if (grp == wave) LSTAMP(3);
        const bool sample = compute && !bad;
        Tap2 ta[8], tb[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (sample) { ta[k] = *(const Tap2 *) (img + off[k]); tb[k] = *(const Tap2 *) (img + off[k] + rowB); }
            else {
#pragma unroll
                for (int i = 0; i < 6; i++) { ta[k].v[i] = 0.0f; tb[k].v[i] = 0.0f; }
            }
        }
        if (grp == wave) LSTAMP(4);
        float energyLeft = 0, wJI2_sum = 0;
        float JI00 = 0, JI11 = 0, JI10 = 0, JabJI00 = 0, JabJI01 = 0, JabJI10 = 0, JabJI11 = 0, Jab00 = 0, Jab01 = 0, Jab11 = 0;
        float JI_r0 = 0, JI_r1 = 0, Jab_r0 = 0, Jab_r1 = 0, rr = 0;
        const bool dump = (B.dumpJ != nullptr) && sample;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // bilinear Vec3f sample of the target image (GlobalFuncs.h:89-103)
            const float dx = fdx[k], dy = fdy[k], dxdy = dx * dy;
            const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
            const float hit0 = ((w11 * tb[k].v[3] + w01 * tb[k].v[0]) + w10 * ta[k].v[3]) + w00 * ta[k].v[0];
            const float hit1 = ((w11 * tb[k].v[4] + w01 * tb[k].v[1]) + w10 * ta[k].v[4]) + w00 * ta[k].v[1];
            const float hit2 = ((w11 * tb[k].v[5] + w01 * tb[k].v[2]) + w10 * ta[k].v[5]) + w00 * ta[k].v[2];
            bad = bad || !isfinite(hit0);
            const float residual = hit0 - (float) (aff0 * color[k] + aff1);
            const float drdA = (color[k] - b0);
            float w_ = sqrtf(S.outlierTHSumComponent / (S.outlierTHSumComponent + (hit1 * hit1 + hit2 * hit2)));
            w_ = 0.5f * (w_ + wgt[k]);
            float hw = fabsf(residual) < S.huberTH ? 1 : S.huberTH / fabsf(residual);
            energyLeft += w_ * w_ * hw * residual * residual * (2 - hw);
            if (hw < 1) hw = sqrtf(hw);
            hw = hw * w_;
            const float gx = hit1 * hw, gy = hit2 * hw;
            const float resF = residual * hw;
            float jab0 = drdA * hw, jab1 = hw;
            JI00 += gx * gx; JI11 += gy * gy; JI10 += gx * gy;
            JabJI00 += drdA * hw * gx; JabJI01 += drdA * hw * gy; JabJI10 += hw * gx; JabJI11 += hw * gy;
            Jab00 += drdA * drdA * hw * hw; Jab01 += drdA * hw * hw; Jab11 += hw * hw;
            wJI2_sum += hw * hw * (gx * gx + gy * gy);
            if (S.affineOptModeA < 0) jab0 = 0;     // J->JabF zeroed AFTER the 2x2 sums (Residuals.cc:184-185)
            if (S.affineOptModeB < 0) jab1 = 0;
            // the residual column of the accumulators: resF (mode 0) or res_toZeroF (MARG, mode 2; Residuals.cc:216-242)
            float resAcc = resF;
            if (MARG) { float rtz = resF; rtz = rtz - gx * Jp_delta_x; rtz = rtz - gy * Jp_delta_y; rtz = rtz - jab0 * dp6; rtz = rtz - jab1 * dp7; resAcc = rtz; }
            JI_r0 += resAcc * gx; JI_r1 += resAcc * gy;
            Jab_r0 += drdA * hw * resAcc; Jab_r1 += hw * resAcc; rr += resAcc * resAcc;
            if (dump) {      // (fields of a residual that turns out-of-bounds at a later pixel are left undefined)
                ldso_rawjac_t &o = B.dumpJ[rflat];
                o.resF[k] = resF; o.JIdx[0][k] = gx; o.JIdx[1][k] = gy; o.JabF[0][k] = jab0; o.JabF[1][k] = jab1;
            }
        }
        Jab_r0 *= z10; Jab_r1 *= z11;      // Jab_r uses the (possibly zeroed) JabF, Jab2 / JabJIdx the un-zeroed sums
        if (compute && bad) { newState = RES_OOB; ret = (double) newEnergy; compute = false; }

        if (compute) {
            newEnergyWO = energyLeft;
            if (energyLeft > thMax || wJI2_sum < 2) { energyLeft = thMax; newState = RES_OUTLIER; }
            else newState = RES_IN;
            newEnergy = energyLeft;
            ret = (double) energyLeft;
            c0 = cKu; c1 = cKv; c2 = new_idepth;
        }

        // ================= applyRes(true) (Residuals.h:70-87) ======================================
        if (doLin && st != RES_OOB) {
            if (newState == RES_IN) {
                activeNew = 1;
                // takeData (Residuals.h:123-128)
                const float v0 = JI00 * Jpdd0 + JI10 * Jpdd1, v1 = JI10 * Jpdd0 + JI11 * Jpdd1;
#pragma unroll
                for (int i = 0; i < 6; i++) jp[i] = x[4 + i] * v0 + y[4 + i] * v1;
                jp[6] = JabJI00 * Jpdd0 + JabJI01 * Jpdd1; jp[7] = JabJI10 * Jpdd0 + JabJI11 * Jpdd1;
            } else {
                activeNew = 0;
            }
            if (FIX) {
                if (activeNew) {
                    if (rnew) {
                        // FullSystem.cc:1518-1534: relative baseline of new residuals
                        const float inf0 = (KRKi0 * pu + KRKi1 * pv) + KRKi2 * 1.0f;
                        const float inf1 = (KRKi3 * pu + KRKi4 * pv) + KRKi5 * 1.0f;
                        const float inf2 = (KRKi6 * pu + KRKi7 * pv) + KRKi8 * 1.0f;
                        const float r0 = inf0 + Kt0 * idp, r1 = inf1 + Kt1 * idp, r2 = inf2 + Kt2 * idp;
                        const float ax = inf0 / inf2 - r0 / r2, ay = inf1 / inf2 - r1 / r2;
                        const float relBS = (float) (0.01 * (double) sqrtf(ax * ax + ay * ay));
                        if (relBS > maxRelBS) maxRelBS = relBS;     // merged across the point's residuals below
                    }
                } else toRemove = 1;
            }
        }
        const unsigned ptShift = (unsigned) (lane & ~(SL - 1));
        const unsigned long long ptMask = (SL == 8) ? 0xFFull : 0xFFFFull;
        if (FIX) {
            const unsigned long long m = __ballot(doLin && st != RES_OOB && activeNew && rnew);
            numGood += __popcll((m >> ptShift) & ptMask);
            maxRelBS = max_point<SL>(maxRelBS);
        }

        // ================= accumulate: active residual, mode 0 / 2 (AccumulatedTopHessian.cc) ==========
        const bool accHere = doLin && activeNew && compute;
        if (accHere) {
            acc13_update(accA, x, y, JI00, JI10, JI11, JabJI00, JabJI01, JabJI10, JabJI11, JI_r0, JI_r1, Jab00, Jab01, Jab_r0, Jab11, Jab_r1, rr);
            nresA++;
        }
        // contributions of this residual to the point sums
        const float Ji2_0 = JI00 * Jpdd0 + JI10 * Jpdd1, Ji2_1 = JI10 * Jpdd0 + JI11 * Jpdd1;
        const float sbd = accHere ? (JI_r0 * Jpdd0 + JI_r1 * Jpdd1) : 0.0f;
        const float sHdd = accHere ? (Ji2_0 * Jpdd0 + Ji2_1 * Jpdd1) : 0.0f;
        const float sHc0 = accHere ? (x[0] * Ji2_0 + y[0] * Ji2_1) : 0.0f, sHc1 = accHere ? (x[1] * Ji2_0 + y[1] * Ji2_1) : 0.0f;
        const float sHc2 = accHere ? (x[2] * Ji2_0 + y[2] * Ji2_1) : 0.0f, sHc3 = accHere ? (x[3] * Ji2_0 + y[3] * Ji2_1) : 0.0f;
        const float bdA = sum_point<SL>(sbd), HddA = sum_point<SL>(sHdd);
        const float HcdA0 = sum_point<SL>(sHc0), HcdA1 = sum_point<SL>(sHc1), HcdA2 = sum_point<SL>(sHc2), HcdA3 = sum_point<SL>(sHc3);
        if (B.dumpJ != nullptr && compute) {
            ldso_rawjac_t &o = B.dumpJ[rflat];
#pragma unroll
            for (int i = 0; i < 6; i++) { o.Jpdxi[0][i] = x[4 + i]; o.Jpdxi[1][i] = y[4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; i++) { o.Jpdc[0][i] = x[i]; o.Jpdc[1][i] = y[i]; }
            o.Jpdd[0] = Jpdd0; o.Jpdd[1] = Jpdd1;
            o.JIdx2[0] = JI00; o.JIdx2[1] = JI10; o.JIdx2[2] = JI10; o.JIdx2[3] = JI11;
            o.JabJIdx[0] = JabJI00; o.JabJIdx[1] = JabJI01; o.JabJIdx[2] = JabJI10; o.JabJIdx[3] = JabJI11;
            o.Jab2[0] = Jab00; o.Jab2[1] = Jab01; o.Jab2[2] = Jab01; o.Jab2[3] = Jab11;
        }

        // ================= linearised residual, mode 1 (AccumulatedTopHessian.cc:29-31,44-63) =======
        float HddL = 0, bdL = 0, HcdL0 = 0, HcdL1 = 0, HcdL2 = 0, HcdL3 = 0;
        if constexpr (HAS_L) {
            const bool accLin = isLin && activeNew;
            float lsbd = 0, lsHdd = 0, lH0 = 0, lH1 = 0, lH2 = 0, lH3 = 0;
            if (accLin) {
                const ldso_rawjac_t &J = B.Jlin[(unsigned) rlidx];
                const float *rtz = B.rtz + (unsigned) rlidx * 8u;
                float lx[10], ly[10];
#pragma unroll
                for (int i = 0; i < 4; i++) { lx[i] = J.Jpdc[0][i]; ly[i] = J.Jpdc[1][i]; }
#pragma unroll
                for (int i = 0; i < 6; i++) { lx[4 + i] = J.Jpdxi[0][i]; ly[4 + i] = J.Jpdxi[1][i]; }
                float dpx = 0, dpy = 0;
#pragma unroll
                for (int i = 0; i < 6; i++) { dpx += lx[4 + i] * dp[i]; dpy += ly[4 + i] * dp[i]; }
                const float dcx = ((lx[0] * cD0 + lx[1] * cD1) + lx[2] * cD2) + lx[3] * cD3;
                const float dcy = ((ly[0] * cD0 + ly[1] * cD1) + ly[2] * cD2) + ly[3] * cD3;
                const float Jdx = dpx + dcx + J.Jpdd[0] * deltaF, Jdy = dpy + dcy + J.Jpdd[1] * deltaF;
                float lJI_r0 = 0, lJI_r1 = 0, lJab_r0 = 0, lJab_r1 = 0, lrr = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const float lgx = J.JIdx[0][k], lgy = J.JIdx[1][k], la0 = J.JabF[0][k], la1 = J.JabF[1][k];
                    float ra = rtz[k];
                    ra = ra + lgx * Jdx; ra = ra + lgy * Jdy; ra = ra + la0 * dp[6]; ra = ra + la1 * dp[7];
                    lJI_r0 += ra * lgx; lJI_r1 += ra * lgy; lJab_r0 += ra * la0; lJab_r1 += ra * la1; lrr += ra * ra;
                }
                const float a = J.JIdx2[0], b = J.JIdx2[1], c = J.JIdx2[3];
                acc13_update(accL, lx, ly, a, b, c, J.JabJIdx[0], J.JabJIdx[1], J.JabJIdx[2], J.JabJIdx[3], lJI_r0, lJI_r1,
                             J.Jab2[0], J.Jab2[1], lJab_r0, J.Jab2[3], lJab_r1, lrr);
                nresL++;
                const float lJi0 = a * J.Jpdd[0] + b * J.Jpdd[1], lJi1 = b * J.Jpdd[0] + c * J.Jpdd[1];
                lsbd = lJI_r0 * J.Jpdd[0] + lJI_r1 * J.Jpdd[1];
                lsHdd = lJi0 * J.Jpdd[0] + lJi1 * J.Jpdd[1];
                lH0 = lx[0] * lJi0 + ly[0] * lJi1; lH1 = lx[1] * lJi0 + ly[1] * lJi1; lH2 = lx[2] * lJi0 + ly[2] * lJi1; lH3 = lx[3] * lJi0 + ly[3] * lJi1;
            }
            bdL = sum_point<SL>(lsbd); HddL = sum_point<SL>(lsHdd);
            HcdL0 = sum_point<SL>(lH0); HcdL1 = sum_point<SL>(lH1); HcdL2 = sum_point<SL>(lH2); HcdL3 = sum_point<SL>(lH3);
        }

        // ================= lifted Schur row: target block of this residual and its share of the host block ==
        const bool lift = exists && activeNew;
        float tgt[8], hostPart[8];
        {
            const float4 *aT = (const float4 *) (sAdT + t * 64), *aH = (const float4 *) (sAdH + t * 64);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float4 t0 = aT[r * 2], t1 = aT[r * 2 + 1], h0 = aH[r * 2], h1 = aH[r * 2 + 1];
                float a = 0.0f, b = 0.0f;
                a = __builtin_fmaf(t0.x, jp[0], a); a = __builtin_fmaf(t0.y, jp[1], a); a = __builtin_fmaf(t0.z, jp[2], a); a = __builtin_fmaf(t0.w, jp[3], a);
                a = __builtin_fmaf(t1.x, jp[4], a); a = __builtin_fmaf(t1.y, jp[5], a); a = __builtin_fmaf(t1.z, jp[6], a); a = __builtin_fmaf(t1.w, jp[7], a);
                b = __builtin_fmaf(h0.x, jp[0], b); b = __builtin_fmaf(h0.y, jp[1], b); b = __builtin_fmaf(h0.z, jp[2], b); b = __builtin_fmaf(h0.w, jp[3], b);
                b = __builtin_fmaf(h1.x, jp[4], b); b = __builtin_fmaf(h1.y, jp[5], b); b = __builtin_fmaf(h1.z, jp[6], b); b = __builtin_fmaf(h1.w, jp[7], b);
                tgt[r] = lift ? a : 0.0f;
                hostPart[r] = sum_point<SL>(lift ? b : 0.0f);
            }
        }
        const int nActive = __popcll((__ballot(lift) >> ptShift) & ptMask);

        // ---- per-residual outputs --------------------------------------------------------------------------------------
        if (pvalid && t < F) {
            ((float4 *) nxt.JpJdF)[slot * 2] = make_float4(jp[0], jp[1], jp[2], jp[3]);
            ((float4 *) nxt.JpJdF)[slot * 2 + 1] = make_float4(jp[4], jp[5], jp[6], jp[7]);
            AT(nxt.state, slot) = newState;
            AT(nxt.active, slot) = activeNew;
            AT(nxt.energy, slot) = newEnergy;
            AT(nxt.newEnergyWO, slot) = doLin ? newEnergyWO : -1.0f;
            if (t == F - 1) AT(nxt.candE, p) = doLin ? newEnergyWO : -1.0f;
            AT(nxt.toRemove, slot) = toRemove;
            if (doLin) energySum += ret;
            AT(nxt.center, slot * 3) = compute ? c0 : cen0; AT(nxt.center, slot * 3 + 1) = compute ? c1 : cen1; AT(nxt.center, slot * 3 + 2) = compute ? c2 : cen2;
        }

        if (grp == wave) LSTAMP(5);
        // ================= per-point Schur quantities (AccumulatedSCHessian.cc:9-31) =====================
        float HdiF = 0, bdSumF = 0, idH = 0;
        const float Hc0 = HcdA0 + HcdL0, Hc1 = HcdA1 + HcdL1, Hc2 = HcdA2 + HcdL2, Hc3 = HcdA3 + HcdL3;
        if (nActive > 0) {
            float Hh = HddA + HddL + priorF;
            if (Hh < 1e-10) Hh = 1e-10;
            idH = Hh;
            HdiF = (float) (1.0 / (double) Hh);
            bdSumF = bdA + bdL;
            if (!MARG) bdSumF += priorF * deltaF;       // shiftPriorToZero = true in accumulateSCF_MT, false in marginalizePointsF
        } else {
            maxRelBS = 0;
        }
        if (pvalid) {
            // ---- store G row: [8*FS frame entries | Hcd 4, bdSum, HdiF, 0, 0] --------------------------------
            float *Grow = nxt.G + (size_t) p * D.GS;
            const bool z = nActive == 0;
            const bool isHost = (t == h);
            float g8[8];
#pragma unroll
            for (int r = 0; r < 8; r++) g8[r] = z ? 0.0f : (isHost ? hostPart[r] : tgt[r]);
            ((float4 *) (Grow + 8 * t))[0] = make_float4(g8[0], g8[1], g8[2], g8[3]);
            ((float4 *) (Grow + 8 * t))[1] = make_float4(g8[4], g8[5], g8[6], g8[7]);
            if (s == 0) {
                ((float4 *) (Grow + 8 * SL))[0] = z ? make_float4(0, 0, 0, 0) : make_float4(Hc0, Hc1, Hc2, Hc3);
                ((float4 *) (Grow + 8 * SL))[1] = make_float4(bdSumF, HdiF, 0.0f, 0.0f);
                AT(nxt.HdiF, p) = HdiF; AT(nxt.bdSumF, p) = bdSumF; AT(nxt.idH, p) = idH;
                AT(nxt.HddA, p) = HddA; AT(nxt.bdA, p) = bdA; AT(nxt.HddL, p) = HddL; AT(nxt.bdL, p) = bdL;
                ((float4 *) nxt.HcdA)[p] = make_float4(HcdA0, HcdA1, HcdA2, HcdA3);
                ((float4 *) nxt.HcdL)[p] = make_float4(HcdL0, HcdL1, HcdL2, HcdL3);
                AT(nxt.maxRelBS, p) = maxRelBS; AT(nxt.numGood, p) = numGood; AT(nxt.nActive, p) = nActive;
                nidSum += fabsf(idp); nidCnt++;
            }
        }
    }   // point groups of this wave

    LSTAMP(6);
    // ================= reduction of the top accumulators: points of the wave (DPP / permute), then the waves of the block (LDS) ===
    {
        float *cell = sRed + (size_t) (wave * SL + s) * LD_TOPN;
#pragma unroll
        for (int i = 0; i < LD_TOPN; i++) { const float v = sum_wave_points<SL>(accA[i]); if (pw == 0) cell[i] = v; }
        if constexpr (HAS_L) {
            float *cellL = sRedL + (size_t) (wave * SL + s) * LD_TOPN;
#pragma unroll
            for (int i = 0; i < LD_TOPN; i++) { const float v = sum_wave_points<SL>(accL[i]); if (pw == 0) cellL[i] = v; }
        }
    }
    {
        double e = energySum;
        int na = nresA, nl = nresL, nc = nidCnt;
        float ns = nidSum;
        for (int o = 32; o > 0; o >>= 1) {
            const double e2 = __shfl_xor(e, o, 64); const int a2 = __shfl_xor(na, o, 64), l2 = __shfl_xor(nl, o, 64), c2_ = __shfl_xor(nc, o, 64); const float s2 = __shfl_xor(ns, o, 64);
            e += e2; na += a2; nl += l2; nc += c2_; ns += s2;
        }
        if (lane == 0) { sE[wave] = e; sC[wave * 4 + 0] = na; sC[wave * 4 + 1] = nl; sC[wave * 4 + 2] = nc; sN[wave] = ns; }
    }
    __syncthreads();
    LSTAMP(7);
    for (int i = tid; i < SL * LD_TOPN; i += NT) {
        float a = 0;
#pragma unroll
        for (int wv = 0; wv < NW; wv++) a += sRed[wv * SL * LD_TOPN + i];
        nxt.topA[(size_t) chunk * SL * LD_TOPN + i] = a;
        if (HAS_L) {
            float l = 0;
#pragma unroll
            for (int wv = 0; wv < NW; wv++) l += sRedL[wv * SL * LD_TOPN + i];
            nxt.topL[(size_t) chunk * SL * LD_TOPN + i] = l;
        }
    }
    if (tid == 0) {
        double e = 0; int na = 0, nl = 0, nc = 0; float ns = 0;
        for (int wv = 0; wv < NW; wv++) { e += sE[wv]; na += sC[wv * 4 + 0]; nl += sC[wv * 4 + 1]; nc += sC[wv * 4 + 2]; ns += sN[wv]; }
        nxt.chunkEnergy[chunk] = e;
        nxt.chunkCnt[chunk * 2 + 0] = na; nxt.chunkCnt[chunk * 2 + 1] = nl;
        nxt.chunkNID[chunk * 2 + 0] = ns; nxt.chunkNID[chunk * 2 + 1] = (float) nc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------------------
static size_t linearize_lds_bytes(int SL, int NW, bool hasL) {
    size_t fl = (size_t) SL * (sizeof(DevPair) / 4) + 2 * (size_t) SL * 64 + (size_t) SL * 8 + 2 * (size_t) SL + (size_t) NW * SL * LD_TOPN * (hasL ? 2 : 1);
    return fl * sizeof(float) + NW * (sizeof(double) + 5 * sizeof(float)) + 64;
}

template <int SL, int NW, bool HAS_L, bool FIX, bool MARG = false>
static hipError_t launch_one(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, int stepMode, const GnInit &gi, hipStream_t st,
                            const int32_t *margFlags = nullptr) {
    const size_t lds = linearize_lds_bytes(SL, NW, HAS_L);
    auto kfn = k_linearize<SL, NW, HAS_L, FIX, MARG>;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *) kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(kfn, dim3(D.nChunks), dim3(64 * NW), lds, st, B, D, cur, nxt, S, stepMode, gi, margFlags);
    return hipGetLastError();
}

template <int SL>
static hipError_t launch_sl(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, bool hasL, bool fix, int stepMode, const GnInit &gi,
                            hipStream_t st) {
    // one-wave workgroups (D.lnw == 1: one point group per workgroup, the latency-optimal shape of small windows) exist for the
    // variants of the GN loop; everything else runs the four-wave shape
    if (D.lnw == 1 && !hasL) return fix ? launch_one<SL, 1, false, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<SL, 1, false, false>(B, D, cur, nxt, S, stepMode, gi, st);
    if (D.lnw == 1) return fix ? launch_one<SL, 1, true, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<SL, 1, true, false>(B, D, cur, nxt, S, stepMode, gi, st);
    if (hasL) return fix ? launch_one<SL, 4, true, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<SL, 4, true, false>(B, D, cur, nxt, S, stepMode, gi, st);
    return fix ? launch_one<SL, 4, false, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<SL, 4, false, false>(B, D, cur, nxt, S, stepMode, gi, st);
}

hipError_t ba_launch_linearize(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S,
                               bool hasL, bool fix, int stepMode, const GnInit &gi, hipStream_t st) {
    if (D.nChunks == 0) return hipSuccess;
    return (D.FS == 8) ? launch_sl<8>(B, D, cur, nxt, S, hasL, fix, stepMode, gi, st) : launch_sl<16>(B, D, cur, nxt, S, hasL, fix, stepMode, gi, st);
}

// marginalizePointsF accumulate for the flagged points (see the MARG note at k_linearize); `nxt` is scratch
hipError_t ba_launch_linearize_marg(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, const int32_t *margFlags,
                                    hipStream_t st) {
    if (D.nChunks == 0) return hipSuccess;
    const GnInit gi{0, 0, 0.0f, -1};
    if (D.FS == 8) return (D.lnw == 1) ? launch_one<8, 1, false, false, true>(B, D, cur, nxt, S, 0, gi, st, margFlags) : launch_one<8, 4, false, false, true>(B, D, cur, nxt, S, 0, gi, st, margFlags);
    return (D.lnw == 1) ? launch_one<16, 1, false, false, true>(B, D, cur, nxt, S, 0, gi, st, margFlags) : launch_one<16, 4, false, false, true>(B, D, cur, nxt, S, 0, gi, st, margFlags);
}
