// ba_solve.hip — the "control" side of a Gauss-Newton iteration (gfx950): everything the reference does sequentially on
// the host between two residual passes, so that an iteration needs no host synchronisation.
//
// Two kernels share the device functions below:
//   k_gn_solve  the forced-accept fast path (ldso_ba_optimize / ldso_ba_enqueue_gn / ldso_ba_gn_solve_reduced): two concurrent
//               workgroups - block 0: solve_core (+ backupState, doStepFromBackup, canbreak, setPrecalcValues) on an LDS mirror of
//               the frames; block 1: statistics of the previous linearizeAll (energy sums, setNewFrameEnergyTH, energy log).
//   k_solve     one workgroup, flag-selected pieces for the step-wise C entry points (LM rejection, debugging, old multi-GPU path):
//     SK_POST     FullSystem::linearizeAll tail: energy sum (FullSystem.cc:1762-1793);  SK_THRESH: setNewFrameEnergyTH
//     SK_ADJ      EnergyFunctional::setAdjointsF (EnergyFunctional.cc:431-489) + gauge nullspace basis
//                 for orthogonalize (EnergyFunctional.cc:685-717, FullSystem.cc:1711-1760)
//     SK_SOLVE    EnergyFunctional::solveSystemF after the assembly (EnergyFunctional.cc:293-351, default solver mode) and the
//                 frame part of resubstituteF_MT (:491-516); HFinal / bFinal come from k_gather (ba_reduce.hip)
//     SK_BACKUP   FullSystem::backupState (FullSystem.cc:1625-1673)
//     SK_STEP     FullSystem::doStepFromBackup frame/calib part + canbreak (FullSystem.cc:1546-1623)
//     SK_LOADBK   FullSystem::loadSateBackup frame/calib part (FullSystem.cc:1675-1692)
//     SK_PRECALC  FullSystem::setPrecalcValues: FrameFramePrecalc::Set for F^2 pairs + setDeltaF
//                 (FullSystem.cc:1423-1431, FrameFramePrecalc.cc:6-35, EnergyFunctional.cc:403-429)
//     SK_REANCHOR end of optimize(): newest frame setEvalPT(PRE_worldToCam, newStateZero) (FullSystem.cc:833-841)
//     SK_COLLECT / SK_LOG / SK_EXPORT / SK_FROMREDUCED: optimize() preamble, energy log, multi-GPU scalar exchange
// All dense math is fp64 like the reference.  The factorisation is an unpivoted LDL^T of the (diag+10)^-1/2-scaled system (SPD
// after the scaling; Eigen's LDLT pivots on the diagonal and gives the same solution up to rounding), see solve_core.
#include <hip/hip_runtime.h>
#include "ba_dev.h"
#include "lie_dev.h"
#include "ba_solve.h"

#include "ba_reduce_body.h"
#define NT 256
#define TH_CAP 8192      // candidate energies staged in LDS up to this many points

static __device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum over the 64 lanes through DPP row operations + four v_readlane (a dependent ds_bpermute costs 30 ns, a DPP step 7 ns):
// for the serial chains that are made of wave sums (the one-sided Jacobi of the nullspace basis)
template <int CTRL> static __device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (u & 0xFFFFFFFFu), CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}
static __device__ __forceinline__ double readlane_d(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (u & 0xFFFFFFFFu), lane), hi = (unsigned) __builtin_amdgcn_readlane((int) (u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}
static __device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);      // row_half_mirror
    v += dpp_f64<0x140>(v);      // row_mirror: every lane holds the sum of its 16-lane row
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// ---------------------------------------------------------------------------------------------------------
// setNewFrameEnergyTH: k-th smallest of the newest frame's residual energies by a 4-pass radix select
// ---------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void post_sums(const BaPtrs &B, const BaDims &D, const ResSet &S, double *sD /*8 doubles*/) {
    const int tid = threadIdx.x;
    // ---- energy sum + counters over the chunks, fixed order -----------------------------------------------
    {
        double e = 0; double na = 0, nl = 0, ns = 0, nc = 0;
        for (int c = tid; c < D.nChunks; c += NT) {
            e += S.chunkEnergy[c]; na += S.chunkCnt[c * 2]; nl += S.chunkCnt[c * 2 + 1];
            ns += (double) S.chunkNID[c * 2]; nc += (double) S.chunkNID[c * 2 + 1];
        }
        e = wave_sum(e); na = wave_sum(na); nl = wave_sum(nl); ns = wave_sum(ns); nc = wave_sum(nc);
        if ((tid & 63) == 0) { sD[(tid >> 6)] = e; sD[4 + (tid >> 6)] = na; }
        __syncthreads();
        if (tid == 0) { B.scalars[0] = sD[0] + sD[1] + sD[2] + sD[3]; B.scalars[1] = sD[4] + sD[5] + sD[6] + sD[7]; }
        __syncthreads();
        if ((tid & 63) == 0) { sD[(tid >> 6)] = nl; sD[4 + (tid >> 6)] = ns; }
        __syncthreads();
        if (tid == 0) { B.scalars[2] = sD[0] + sD[1] + sD[2] + sD[3]; B.scalars[6] = sD[4] + sD[5] + sD[6] + sD[7]; }
        __syncthreads();
        if ((tid & 63) == 0) sD[(tid >> 6)] = nc;
        __syncthreads();
        if (tid == 0) B.scalars[7] = sD[0] + sD[1] + sD[2] + sD[3];
        __syncthreads();
    }
}

// resInA / resInL of the accumulate that produced this set (EnergyFunctional.cc:558,573)
static __device__ __forceinline__ void res_counts(const BaPtrs &B, const BaDims &D, const ResSet &S, double *sD /*8 doubles*/) {
    const int tid = threadIdx.x;
    double na = 0, nl = 0;
    for (int c = tid; c < D.nChunks; c += NT) { na += S.chunkCnt[c * 2]; nl += S.chunkCnt[c * 2 + 1]; }
    na = wave_sum(na); nl = wave_sum(nl);
    if ((tid & 63) == 0) { sD[tid >> 6] = na; sD[4 + (tid >> 6)] = nl; }
    __syncthreads();
    if (tid == 0) { B.scalars[9] = sD[0] + sD[1] + sD[2] + sD[3]; B.scalars[10] = sD[4] + sD[5] + sD[6] + sD[7]; }
    __syncthreads();
}

// candidates: active-set residuals targeting the newest frame with state_NewEnergyWithOutlier >= 0,
// written compactly by the linearize kernel (S.candE[p], -1 = no candidate).
// extE (multi-GPU): all-reduced array of P doubles holding value+1 for candidates and 0 otherwise.
static __device__ __forceinline__ void post_thresh(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St,
                                   const double *extE, float *sVal /*LDS, TH_CAP floats*/, int *sHist /*256 ints*/, int *sI /*8 ints*/) {
    const int tid = threadIdx.x;
    const int F = D.F;
    const int tN = F - 1;
    const int nItems = D.P;
    const bool inLds = nItems <= TH_CAP;
    // stage the candidate values (negative = not a candidate)
    for (int i0 = tid; i0 < nItems; i0 += NT * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int i = i0 + u * NT;
            v[u] = -1.0f;
            if (i < nItems) {
                if (extE != nullptr) { double d = extE[i]; v[u] = (d > 0.0) ? (float) (d - 1.0) : -1.0f; }
                else if (i >= D.pBegin && i < D.pEnd) v[u] = S.candE[i];
            }
        }
        if (inLds) {
#pragma unroll
            for (int u = 0; u < 8; u++) { int i = i0 + u * NT; if (i < nItems) sVal[i] = v[u]; }
        }
    }
    __syncthreads();
    auto value = [&](int i) -> float {
        if (inLds) return sVal[i];
        if (extE != nullptr) { double d = extE[i]; return (d > 0.0) ? (float) (d - 1.0) : -1.0f; }
        return (i >= D.pBegin && i < D.pEnd) ? S.candE[i] : -1.0f;
    };
    int cnt = 0;
    for (int i = tid; i < nItems; i += NT) if (value(i) >= 0.0f) cnt++;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) sI[tid >> 6] = cnt;
    __syncthreads();
    const int total = sI[0] + sI[1] + sI[2] + sI[3];
    __syncthreads();
    DevFrame &nf = B.frames[tN];
    if (total == 0) {
        if (tid == 0) nf.frameEnergyTH = 12 * 12 * 8;
        __syncthreads();
    } else {
        int kth = (int) (St.frameEnergyTHN * (float) total);
        unsigned prefix = 0, mask = 0;
        for (int pass = 3; pass >= 0; pass--) {
            sHist[tid] = 0;
            __syncthreads();
            const int shift = pass * 8;
            for (int i = tid; i < nItems; i += NT) {
                float v = value(i);
                if (v >= 0.0f) {
                    unsigned u = __float_as_uint(v);
                    if ((u & mask) == prefix) atomicAdd(&sHist[(u >> shift) & 0xFF], 1);
                }
            }
            __syncthreads();
            {
                const int c = sHist[tid];
                int inc = c;
                for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += v; }
                if ((tid & 63) == 63) sI[tid >> 6] = inc;
                __syncthreads();
                int base = 0;
                for (int wv = 0; wv < (tid >> 6); wv++) base += sI[wv];
                inc += base;
                const int exc = inc - c;
                __syncthreads();
                if (c > 0 && exc <= kth && kth < inc) { sI[4] = tid; sI[5] = kth - exc; }
            }
            __syncthreads();
            prefix |= ((unsigned) sI[4]) << shift;
            mask |= 0xFFu << shift;
            kth = sI[5];
            __syncthreads();
        }
        if (tid == 0) {
            float nthElement = sqrtf(__uint_as_float(prefix));
            float th = nthElement * St.frameEnergyTHFacMedian;
            th = 26.0f * St.frameEnergyTHConstWeight + th * (1 - St.frameEnergyTHConstWeight);
            th = th * th;
            th *= St.overallEnergyTHWeight * St.overallEnergyTHWeight;
            nf.frameEnergyTH = th;
        }
        __syncthreads();
    }
    // refresh thMax of the pairs (the next linearize reads it)
    for (int i = tid; i < F * F; i += NT) {
        int h = i / F, t = i % F;
        B.pairs[i].thMax = fmaxf(B.frames[h].frameEnergyTH, B.frames[t].frameEnergyTH);
    }
    __syncthreads();
}

static __device__ void frame_nullspaces(DevFrame &f) { ld::frame_nullspaces(f.evalPT, f.state_zero[6], f.ab_exposure, f.ns_pose, f.ns_scale, f.ns_affine); }

static __device__ __forceinline__ void aff_from_to(float expF, float expT, float aF, float bF, float aT, float bT, float &a, float &b) {
    if (expF == 0 || expT == 0) { expT = expF = 1; }
    a = expf(aT - aF) * expT / expF;
    b = bT - a * bF;
}

// setAdjointsF + nullspace basis U (n x 7, columns with dropped singular values zeroed)
static __device__ __forceinline__ void set_adjoints(const BaPtrs &B, const BaDims &D, const ldso_settings_t &St, double *sW /*LDS scratch >= 7*n + 64 doubles*/, bool withNullspace,
                                    double *sBig /*LDS >= 37 * F * F doubles (the solve-core region, idle here)*/) {
    const int tid = threadIdx.x, F = D.F, n = D.n;
    // (1) one lane per pair: Adj(T_t T_h^-1) at the evaluation points and the affine factor -> LDS
    for (int i = tid; i < F * F; i += NT) {
        int h = i % F, t = i / F;      // slot h + t*F
        const DevFrame &fh = B.frames[h], &ft = B.frames[t];
        double Ti[12], T[12], Adj[36];
        ld::se3_inv(fh.evalPT, Ti);
        ld::se3_mul(ft.evalPT, Ti, T);
        ld::se3_adj(T, Adj);
        float a, b;
        aff_from_to(fh.ab_exposure, ft.ab_exposure, (float) (fh.state_zero[6] * 10.0), (float) (fh.state_zero[7] * 1000.0),
                    (float) (ft.state_zero[6] * 10.0), (float) (ft.state_zero[7] * 1000.0), a, b);
#pragma unroll
        for (int e = 0; e < 36; e++) sBig[i * 37 + e] = Adj[e];
        sBig[i * 37 + 36] = (double) a;
    }
    __syncthreads();
    // (2) one thread per entry of the two 8x8 adjoints (EnergyFunctional.cc:431-489), coalesced stores of the four tables
    for (int idx = tid; idx < F * F * 64; idx += NT) {
        const int i = idx >> 6, e = idx & 63, r = e >> 3, c = e & 7;
        const double *Adj = sBig + i * 37;
        const double a = Adj[36];
        double AH = (r == c) ? 1.0 : 0.0, AT = AH;
        if (r < 6 && c < 6) AH = -Adj[c * 6 + r];
        if (e == 6 * 8 + 6) { AT = -a; AH = a; }
        if (e == 7 * 8 + 7) { AT = -1.0; AH = a; }
        const double rs = (r < 3) ? 0.5 : (r < 6) ? 1.0 : (r == 6) ? 10.0 : 1000.0;
        AH *= rs; AT *= rs;
        B.adHost[idx] = AH; B.adTarget[idx] = AT;
        B.adHostF[idx] = (float) AH; B.adTargetF[idx] = (float) AT;
    }
    if (!withNullspace) { __syncthreads(); return; }
    // ---- N = [6 pose | 1 scale] nullspaces, columns normalised (FullSystem.cc:1711-1760, EF.cc:691-694) -----
    double *N = sW;               // column-major n x 7
    for (int i = tid; i < n * 7; i += NT) {
        int c = i / n, r = i % n;
        double v = 0;
        if (r >= 4) {
            int f = (r - 4) / 8, o = (r - 4) % 8;
            if (o < 6) {
                v = (c < 6) ? B.frames[f].ns_pose[o * 6 + c] : B.frames[f].ns_scale[o];
                v *= (o < 3) ? (double) (1.0f / 0.5f) : (double) (1.0f / 1.0f);
            }
        }
        N[c * n + r] = v;
    }
    __syncthreads();
    if (tid < 64) {
        // one wave: normalise, then one-sided Jacobi (Hestenes) on the 7 columns
        const int lane = tid;
        for (int c = 0; c < 7; c++) {
            double s = 0;
            for (int r = lane; r < n; r += 64) s += N[c * n + r] * N[c * n + r];
            s = sqrt(wave_sum_dpp(s));
            for (int r = lane; r < n; r += 64) N[c * n + r] /= s;
        }
        for (int sweep = 0; sweep < 30; sweep++) {
            double off = 0;
            for (int p = 0; p < 6; p++)
                for (int q = p + 1; q < 7; q++) {
                    double al = 0, be = 0, ga = 0;
                    for (int r = lane; r < n; r += 64) { double a = N[p * n + r], b = N[q * n + r]; al += a * a; be += b * b; ga += a * b; }
                    al = wave_sum_dpp(al); be = wave_sum_dpp(be); ga = wave_sum_dpp(ga);
                    if (ga == 0) continue;
                    off = fmax(off, fabs(ga) / sqrt(al * be + 1e-300));
                    double zeta = (be - al) / (2 * ga);
                    double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
                    double cs = 1 / sqrt(1 + tt * tt), sn = cs * tt;
                    for (int r = lane; r < n; r += 64) { double a = N[p * n + r], b = N[q * n + r]; N[p * n + r] = cs * a - sn * b; N[q * n + r] = sn * a + cs * b; }
                }
            if (off < 1e-15) break;
        }
        double sv[7], maxSv = 0;
        for (int c = 0; c < 7; c++) {
            double s = 0;
            for (int r = lane; r < n; r += 64) s += N[c * n + r] * N[c * n + r];
            sv[c] = sqrt(wave_sum_dpp(s));
            maxSv = fmax(maxSv, sv[c]);
        }
        for (int c = 0; c < 7; c++) {
            bool keep = sv[c] > St.solverModeDelta * maxSv;
            for (int r = lane; r < n; r += 64) B.nsProj[c * n + r] = keep ? N[c * n + r] / sv[c] : 0.0;
        }
    }
    __syncthreads();
}

// canbreak of doStepFromBackup (FullSystem.cc:1604-1622) in two parts so that it can overlap setPrecalcValues: the four
// sums are accumulated by lanes of waves 1..3 (wave 0 is busy with the frame exponentials), the decision is taken by one
// thread after the next block barrier.
static __device__ __forceinline__ void canbreak_partial(const DevFrame *fr, int F, float *sF /*4 floats of LDS*/) {
    const int tid = threadIdx.x;
    const int which = (tid == 64) ? 0 : (tid == 65) ? 1 : (tid == 128) ? 2 : (tid == 192) ? 3 : -1;
    if (which < 0) return;
    float acc = 0;
    for (int f = 0; f < F; f++) {
        const double *s = fr[f].step;
        if (which == 0) acc += s[6] * s[6];
        else if (which == 1) acc += s[7] * s[7];
        else if (which == 2) acc += s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
        else acc += s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
    }
    acc /= F;
    sF[which] = acc;
}
// the same four sums from the solution vector x (step = -x: the squares are the same numbers), by lanes 0..3 of the calling wavefront.  No branch on
// `which` around the loads: four lanes taking four different branches, each with its own LDS round trips inside the loop over the frames, took 2.8 us
static __device__ __forceinline__ float canbreak_partial_x(const double *x, int F, int which) {
    const int base = (which == 0) ? 6 : (which == 1) ? 7 : (which == 2) ? 0 : 3;
    const bool three = which >= 2;
    const int o1 = three ? base + 1 : base, o2 = three ? base + 2 : base;
    float acc = 0;
    for (int f0 = 0; f0 < F; f0 += 8) {
        double a[8], b[8], c[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const double *s = x + 4 + 8 * min(f0 + u, F - 1); a[u] = s[base]; b[u] = s[o1]; c[u] = s[o2]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double v1 = a[u] * a[u], v3 = a[u] * a[u] + b[u] * b[u] + c[u] * c[u];
            if (f0 + u < F) acc += three ? v3 : v1;
        }
    }
    acc /= F;
    return acc;
}
static __device__ __forceinline__ void canbreak_final(const BaPtrs &B, const ldso_settings_t &St, const float *sF, float sumNID) {
    const float sumA = sF[0], sumB = sF[1], sumT = sF[2], sumR = sF[3];
    bool cb = sqrtf(sumA) < 0.0005 * St.thOptIterations && sqrtf(sumB) < 0.00005 * St.thOptIterations &&
              sqrtf(sumR) < 0.00005 * St.thOptIterations && sqrtf(sumT) * sumNID < 0.00005 * St.thOptIterations;
    B.scalars[3] = cb ? 1.0 : 0.0;
}

// ---- pieces of setPrecalcValues shared by the step-wise path (set_precalc) and the GN tail (gn_tail) ----
// one frame: PRE_worldToCam = exp(state) evalPT, its inverse, the deltas, from the state st[8] the caller holds in registers.  Everything is computed in
// registers and stored afterwards - written straight into the frame, every store would force the loads that follow it to wait for it
static __device__ __forceinline__ void frame_pose(DevFrame &f, const double *st) {
    double sz[8], ev[12];
#pragma unroll
    for (int i = 0; i < 8; i++) sz[i] = f.state_zero[i];
#pragma unroll
    for (int i = 0; i < 12; i++) ev[i] = f.evalPT[i];
    const double ss[6] = {0.5 * st[0], 0.5 * st[1], 0.5 * st[2], 1.0 * st[3], 1.0 * st[4], 1.0 * st[5]};
    double E[12], P[12], Pi[12];
    ld::se3_exp(ss, E);
    ld::se3_mul(E, ev, P);
    ld::se3_inv(P, Pi);
#pragma unroll
    for (int i = 0; i < 12; i++) { f.PRE_w2c[i] = P[i]; f.PRE_c2w[i] = Pi[i]; }
#pragma unroll
    for (int i = 0; i < 8; i++) { f.delta[i] = st[i] - sz[i]; f.delta_prior[i] = st[i]; }
}
// CalibHessian::setValue derived floats (CalibHessian.h:71-85) from the calibration value v[4]
static __device__ __forceinline__ void calib_derived(DevCalib &C, const double *v) {
    float sf[4], dl[4];
    for (int i = 0; i < 4; i++) { sf[i] = (float) (50.0 * v[i]); dl[i] = (float) (v[i] - C.value_zero[i]); }
    for (int i = 0; i < 4; i++) { C.sf[i] = sf[i]; C.cDeltaF[i] = dl[i]; }
    C.si[0] = 1.0f / sf[0]; C.si[1] = 1.0f / sf[1]; C.si[2] = -sf[2] / sf[0]; C.si[3] = -sf[3] / sf[1];
}
// one pair record.  The affine parameters come from delta_prior (= state, written by frame_pose): inside gn_tail the states themselves are being
// stored by other wavefronts while this runs.  FULL = false (inside a GN iteration): the linearisation-point part of a pair (R0, t0, b0: functions of
// evalPT and state_zero only) is left untouched.
// K^-1 by Eigen's 3x3 cofactor inverse (float)
static __device__ __forceinline__ void k_inverse(float fx, float fy, float cx, float cy, float *Ki) {
    const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    auto cof = [&](int a, int b) { int a1 = (a + 1) % 3, a2 = (a + 2) % 3, b1 = (b + 1) % 3, b2 = (b + 2) % 3; return K[a1 * 3 + b1] * K[a2 * 3 + b2] - K[a1 * 3 + b2] * K[a2 * 3 + b1]; };
    float k0 = cof(0, 0), k1 = cof(1, 0), k2 = cof(2, 0);
    float det = (k0 * K[0] + k1 * K[3]) + k2 * K[6];
    float invdet = 1.0f / det;
    Ki[0] = k0 * invdet; Ki[1] = k1 * invdet; Ki[2] = k2 * invdet;
    Ki[3] = cof(0, 1) * invdet; Ki[4] = cof(1, 1) * invdet; Ki[5] = cof(2, 1) * invdet;
    Ki[6] = cof(0, 2) * invdet; Ki[7] = cof(1, 2) * invdet; Ki[8] = cof(2, 2) * invdet;
}
// KiShared: K^-1 of the current calibration where the caller has it (gn_tail: computed once beside the frame poses), or nullptr
template <bool FULL>
static __device__ __forceinline__ void pair_record(const BaPtrs &B, const DevFrame *fr, const DevCalib &C, int F, int i, const float *KiShared = nullptr) {
    const int h = i / F, t = i % F;
    const DevFrame &fh = fr[h], &ft = fr[t];
    DevPair &o = B.pairs[i];
    // every operand into registers before the first store of the record
    double Tw[12], Hc[12], T[12];
#pragma unroll
    for (int q = 0; q < 12; q++) { Tw[q] = ft.PRE_w2c[q]; Hc[q] = fh.PRE_c2w[q]; }
    const float expH = fh.ab_exposure, expT = ft.ab_exposure;
    const float aH = (float) (10.0f * fh.delta_prior[6]), bH = (float) (1000.0f * fh.delta_prior[7]), aT = (float) (10.0f * ft.delta_prior[6]), bT = (float) (1000.0f * ft.delta_prior[7]);
    const float fx = C.sf[0], fy = C.sf[1], cx = C.sf[2], cy = C.sf[3];
    ld::se3_mul(Tw, Hc, T);
    float R[9], tt[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = (float) T[r * 4 + c]; tt[r] = (float) T[r * 4 + 3]; }
    // PRE_RTll / PRE_tTll for point activation: only where the host can call it (after set_frames / optimize), not inside GN iterations
    if (FULL) { float *rt = B.pairRt + (size_t) i * 12; for (int q = 0; q < 9; q++) rt[q] = R[q]; for (int q = 0; q < 3; q++) rt[9 + q] = tt[q]; }
    if (FULL) {
        double Ti[12], T0[12];
        ld::se3_inv(fh.evalPT, Ti);
        ld::se3_mul(ft.evalPT, Ti, T0);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) o.R0[r * 3 + c] = (float) T0[r * 4 + c]; o.t0[r] = (float) T0[r * 4 + 3]; }
        o.b0 = (float) (fh.state_zero[7] * 1000.0);
        o.thMax = fmaxf(fh.frameEnergyTH, ft.frameEnergyTH);
    }
    const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    float Ki[9];
    if (KiShared != nullptr) { for (int q = 0; q < 9; q++) Ki[q] = KiShared[q]; }
    else k_inverse(fx, fy, cx, cy, Ki);
    float KR[9], KRKi[9], Kt[3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) KR[r * 3 + c] = (K[r * 3 + 0] * R[0 * 3 + c] + K[r * 3 + 1] * R[1 * 3 + c]) + K[r * 3 + 2] * R[2 * 3 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) KRKi[r * 3 + c] = (KR[r * 3 + 0] * Ki[0 * 3 + c] + KR[r * 3 + 1] * Ki[1 * 3 + c]) + KR[r * 3 + 2] * Ki[2 * 3 + c];
    for (int r = 0; r < 3; r++) Kt[r] = (K[r * 3 + 0] * tt[0] + K[r * 3 + 1] * tt[1]) + K[r * 3 + 2] * tt[2];
    float a0_, b0_;
    aff_from_to(expH, expT, aH, bH, aT, bT, a0_, b0_);
    for (int q = 0; q < 9; q++) o.KRKi[q] = KRKi[q];
    for (int q = 0; q < 3; q++) o.Kt[q] = Kt[q];
    o.aff[0] = a0_; o.aff[1] = b0_;
}

// setPrecalcValues of the step-wise path and of the host entry points: frames' PRE poses, pair precalc (with the linearisation-point part), deltas
static __device__ __forceinline__ void set_precalc(const BaPtrs &B, const BaDims &D, DevFrame *fr, DevCalib *cal, const float *adH, const float *adT) {
    const int tid = threadIdx.x, F = D.F;
    DevCalib &C = *cal;
    if (tid < F) {
        double st[8];
        for (int i = 0; i < 8; i++) st[i] = fr[tid].state[i];
        frame_pose(fr[tid], st);
    }
    if (tid == 64) {
        double v[4];
        for (int i = 0; i < 4; i++) v[i] = C.value[i];
        calib_derived(C, v);
    }
    __syncthreads();
    for (int i = tid; i < F * F; i += NT) pair_record<true>(B, fr, C, F, i);
    // adHTdeltaF[h + t*F] = delta_h^T adHostF + delta_t^T adTargetF  (EnergyFunctional.cc:403-414), one thread per component; the frames' delta is
    // state - state_zero in double, stored exactly: converting it is converting the difference
    for (int i = tid; i < F * F * 8; i += NT) {
        const int c = i & 7, pr = i >> 3, h = pr / F, t = pr % F;
        const DevFrame &fh = fr[h], &ft = fr[t];
        const float *AH = adH + (size_t) (h + t * F) * 64, *AT = adT + (size_t) (h + t * F) * 64;
        float s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s1 += (float) fh.delta[k] * AH[k * 8 + c];
#pragma unroll
        for (int k = 0; k < 8; k++) s2 += (float) ft.delta[k] * AT[k * 8 + c];
        B.pairs[pr].dp[c] = s1 + s2;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// solve_core — EnergyFunctional::solveSystemF after the assembly (EnergyFunctional.cc:293-351): scale by
// (diag+10)^-1/2, LDL^T, back substitution, unscale, orthogonalise against the gauge nullspaces, then the frame /
// calibration part of resubstituteF_MT (:491-516).
//
// The matrix is symmetric positive (semi-)definite after the scaling, so the factorisation runs WITHOUT pivoting
// (Eigen's LDLT pivots on the diagonal; for SPD input both give the same solution up to rounding).  Zero pivots
// (all-zero row/column of a PSD matrix) are skipped, like Eigen's D^+ pseudo-inverse.
//
// Layout: the (n+1)x(n+1) augmented system (row n = right-hand side, so the forward substitution is part of the
// factorisation) is padded to M = 16*NB and its lower triangle lives in REGISTERS: thread (ty,tx) of the 16x16
// block owns element (ty+16a, tx+16b) of every tile a >= b.  C (= 4) columns are eliminated per round:
//   phase 1: one thread per row replays the four scalar elimination steps on its 4 panel entries (private copy of
//            the 4x4 pivot block) -> multipliers F[i][q] = L[i][k+q] and pivot-row values G[j][q] into LDS;
//   phase 2: every thread applies the rank-4 update v -= sum_q F[i][q] G[j][q] to its register tiles and the
//            owners of the next four columns publish them as the next panel.
// Two barriers per round, no global or matrix LDS traffic inside the loop.  Every global read of the solve is
// issued in the prologue (one latency level).
// ---------------------------------------------------------------------------------------------------------
// reciprocal: hardware estimate (v_rcp_f64, ~2^-24) + ONE Newton step -> relative error <= 2.2e-15 (measured on gfx950 over
// 2^20 samples spanning 26 decades; a second step reaches 1.1e-16).  The pivot reciprocals sit on the serial critical path of the
// factorisation (fp64 VALU results have a ~30-cycle dependent-issue latency), and a 2e-15 perturbation of a multiplier is far
// below the conditioning of the system.
static __device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
}

static __device__ __forceinline__ double readlane_f64(double v, int lane) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (u & 0xFFFFFFFFu), lane);
    unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo);
}

// reciprocal square root: hardware estimate (v_rsq_f64) + one Newton step (relative error <= 4.2e-15; only used as a scaling)
static __device__ __forceinline__ double fast_rsqrt(double d) {
    double r = __builtin_amdgcn_rsq(d);
    double e = __builtin_fma(-(d * r), r, 1.0);
    return __builtin_fma(0.5 * r, e, r);
}
static __device__ __forceinline__ double2 ld2(const double *p) { return *(const double2 *) p; }
static __device__ __forceinline__ void st2(double *p, double a, double b) { *(double2 *) p = make_double2(a, b); }

// Storage of L for the back substitution: L[i][c] (i > c) at sL[LIX(i, c)].
//   NB == 4: zero-initialised square [c][i] with row pitch M+2, so that lane c reads its whole column with 16-byte loads;
//   NB  > 4: packed columns (column c holds rows c..M-1).
#define LIX(i, c) ((NB == 4) ? ((c) * (M + 2) + (i)) : ((c) * M - (((c) * ((c) -1)) >> 1) + ((i) - (c))))

// LD_MF7 = 1: the trailing update of the 112 x 112 system (9..13 key frames, C5) on the fp64 matrix cores like the 64 x 64 one (28 tiles, seven per wavefront, two
// matrix rows per lane in phase 1).  Built, correct (24 GPU tests), and NOT faster: k_gn_solve 61.2 us with it, 61.2 without (round 6, one box) - a round of the
// factorisation is 1.43 us of phase-1 issue, LDS round trips and the barrier either way (25 rounds = 36 us), the matrix cores only replace the cheapest part.
#ifndef LD_MF7
#define LD_MF7 0
#endif
static __host__ __device__ inline size_t solve_core_lds_doubles(int NB, int n) {
    size_t M = 16 * NB;
    size_t L = (NB == 4) ? M * (M + 2) : M * (M + 1) / 2;
    // NB == 4 (MFMA variant): two panel buffers [M][6] + per-wave private copies of F and G (4 waves x 2 x [64][4]) = 44 M
    // (matrix-core variants, NB == 4 and NB == 7: two panel buffers [M][6] + per-wave private copies of F and G, 4 waves x 2 x [M][4], = 44 M)
    return L + 2 * (M + 8) /*D,Y*/ + ((NB == 4 || (NB == 7 && LD_MF7)) ? 46 : 30) * M /*F,G,panel (up to 8 columns, pitch 10)*/ + 2 * M /*scale,x*/ + 7 * (size_t) n + 16;
}

// GN = true (k_gn_solve): the prologue also mirrors the frames / calibration (and, when they fit, the float
// adjoints) into LDS and reduces sumNID, so that the whole control step has ONE global-load latency level; the
// frame / calibration part of backupState + doStepFromBackup is fused into the output pass.
// LDS copies of the float adjoints: 72 floats per 8 x 8 matrix.  At 64 the column reads of xAd / adHTdeltaF (lane = (pair, column), one row per instruction) put the
// eight pairs of a wavefront on the same eight banks - an 8-way conflict on each of the 32 reads of an item.  The copies are ordered by pr = h F + t (the order the
// items walk them in; the global tables are indexed h + F t), so that eight consecutive items always read eight different bank groups
#define LD_AD_LDS_PITCH 72
struct SolveIO {
    DevFrame *fr;          // working copy of the frames (global, or the LDS mirror)
    DevCalib *cal;
    const float *adH, *adT;  // float adjoints [F*F][adPitch] (global: 64, or the LDS copies: LD_AD_LDS_PITCH)
    int adPitch;
    float *ldsAd;          // LDS room for both adjoint tables (GN, may be null)
    double *sRed;          // 16 doubles of LDS scratch
    float sumNID;          // out (GN)
    double lambda;         // GN: LM lambda as passed to solveSystem
    int hasPrior;          // GN: HM / bM present
    const double *redScalars;  // GN, multi-GPU: all-reduced scalar sums (see k_gn_export), nullptr on one GPU
    const double *sx;          // out (GN): the solution in LDS (steps are its negative)
    float *sTail;              // out (GN): LDS scratch for gn_tail, 3 x 2 x LD_XFP floats (the panel buffers, free after the factorisation)
    int *waitCtr;              // GN, fused kernel: HFinal / bFinal are complete when *waitCtr reaches waitTarget (nullptr: already complete)
    int waitTarget;
};

typedef double __attribute__((ext_vector_type(4))) ld_d4;
#ifndef LD_SKIP_DONE
#define LD_SKIP_DONE 0          // experiment (round 6): finished tile columns of the 112 x 112 trailing update skipped (uniform branches): k_gn_solve 60.8 -> 63.5 us at C5 - not used
#endif
#ifndef LD_C7
#define LD_C7 4          // columns per round of the 112 x 112 factorisation (VALU variant)
#endif

template <int NB, int C, bool GN, bool WAIT = false, bool MF = false>
static __device__ __forceinline__ void solve_core(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, int iteration, double *sm, SolveIO &io) {
    constexpr int M = 16 * NB;
    constexpr int CP = C + 2;      // row pitch of the panel buffers: 16-byte aligned rows, conflict-free 16-byte accesses at stride CP
    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int LSZ = (NB == 4) ? M * (M + 2) : M * (M + 1) / 2;
    const double TINY = 2.2250738585072014e-308;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, F = D.F, n = D.n;
    double *sL = sm;                       // L (see LIX)
    double *sD = sL + LSZ;                 // [M + 8]
    double *sY = sD + M + 8;               // [M + 8]
    double *sFp = sY + M + 8;              // [M][CP]        (MF: panel buffer A)
    double *sGp = sFp + CP * M;            // [M][CP]        (MF: panel buffer B)
    double *sPn = sGp + CP * M;            // [M][CP]        (MF: per-wave copies of F, then of G: 2 x 4 x [64][4])
    double *sSc = sPn + (MF ? 2 * 4 * M * 4 : CP * M);            // [M]
    double *sx = sSc + M;                  // [M]
    double *sNs = sx + M;                  // [7][n]
    const double *HF = B.sys + 3 * (n * n + n), *bF = HF + n * n;      // assembled by k_gather (step-wise path)
    const bool ortho = (St.solverMode & LDSO_SOLVER_ORTHOGONALIZE_X) || (iteration >= 2 && (St.solverMode & LDSO_SOLVER_ORTHOGONALIZE_X_LATER));

    // ---------------- prologue: every global load of the control step, issued back to back ----------------
    // GN: HFinal / bFinal (lower triangle) come straight from the accumulator k_reduce added into (B.acc)
    double v[NTILE], dI[NB], dJ[NB], dS = 0.0;
    // MF (n + 1 <= 64): the trailing update runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).  The ten 16x16 tiles of the lower
    // triangle live in MFMA accumulator layout, three slots per wavefront: lane l, register r of a slot <-> element
    // (16 ta + (l >> 4) + 4 r, 16 tb + (l & 15)) of tile (ta, tb); wave w holds (w,0) | (w+1,w... see the packed tables) - 15 = no tile.
    static_assert(!MF || ((NB == 4 || NB == 7) && C == 4), "the MFMA variant is written for the 64x64 and the 112x112 system, 4 columns per round");
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // NS tile slots per wavefront.  NB == 4: the hand-placed ten tiles (three slots; 15 = no tile).  NB == 7 (round 6, windows of 9..13 key frames - C5): the 28
    // tiles of the lower triangle numbered column by column, wave w holds the tiles w, w + 4, w + 8, ... - as tile columns finish from the left every wave loses
    // tiles at the same rate; RPL = 2 rows of the matrix per lane in phase 1 (112 rows on 64 lanes).
    constexpr int NS = !MF ? 1 : (NB == 4 ? 3 : (NTILE + 3) / 4), RPL = (M + 63) / 64;
    int mta[NS], mtb[NS];
    if constexpr (MF && NB == 4) {
        mta[0] = (0x3210 >> (4 * wv)) & 15; mta[1] = (0xF321 >> (4 * wv)) & 15; mta[2] = (0xF332 >> (4 * wv)) & 15;
        mtb[0] = 0; mtb[1] = (0xF111 >> (4 * wv)) & 15; mtb[2] = (0xF322 >> (4 * wv)) & 15;
    } else if constexpr (MF) {
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            const int t = wv + 4 * s_;          // tile number, column-major over the lower triangle: column b starts at b NB - b (b - 1) / 2
            int b = 0;
#pragma unroll
            for (int q = 1; q < NB; q++) b += (t >= q * NB - (q * (q - 1)) / 2) ? 1 : 0;
            mtb[s_] = b; mta[s_] = (t < NTILE) ? b + (t - (b * NB - (b * (b - 1)) / 2)) : 15;
        }
    } else { mta[0] = 15; mtb[0] = 0; }
    ld_d4 Dv[NS];
    if (GN) { HF = B.acc; bF = HF + (size_t) n * n; }
// branch-free (clamped offsets, masked values): bFinal follows HFinal in memory, so row n of the augmented system is HF[n*n + j].
// Straight-line code matters here: in the fused kernel these loads are the first instructions after the wait (cold instruction cache).
#define LD_LOAD_H() do { \
        const int nn_ = n * n + n - 1; \
_Pragma("unroll") \
        for (int a = 0; a < NB; a++) { \
            const int i = ty + 16 * a, j = tx + 16 * a; \
            const double vi = HF[min(i * n + i, nn_)], vj = HF[min(j * n + j, nn_)]; \
            dI[a] = (i < n) ? vi : 0.0; \
            dJ[a] = (j < n) ? vj : 0.0; \
        } \
_Pragma("unroll") \
        for (int a = 0; a < NB; a++) \
_Pragma("unroll") \
            for (int b = 0; b <= a; b++) { \
                const int i = ty + 16 * a, j = tx + 16 * b; \
                const double q = HF[min(i * n + j, nn_)]; \
                v[a * (a + 1) / 2 + b] = (j < n && (!GN || j <= i) && i <= n) ? q : 0.0; \
            } \
        { const double q = HF[min(tid * n + tid, nn_)]; dS = (tid < n) ? q : 0.0; } \
    } while (0)
#define LD_LOAD_H_MF() do { \
        const int nn_ = n * n + n - 1; \
_Pragma("unroll") \
        for (int s_ = 0; s_ < NS; s_++) \
_Pragma("unroll") \
            for (int r = 0; r < 4; r++) { \
                const int i = 16 * mta[s_] + (lane >> 4) + 4 * r, j = 16 * mtb[s_] + (lane & 15); \
                const double q = HF[min(i * n + j, nn_)]; \
                Dv[s_][r] = (mta[s_] < NB && j < n && (!GN || j <= i) && i <= n) ? q : 0.0; \
            } \
        { const double q = HF[min(tid * n + tid, nn_)]; dS = (tid < n) ? q : 0.0; } \
    } while (0)
    // fused kernel (io.waitCtr): the system is still being accumulated by the reduce workgroups of this launch - everything that does
    // not depend on it is loaded and staged first, the system after the wait below
    constexpr bool waitH = GN && WAIT;
    if constexpr (!waitH) { if constexpr (MF) { LD_LOAD_H_MF(); } else { LD_LOAD_H(); } }
    double nsv[4];
    if (ortho) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = tid + u * NT; nsv[u] = (i < 7 * n) ? B.nsProj[i] : 0.0; }      // 7n <= 924
    }
    if (GN) {
        constexpr int FW = (int) (sizeof(DevFrame) / 4), CW = (int) (sizeof(DevCalib) / 4), MW = (LD_MAXF * FW + NT - 1) / NT;
        unsigned mw[MW];
        const unsigned *gF = (const unsigned *) B.frames;
#pragma unroll
        for (int u = 0; u < MW; u++) { const int i = tid + u * NT; mw[u] = (i < F * FW) ? gF[i] : 0u; }
        const unsigned cw = (tid < CW) ? ((const unsigned *) B.calib)[tid] : 0u;
        float4 ah[4], at[4];
        if (io.ldsAd != nullptr) {      // F <= 8: F*F*16 <= 1024 float4 per table
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = tid + u * NT;
                ah[u] = (i < F * F * 16) ? ((const float4 *) B.adHostF)[i] : make_float4(0, 0, 0, 0);
                at[u] = (i < F * F * 16) ? ((const float4 *) B.adTargetF)[i] : make_float4(0, 0, 0, 0);
            }
        }
        double ns = 0, nc = 0;      // sumNID of doStepFromBackup, same summation order as post_sums
        for (int c = tid; c < D.nChunks; c += NT) { ns += (double) S.chunkNID[c * 2]; nc += (double) S.chunkNID[c * 2 + 1]; }
        // ---- consume ----
        unsigned *lF = (unsigned *) io.fr;
#pragma unroll
        for (int u = 0; u < MW; u++) { const int i = tid + u * NT; if (i < F * FW) lF[i] = mw[u]; }
        if (tid < CW) ((unsigned *) io.cal)[tid] = cw;
        if (io.ldsAd != nullptr) {
            float4 *lh = (float4 *) io.ldsAd, *lt = (float4 *) (io.ldsAd + F * F * LD_AD_LDS_PITCH);
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = tid + u * NT, sl = i >> 4, j = ((sl % F) * F + sl / F) * (LD_AD_LDS_PITCH / 4) + (i & 15); if (i < F * F * 16) { lh[j] = ah[u]; lt[j] = at[u]; } }
            io.adH = io.ldsAd; io.adT = io.ldsAd + F * F * LD_AD_LDS_PITCH; io.adPitch = LD_AD_LDS_PITCH;
        }
        for (int o = 32; o > 0; o >>= 1) { double a_ = __shfl_xor(ns, o, 64), b_ = __shfl_xor(nc, o, 64); ns += a_; nc += b_; }
        if ((tid & 63) == 0) { io.sRed[tid >> 6] = ns; io.sRed[4 + (tid >> 6)] = nc; }
    }
    if (ortho) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = tid + u * NT; if (i < 7 * n) sNs[i] = nsv[u]; }
    }
    if (NB == 4) {      // zero the square L buffer
        for (int i = tid; i < LSZ / 2; i += NT) st2(&sL[2 * i], 0.0, 0.0);
    }
    if constexpr (waitH) {
        if (LD_STAMP_ON && tid == 0) B.energyLog[47] = (double) wall_clock64();
        if (tid == 0) {
            while (__hip_atomic_load(io.waitCtr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < io.waitTarget) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(io.waitCtr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every producer has incremented: re-arm for the next launch
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (LD_STAMP_ON && tid == 0) B.energyLog[48] = (double) wall_clock64();
        if constexpr (MF) { LD_LOAD_H_MF(); } else { LD_LOAD_H(); }
    }
    if (tid < M) sSc[tid] = fast_rsqrt(dS + 10.0);
    if constexpr (MF) {
        __syncthreads();
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            const double sj = sSc[min(16 * mtb[s_] + (lane & 15), M - 1)];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * mta[s_] + (lane >> 4) + 4 * r;
                const double si = (i == n) ? 1.0 : sSc[min(i, M - 1)];
                Dv[s_][r] = si * Dv[s_][r] * sj;
            }
        }
        // first panel (columns 0..3): the tiles of tile column 0 (NB == 4: slot 0 of every wave)
#pragma unroll
        for (int s_ = 0; s_ < NS; s_++) {
            if (mta[s_] >= NB || mtb[s_] != 0) continue;          // uniform per wave
            if ((lane & 15) < C) {
#pragma unroll
                for (int r = 0; r < 4; r++) sFp[(16 * mta[s_] + (lane >> 4) + 4 * r) * CP + (lane & 15)] = Dv[s_][r];
            }
        }
    } else {
        double sI[NB], sJ[NB];
#pragma unroll
        for (int a = 0; a < NB; a++) { sI[a] = (ty + 16 * a == n) ? 1.0 : fast_rsqrt(dI[a] + 10.0); sJ[a] = fast_rsqrt(dJ[a] + 10.0); }
#pragma unroll
        for (int a = 0; a < NB; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) v[a * (a + 1) / 2 + b] = sI[a] * v[a * (a + 1) / 2 + b] * sJ[b];
    }
    // first panel (columns 0..C-1)
    if (!MF && tx < C) {
#pragma unroll
        for (int a = 0; a < NB; a++) sPn[(ty + 16 * a) * CP + tx] = v[a * (a + 1) / 2];
    }
    __syncthreads();
    if (LD_STAMP_ON && GN && tid == 0) B.energyLog[41] = (double) wall_clock64();
    if (GN) io.sumNID = (io.redScalars != nullptr) ? (float) io.redScalars[3] / (float) io.redScalars[4]
                                                   : (float) (io.sRed[0] + io.sRed[1] + io.sRed[2] + io.sRed[3]) / (float) (io.sRed[4] + io.sRed[5] + io.sRed[6] + io.sRed[7]);

#if LD_STAMP_ON
    long long rc_[6] = {0, 0, 0, 0, 0, 0}, rt_ = 0;
#define RCYC(i, WAITLDS) do { if (WAITLDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long now_ = (long long) __builtin_readcyclecounter(); \
        rc_[i] += now_ - rt_; rt_ = now_; } while (0)
    rt_ = (long long) __builtin_readcyclecounter();
#else
#define RCYC(i, WAITLDS) do { } while (0)
#endif
#pragma nounroll
    for (int k = 0; k < n; k += C) {
        // ---------------- phase 1: LDL^T of the CxC pivot block (replicated in every lane) + this row's multipliers -------
        // c[r][q] (r >= q) = column q of the block after the eliminations 0..q-1 (unscaled), d_q = c[q][q], inv_q = 1/d_q;
        // g_q = A[i][k+q] after the eliminations 0..q-1, f_q = g_q / d_q = L[i][k+q].  Every update is written as
        // x -= (product of already known values) * inv_q, so that pivot d_{q+1} is ONE fma behind the reciprocal of d_q.
        // MF: every wavefront replays phase 1 for all 64 rows (lane = row) - the four copies run side by side on the four SIMDs and
        // spare the hand-over of F / G through a workgroup barrier; the panel buffers alternate (one barrier per round)
        if (MF || tid < M) {
            const double *sPr = MF ? (((k >> 2) & 1) ? sGp : sFp) : sPn;
            double c[C][C], inv[C];
#pragma unroll
            for (int r = 0; r < C; r++)
#pragma unroll
                for (int q2 = 0; q2 <= r; q2 += 2) {
                    const double2 w = ld2(&sPr[(k + r) * CP + q2]);
                    c[r][q2] = w.x; if (q2 + 1 < C) c[r][q2 + 1] = w.y;
                }
            // this lane's row(s) of the panel: one (the row of the thread / of the lane), or - matrix-core variant above 64 rows - the rows lane and lane + 64
            constexpr int NR = MF ? RPL : 1;
            double g[NR][C], f[NR][C];
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const int i = MF ? min(lane + 64 * rr, M - 1) : tid;
#pragma unroll
                for (int q2 = 0; q2 < C; q2 += 2) { const double2 w = ld2(&sPr[i * CP + q2]); g[rr][q2] = w.x; g[rr][q2 + 1] = w.y; }
            }
            RCYC(0, true);          // panel in registers
            // l_r = c[r][q] / d_q once per pivot, every update ONE fma on it, this row's g decoupled from f: 38 fp64 instructions per round.  (Until round 4
            // every update was written as x -= (product of known values) * inv_q, which puts pivot d_{q+1} one fma behind the reciprocal of d_q but takes
            // 48 instructions: 25.4 -> 25.7 k GN iterations/s at C3 for the short form, same box - the round is bound by issue, not by the pivot chain.)
#pragma unroll
            for (int q = 0; q < C; q++) {
                const double d = c[q][q];
                inv[q] = (fabs(d) > TINY) ? fast_rcp(d) : 0.0;
                double l_[C];
#pragma unroll
                for (int r = q + 1; r < C; r++) l_[r] = c[r][q] * inv[q];
#pragma unroll
                for (int r = q + 1; r < C; r++) {
#pragma unroll
                    for (int s_ = q + 1; s_ <= r; s_++) c[r][s_] = __builtin_fma(-l_[r], c[s_][q], c[r][s_]);
#pragma unroll
                    for (int rr = 0; rr < NR; rr++) g[rr][r] = __builtin_fma(-g[rr][q], l_[r], g[rr][r]);
                }
            }
#pragma unroll
            for (int rr = 0; rr < NR; rr++)
#pragma unroll
                for (int q = 0; q < C; q++) f[rr][q] = g[rr][q] * inv[q];
#if LD_STAMP_ON
            asm volatile("" :: "v"(f[0][C - 1]), "v"(g[0][C - 1]));
#endif
            RCYC(1, false);         // 4 x 4 block and this row's multipliers issued
#pragma unroll
            for (int rr = 0; rr < NR; rr++) {
                const int i = MF ? lane + 64 * rr : tid;
                if (MF && i >= M) continue;
                const bool below = (i >= k + C);
                const bool rowOn = below && (i <= n), colOn = below && (i < n);
                if constexpr (MF) {
                    double *Fw = sPn + wv * (M * 4), *Gw = sPn + 4 * M * 4 + wv * (M * 4);       // this wave's own copies: [M rows][4]
#pragma unroll
                    for (int q2 = 0; q2 < C; q2 += 2) {
                        st2(&Fw[i * 4 + q2], rowOn ? f[rr][q2] : 0.0, rowOn ? f[rr][q2 + 1] : 0.0);
                        st2(&Gw[i * 4 + q2], colOn ? g[rr][q2] : 0.0, colOn ? g[rr][q2 + 1] : 0.0);
                    }
                } else {
#pragma unroll
                    for (int q2 = 0; q2 < C; q2 += 2) {
                        st2(&sFp[i * CP + q2], rowOn ? f[rr][q2] : 0.0, rowOn ? f[rr][q2 + 1] : 0.0);
                        st2(&sGp[i * CP + q2], colOn ? g[rr][q2] : 0.0, colOn ? g[rr][q2 + 1] : 0.0);
                    }
                }
                // L, y, D are stored once (MF: by wave 3 - it holds ONE tile, wave 0 holds three for the longest and is the wave the round's barrier waits for:
                // 36.9 -> 35.6 us per iteration at C3, A/B x 3 on one box, profiles/r05_call0_solve_variants.log; skipping phase 1 in waves without a live tile was slower)
                const bool keeper = !MF || wv == 3;
                if (keeper && i > k && i < n) {       // L[i][k+q] for the rows of the pivot block (q < i-k) and all rows below it
#pragma unroll
                    for (int q = 0; q < C; q++) if (i > k + q) sL[LIX(i, k + q)] = f[rr][q];
                }
                if (keeper && i == n) {               // forward-substituted rhs (k + C <= n + C - 1 < M + C: sY has C spare entries)
#pragma unroll
                    for (int q2 = 0; q2 < C; q2 += 2) st2(&sY[k + q2], g[rr][q2], g[rr][q2 + 1]);
                }
                if (keeper && i == k) {
#pragma unroll
                    for (int q2 = 0; q2 < C; q2 += 2) st2(&sD[k + q2], c[q2][q2], c[q2 + 1][q2 + 1]);
                }
            }
        }
        if constexpr (MF) {
            // ---------------- phase 2 on the matrix cores: one v_mfma_f64_16x16x4_f64 per live tile (D -= F_rows G_cols^T) -------------
            // A operand: lane l supplies F[16 ta + (l & 15)][l >> 4], B operand: G[16 tb + (l & 15)][l >> 4] - read back from this
            // wave's own LDS copies (same wave wrote them: no barrier).  Tile columns left of the next panel are finished.
            RCYC(2, true);          // F, G, L, y, D stored
            const int bDone = (k + C) >> 4, c0 = (k + C) & 15;
            const double *Fw = sPn + wv * (M * 4), *Gw = sPn + 4 * M * 4 + wv * (M * 4);
            double *sPw = ((k >> 2) & 1) ? sFp : sGp;
#pragma unroll
            for (int s_ = 0; s_ < NS; s_++) {
                if (mta[s_] >= NB || mtb[s_] < bDone) continue;          // uniform per wave
                const double aop = -Fw[(16 * mta[s_] + (lane & 15)) * 4 + (lane >> 4)];
                const double bop = Gw[(16 * mtb[s_] + (lane & 15)) * 4 + (lane >> 4)];
                Dv[s_] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, Dv[s_], 0, 0, 0);
                // owners of columns k+C .. k+2C-1 publish them as the next panel
                if (mtb[s_] == bDone && (lane & 15) >= c0 && (lane & 15) < c0 + C) {
#pragma unroll
                    for (int r = 0; r < 4; r++) sPw[(16 * mta[s_] + (lane >> 4) + 4 * r) * CP + (lane & 15) - c0] = Dv[s_][r];
                }
            }
            RCYC(3, true);          // operands read back, matrix cores, next panel published
            __syncthreads();
            RCYC(4, false);         // barrier
            continue;
        }
        __syncthreads();
        // ---------------- phase 2 (branch free: finished columns have G = 0, finished rows F = 0) ----------------
        const int a0 = (k + C) >> 4, c0 = (k + C) & 15;
        double fi[NB][C], gj[NB][C];
#pragma unroll
        for (int a = 0; a < NB; a++) {
            if (NB > 4 && LD_SKIP_DONE && a < a0) {          // uniform: no live tile in this tile row / column any more
#pragma unroll
                for (int q = 0; q < C; q++) { fi[a][q] = 0.0; gj[a][q] = 0.0; }
                continue;
            }
#pragma unroll
            for (int q2 = 0; q2 < C; q2 += 2) {
                const double2 f0 = ld2(&sFp[(ty + 16 * a) * CP + q2]), g0 = ld2(&sGp[(tx + 16 * a) * CP + q2]);
                fi[a][q2] = f0.x; fi[a][q2 + 1] = f0.y; gj[a][q2] = g0.x; gj[a][q2 + 1] = g0.y;
            }
        }
        // (tile columns left of the next panel are finished, their G is zero; skipping them - LD_SKIP_DONE - is slower: the round is a latency chain, not fma issue)
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (NB > 4 && LD_SKIP_DONE && b < a0) continue;          // uniform
#pragma unroll
            for (int a = b; a < NB; a++) {
                double w = v[a * (a + 1) / 2 + b];
#pragma unroll
                for (int q = 0; q < C; q++) w = __builtin_fma(-fi[a][q], gj[b][q], w);
                v[a * (a + 1) / 2 + b] = w;
            }
        }
        // owners of columns k+C .. k+2C-1 publish them as the next panel
        const bool pub = (tx >= c0) && (tx < c0 + C);
#pragma unroll
        for (int a = 0; a < NB; a++) {
            double w = v[a * (a + 1) / 2];
#pragma unroll
            for (int b = 1; b <= a; b++) w = (a0 == b) ? v[a * (a + 1) / 2 + b] : w;
            if (pub && a >= a0) sPn[(ty + 16 * a) * CP + (tx - c0)] = w;
        }
        __syncthreads();
    }
    if (LD_STAMP_ON && GN && tid == 0) B.energyLog[42] = (double) wall_clock64();
#if LD_STAMP_ON
    if (GN && MF && tid == 0) for (int u = 0; u < 5; u++) B.energyLog[53 + u] = (double) rc_[u];
#endif

    // ---------------- back substitution by wave 0: x = L^-T D^+ y ----------------
    if (tid < 64) {
        const int lane = tid;
        if (NB == 4) {
            // lane i owns x_i and column i of L in registers (zero outside the strict lower triangle); x_k is broadcast
            // with v_readlane
            double row[64];
#pragma unroll
            for (int kk = 0; kk < 64; kk += 2) { const double2 q = ld2(&sL[lane * (M + 2) + kk]); row[kk] = q.x; row[kk + 1] = q.y; }
            const double d = sD[lane], z = sY[lane];
            double xi = (lane < n && fabs(d) > TINY) ? z * fast_rcp(d) : 0.0;
            // (tried, round 4: 16-lane blocks, x_k to the lanes of its row by DPP row_newbcast - 2 moves + 1 fma per column, no v_readlane - and the 16
            // finished values of a block to the lanes below through LDS: bitwise the same x, 2.04 against 1.78 us.  A column is one broadcast + one fp64 fma
            // on the dependent chain either way, ~50 cycles; the DPP moves wait for the fma like the readlanes do, and the LDS hand-overs come on top.)
#pragma unroll
            for (int kk = 63; kk >= 1; kk--) { double xk = readlane_f64(xi, kk); xi = __builtin_fma(-row[kk], xk, xi); }
            if (lane < n) sx[lane] = xi * sSc[lane];
        } else {
            // lane l owns x_i for i = l, l + 64, (l + 128) in registers; x_k is broadcast with v_readlane (no LDS round trip and no
            // wave barrier per column: the column-oriented LDS version took 15 us at n = 100); the L entries do not depend on x and
            // are loaded ahead of the dependent chain (the loop carries no LDS store)
            constexpr int NSEG = (M + 63) / 64;
            double xr[NSEG];
#pragma unroll
            for (int sg = 0; sg < NSEG; sg++) {
                const int i = lane + 64 * sg;
                const double d = sD[min(i, M - 1)];
                xr[sg] = (i < n && fabs(d) > TINY) ? sY[min(i, M - 1)] / d : 0.0;
            }
            // (round 6, measured and not kept: the L entries fetched in blocks of 16 rows one block ahead of the chain, 32 + 32 doubles in registers - k_gn_solve at C5
            // 61.2 -> 67.8 us; 8 columns per round in the factorisation, LD_C7 = 8: - 1 us)
#pragma unroll 4
            for (int kk = n - 1; kk > 0; kk--) {           // x_i -= L[k][i] x_k for i < k
                double lk[NSEG];
#pragma unroll
                for (int sg = 0; sg < NSEG; sg++) { const int i = lane + 64 * sg; lk[sg] = (i < kk) ? sL[LIX(kk, min(i, kk))] : 0.0; }
                double xk = 0.0;
#pragma unroll
                for (int sg = 0; sg < NSEG; sg++) if ((kk >> 6) == sg) xk = readlane_f64(xr[sg], kk & 63);      // uniform
#pragma unroll
                for (int sg = 0; sg < NSEG; sg++) xr[sg] = __builtin_fma(-lk[sg], xk, xr[sg]);
            }
#pragma unroll
            for (int sg = 0; sg < NSEG; sg++) { const int i = lane + 64 * sg; if (i < n) sx[i] = xr[sg] * sSc[i]; }
        }
        // orthogonalize x against the gauge nullspaces (x -= U U^T x): 8 lanes per nullspace vector
        if (ortho) {
            __builtin_amdgcn_wave_barrier();
            const int kk = lane >> 3, j = lane & 7;
            double c = 0;
            if (kk < 7) for (int r = j; r < n; r += 8) c = __builtin_fma(sNs[kk * n + r], sx[r], c);
            c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
            double cc[7];
#pragma unroll
            for (int q = 0; q < 7; q++) cc[q] = readlane_f64(c, q * 8);
            for (int r = lane; r < n; r += 64) {
                double s_ = 0;
#pragma unroll
                for (int q = 0; q < 7; q++) s_ = __builtin_fma(sNs[q * n + r], cc[q], s_);
                sx[r] -= s_;
            }
        }
    }
    __syncthreads();
    if (LD_STAMP_ON && GN && tid == 0) B.energyLog[43] = (double) wall_clock64();
    // ---------------- outputs: x, steps, xAd ----------------
    DevFrame *fr = io.fr;
    io.sx = sx;
    auto x_ad = [&](int i) {
        int c = i & 7, pr = i >> 3, h = pr / F, t = pr % F;
        const int sl = (io.adPitch == 64) ? h + F * t : pr;          // the LDS copies are ordered by pr = h F + t
        const float *AH = io.adH + (size_t) sl * io.adPitch, *AT = io.adT + (size_t) sl * io.adPitch;
        float ah[8], at[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { ah[kk] = AH[kk * 8 + c]; at[kk] = AT[kk * 8 + c]; }
        float s1 = 0, s2 = 0;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) s1 += (float) sx[4 + 8 * h + kk] * ah[kk];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) s2 += (float) sx[4 + 8 * t + kk] * at[kk];
        B.xAd[i] = s1 + s2;
    };
    if constexpr (GN) { io.sTail = (float *) sFp; return; }          // the panel buffers are free now; gn_tail does the rest
    bool bad = false;
    for (int i = tid; i < n; i += NT) { B.x[i] = sx[i]; if (!isfinite(sx[i])) bad = true; }
    if (bad) B.scalars[4] = 1.0;
    if (tid < 4) {
        const double st = -sx[tid];
        io.cal->step[tid] = st; B.xc[tid] = (float) sx[tid];
    }
    for (int i = tid; i < F * 10; i += NT) {
        int f = i / 10, a = i % 10;
        const double st = (a < 8) ? -sx[4 + 8 * f + a] : 0.0;
        fr[f].step[a] = st;
    }
    for (int i = tid; i < F * F * 8; i += NT) x_ad(i);
    __syncthreads();
}

template <bool GN, bool WAIT = false>
static __device__ __forceinline__ void solve_core_dispatch(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, int iteration, double *sm, SolveIO &io) {
    if (D.n + 1 <= 64) solve_core<4, 4, GN, WAIT, true>(B, D, S, St, iteration, sm, io);
    else if (D.n + 1 <= 112) solve_core<7, (LD_MF7 != 0) ? 4 : LD_C7, GN, WAIT, LD_MF7 != 0>(B, D, S, St, iteration, sm, io);
    else solve_core<9, 4, GN, WAIT>(B, D, S, St, iteration, sm, io);
}

// frame / calibration part of backupState, doStepFromBackup (+ canbreak), loadSateBackup on the working copies
static __device__ void frames_backup(DevFrame *fr, DevCalib *cal, int F) {
    const int tid = threadIdx.x;
    if (tid < F) for (int i = 0; i < 10; i++) fr[tid].state_backup[i] = fr[tid].state[i];
    if (tid == 0) for (int i = 0; i < 4; i++) cal->value_backup[i] = cal->value[i];
    __syncthreads();
}
static __device__ __forceinline__ void frames_step(const BaPtrs &B, const ldso_settings_t &St, DevFrame *fr, DevCalib *cal, int F, float sumNID, double *sRed) {
    const int tid = threadIdx.x;
    if (tid < F) for (int i = 0; i < 10; i++) fr[tid].state[i] = fr[tid].state_backup[i] + fr[tid].step[i];
    if (tid == 0) for (int i = 0; i < 4; i++) cal->value[i] = cal->value_backup[i] + cal->step[i] * (double) 1.0f;
    canbreak_partial(fr, F, (float *) (sRed + 8));
    __syncthreads();
    if (tid == 0) canbreak_final(B, St, (const float *) (sRed + 8), sumNID);
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_solve(BaPtrs B, BaDims D, ResSet S, ldso_settings_t St, SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, F = D.F, n = D.n;
    const int NBsel = (n + 1 <= 64) ? 4 : (n + 1 <= 112) ? 7 : 9;
    double *sW = sm + solve_core_lds_doubles(NBsel, n);      // scratch: 7n + 64
    // the threshold pass (histogram + staged candidates) runs before the solve and is over when solve_core starts: it aliases its LDS
    int *sHist = (int *) sm;                 // 256
    int *sI = sHist + 256;                   // 8 (+ TH_CAP floats of candidate staging behind it)
    // two workgroups (ba_launch_solve): the statistics of the last linearizeAll (POST / THRESH / LOG) run next to the control part;
    // they touch disjoint data (frameEnergyTH is only written by the statistics and only read by k_linearize)
    unsigned fl = A.flags;
    if (gridDim.x == 2) fl &= (blockIdx.x == 1) ? (SK_POST | SK_THRESH | SK_LOG) : ~(unsigned) (SK_POST | SK_THRESH | SK_LOG);

    if (fl & SK_COLLECT) {
        // FullSystem::optimize preamble: resetOOB on every non-linearised residual (FullSystem.cc:744-748)
        for (int i = tid; i < D.P * D.FS; i += NT)
            if (B.rtab[i].rflat >= 0 && !B.rtab[i].rlin) { S.slot[i].e[LD_SM_STATE].m.i = 0; S.slot[i].e[LD_SM_ENERGY].m.f = 0.0f; }
        __syncthreads();
    }
    if (fl & SK_POST) post_sums(B, D, S, sW);

    if (fl & SK_REANCHOR) {
        // FrameHessian::setEvalPT + setStateZero of the newest frame (FrameHessian.cc:12-42).  The numeric nullspaces are 14 independent
        // exp / mul / mul / log chains (6 directions x +-eps, scale up / down): one lane each - the same operations as ld::frame_nullspaces
        // (bit-identical), a fourteenth of its latency (it was 30 us of the 39 us tail on one lane)
        DevFrame &f = B.frames[F - 1];
        double *sLp = sW;                  // [14][6] logs
        double *sEv = sW + 96;             // the new evalPT
        if (tid < 12) sEv[tid] = f.PRE_w2c[tid];
        __syncthreads();
        if (tid < 14) {
            double Ti[12], Am[12], Bm[12], lg[6];
            ld::se3_inv(sEv, Ti);
            if (tid < 12) {
                double eps[6] = {0, 0, 0, 0, 0, 0}, E[12];
                const double e_ = (tid & 1) ? -1e-3 : 1e-3;
#pragma unroll
                for (int q = 0; q < 6; q++) eps[q] = (q == (tid >> 1)) ? e_ : 0.0;
                ld::se3_exp(eps, E); ld::se3_mul(sEv, E, Am); ld::se3_mul(Am, Ti, Bm); ld::se3_log(Bm, lg);
            } else {
                double Tp[12];
                for (int i = 0; i < 12; i++) Tp[i] = sEv[i];
                if (tid == 12) { for (int i = 0; i < 3; i++) Tp[i * 4 + 3] *= 1.00001; } else { for (int i = 0; i < 3; i++) Tp[i * 4 + 3] /= 1.00001; }
                ld::se3_mul(Tp, Ti, Am); ld::se3_log(Am, lg);
            }
#pragma unroll
            for (int r = 0; r < 6; r++) sLp[tid * 6 + r] = lg[r];
        }
        __syncthreads();
        if (tid < 36) { const int r = tid / 6, i = tid % 6; f.ns_pose[r * 6 + i] = (sLp[(2 * i) * 6 + r] - sLp[(2 * i + 1) * 6 + r]) / 2e-3; }
        else if (tid < 42) { const int r = tid - 36; f.ns_scale[r] = (sLp[12 * 6 + r] - sLp[13 * 6 + r]) / 2e-3; }
        else if (tid < 54) f.evalPT[tid - 42] = sEv[tid - 42];
        else if (tid == 54) {
            const double a = f.state[6], b = f.state[7];
            for (int i = 0; i < 10; i++) { f.state[i] = 0; f.state_zero[i] = 0; }
            f.state[6] = a; f.state[7] = b; f.state_zero[6] = a; f.state_zero[7] = b;
            for (int i = 0; i < 8; i++) f.ns_affine[i] = 0;
            f.ns_affine[0] = 1;
            f.ns_affine[3] = (double) (expf((float) (a * 10.0)) * f.ab_exposure);
        }
        __syncthreads();
    }
    if (fl & SK_ADJ) set_adjoints(B, D, St, sW, !(fl & SK_NONULLSPACE), sm);

    if (fl & SK_EXPORT) {
        // multi-GPU: rank-local scalar sums ride in the all-reduce buffer
        double na = 0, nl = 0;
        for (int c = tid; c < D.nChunks; c += NT) { na += S.chunkCnt[c * 2]; nl += S.chunkCnt[c * 2 + 1]; }
        na = wave_sum(na); nl = wave_sum(nl);
        if ((tid & 63) == 0) { sW[tid >> 6] = na; sW[4 + (tid >> 6)] = nl; }
        __syncthreads();
        if (tid == 0) { double *sc = A.reduceOut + 3 * (n * n + n); sc[0] = B.scalars[0]; sc[1] = B.scalars[1]; sc[2] = B.scalars[2]; sc[3] = B.scalars[6]; sc[4] = B.scalars[7];
                        sc[5] = sW[0] + sW[1] + sW[2] + sW[3]; sc[6] = sW[4] + sW[5] + sW[6] + sW[7]; sc[7] = 0; }
        __syncthreads();
    }
    if (fl & SK_FROMREDUCED) {
        const double *sc = A.reduceIn + 3 * (n * n + n);
        if (tid == 0) { B.scalars[0] = sc[0]; B.scalars[1] = sc[1]; B.scalars[2] = sc[2]; B.scalars[6] = sc[3]; B.scalars[7] = sc[4]; B.scalars[9] = sc[5]; B.scalars[10] = sc[6]; }
        __syncthreads();
    }
    if (fl & SK_THRESH) post_thresh(B, D, S, St, (fl & SK_FROMREDUCED) ? (A.reduceIn + 3 * (n * n + n) + 8) : nullptr, (float *) (sI + 8), sHist, sI);
    if (fl & SK_LOG) { if (tid == 0 && A.logIdx >= 0 && A.logIdx < 64) B.energyLog[A.logIdx] = B.scalars[0]; __syncthreads(); }

    if (fl & SK_SOLVE) {
        // HFinal / bFinal were assembled by k_gather (ba_reduce.hip)
        if (!(fl & SK_FROMREDUCED)) res_counts(B, D, S, sW);
        SolveIO io;
        io.fr = B.frames; io.cal = B.calib; io.adH = B.adHostF; io.adT = B.adTargetF; io.adPitch = 64; io.ldsAd = nullptr; io.sRed = sW; io.sumNID = 0; io.lambda = A.lambda; io.hasPrior = A.hasPrior; io.redScalars = nullptr;
        solve_core_dispatch<false>(B, D, S, St, A.iteration, sm, io);
    }
    if (fl & SK_BACKUP) frames_backup(B.frames, B.calib, F);
    if (fl & SK_STEP) frames_step(B, St, B.frames, B.calib, F, (float) B.scalars[6] / (float) B.scalars[7], sW);
    if (fl & SK_LOADBK) {
        if (tid < F) for (int i = 0; i < 10; i++) B.frames[tid].state[i] = B.frames[tid].state_backup[i];
        if (tid == 0) for (int i = 0; i < 4; i++) B.calib->value[i] = B.calib->value_backup[i];
        __syncthreads();
    }
    if (fl & SK_PRECALC) set_precalc(B, D, B.frames, B.calib, B.adHostF, B.adTargetF);
}

// ---------------------------------------------------------------------------------------------------------
// k_gn_solve — the control step of ONE forced-accept Gauss-Newton iteration, two concurrent workgroups:
//   block 0 (critical path): solve_core + backupState + doStepFromBackup + setPrecalcValues, working on an LDS
//            mirror of the frames / calibration that is written back once at the end;
//   block 1 (statistics of the previous linearizeAll): energy sums, setNewFrameEnergyTH, energy log.
// The two blocks touch disjoint data: block 0 never reads frameEnergyTH (the linearize kernel takes the pair
// maximum itself) and skips that word in its write-back; canbreak's sumNID is recomputed from the chunk sums.
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// gn_tail - everything of the control step behind the back substitution: x, xAd, backupState + doStepFromBackup (frames, calibration), canbreak,
// setPrecalcValues (frame poses, pair records, adHTdeltaF).  One workgroup on the critical path of the iteration, so it is laid out by WAVEFRONT around the
// one chain that is long:  new states -> frame poses (a lane per frame, a few hundred dependent fp64 instructions) -> pair records (a lane per pair).
// Wave 0 runs that chain and nothing else: it forms the new states in registers, touches no memory another wave writes, and (windows of up to 8 frames:
// all pairs fit its lanes) needs no workgroup barrier between the poses and the pair records.  Waves 1..3 do the wide work beside it: float copies of x
// and of the new deltas (private to each wave), per item (pair, column) xAd AND adHTdeltaF in one pass over the adjoint columns (the same loads; waves 1
// and 2), the calibration (its step, the derived floats, K^-1: one lane, handed to wave 0 through a flag), the remainder of the items and canbreak (wave 3),
// the frames' step / state_backup (wave 2).  The OLD states stay in the working copies - every
// wave reads them - and the new ones reach memory through delta_prior (see the end).  One workgroup barrier, at the end.
// Device stamps, C3, us after the back substitution - before: outputs 1.2, poses 2.0 - 2.8, pair records 1.0, adHTdeltaF 0.85 = 5.6 (the poses stored through
// LDS between dependent products, canbreak walked four divergent branches with LDS round trips inside, and every wave walked every phase); now: see DESIGN 5.
// ---------------------------------------------------------------------------------------------------------
#define LD_XFP (8 * LD_MAXF + 8)
// LDS scratch of the control workgroup (gn_solve_body: sW, LD_SW_DOUBLES doubles): doubles [0, LD_SW_RED) are io.sRed (solve_core's partial sums of sumNID - the
// only part of it the solve may touch), the floats behind them belong to gn_tail: canbreak sums 0..3, K^-1 8..16, the calibration hand-over flag LD_TAIL_FLAG
#define LD_SW_DOUBLES 64
#define LD_SW_RED 8
#define LD_TAIL_KI 8
#define LD_TAIL_FLAG 20          // float index behind cbF of the calibration hand-over flag
static_assert(NT == 256, "gn_tail is laid out for exactly four wavefronts: wave 0 waits for a flag that wave 3 sets");
static_assert(LD_TAIL_KI + 9 <= LD_TAIL_FLAG && (LD_TAIL_FLAG + 1) * 4 <= (LD_SW_DOUBLES - LD_SW_RED) * 8, "gn_tail's floats must fit the scratch behind io.sRed");
static __device__ __forceinline__ void gn_tail(const BaPtrs &B, const BaDims &D, DevFrame *fr, DevCalib *cal, const SolveIO &io, const ldso_settings_t &St, float *cbF,
                                               int cbIter, int *hostStop, int lastIt) {
    const int tid = threadIdx.x, F = D.F, n = D.n, wave = tid >> 6, lane = tid & 63;
    constexpr int W2 = 128;          // waves 1 and 2 walk the items; wave 3 takes what is left of them, and canbreak
    const double *sx = io.sx;
    DevCalib &C = *cal;
    float *sKi = cbF + LD_TAIL_KI;          // 9 floats of the scratch behind the canbreak sums
    int *sFlag = (int *) (cbF + LD_TAIL_FLAG);          // cleared by the caller before the solve
    const bool split = F * F <= 64;          // one wavefront holds all pairs
    const int items = F * F * 8, nMain = (items / W2) * W2;
    float *xf = io.sTail + (wave > 0 ? wave - 1 : 0) * 2 * LD_XFP, *df = xf + LD_XFP;
    const float *adH = io.adH, *adT = io.adT;
    const int adPitch = io.adPitch;
    // xAd[i] = x_h^T adHostF[:, c] + x_t^T adTargetF[:, c] (EnergyFunctional.cc:491-516), adHTdeltaF likewise from the deltas (:403-414)
    auto item = [&](int i) {
        const int c = i & 7, pr = i >> 3, h = pr / F, t = pr % F;
        const int sl = (adPitch == 64) ? h + F * t : pr;          // the LDS copies are ordered by pr = h F + t
        const float *AH = adH + (size_t) sl * adPitch, *AT = adT + (size_t) sl * adPitch;
        float ah[8], at[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { ah[k] = AH[k * 8 + c]; at[k] = AT[k * 8 + c]; }
        const float4 xh0 = *(const float4 *) (xf + 4 + 8 * h), xh1 = *(const float4 *) (xf + 8 + 8 * h), xt0 = *(const float4 *) (xf + 4 + 8 * t), xt1 = *(const float4 *) (xf + 8 + 8 * t);
        const float4 dh0 = *(const float4 *) (df + 4 + 8 * h), dh1 = *(const float4 *) (df + 8 + 8 * h), dt0 = *(const float4 *) (df + 4 + 8 * t), dt1 = *(const float4 *) (df + 8 + 8 * t);
        const float xh[8] = {xh0.x, xh0.y, xh0.z, xh0.w, xh1.x, xh1.y, xh1.z, xh1.w}, xt[8] = {xt0.x, xt0.y, xt0.z, xt0.w, xt1.x, xt1.y, xt1.z, xt1.w};
        const float dh[8] = {dh0.x, dh0.y, dh0.z, dh0.w, dh1.x, dh1.y, dh1.z, dh1.w}, dt[8] = {dt0.x, dt0.y, dt0.z, dt0.w, dt1.x, dt1.y, dt1.z, dt1.w};
        float s1 = 0, s2 = 0, d1 = 0, d2 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { s1 += xh[k] * ah[k]; d1 += dh[k] * ah[k]; }
#pragma unroll
        for (int k = 0; k < 8; k++) { s2 += xt[k] * at[k]; d2 += dt[k] * at[k]; }
        B.xAd[i] = s1 + s2;
        B.pairs[pr].dp[c] = d1 + d2;
    };
    if (wave == 0) {
        if (lane < F) {
            double st[8];
#pragma unroll
            for (int a = 0; a < 8; a++) st[a] = fr[lane].state[a] + (-sx[4 + 8 * lane + a]);          // doStepFromBackup: state = backup + step, step = -x
            frame_pose(fr[lane], st);
        }
        if (LD_STAMP_ON && tid == 0) B.energyLog[59] = (double) wall_clock64();          // frame poses done
        if (split) {
            // poses, calibration floats and pair records all belong to this wave: no workgroup barrier between them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // wave 3: calibration floats, K^-1 (long done).  Bounded like the p2p polls: a flag that never comes (a launch shape this was not written for) ends
            // in the non-finite status of the iteration (scalars[4]) instead of a hung kernel
            int spins = 0;
            while (__hip_atomic_load(sFlag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 22) && lane == 0) B.scalars[4] = 1.0;
            if (lane < F * F) pair_record<false>(B, fr, C, F, lane, sKi);
        }
        if (LD_STAMP_ON && tid == 0) B.energyLog[62] = (double) wall_clock64();          // wave 0 at the barrier
    } else {
        bool bad = false;
        for (int i = lane; i < n; i += 64) {
            const double x = sx[i];
            double d = 0.0;
            if (i >= 4) {
                const int f = (i - 4) >> 3, a = (i - 4) & 7;
                const double bk = fr[f].state[a];
                d = (bk + (-x)) - fr[f].state_zero[a];
                if (wave == 2) { fr[f].step[a] = -x; fr[f].state_backup[a] = bk; }          // backupState, the step: written only, by nobody else
            }
            xf[i] = (float) x; df[i] = (float) d;
            if (wave == 1) { B.x[i] = x; if (!isfinite(x)) bad = true; }
        }
        if (wave == 2 && lane < 2 * F) { const int f = lane >> 1, a = 8 + (lane & 1); fr[f].step[a] = 0.0; fr[f].state_backup[a] = fr[f].state[a]; }
        if (bad) B.scalars[4] = 1.0;
        // the copies are private to the wave; LDS executes a wavefront's accesses in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (wave < 3) for (int i = tid - 64; i < nMain; i += W2) item(i);
        if (wave == 3) {
            // the calibration: backupState + doStepFromBackup, the derived floats, K^-1 for the pair records - one lane, off wave 0's chain (a branch of
            // its own there ran behind the poses: +0.25 us); handed to wave 0 through a flag in LDS (a wavefront's LDS accesses complete in order)
            if (lane == 63) {
                double v[4];
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const double stp = -sx[a], bk = C.value[a];
                    v[a] = bk + stp * (double) 1.0f;
                    C.step[a] = stp; C.value_backup[a] = bk; C.value[a] = v[a]; B.xc[a] = (float) sx[a];
                }
                calib_derived(C, v);
                float Ki[9];
                k_inverse((float) (50.0 * v[0]), (float) (50.0 * v[1]), (float) (50.0 * v[2]), (float) (50.0 * v[3]), Ki);          // = C.sf
                for (int q = 0; q < 9; q++) sKi[q] = Ki[q];
                __hip_atomic_store(sFlag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            for (int i = nMain + lane; i < items; i += 64) item(i);
            // canbreak (FullSystem.cc:1604-1622): the four sums, their square roots and comparisons on four lanes side by side, the conjunction by ballot
            // (one lane walking sqrt -> compare -> branch four times in a row, behind a round trip through LDS, took as long as the sums)
            bool ok = true;
            if (lane < 4) {
                float v = sqrtf(canbreak_partial_x(sx, F, lane));
                if (lane == 2) v = v * io.sumNID;
                ok = v < ((lane == 0) ? 0.0005 : 0.00005) * St.thOptIterations;
            }
            const bool cb = (__ballot(ok) & 0xFull) == 0xFull;
            if (lane == 0) {
                B.scalars[3] = cb ? 1.0 : 0.0;
                // un-forced optimize(): end the loop after this iteration (FullSystem.cc:829)
                if (cbIter >= 0 && cb && cbIter >= St.minOptIterations && (double) cbIter < B.scalars[LD_SC_STOP]) B.scalars[LD_SC_STOP] = (double) cbIter;
                // tell the host at once which iteration ended the loop (it then enqueues the tail behind the iterations that turn into no-ops
                // instead of synchronising with the stream first); iterations after the stop return at their first instruction and never get here
                if (hostStop != nullptr && cbIter >= 0 && (B.scalars[LD_SC_STOP] == (double) cbIter || cbIter == lastIt))
                    __hip_atomic_store(hostStop, cbIter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (LD_STAMP_ON && lane == 0) B.energyLog[63] = (double) wall_clock64();          // wave 3 at the barrier
        }
        if (LD_STAMP_ON && tid == 64) B.energyLog[58] = (double) wall_clock64();          // wave 1 at the barrier
    }
    if (!split) {
        __syncthreads();
        for (int i = tid; i < F * F; i += NT) pair_record<false>(B, fr, C, F, i, sKi);
    }
    // The frames' NEW states are in delta_prior (= state, setPrecalcValues); the state fields of the working copies still hold the old ones - they were
    // read by every wave up to here.  The caller's write-back of the copies takes state[0..7] from delta_prior (gn_solve_body).
    __syncthreads();
}

template <bool WAIT>
static __device__ __forceinline__ void gn_solve_body(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, const int role) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, F = D.F, n = D.n;
    const int NBsel = (n + 1 <= 64) ? 4 : (n + 1 <= 112) ? 7 : 9;
    double *sW = sm + solve_core_lds_doubles(NBsel, n);      // LD_SW_DOUBLES doubles of scratch (layout: above gn_tail)
    if (LD_ITER_SKIPPED(B, A.itCheck)) return;
    const long long t0_ = wall_clock64();
#define GSTAMP(i) do { if (LD_STAMP_ON && tid == 0) B.energyLog[40 + (i)] = (double) (wall_clock64() - t0_); } while (0)
    if (role == 1) {
        double *sW1 = sm;                        // block 1 never runs solve_core: its scratch starts at the base
        int *sHist = (int *) (sW1 + 64);
        int *sI = sHist + 256;
        if (A.reduceIn != nullptr) {
            // multi-GPU: the sums over all ranks arrive in the all-reduce buffer (k_gn_export layout), the candidates behind them
            const double *sc = A.reduceIn;
            if (tid == 0) { B.scalars[0] = sc[0]; B.scalars[1] = sc[1]; B.scalars[2] = sc[2]; B.scalars[6] = sc[3]; B.scalars[7] = sc[4]; B.scalars[9] = sc[5]; B.scalars[10] = sc[6]; }
            __syncthreads();
            post_thresh(B, D, S, St, A.reduceIn + 8, (float *) (sI + 8), sHist, sI);
        } else {
            post_sums(B, D, S, sW1);
            GSTAMP(10);
            res_counts(B, D, S, sW1);
            GSTAMP(11);
            post_thresh(B, D, S, St, nullptr, (float *) (sI + 8), sHist, sI);
            GSTAMP(12);
        }
        if (tid == 0 && A.logIdx >= 0 && A.logIdx < 64) B.energyLog[A.logIdx] = B.scalars[0];
        return;
    }
    DevFrame *sFr = (DevFrame *) (sW + LD_SW_DOUBLES);
    DevCalib *sCal = (DevCalib *) (sFr + F);
    if (LD_STAMP_ON && tid == 0) B.energyLog[39] = (double) t0_;
    SolveIO io;
    io.fr = sFr; io.cal = sCal; io.adH = B.adHostF; io.adT = B.adTargetF; io.adPitch = 64; io.sRed = sW; io.sumNID = 0; io.lambda = A.lambda; io.hasPrior = A.hasPrior; io.redScalars = A.reduceIn; io.waitCtr = A.waitCtr; io.waitTarget = A.waitTarget;
    io.ldsAd = (F <= 8) ? (float *) (sCal + 1) : nullptr;
    if (tid == 0) *(int *) ((float *) (sW + LD_SW_RED) + LD_TAIL_FLAG) = 0;          // gn_tail's hand-over flag (the solve's barriers publish it)
    solve_core_dispatch<true, WAIT>(B, D, S, St, A.iteration, sm, io);      // + mirrors
    GSTAMP(4);
    GSTAMP(5);
    gn_tail(B, D, sFr, sCal, io, St, (float *) (sW + LD_SW_RED), A.itCheck, A.hostStop, A.lastIt);      // x, xAd, backupState + doStepFromBackup + canbreak, setPrecalcValues
    GSTAMP(6);
    // write the mirrors back (all but frameEnergyTH, which block 1 owns)
    {
        unsigned *gF = (unsigned *) B.frames; const unsigned *lF = (const unsigned *) sFr;
        const int W = (int) (sizeof(DevFrame) / 4), skip = (int) (offsetof(DevFrame, frameEnergyTH) / 4);
        const int s0 = (int) (offsetof(DevFrame, state) / 4), dp0 = (int) (offsetof(DevFrame, delta_prior) / 4);
        for (int i = tid; i < F * W; i += NT) {
            const int w = i % W;
            if (w != skip) gF[i] = lF[(w >= s0 && w < s0 + 16) ? i - s0 + dp0 : i];          // doStepFromBackup: state[0..7] = the new states (gn_tail)
        }
        unsigned *gC = (unsigned *) B.calib; const unsigned *lC = (const unsigned *) sCal;
        for (int i = tid; i < (int) (sizeof(DevCalib) / 4); i += NT) gC[i] = lC[i];
    }
}

__global__ __launch_bounds__(NT) void k_gn_solve(BaPtrs B, BaDims D, ResSet S, ldso_settings_t St, SolveArgs A) {
    // the 672 bytes of arguments into the scalar cache with one wait (ba_dev.h) - for the systems up to 64 x 64 only: measured (round 4, A/B on one box)
    // 27.2 -> 26.5 us at C3 (n = 60), but 62.1 -> 63.6 us at C5 (n = 100, the generic factorisation)
    static_assert(sizeof(BaPtrs) + sizeof(BaDims) + sizeof(ResSet) + sizeof(ldso_settings_t) + sizeof(SolveArgs) >= 10 * 64 - 60, "k_gn_solve: ld_touch_kernarg<10> must stay inside the arguments");
    if (D.n + 1 <= 64) ld_touch_kernarg<10>();
    gn_solve_body<false>(B, D, S, St, A, (int) blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// k_reduce_solve — k_reduce and k_gn_solve of one GN iteration in ONE launch (single-GPU fast path): workgroup 0 is the control
// step, workgroup 1 the statistics, workgroups 2.. the reduce workgroups.  The control workgroup stages everything that does not
// depend on the system (frames, calibration, adjoints, nullspace projector, LDS set-up) while the reduce workgroups run, waits
// until all of them have signalled (increment of a device counter once their atomics are acknowledged; acquire fence on the other side)
// and only then loads HFinal / bFinal.  This removes one kernel boundary and hides the launch-to-first-data latency of the control
// step (about 5 us of a 50 us iteration).  All workgroups of the launch are resident at once (F*F + 4*tiles + 3 <= 256 CUs), and the
// waiting workgroup has the lowest index, so it never keeps a producer from being scheduled.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_reduce_solve(BaPtrs B, BaDims D, ResSet S, ldso_settings_t St, SolveArgs A, ChunkStarts chunkStart, int atomicMode,
                                                     float calibPrior, double l1, double il) {
    static_assert(sizeof(BaPtrs) + sizeof(BaDims) + sizeof(ResSet) + sizeof(ldso_settings_t) + sizeof(SolveArgs) + sizeof(ChunkStarts) + 24 >= 12 * 64 - 60, "k_reduce_solve: ld_touch_kernarg<12> must stay inside the arguments");
    ld_touch_kernarg<12>();          // 768 bytes of arguments into the scalar cache with one wait (ba_dev.h): 31.8 -> 31.25 us at C3
    if (blockIdx.x >= 2) {
        const long long tStart_ = LD_STAMP_ON ? wall_clock64() : 0;
        if (LD_ITER_SKIPPED(B, A.itCheck)) return;
        // dispatch order = index order: the Schur tile workgroups (the longest) get the lowest indices, then the pair workgroups,
        // the extras workgroup last; reduce_body numbers them pairs | tiles | extras
        const int nT_ = A.GSP / 16, nTiles = D.ks * nT_ * (nT_ + 1) / 2, nPair = D.F * D.F * (A.hasL ? 2 : 1);
        const int q = (int) blockIdx.x - 2;
        const int bid = (q < nTiles) ? nPair + q : (q < nTiles + nPair) ? q - nTiles : q;
        reduce_body(B, D, S, chunkStart, A.hasL, A.GSP, atomicMode, A.hasPrior, calibPrior, l1, il, -1, bid);
        // Everything a reduce workgroup hands to the control workgroup went through device-scope atomics (performed at the memory
        // side): waiting for their acknowledgement is enough - a release FENCE would also write the whole L2 back (the outputs of
        // the previous k_linearize are still dirty there), which costs more than the kernel boundary this fusion removes.
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int prev = __hip_atomic_fetch_add(A.waitCtr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (LD_STAMP_ON && prev == A.waitTarget - 1) { B.energyLog[29] = (double) wall_clock64(); B.energyLog[30] = (double) bid; B.energyLog[31] = (double) tStart_; }      // who was last, and when
        }
        return;
    }
    gn_solve_body<true>(B, D, S, St, A, (int) blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// Batched windows: k_reduce and the control step of nWin independent windows, one launch each (the fused k_reduce_solve needs
// every workgroup of a window resident at once - with many windows per launch a kernel boundary orders the two instead).
// ---------------------------------------------------------------------------------------------------------
// Two register allocations of the same kernel.  Its workgroups are latency-bound (each lives ~10 us whatever the load), so for a launch
// that queues many of them per CU residency is throughput, while a launch that fits the chip in one go only sees the spill code.  Measured,
// B = 32 (2080 workgroups per half-batch launch, 8 per CU): 3 per CU (132 VGPRs, the unconstrained allocation) and 4 (126) 101.4 k
// window-iterations/s, 5 (96 VGPRs, 30 spilled; also fits beside the two resident k_linearize_batch wavefronts of a SIMD, 2 x 205 + 96 <= 512, so
// the reduce of one half-batch overlaps the linearisation of the other) 104.8 k, 6 (80 VGPRs, 48 spilled) 83.0 k; B = 8 (520 workgroups per
// launch, 2 per CU): 82 k unconstrained, 75.7 k with the 96-register allocation (different boxes, same day).
#ifndef LD_REDB_BLOCKS
#define LD_REDB_BLOCKS 4             // the dense variant: workgroups per CU its register allocation leaves room for (round 4, against the record-layout
                                     // k_linearize_batch of 187 VGPRs, B = 32 different windows: 3 -> 122.4 k, 4 -> 122.4 k, 5 -> 96.8 k, 6 -> 87.5 k window-iterations/s,
                                     // unconstrained 107.8 k; with round 3's 205-VGPR kernel 5 had been the best: 104.8 k)
#endif
#ifndef LD_REDB_DENSE_PER_CU
#define LD_REDB_DENSE_PER_CU 6       // launches with at least this many workgroups per CU take the dense variant
#endif
static __device__ __forceinline__ void reduce_batch_body(const BatchItem *__restrict__ items, int nWin, int cur, float calibPrior, double l1, double il) {
    int w = 0;
    for (int i = 1; i < nWin; i++) if ((int) blockIdx.x >= items[i].redBlock0) w = i;
    const BatchItem &it = items[w];
    reduce_body(it.B, it.D, it.set[cur], it.cs, 0, it.GSP, 1, it.hasPrior, calibPrior, l1, il, -1, (int) blockIdx.x - it.redBlock0);
}
__global__ __launch_bounds__(NT) void k_reduce_batch(const BatchItem *__restrict__ items, int nWin, int cur, float calibPrior, double l1, double il) {
    reduce_batch_body(items, nWin, cur, calibPrior, l1, il);
}
__global__ __launch_bounds__(NT, LD_REDB_BLOCKS) void k_reduce_batch_dense(const BatchItem *__restrict__ items, int nWin, int cur, float calibPrior, double l1, double il) {
    reduce_batch_body(items, nWin, cur, calibPrior, l1, il);
}

__global__ __launch_bounds__(NT) void k_gn_solve_batch(const BatchItem *__restrict__ items, int cur, ldso_settings_t St, int iteration, double lambda) {
    const BatchItem &it = items[blockIdx.x >> 1];
    SolveArgs A;
    A.flags = 0; A.iteration = iteration; A.lambda = lambda; A.hasL = 0; A.hasPrior = it.hasPrior; A.GSP = it.GSP; A.logIdx = -1;
    A.reduceOut = nullptr; A.reduceIn = nullptr; A.itCheck = -1; A.waitCtr = nullptr; A.waitTarget = 0; A.hostStop = nullptr; A.lastIt = -1;
    gn_solve_body<false>(it.B, it.D, it.set[cur], St, A, (int) (blockIdx.x & 1));
}

hipError_t ba_launch_reduce_batch(const BatchItem *d_items, int nWin, int totalBlocks, int cur, float calibPrior, double l1, double il, hipStream_t st) {
    const size_t lds = (size_t) (2 * SCT_SLAB * 16 + SCT_SLAB) * sizeof(float);
    static int numCU[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (numCU[dev] == 0) { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256; numCU[dev] = cu; }
    if (lds > 48 * 1024) {
        (void) hipFuncSetAttribute((const void *) k_reduce_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        (void) hipFuncSetAttribute((const void *) k_reduce_batch_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    }
    if (totalBlocks >= LD_REDB_DENSE_PER_CU * numCU[dev]) hipLaunchKernelGGL(k_reduce_batch_dense, dim3(totalBlocks), dim3(NT), lds, st, d_items, nWin, cur, calibPrior, l1, il);
    else hipLaunchKernelGGL(k_reduce_batch, dim3(totalBlocks), dim3(NT), lds, st, d_items, nWin, cur, calibPrior, l1, il);
    return hipGetLastError();
}

// Dmax: the window of the batch with the most frames (LDS of the control step)
hipError_t ba_launch_gn_solve_batch(const BatchItem *d_items, int nWin, const BaDims &Dmax, int cur, const ldso_settings_t &St, int iteration, double lambda, hipStream_t st) {
    size_t mirror = (size_t) Dmax.F * sizeof(DevFrame) + sizeof(DevCalib) + (Dmax.F <= 8 ? (size_t) Dmax.F * Dmax.F * 2 * LD_AD_LDS_PITCH * sizeof(float) : 0), stats = 64 * sizeof(double) + (256 + 8) * sizeof(int) + TH_CAP * sizeof(float);
    const int n = Dmax.n, NBsel = (n + 1 <= 64) ? 4 : (n + 1 <= 112) ? 7 : 9;
    size_t lds0 = solve_core_lds_doubles(NBsel, n) * sizeof(double) + 64 * sizeof(double) + mirror + 64;
    size_t lds = lds0 > stats + 64 ? lds0 : stats + 64;
    if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_gn_solve_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(k_gn_solve_batch, dim3(2 * nWin), dim3(NT), lds, st, d_items, cur, St, iteration, lambda);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// k_gn_export (multi-GPU fast path): rank-local scalar sums and newest-frame energy candidates into the tail of the all-reduce
// buffer  [HFinal lower | bFinal | 8 scalars | P candidates (value+1, 0 = none)]  whose head k_reduce accumulated (B.acc).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_gn_export(BaPtrs B, BaDims D, ResSet S, double *tail) {
    __shared__ double sW[16];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        post_sums(B, D, S, sW);
        res_counts(B, D, S, sW);
        if (tid == 0) { tail[0] = B.scalars[0]; tail[1] = B.scalars[1]; tail[2] = B.scalars[2]; tail[3] = B.scalars[6]; tail[4] = B.scalars[7];
                        tail[5] = B.scalars[9]; tail[6] = B.scalars[10]; tail[7] = 0; }
        return;
    }
    const int i = (blockIdx.x - 1) * NT + tid;
    if (i < D.P) {
        double v = 0.0;
        if (i >= D.pBegin && i < D.pEnd) { const float c = S.candE[i]; if (c >= 0.0f) v = (double) c + 1.0; }
        tail[8 + i] = v;
    }
}

hipError_t ba_launch_gn_export(const BaPtrs &B, const BaDims &D, const ResSet &S, double *tail, hipStream_t st) {
    hipLaunchKernelGGL(k_gn_export, dim3(1 + (D.P + NT - 1) / NT), dim3(NT), 0, st, B, D, S, tail);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// point part of resubstituteFPt + doStepFromBackup / backupState / loadSateBackup
//   (EnergyFunctional.cc:518-547, FullSystem.cc:1585-1602)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_point_step(BaPtrs B, BaDims D, ResSet S, int mode) {
    const int p = D.pBegin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.pEnd) return;
    PtGeo &Gp = B.pgeo[p];
    if (mode & PS_LOAD) { float b = Gp.idepth_backup; Gp.idepth = b; Gp.idepth_zero = b; return; }   // loadSateBackup
    const int F = D.F, FS = D.FS, h = B.phost[p];
    float step = Gp.step;
    PtRec &R = S.pt[p];
    if (mode & PS_RESUB) { Gp.lastHdiF = R.HdiF; Gp.lastBdSumF = R.bdSumF; Gp.lastIdH = R.idH; }      // what this solve's accumulateSCF_MT left in the point
    if ((mode & PS_RESUB) && R.nActive <= 0) { step = 0.0f; R.maxRelBS = 0.0f; }      // AccumulatedSCHessian.cc:14-21 (zeroed by the solve)
    if ((mode & PS_RESUB) && R.nActive > 0) {
        float b = R.bdSumF;
        float dot = 0;
        for (int i = 0; i < 4; i++) dot += B.xc[i] * (R.HcdA[i] + R.HcdL[i]);
        b -= dot;
        bool finite = true;
        for (int t = 0; t < F; t++) {
            int slot = p * FS + t;
            const SlotRec &sr = S.slot[slot];
            if (B.rtab[slot].rflat < 0 || !sr.e[LD_SM_ACTIVE].m.i) continue;
            const float *xa = B.xAd + (size_t) (h * F + t) * 8;
            float s = 0;
            for (int i = 0; i < 8; i++) s += xa[i] * sr.e[i].jp;
            b -= s;
        }
        if (!isfinite(b)) finite = false;
        if (finite) step = -b * R.HdiF; else { step = Gp.step; B.scalars[4] = 1.0; }
    }
    Gp.step = step;
    if (mode & PS_BACKUP) Gp.idepth_backup = Gp.idepth;
    if (mode & PS_STEP) {     // doStepFromBackup (stepfacD = 1)
        float ni = Gp.idepth_backup + 1.0f * step;
        Gp.idepth = ni;
        Gp.idepth_zero = ni;
    }
}

// ---------------------------------------------------------------------------------------------------------
// k_lm_energies — the two energies of the LM accept test (FullSystem.cc:805-826):
//   scalars[12] = EnergyFunctional::calcMEnergyF (EnergyFunctional.cc:353-359):  delta^T (2 b_M + H_M delta)
//   scalars[13] = EnergyFunctional::calcLEnergyF_MT (:361-378, calcLEnergyPt :627-682): prior terms of frames / calibration /
//                 points + the linearised residuals' (2 res_toZeroF + J delta) . J delta
// One workgroup; sums in double (the reference accumulates the point part in float Accumulator11: same value to ~1e-6).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lm_energies(BaPtrs B, BaDims D, ResSet S, float calibPrior, int hasPrior) {
    __shared__ double sDelta[8 * LD_MAXF + 4];
    __shared__ double sRed[8];
    const int tid = threadIdx.x, F = D.F, n = D.n;
    for (int i = tid; i < n; i += 256) sDelta[i] = (i < 4) ? (double) B.calib->cDeltaF[i] : B.frames[(i - 4) >> 3].delta[(i - 4) & 7];
    __syncthreads();
    // ---- M energy ----
    double em = 0.0;
    if (hasPrior) for (int i = tid; i < n; i += 256) {
        double hd = 0.0;
        for (int j = 0; j < n; j++) hd += B.HM[(size_t) i * n + j] * sDelta[j];
        em += sDelta[i] * (2.0 * B.bM[i] + hd);
    }
    // ---- L energy: frame and calibration priors ----
    double el = 0.0;
    for (int i = tid; i < F * 8; i += 256) { const DevFrame &f = B.frames[i >> 3]; const double dp = f.delta_prior[i & 7]; el += dp * f.prior[i & 7] * dp; }
    if (tid < 4) { const float c = B.calib->cDeltaF[tid]; el += (double) (c * calibPrior * c); }
    // ---- L energy: points ----
    const float cD0 = B.calib->cDeltaF[0], cD1 = B.calib->cDeltaF[1], cD2 = B.calib->cDeltaF[2], cD3 = B.calib->cDeltaF[3];
    for (int p = D.pBegin + tid; p < D.pEnd; p += 256) {
        const float dd = B.pgeo[p].idepth - B.pgeo[p].idepth_zero;
        const int h = B.phost[p];
        double e = 0.0;
        for (int t = 0; t < F; t++) {
            const int slot = p * D.FS + t;
            const SlotTab tb = B.rtab[slot];
            if (tb.rflat < 0 || !tb.rlin || !S.slot[slot].e[LD_SM_ACTIVE].m.i) continue;
            const ldso_rawjac_t &J = B.Jlin[tb.rlidx];
            const float *rtz = B.rtz + (size_t) tb.rlidx * 8;
            const float *dp = B.pairs[h * F + t].dp;
            float jx = 0, jy = 0;
            for (int i = 0; i < 6; i++) { jx += J.Jpdxi[0][i] * dp[i]; jy += J.Jpdxi[1][i] * dp[i]; }
            jx = jx + (((J.Jpdc[0][0] * cD0 + J.Jpdc[0][1] * cD1) + J.Jpdc[0][2] * cD2) + J.Jpdc[0][3] * cD3) + J.Jpdd[0] * dd;
            jy = jy + (((J.Jpdc[1][0] * cD0 + J.Jpdc[1][1] * cD1) + J.Jpdc[1][2] * cD2) + J.Jpdc[1][3] * cD3) + J.Jpdd[1] * dd;
            for (int i = 0; i < 8; i++) {
                const float Jd = ((J.JIdx[0][i] * jx + J.JIdx[1][i] * jy) + J.JabF[0][i] * dp[6]) + J.JabF[1][i] * dp[7];
                e += (double) (Jd * ((rtz[i] + rtz[i]) + Jd));
            }
        }
        e += (double) (dd * dd * B.pgeo[p].priorF);
        el += e;
    }
    em = wave_sum(em); el = wave_sum(el);
    if ((tid & 63) == 0) { sRed[tid >> 6] = em; sRed[4 + (tid >> 6)] = el; }
    __syncthreads();
    if (tid == 0) { B.scalars[12] = sRed[0] + sRed[1] + sRed[2] + sRed[3]; B.scalars[13] = sRed[4] + sRed[5] + sRed[6] + sRed[7]; }
}

hipError_t ba_launch_lm_energies(const BaPtrs &B, const BaDims &D, const ResSet &S, float calibPrior, bool hasPrior, hipStream_t st) {
    hipLaunchKernelGGL(k_lm_energies, dim3(1), dim3(256), 0, st, B, D, S, calibPrior, hasPrior ? 1 : 0);
    return hipGetLastError();
}

static size_t solve_lds_common(const BaDims &D) {
    const int n = D.n, NBsel = (n + 1 <= 64) ? 4 : (n + 1 <= 112) ? 7 : 9;
    return solve_core_lds_doubles(NBsel, n) * sizeof(double);
}

hipError_t ba_launch_solve(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, hipStream_t st) {
    size_t stats = (256 + 8) * sizeof(int) + TH_CAP * sizeof(float), core = solve_lds_common(D);
    size_t lds = (core > stats ? core : stats) + (7 * (size_t) D.n + 64) * sizeof(double) + 64;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *) k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    // the tail of optimize(): statistics and the re-anchor / adjoint / precalc part are independent -> two workgroups
    const unsigned fstats = SK_POST | SK_THRESH | SK_LOG, fctl = SK_REANCHOR | SK_ADJ | SK_NONULLSPACE | SK_PRECALC;
    const bool split = (A.flags & fstats) && (A.flags & fctl) && !(A.flags & ~(fstats | fctl));
    hipLaunchKernelGGL(k_solve, dim3(split ? 2 : 1), dim3(NT), lds, st, B, D, S, St, A);
    return hipGetLastError();
}

hipError_t ba_launch_gn_solve(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, hipStream_t st) {
    size_t mirror = (size_t) D.F * sizeof(DevFrame) + sizeof(DevCalib) + (D.F <= 8 ? (size_t) D.F * D.F * 2 * LD_AD_LDS_PITCH * sizeof(float) : 0), stats = 64 * sizeof(double) + (256 + 8) * sizeof(int) + TH_CAP * sizeof(float);
    size_t lds0 = solve_lds_common(D) + 64 * sizeof(double) + mirror + 64;      // block 0
    size_t lds = lds0 > stats + 64 ? lds0 : stats + 64;                         // block 1 aliases the base
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *) k_gn_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(k_gn_solve, dim3(2), dim3(NT), lds, st, B, D, S, St, A);
    return hipGetLastError();
}

// nReduce = workgroups of ba_launch_reduce in atomic mode (see there)
hipError_t ba_launch_reduce_solve(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, const ChunkStarts &chunkStart,
                                  int atomicMode, float calibPrior, double l1, double il, hipStream_t st) {
    const int nT = A.GSP / 16;
    const int nReduce = D.F * D.F * (A.hasL ? 2 : 1) + D.ks * nT * (nT + 1) / 2 + 1;
    size_t mirror = (size_t) D.F * sizeof(DevFrame) + sizeof(DevCalib) + (D.F <= 8 ? (size_t) D.F * D.F * 2 * LD_AD_LDS_PITCH * sizeof(float) : 0), stats = 64 * sizeof(double) + (256 + 8) * sizeof(int) + TH_CAP * sizeof(float);
    size_t lds0 = solve_lds_common(D) + 64 * sizeof(double) + mirror + 64;
    size_t lds = lds0 > stats + 64 ? lds0 : stats + 64;
    const size_t ldsR = (size_t) (2 * SCT_SLAB * 16 + SCT_SLAB) * sizeof(float);
    if (ldsR > lds) lds = ldsR;
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *) k_reduce_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    SolveArgs A2 = A;
    A2.waitTarget = nReduce;
    hipLaunchKernelGGL(k_reduce_solve, dim3(nReduce + 2), dim3(NT), lds, st, B, D, S, St, A2, chunkStart, atomicMode, calibPrior, l1, il);
    return hipGetLastError();
}

hipError_t ba_launch_point_step(const BaPtrs &B, const BaDims &D, const ResSet &S, int mode, hipStream_t st) {
    int np = D.pEnd - D.pBegin;
    if (np <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_point_step, dim3((np + 255) / 256), dim3(256), 0, st, B, D, S, mode);
    return hipGetLastError();
}
