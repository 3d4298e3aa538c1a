// ba_reduce.hip — second stage of the Hessian accumulation (gfx950).
//
// Part A (one block per (host,target) pair; reference AccumulatedTopHessianSSE::stitchDoubleInternal,
//   src/internal/OptimizationBackend/AccumulatedTopHessian.cc:193-255): sum the per-block 91-entry
//   partials of the pair in double, rebuild the 13x13 block and lift it through the adjoints:
//   Ad_h A88 Ad_h^T, Ad_t A88 Ad_t^T, Ad_h A88 Ad_t^T, Ad_h A8c, Ad_t A8c, Acc and the b parts.
// Part B (Schur complement; reference AccumulatedSCHessianSSE::addPoint + stitchDoubleInternal,
//   .../AccumulatedSCHessian.cc:9-119): with the lifted rows g_p produced by the linearize kernel the
//   whole F^3-block stitch collapses into one symmetric rank-P update  M = sum_p HdiF_p g_p g_p^T,
//   computed here with exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32).  This is the only MFMA use on the path ("the final
//   small dense Hessian accumulate").  Fast path: one block per (16x16 tile, K-split), results added with fp64 atomics straight
//   into HFinal / bFinal (B.acc); step-wise path: LD_SC_SPLITS K-range blocks write partial matrices for k_gather.
// Also here: k_gather (assembly of H_A,b_A,H_L,b_L,H_sc,b_sc,HFinal,bFinal for the step-wise path and the old all-reduce layout),
// k_marg_update / k_marg_frame (EnergyFunctional::marginalizePointsF tail, marginalizeFrame), k_acc_init.
#include <hip/hip_runtime.h>
#include "ba_dev.h"

#include "ba_reduce_body.h"

__global__ __launch_bounds__(256) void k_reduce(BaPtrs B, BaDims D, ResSet S, ChunkStarts chunkStart, int hasL, int GSP, int atomicMode,
                                                int hasPrior, float calibPrior, double l1, double il, int itCheck) {
    static_assert(sizeof(BaPtrs) + sizeof(BaDims) + sizeof(ResSet) + sizeof(ChunkStarts) >= 8 * 64 - 60, "k_reduce: ld_touch_kernarg<8> must stay inside the arguments");
    ld_touch_kernarg<8>();           // 620 bytes of arguments: the first 512 into the scalar cache with one wait (ba_dev.h): 9.24 -> 9.03 us at C3
    reduce_body(B, D, S, chunkStart, hasL, GSP, atomicMode, hasPrior, calibPrior, l1, il, itCheck, (int) blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// k_gather — one thread per entry of the stitched systems (n*n + n entries): sums the pair contributions
// (fixed order) and the Schur K-split partials, adds priors, and assembles HFinal / bFinal of
// EnergyFunctional::solveSystemF (EnergyFunctional.cc:257-291, default solver mode):
//   HFinal = (H_L + H_M + H_A) with diag*(1+lambda) - H_sc/(1+lambda);  bFinal = b_L + (b_M + H_M delta) + b_A - b_sc.
// Reference: AccumulatedTopHessian.h:64-105 (symmetrisation), AccumulatedSCHessian.h:64-98.
// mode 0: single GPU;  mode 1: export rank-local sums into the all-reduce buffer (no HFinal);
// mode 2: import all-reduced sums, then assemble HFinal/bFinal.
// sys layout: HA n*n | bA n | HL | bL | Hsc | bsc | HFinal | bFinal.
// ---------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ int g_col(int i, int FS) { return (i < 4) ? (8 * FS + i) : (i - 4); }

// sum over x = 0..F-1 of PC[(base + x*stride) * LD_PAIRC + off], fixed order, all loads issued before the adds
static __device__ __forceinline__ double sum_f(const double *PC, int F, int base, int stride, int off, double init = 0.0) {
    double q[LD_MAXF];
#pragma unroll
    for (int x = 0; x < LD_MAXF; x++) q[x] = (x < F) ? PC[(size_t) (base + x * stride) * LD_PAIRC + off] : 0.0;
    double v = init;
#pragma unroll
    for (int x = 0; x < LD_MAXF; x++) if (x < F) v += q[x];
    return v;
}
// sum over all F*F pairs, 16 loads in flight
static __device__ __forceinline__ double sum_pairs(const double *PC, int F, int off) {
    double v = 0;
    for (int p0 = 0; p0 < F * F; p0 += 16) {
        double q[16];
#pragma unroll
        for (int x = 0; x < 16; x++) q[x] = (p0 + x < F * F) ? PC[(size_t) (p0 + x) * LD_PAIRC + off] : 0.0;
#pragma unroll
        for (int x = 0; x < 16; x++) if (p0 + x < F * F) v += q[x];
    }
    return v;
}

static __device__ __forceinline__ double top_entry(const double *PC, int F, int i, int j) {
    // i, j in reference ordering [calib 4 | frames 8F]; j == -1 -> b entry
    if (j < 0) {
        if (i < 4) return sum_pairs(PC, F, 288 + i);
        int f = (i - 4) >> 3, a = (i - 4) & 7;
        double v = sum_f(PC, F, f * F, 1, 272 + a);
        return sum_f(PC, F, f, F, 280 + a, v);          // host part first, then target part
    }
    if (i < 4 && j < 4) return sum_pairs(PC, F, 256 + i * 4 + j);
    if (i < 4 || j < 4) {
        int fi = (i < 4) ? j : i, c = (i < 4) ? i : j;
        int f = (fi - 4) >> 3, a = (fi - 4) & 7;
        double v = sum_f(PC, F, f * F, 1, 192 + a * 4 + c);
        return sum_f(PC, F, f, F, 224 + a * 4 + c, v);
    }
    int f = (i - 4) >> 3, a = (i - 4) & 7, g = (j - 4) >> 3, c = (j - 4) & 7;
    if (f == g) {
        double v = sum_f(PC, F, f * F, 1, a * 8 + c);
        return sum_f(PC, F, f, F, 64 + a * 8 + c, v);
    }
    if (f < g) return PC[(size_t) (f * F + g) * LD_PAIRC + 128 + a * 8 + c] + PC[(size_t) (g * F + f) * LD_PAIRC + 128 + c * 8 + a];
    return PC[(size_t) (g * F + f) * LD_PAIRC + 128 + c * 8 + a] + PC[(size_t) (f * F + g) * LD_PAIRC + 128 + a * 8 + c];
}

__global__ __launch_bounds__(256) void k_gather(BaPtrs B, BaDims D, ResSet S, int hasL, int hasPrior, int GSP, double lambdaIn, int solverMode,
                                                float calibPrior, int mode, double *rbuf) {
    const int F = D.F, n = D.n, FS = D.FS;
    const int N = n * n + n;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double lambda = lambdaIn;
    if (solverMode & LDSO_SOLVER_USE_GN) lambda = 0;
    if (solverMode & LDSO_SOLVER_FIX_LAMBDA) lambda = 1e-5;
    const size_t blk = (size_t) n * n + n;
    double *HA = B.sys, *bA = HA + n * n, *HL = bA + n, *bL = HL + n * n, *Hsc = bL + n, *bsc = Hsc + n * n, *HF = bsc + n, *bF = HF + n * n;
    if (mode == 1 && e >= N && e < N + D.P) {
        // newest-frame energy candidates of the local points: value+1, zero elsewhere (see post_thresh)
        int i = e - N;
        double v = 0.0;
        if (i >= D.pBegin && i < D.pEnd) { float c = S.candE[i]; if (c >= 0.0f) v = (double) c + 1.0; }
        rbuf[3 * blk + 8 + i] = v;
    }
    if (e >= N) return;
    const int i = (e < n * n) ? e / n : (e - n * n), j = (e < n * n) ? e % n : -1;
    double prH = 0, prb = 0;      // prior contributions (L pass only, AccumulatedTopHessian.cc:246-254)
    if (j < 0) prb = (i < 4) ? (double) calibPrior * (double) B.calib->cDeltaF[i] : B.frames[(i - 4) >> 3].prior[(i - 4) & 7] * B.frames[(i - 4) >> 3].delta_prior[(i - 4) & 7];
    else if (i == j) prH = (i < 4) ? (double) calibPrior : B.frames[(i - 4) >> 3].prior[(i - 4) & 7];

    double vA, vL, vS;
    if (mode == 2) {
        vA = rbuf[e]; vL = rbuf[blk + e]; vS = rbuf[2 * blk + e];
    } else {
        vA = top_entry(B.pairC, F, i, j);
        vL = hasL ? top_entry(B.pairC + (size_t) F * F * LD_PAIRC, F, i, j) : 0.0;
        // Schur: only upper-triangular 16x16 tiles are stored; a diagonal tile holds both triangles
        int ci = g_col(i, FS), cj = (j >= 0) ? g_col(j, FS) : (8 * FS + 4);
        int r = min(ci, cj), c = max(ci, cj);
        size_t off = (r / 16 == c / 16) ? ((size_t) ci * GSP + cj) : ((size_t) r * GSP + c);
        float q[LD_SC_SPLITS];
#pragma unroll
        for (int sp = 0; sp < LD_SC_SPLITS; sp++) q[sp] = B.scPart[(size_t) sp * GSP * GSP + off];
        vS = 0;
#pragma unroll
        for (int sp = 0; sp < LD_SC_SPLITS; sp++) vS += (double) q[sp];
    }
    if (mode == 1) { rbuf[e] = vA; rbuf[blk + e] = vL; rbuf[2 * blk + e] = vS; return; }
    if (j >= 0) { HA[e] = vA; HL[e] = vL + prH; Hsc[e] = vS; }
    else { bA[i] = vA; bL[i] = vL + prb; bsc[i] = vS; }
    // HFinal / bFinal
    if (j >= 0) {
        double v = (vL + prH) + (hasPrior ? B.HM[e] : 0.0) + vA;
        if (i == j) v *= (1 + lambda);
        v -= vS * (double) (1.0f / (1 + lambda));
        HF[e] = v;
    } else {
        double s = 0;
        if (hasPrior) {
            s = B.bM[i];
            for (int jj = 0; jj < n; jj++) {
                double dj = (jj < 4) ? (double) B.calib->cDeltaF[jj] : B.frames[(jj - 4) >> 3].delta[(jj - 4) & 7];
                s += B.HM[(size_t) i * n + jj] * dj;
            }
        }
        bF[i] = (vL + prb) + s + vA - vS;
    }
}

hipError_t ba_launch_gather(const BaPtrs &B, const BaDims &D, const ResSet &S, bool hasL, bool hasPrior, int GSP, double lambda,
                            const ldso_settings_t &St, int mode, double *rbuf, hipStream_t st) {
    int N = D.n * D.n + D.n + (mode == 1 ? D.P : 0);
    hipLaunchKernelGGL(k_gather, dim3((N + 255) / 256), dim3(256), 0, st, B, D, S, hasL ? 1 : 0, hasPrior ? 1 : 0, GSP, lambda, St.solverMode,
                       St.initialCalibHessian, mode, rbuf);
    return hipGetLastError();
}

hipError_t ba_launch_reduce(const BaPtrs &B, const BaDims &D, const ResSet &S, const ChunkStarts &chunkStart, bool hasL, int GSP, int atomicMode, bool hasPrior,
                            float calibPrior, double l1, double il, int itCheck, hipStream_t st) {
    const int nT = GSP / 16;
    int nb = D.F * D.F * (hasL ? 2 : 1) + (atomicMode ? D.ks * nT * (nT + 1) / 2 + 1 : LD_SC_SPLITS);
    size_t lds = atomicMode ? (size_t) (2 * SCT_SLAB * 16 + SCT_SLAB) * sizeof(float) : (size_t) (SC_SLAB * GSP) * sizeof(float);
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *) k_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(k_reduce, dim3(nb), dim3(256), lds, st, B, D, S, chunkStart, hasL ? 1 : 0, GSP, atomicMode, hasPrior ? 1 : 0, calibPrior, l1, il, itCheck);
    return hipGetLastError();
}

// EnergyFunctional::marginalizePointsF tail (EnergyFunctional.cc:203-216): HM += margWeightFac (M - Msc), bM += margWeightFac (Mb - Mbsc),
// with M / Msc the top / Schur systems of the marginalised points assembled by k_gather (mode 0) in B.sys
__global__ __launch_bounds__(256) void k_marg_update(BaPtrs B, BaDims D, double w) {
    const int n = D.n, e = blockIdx.x * blockDim.x + threadIdx.x;
    const double *HA = B.sys, *bA = HA + n * n, *Hsc = bA + n + (size_t) n * n + n, *bsc = Hsc + n * n;
    if (e < n * n) B.HM[e] += (HA[e] - Hsc[e]) * w;
    else if (e < n * n + n) B.bM[e - n * n] += (bA[e - n * n] - bsc[e - n * n]) * w;
}

hipError_t ba_launch_marg_update(const BaPtrs &B, const BaDims &D, double w, hipStream_t st) {
    const int N = D.n * D.n + D.n;
    hipLaunchKernelGGL(k_marg_update, dim3((N + 255) / 256), dim3(256), 0, st, B, D, w);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// EnergyFunctional::marginalizeFrame (EnergyFunctional.cc:72-151) on the device prior H_M / b_M: move the frame's 8 rows /
// columns to the end, add its prior, scale by (|diag|+10)^-1/2, eliminate the 8x8 block (inverse by partial-pivot LU like Eigen's
// fixed-size inverse()), unscale, symmetrise.  One workgroup; the matrices live in global memory (n <= 132), W = n*n + n doubles
// of scratch.  Output: (n-8)^2 row-major + (n-8).  Per key frame, not per iteration: written for clarity, not speed.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_marg_frame(BaPtrs B, BaDims D, int idx, double *W, double *outH, double *outb) {
    const int tid = threadIdx.x, n = D.n, nd = n - 8;
    double *Hs = W, *bs = W + (size_t) n * n;
    __shared__ double sSV[8 * LD_MAXF + 4], sHpi[64], sLU[64], sBLI[(8 * LD_MAXF + 4) * 8];
    __shared__ int sPiv[8];
    auto perm = [&](int i) { const int io = 4 + 8 * idx; return (i < io) ? i : (i < nd) ? i + 8 : io + (i - nd); };
    // permuted copy + the frame's prior on its (now trailing) diagonal block
    const DevFrame &fh = B.frames[idx];
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        double v = B.HM[(size_t) perm(i) * n + perm(j)];
        if (i == j && i >= nd) v += fh.prior[i - nd];
        Hs[e] = v;
    }
    for (int i = tid; i < n; i += 256) bs[i] = B.bM[perm(i)] + ((i >= nd) ? fh.prior[i - nd] * fh.delta_prior[i - nd] : 0.0);
    __syncthreads();
    for (int i = tid; i < n; i += 256) sSV[i] = sqrt(fabs(Hs[(size_t) i * n + i]) + 10.0);
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e % n; Hs[e] = (1.0 / sSV[i]) * Hs[e] * (1.0 / sSV[j]); }
    for (int i = tid; i < n; i += 256) bs[i] = (1.0 / sSV[i]) * bs[i];
    __syncthreads();
    // hpi = inverse(0.5 (hpi + hpi^T)) by LU with partial pivoting (thread 0, LDS), then symmetrised again
    if (tid < 64) { const int r = tid >> 3, c = tid & 7; sLU[tid] = 0.5 * (Hs[(size_t) (nd + r) * n + nd + c] + Hs[(size_t) (nd + c) * n + nd + r]); }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 8; k++) {
            int p = k; double best = fabs(sLU[k * 8 + k]);
            for (int i = k + 1; i < 8; i++) if (fabs(sLU[i * 8 + k]) > best) { best = fabs(sLU[i * 8 + k]); p = i; }
            sPiv[k] = p;
            if (p != k) for (int j = 0; j < 8; j++) { double t = sLU[k * 8 + j]; sLU[k * 8 + j] = sLU[p * 8 + j]; sLU[p * 8 + j] = t; }
            for (int i = k + 1; i < 8; i++) {
                sLU[i * 8 + k] /= sLU[k * 8 + k];
                for (int j = k + 1; j < 8; j++) sLU[i * 8 + j] -= sLU[i * 8 + k] * sLU[k * 8 + j];
            }
        }
    }
    __syncthreads();
    if (tid < 8) {      // column tid of the inverse: solve LU x = P e_tid
        double x[8];
        for (int i = 0; i < 8; i++) x[i] = (i == tid) ? 1.0 : 0.0;
        for (int k = 0; k < 8; k++) { const int p = sPiv[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
        for (int i = 0; i < 8; i++) for (int j = 0; j < i; j++) x[i] -= sLU[i * 8 + j] * x[j];
        for (int i = 7; i >= 0; i--) { for (int j = i + 1; j < 8; j++) x[i] -= sLU[i * 8 + j] * x[j]; x[i] /= sLU[i * 8 + i]; }
        for (int i = 0; i < 8; i++) sHpi[i * 8 + tid] = x[i];
    }
    __syncthreads();
    double hsym = 0.0;
    if (tid < 64) { const int r = tid >> 3, c = tid & 7; hsym = 0.5 * (sHpi[r * 8 + c] + sHpi[c * 8 + r]); }
    __syncthreads();
    if (tid < 64) sHpi[tid] = hsym;
    __syncthreads();
    // Schur complement of the trailing block: row i of the result needs bli[i][:] = sum_k Hs[nd+k][i] hpi[k][:]
    // (the trailing rows nd.. are only read: all rows of bli first, then every entry of the result on its own thread - the same sums in
    // the same order as the row-by-row loop, without its 2 nd barriers)
    for (int e = tid; e < nd * 8; e += 256) {
        const int i = e >> 3, c = e & 7;
        double s_ = 0;
        for (int k = 0; k < 8; k++) s_ += Hs[(size_t) (nd + k) * n + i] * sHpi[k * 8 + c];
        sBLI[e] = s_;
    }
    __syncthreads();
    for (int e = tid; e < nd * (nd + 1); e += 256) {
        const int i = e / (nd + 1), j = e % (nd + 1);
        const double *bl = sBLI + i * 8;
        double s_ = 0;
        if (j < nd) { for (int k = 0; k < 8; k++) s_ += bl[k] * Hs[(size_t) (nd + k) * n + j]; Hs[(size_t) i * n + j] -= s_; }
        else { for (int k = 0; k < 8; k++) s_ += bl[k] * bs[nd + k]; bs[i] -= s_; }
    }
    __syncthreads();
    // unscale and symmetrise
    for (int e = tid; e < nd * nd; e += 256) {
        const int i = e / nd, j = e % nd;
        outH[e] = 0.5 * (sSV[i] * Hs[(size_t) i * n + j] * sSV[j] + sSV[j] * Hs[(size_t) j * n + i] * sSV[i]);
    }
    for (int i = tid; i < nd; i += 256) outb[i] = sSV[i] * bs[i];
}

hipError_t ba_launch_marg_frame(const BaPtrs &B, const BaDims &D, int idx, double *work, double *outH, double *outb, hipStream_t st) {
    hipLaunchKernelGGL(k_marg_frame, dim3(1), dim3(256), 0, st, B, D, idx, work, outH, outb);
    return hipGetLastError();
}

// initialisation of the HFinal / bFinal accumulator outside k_linearize (used when the accumulator is re-pointed at a
// caller's all-reduce buffer): mode 1 = H_M + diagonal priors (lower triangle), mode 2 = zeros
__global__ __launch_bounds__(256) void k_acc_init(BaPtrs B, BaDims D, GnInit gi) {
    const int n = D.n, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n + n) return;
    double v = 0.0;
    if (e < n * n && gi.enable == 1) {
        const int i = e / n, j = e % n;
        if (j <= i) {
            if (gi.hasPrior) v = B.HM[e];
            if (i == j) v += (i < 4) ? (double) gi.calibPrior : B.frames[(i - 4) >> 3].prior[(i - 4) & 7];
        }
    }
    B.acc[e] = v;
}

hipError_t ba_launch_acc_init(const BaPtrs &B, const BaDims &D, const GnInit &gi, hipStream_t st) {
    const int N = D.n * D.n + D.n;
    hipLaunchKernelGGL(k_acc_init, dim3((N + 255) / 256), dim3(256), 0, st, B, D, gi);
    return hipGetLastError();
}
