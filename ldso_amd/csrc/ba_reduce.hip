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

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SC_SLAB 128     // points staged per LDS slab (one slab per split at the C3 window: a single load level)
#define PA_SLICES 2     // Part A: interleaved slices of a pair's chunk range (2 x 91 threads)
#define PA_UNROLL 24    // Part A: partial loads in flight per thread
#define SCT_KS 4         // atomic mode: K-splits per Schur tile
#define SCT_SLAB 512    // atomic mode: points staged per LDS slab of a tile block
#define SC_MAXT 12      // tiles per wave: GSP=144 (FS=16) -> 45 upper tiles / 4 waves

__host__ __device__ constexpr int tri13r(int r, int c) { return r * 13 - (r * (r - 1)) / 2 + (c - r); }

// chunkStart[h]..chunkStart[h+1]: chunks of host h (chunks are host-major)
// atomicMode (GN fast path): instead of pairC / scPart the results are added (fp64 atomics) straight into the lower triangle of
// HFinal / bFinal (B.acc, initialised by k_linearize with the H_M / prior terms; EnergyFunctional.cc:257-291):
//   HFinal = (H_A + H_L + priors + H_M) with diag * (1+lambda) - H_sc / (1+lambda)   ->  top terms are scaled by l1 = 1+lambda on the
//   diagonal, Schur terms by -il = -1/(1+lambda);  bFinal = b_A + b_L + (prior delta + b_M + H_M delta) - b_sc.
static __device__ __forceinline__ void acc_add(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NT, int W> static __device__ __forceinline__ void schur_wave(const float *sG, int GSP, int cnt, int wcol, int li, int lk, f32x4 *acc);

__global__ __launch_bounds__(256) void k_reduce(BaPtrs B, BaDims D, ResSet S, ChunkStarts chunkStart, int hasL, int GSP, int atomicMode,
                                                int hasPrior, float calibPrior, double l1, double il, int itCheck) {
    if (LD_ITER_SKIPPED(B, itCheck)) return;
    const int F = D.F, FS = D.FS;
    const int nPairBlocks = F * F * (hasL ? 2 : 1);
    const int tid = threadIdx.x;
    __shared__ double sA[13 * 13];
    __shared__ double sT[2][64];
    const long long t0_ = wall_clock64();
#define RSTAMP(i) do { if (LD_STAMP_ON && tid == 0) B.energyLog[(i)] = (double) (wall_clock64() - t0_); } while (0)

    if ((int) blockIdx.x < nPairBlocks) {
        // ------------------------------- Part A ---------------------------------------------------------
        const int which = blockIdx.x / (F * F);     // 0 = A, 1 = L
        const int pair = blockIdx.x % (F * F);
        const int h = pair / F, t = pair % F;        // pairC index [h*F + t]
        const float *part = which ? S.topL : S.topA;
        const int c0 = chunkStart.v[h], c1 = chunkStart.v[h + 1];
        // adjoints of this pair -> LDS (issued together with the partial loads: one latency level)
        __shared__ double sAH[64], sAT[64];
        __shared__ double sPart[PA_SLICES][LD_TOPN];
        if (tid >= 128 && tid < 192) { sAH[tid - 128] = B.adHost[(size_t) (h + t * F) * 64 + (tid - 128)]; }
        if (tid >= 192) { sAT[tid - 192] = B.adTarget[(size_t) (h + t * F) * 64 + (tid - 192)]; }
        // the per-chunk partials of this pair: PA_SLICES interleaved slices of the chunk range, PA_UNROLL loads in flight
        const int slice = tid / LD_TOPN, ent = tid % LD_TOPN;
        if (slice < PA_SLICES) {
            double a = 0;
            for (int cb = c0 + slice; cb < c1; cb += PA_SLICES * PA_UNROLL) {
                float q[PA_UNROLL];
#pragma unroll
                for (int u = 0; u < PA_UNROLL; u++) { int c = cb + u * PA_SLICES; q[u] = (c < c1) ? part[((size_t) c * FS + t) * LD_TOPN + ent] : 0.0f; }
#pragma unroll
                for (int u = 0; u < PA_UNROLL; u++) a += (double) q[u];
            }
            sPart[slice][ent] = a;
        }
        __syncthreads();
        if (blockIdx.x == 0) RSTAMP(20);
        if (tid < LD_TOPN) {
            double a = sPart[0][tid];
#pragma unroll
            for (int sl = 1; sl < PA_SLICES; sl++) a += sPart[sl][tid];
            // unpack to symmetric 13x13
            int r = 0, rem = tid;
            while (rem >= 13 - r) { rem -= 13 - r; r++; }
            int cidx = r + rem;
            sA[r * 13 + cidx] = a;
            sA[cidx * 13 + r] = a;
        }
        __syncthreads();
        double *out = B.pairC + ((size_t) which * F * F + pair) * LD_PAIRC;
        const double *AH = sAH, *AT = sAT;
        if (tid < 64) {
            int i = tid >> 3, j = tid & 7;
            double th = 0, tt = 0;
            for (int m = 0; m < 8; m++) { th += AH[i * 8 + m] * sA[(4 + m) * 13 + 4 + j]; tt += AT[i * 8 + m] * sA[(4 + m) * 13 + 4 + j]; }
            sT[0][tid] = th; sT[1][tid] = tt;
        }
        __syncthreads();
        const int n = D.n;
        double *accT = B.acc, *accb = B.acc + (size_t) n * n;
        const int rh = 4 + 8 * h, rt = 4 + 8 * t;      // first rows of the two frames in the reference ordering
        if (tid < 64) {
            int i = tid >> 3, j = tid & 7;
            double hh = 0, tt = 0, ht = 0;
            for (int m = 0; m < 8; m++) { hh += sT[0][i * 8 + m] * AH[j * 8 + m]; tt += sT[1][i * 8 + m] * AT[j * 8 + m]; ht += sT[0][i * 8 + m] * AT[j * 8 + m]; }
            if (!atomicMode) { out[tid] = hh; out[64 + tid] = tt; out[128 + tid] = ht; }
            else {
                if (i >= j) { const double dsc = (i == j) ? l1 : 1.0; acc_add(&accT[(size_t) (rh + i) * n + rh + j], hh * dsc); acc_add(&accT[(size_t) (rt + i) * n + rt + j], tt * dsc); }
                if (h > t) acc_add(&accT[(size_t) (rh + i) * n + rt + j], ht);
                else if (h < t) acc_add(&accT[(size_t) (rt + j) * n + rh + i], ht);
            }
        } else if (tid < 64 + 32) {
            int e = tid - 64, i = e >> 2, c = e & 3;
            double hc = 0, tc = 0;
            for (int m = 0; m < 8; m++) { hc += AH[i * 8 + m] * sA[(4 + m) * 13 + c]; tc += AT[i * 8 + m] * sA[(4 + m) * 13 + c]; }
            if (!atomicMode) { out[192 + e] = hc; out[224 + e] = tc; }
            else { acc_add(&accT[(size_t) (rh + i) * n + c], hc); acc_add(&accT[(size_t) (rt + i) * n + c], tc); }
        } else if (tid < 96 + 16) {
            int e = tid - 96;
            if (!atomicMode) out[256 + e] = sA[(e >> 2) * 13 + (e & 3)];
            else if ((e >> 2) >= (e & 3)) acc_add(&accT[(size_t) (e >> 2) * n + (e & 3)], sA[(e >> 2) * 13 + (e & 3)] * (((e >> 2) == (e & 3)) ? l1 : 1.0));
        } else if (tid < 112 + 8) {
            int i = tid - 112;
            double bh = 0, bt = 0;
            for (int m = 0; m < 8; m++) { bh += AH[i * 8 + m] * sA[(4 + m) * 13 + 12]; bt += AT[i * 8 + m] * sA[(4 + m) * 13 + 12]; }
            if (!atomicMode) { out[272 + i] = bh; out[280 + i] = bt; }
            else { acc_add(&accb[rh + i], bh); acc_add(&accb[rt + i], bt); }
        } else if (tid < 120 + 4) {
            int i = tid - 120;
            if (!atomicMode) out[288 + i] = sA[i * 13 + 12];
            else acc_add(&accb[i], sA[i * 13 + 12]);
        }
        if (blockIdx.x == 0) RSTAMP(21);
        return;
    }

    const int nSplitBase = nPairBlocks;
    if (atomicMode && (int) blockIdx.x < nSplitBase + SCT_KS * (GSP / 16) * (GSP / 16 + 1) / 2) {
        // ------------------------------- Part B, atomic mode: one block per (16x16 tile, K-split) ---------------------
        // The block stages the two 16-column blocks of its G rows (and the weights HdiF) in LDS with 16-byte loads, its four
        // waves interleave the k-steps of v_mfma_f32_16x16x4_f32, the four partial tiles are summed through LDS and each
        // thread adds ONE element (scaled by -1/(1+lambda)) into HFinal / bFinal: SCT_KS-way contention per address.
        extern __shared__ __attribute__((aligned(16))) float sT_[];
        float *sAc = sT_, *sBc = sAc + SCT_SLAB * 16, *sWc = sBc + SCT_SLAB * 16;      // [SLAB][16], [SLAB][16], [SLAB]
        const int bb = blockIdx.x - nSplitBase, tile = bb / SCT_KS, ks = bb % SCT_KS;
        const int nT = GSP / 16, GS = D.GS, n = D.n;
        int ti = 0, rem = tile;
        while (rem >= nT - ti) { rem -= nT - ti; ti++; }
        const int tj = ti + rem;
        const int P0 = D.pBegin, Pn = D.pEnd - D.pBegin, per = ((Pn + SCT_KS - 1) / SCT_KS + 3) & ~3;
        const int pa = P0 + ks * per, pb = min(P0 + Pn, pa + per);
        const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int wcol = 8 * FS + 5;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int base = pa; base < pb; base += SCT_SLAB) {
            const int cnt = min(SCT_SLAB, pb - base), rows = (cnt + 3) & ~3;
            __syncthreads();
            {
                // float4 e of the slab: row r = e / 8, half = (e / 4) & 1 (A or B column block), c4 = e & 3
                float4 q[16];
                float wq[2];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int e = tid + u * 256, r = e >> 3, half = (e >> 2) & 1, c4 = e & 3;
                    q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int col = (half ? tj : ti) * 16 + c4 * 4;
                    if (r < cnt && col < GS) q[u] = *(const float4 *) (S.G + (size_t) (base + r) * GS + col);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) { const int r = tid + u * 256; wq[u] = (r < cnt) ? S.G[(size_t) (base + r) * GS + wcol] : 0.f; }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int e = tid + u * 256, r = e >> 3, half = (e >> 2) & 1, c4 = e & 3;
                    if (r < rows) *(float4 *) ((half ? sBc : sAc) + r * 16 + c4 * 4) = q[u];
                }
#pragma unroll
                for (int u = 0; u < 2; u++) { const int r = tid + u * 256; if (r < rows) sWc[r] = wq[u]; }
            }
            __syncthreads();
#pragma unroll 4
            for (int k0 = wave * 4; k0 < rows; k0 += 16) {
                const float a = sAc[(k0 + lk) * 16 + li] * sWc[k0 + lk];
                const float b = sBc[(k0 + lk) * 16 + li];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        }
        __syncthreads();
        float *sR = sT_;      // [4 waves][256]
#pragma unroll
        for (int r = 0; r < 4; r++) sR[wave * 256 + (lk * 4 + r) * 16 + li] = acc[r];
        __syncthreads();
        {
            const float v = ((sR[tid] + sR[256 + tid]) + sR[512 + tid]) + sR[768 + tid];
            const int rr = ti * 16 + (tid >> 4), cc = tj * 16 + (tid & 15);
            const int F8 = 8 * D.F, FS8 = 8 * FS;
            // G column -> index in the reference ordering [calib 4 | frames 8F]; n = right-hand side; -1 = padding
            const int I = (rr < F8) ? 4 + rr : (rr >= FS8 && rr < FS8 + 4) ? rr - FS8 : (rr == FS8 + 4) ? n : -1;
            const int J = (cc < F8) ? 4 + cc : (cc >= FS8 && cc < FS8 + 4) ? cc - FS8 : (cc == FS8 + 4) ? n : -1;
            const bool use = !(ti == tj && rr > cc) && I >= 0 && J >= 0 && I != n;
            if (use) {
                if (J == n) acc_add(&B.acc[(size_t) n * n + I], -(double) v);
                else acc_add(&B.acc[(size_t) max(I, J) * n + min(I, J)], -(double) v * il);
            }
        }
        return;
    }
    if (atomicMode == 2) return;      // ranks > 0 of a sharded window: the prior / lambda terms are added once, by rank 0
    if (atomicMode) {
        // ------------------------------- extras of the GN fast path --------------------------------------------
        // bExtra = prior * delta_prior + (bM + HM delta), priorDiag   (AccumulatedTopHessian.cc:246-254, EnergyFunctional.cc:279)
        const int n = D.n;
        __shared__ double sDelta[8 * LD_MAXF + 4];
        if (tid < n) sDelta[tid] = (tid < 4) ? (double) B.calib->cDeltaF[tid] : B.frames[(tid - 4) >> 3].delta[(tid - 4) & 7];
        double prb = 0, prH = 0, bm = 0;
        if (tid < n) {
            if (tid < 4) { prH = (double) calibPrior; prb = (double) calibPrior * (double) B.calib->cDeltaF[tid]; }
            else { const DevFrame &f = B.frames[(tid - 4) >> 3]; prH = f.prior[(tid - 4) & 7]; prb = prH * f.delta_prior[(tid - 4) & 7]; }
            if (hasPrior) bm = B.bM[tid];
        }
        __syncthreads();
        if (tid < n) {
            double s_ = 0;
            if (hasPrior) {
                s_ = bm;
                for (int j0 = 0; j0 < n; j0 += 16) {
                    double q[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) q[u] = (j0 + u < n) ? B.HM[(size_t) tid * n + j0 + u] : 0.0;
#pragma unroll
                    for (int u = 0; u < 16; u++) if (j0 + u < n) s_ += q[u] * sDelta[j0 + u];
                }
            }
            acc_add(&B.acc[(size_t) n * n + tid], prb + s_);
            // lambda scaling of the diagonal terms k_linearize put there: (H_M + prior)_ii * (l1 - 1)
            const double dterm = prH + (hasPrior ? B.HM[(size_t) tid * n + tid] : 0.0);
            acc_add(&B.acc[(size_t) tid * n + tid], dterm * (l1 - 1.0));
        }
        return;
    }

    // ----------------------------------- Part B -----------------------------------------------------------
    // M[sp] = sum over this split's points of  w_p * r_p r_p^T  with r_p = G row (GS entries, zero padded
    // to GSP = multiple of 16) and w_p = HdiF_p = r_p[8*FS+5].  Upper-triangular 16x16 tiles only.
    // G rows are staged through LDS in slabs of SC_SLAB points (coalesced row copies), then consumed by
    // v_mfma_f32_16x16x4_f32: A[i][k] = w_k r_k[ti*16+i], B[k][j] = r_k[tj*16+j].  Tile t (row-major over the upper
    // triangle) belongs to wave t % 4; the tile lists are compile-time (schur_wave<NT, W>).
    extern __shared__ __attribute__((aligned(16))) float sG[];      // [SC_SLAB][GSP]
    const int sp = blockIdx.x - nSplitBase;
    const int wave = tid >> 6, lane = tid & 63;
    const int nT = GSP / 16;
    const int P0 = D.pBegin, Pn = D.pEnd - D.pBegin;
    const int per = (Pn + LD_SC_SPLITS - 1) / LD_SC_SPLITS;
    const int pa = P0 + sp * per, pb = min(P0 + Pn, pa + per);
    const int GS = D.GS;
    const int li = lane & 15, lk = lane >> 4;
    float *Mout = B.scPart + (size_t) sp * GSP * GSP;
    f32x4 acc[SC_MAXT];
#pragma unroll
    for (int q = 0; q < SC_MAXT; q++) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int base = pa; base < pb; base += SC_SLAB) {
        const int cnt = min(SC_SLAB, pb - base);
        __syncthreads();
        {
            // coalesced 16-byte row copies (GS is a multiple of 8 floats); the pad columns GS..GSP-1 and the unused rows are zeroed
            const int g4 = GS >> 2, p4 = GSP >> 2;
            const int rows = (cnt + 3) & ~3;          // the MFMA loop consumes rows in groups of 4
            for (int e0 = tid; e0 < rows * p4; e0 += 256 * 8) {
                float4 q[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u * 256, r = e / p4, c = e % p4;
                    q[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < rows * p4 && r < cnt && c < g4) q[u] = ((const float4 *) (S.G + (size_t) (base + r) * GS))[c];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) { const int e = e0 + u * 256; if (e < rows * p4) ((float4 *) sG)[e] = q[u]; }
            }
        }
        __syncthreads();
        if (sp == 0) RSTAMP(22);
        const int wcol = 8 * FS + 5;
        if (nT == 5) {
            if (wave == 0) schur_wave<5, 0>(sG, GSP, cnt, wcol, li, lk, acc); else if (wave == 1) schur_wave<5, 1>(sG, GSP, cnt, wcol, li, lk, acc);
            else if (wave == 2) schur_wave<5, 2>(sG, GSP, cnt, wcol, li, lk, acc); else schur_wave<5, 3>(sG, GSP, cnt, wcol, li, lk, acc);
        } else {
            if (wave == 0) schur_wave<9, 0>(sG, GSP, cnt, wcol, li, lk, acc); else if (wave == 1) schur_wave<9, 1>(sG, GSP, cnt, wcol, li, lk, acc);
            else if (wave == 2) schur_wave<9, 2>(sG, GSP, cnt, wcol, li, lk, acc); else schur_wave<9, 3>(sG, GSP, cnt, wcol, li, lk, acc);
        }
    }
    if (sp == 0) RSTAMP(23);
    // C/D layout: col = lane&15, row = (lane>>4)*4 + r.  Tile list again (runtime walk, once).
    {
        const int n = D.n, F8 = 8 * D.F, FS8 = 8 * FS;
        double *accS = B.acc, *accSb = accS + (size_t) n * n;
        const double *unused_ = nullptr; (void) unused_;
        for (int pass = 0; pass < 1; pass++) {
            int tileIdx = 0;
            for (int ti = 0; ti < nT; ti++)
                for (int tj = ti; tj < nT; tj++, tileIdx++) {
                    if ((tileIdx & 3) != wave) continue;
                    const int q = tileIdx >> 2;
                    f32x4 a4 = acc[0];
#pragma unroll
                    for (int u = 1; u < SC_MAXT; u++) if (u == q) a4 = acc[u];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int rr = ti * 16 + lk * 4 + r, cc = tj * 16 + li;
                        if (!atomicMode) { Mout[(size_t) rr * GSP + cc] = a4[r]; continue; }
                        if (ti == tj && rr > cc) continue;
                        // G column -> index in the reference ordering [calib 4 | frames 8F]; n = right-hand side; -1 = padding
                        const int I = (rr < F8) ? 4 + rr : (rr >= FS8 && rr < FS8 + 4) ? rr - FS8 : (rr == FS8 + 4) ? n : -1;
                        const int J = (cc < F8) ? 4 + cc : (cc >= FS8 && cc < FS8 + 4) ? cc - FS8 : (cc == FS8 + 4) ? n : -1;
                        if (I < 0 || J < 0 || I == n) continue;
                        if (J == n) acc_add(&accSb[I], -(double) a4[r]);
                        else acc_add(&accS[(size_t) max(I, J) * n + min(I, J)], -(double) a4[r] * il);
                    }
                }
        }
    }
    if (sp == 0) RSTAMP(24);
}

// one wave's share of the rank-cnt update: tiles t of the upper triangle with t % 4 == W accumulate into acc[t / 4]
template <int NT, int W>
static __device__ __forceinline__ void schur_wave(const float *sG, int GSP, int cnt, int wcol, int li, int lk, f32x4 *acc) {
#pragma unroll 2
    for (int k0 = 0; k0 < cnt; k0 += 4) {
        const float *row = sG + (k0 + lk) * GSP;
        float val[NT];
#pragma unroll
        for (int c = 0; c < NT; c++) val[c] = row[c * 16 + li];
        const float w = row[wcol];
        int idx = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = ti; tj < NT; tj++, idx++)
                if ((idx & 3) == W) acc[idx >> 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(val[ti] * w, val[tj], acc[idx >> 2], 0, 0, 0);
    }
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
    int nb = D.F * D.F * (hasL ? 2 : 1) + (atomicMode ? SCT_KS * nT * (nT + 1) / 2 + 1 : LD_SC_SPLITS);
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
    __shared__ double sSV[8 * LD_MAXF + 4], sHpi[64], sLU[64], sBli[8];
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
    for (int i = 0; i < nd; i++) {
        if (tid < 8) { double s_ = 0; for (int k = 0; k < 8; k++) s_ += Hs[(size_t) (nd + k) * n + i] * sHpi[k * 8 + tid]; sBli[tid] = s_; }
        __syncthreads();
        for (int j = tid; j <= nd; j += 256) {
            double s_ = 0;
            if (j < nd) { for (int k = 0; k < 8; k++) s_ += sBli[k] * Hs[(size_t) (nd + k) * n + j]; Hs[(size_t) i * n + j] -= s_; }
            else { for (int k = 0; k < 8; k++) s_ += sBli[k] * bs[nd + k]; bs[i] -= s_; }
        }
        __syncthreads();
    }
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
