// ba_reduce_body.h — the body of k_reduce as a device function, shared by the stand-alone kernel (ba_reduce.hip) and the fused
// reduce + control-step kernel of the single-GPU GN fast path (k_reduce_solve, ba_solve.hip).  See ba_reduce.hip for the algorithm.
#pragma once
#include <hip/hip_runtime.h>
#include "ba_dev.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SC_SLAB 128     // points staged per LDS slab (one slab per split at the C3 window: a single load level)
#define PA_SLICES 2     // Part A: interleaved slices of a pair's chunk range (2 x 91 threads)
#define PA_UNROLL 24    // Part A: partial loads in flight per thread
#define SCT_KS LD_SCT_KS // atomic mode: K-splits per Schur tile
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

static __device__ __forceinline__ void reduce_body(const BaPtrs &B, const BaDims &D, const ResSet &S, const ChunkStarts &chunkStart, int hasL, int GSP, int atomicMode,
                                   int hasPrior, float calibPrior, double l1, double il, int itCheck, const int bid) {
    if (LD_ITER_SKIPPED(B, itCheck)) return;
    const int F = D.F, FS = D.FS;
    const int nPairBlocks = F * F * (hasL ? 2 : 1);
    const int tid = threadIdx.x;
    __shared__ double sA[13 * 13];
    __shared__ double sT[2][64];
    const long long t0_ = wall_clock64();
#define RSTAMP(i) do { if (LD_STAMP_ON && tid == 0) B.energyLog[(i)] = (double) (wall_clock64() - t0_); } while (0)

    if (bid < nPairBlocks) {
        // ------------------------------- Part A ---------------------------------------------------------
        const int which = bid / (F * F);     // 0 = A, 1 = L
        const int pair = bid % (F * F);
        const int h = pair / F, t = pair % F;        // pairC index [h*F + t]
        const float *part = which ? S.topL : S.topA;
        const int c0 = chunkStart.v[h], c1 = chunkStart.v[h + 1];
        if (bid == 0 && c1 > c0) RSTAMP(22);
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
                for (int u = 0; u < PA_UNROLL; u++) {
                    // branch-free: clamp the chunk index, mask the value (a branch per load breaks the sequential instruction
                    // prefetch right at the cold start of the kernel)
                    const int c = cb + u * PA_SLICES, cc = max(min(c, c1 - 1), 0);
                    const float v = part[((unsigned) cc * (unsigned) FS + (unsigned) t) * LD_TOPN + (unsigned) ent];
                    q[u] = (c < c1) ? v : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < PA_UNROLL; u++) a += (double) q[u];
            }
            sPart[slice][ent] = a;
            if (bid == 0 && a != 12345.678) RSTAMP(23);      // (stamps builds) the comparison makes the stamp wait for the loaded partials
        }
        __syncthreads();
        if (bid == 0) RSTAMP(20);
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
        if (bid == 0) RSTAMP(21);
        return;
    }

    const int nSplitBase = nPairBlocks;
    const int KS = D.ks;          // K-splits per Schur tile: a field of the window's dimensions since round 6 (a batch runs its windows with fewer)
    if (atomicMode && bid < nSplitBase + KS * (GSP / 16) * (GSP / 16 + 1) / 2) {
        // ------------------------------- Part B, atomic mode: one block per (16x16 tile, K-split) ---------------------
        // The block stages the two 16-column blocks of its G rows (and the weights HdiF) in LDS with 16-byte loads, its four
        // waves interleave the k-steps of v_mfma_f32_16x16x4_f32, the four partial tiles are summed through LDS and each
        // thread adds ONE element (scaled by -1/(1+lambda)) into HFinal / bFinal: KS-way contention per address.
        extern __shared__ __attribute__((aligned(16))) float sT_[];
        float *sAc = sT_, *sBc = sAc + SCT_SLAB * 16, *sWc = sBc + SCT_SLAB * 16;      // [SLAB][16], [SLAB][16], [SLAB]
        const int bb = bid - nSplitBase, tile = bb / KS, ks = bb % KS;
        const int nT = GSP / 16, GS = D.GS, n = D.n;
        int ti = 0, rem = tile;
        while (rem >= nT - ti) { rem -= nT - ti; ti++; }
        const int tj = ti + rem;
        const int P0 = D.pBegin, Pn = D.pEnd - D.pBegin, per = ((Pn + KS - 1) / KS + 3) & ~3;
        const int pa = P0 + ks * per, pb = min(P0 + Pn, pa + per);
        const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
        const int wcol = 8 * FS + 5;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc2 = acc;
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
                    const int col = (half ? tj : ti) * 16 + c4 * 4;
                    // branch-free (clamped address, masked value): straight-line code keeps the instruction prefetch going at the
                    // cold start of the kernel
                    const float4 v = *(const float4 *) (S.G + (unsigned) ((base + min(r, cnt - 1)) * GS + min(col, GS - 4)));
                    const bool in = r < cnt && col < GS;
                    q[u] = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) { const int r = tid + u * 256; const float v = S.G[(unsigned) ((base + min(r, cnt - 1)) * GS + wcol)]; wq[u] = (r < cnt) ? v : 0.f; }
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int e = tid + u * 256, r = e >> 3, half = (e >> 2) & 1, c4 = e & 3;
                    if (r < rows) *(float4 *) ((half ? sBc : sAc) + r * 16 + c4 * 4) = q[u];
                }
#pragma unroll
                for (int u = 0; u < 2; u++) { const int r = tid + u * 256; if (r < rows) sWc[r] = wq[u]; }
            }
            __syncthreads();
            if (bb == 0) RSTAMP(25);
            // two independent accumulation chains per wave (an MFMA depends on the previous one through its accumulator)
#pragma unroll 4
            for (int k0 = wave * 4; k0 < rows; k0 += 32) {
                const int k1 = k0 + 16;
                const float a0 = sAc[(k0 + lk) * 16 + li] * sWc[k0 + lk], b0 = sBc[(k0 + lk) * 16 + li];
                const bool in1 = k1 < rows;
                const int k1c = in1 ? k1 : k0;
                const float a1 = in1 ? sAc[(k1c + lk) * 16 + li] * sWc[k1c + lk] : 0.f, b1 = sBc[(k1c + lk) * 16 + li];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc2, 0, 0, 0);
            }
        }
        __syncthreads();
        if (bb == 0) RSTAMP(26);
        float *sR = sT_;      // [4 waves][256]
#pragma unroll
        for (int r = 0; r < 4; r++) sR[wave * 256 + (lk * 4 + r) * 16 + li] = acc[r] + acc2[r];
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
        if (bb == 0) RSTAMP(27);
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
        // H_M delta with the whole workgroup: 4 threads per row, a quarter of the row each, ALL loads of a thread issued at once (one memory latency; a thread
        // per row walked its row in four dependent batches - this workgroup signalled last in four iterations of five, 6.6 us after the launch), the four
        // partial sums of a row combined in the quad
        const int rowT = tid >> 2, quarter = tid & 3;
        constexpr int QW = 16;          // columns per quarter: n <= 64
        double hq[QW];
        const int per = (n + 3) >> 2, j0 = quarter * per;
        const bool rowOn = hasPrior && rowT < n && 4 * n <= 256;          // 4 threads per row: systems up to 64 x 64; larger ones take the loop below
#pragma unroll
        for (int u = 0; u < QW; u++) hq[u] = (rowOn && u < per && j0 + u < n) ? B.HM[(size_t) rowT * n + j0 + u] : 0.0;
        __syncthreads();
        double part = 0;
#pragma unroll
        for (int u = 0; u < QW; u++) if (u < per && j0 + u < n) part += hq[u] * sDelta[j0 + u];
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
        __shared__ double sHd[8 * LD_MAXF + 4];
        if (quarter == 0 && rowT < n) sHd[rowT] = part;
        __syncthreads();
        if (tid < n) {
            double s_ = 0;
            if (hasPrior) {
                s_ = bm;
                if (4 * n <= 256) s_ += sHd[tid];
                else for (int j0_ = 0; j0_ < n; j0_ += 16) {
                    double q[16];
#pragma unroll
                    for (int u = 0; u < 16; u++) q[u] = (j0_ + u < n) ? B.HM[(size_t) tid * n + j0_ + u] : 0.0;
#pragma unroll
                    for (int u = 0; u < 16; u++) if (j0_ + u < n) s_ += q[u] * sDelta[j0_ + u];
                }
            }
            acc_add(&B.acc[(size_t) n * n + tid], prb + s_);
            // lambda scaling of the diagonal terms k_linearize put there: (H_M + prior)_ii * (l1 - 1)
            const double dterm = prH + (hasPrior ? B.HM[(size_t) tid * n + tid] : 0.0);
            acc_add(&B.acc[(size_t) tid * n + tid], dterm * (l1 - 1.0));
        }
        RSTAMP(28);
        return;
    }

    // ----------------------------------- Part B -----------------------------------------------------------
    // M[sp] = sum over this split's points of  w_p * r_p r_p^T  with r_p = G row (GS entries, zero padded
    // to GSP = multiple of 16) and w_p = HdiF_p = r_p[8*FS+5].  Upper-triangular 16x16 tiles only.
    // G rows are staged through LDS in slabs of SC_SLAB points (coalesced row copies), then consumed by
    // v_mfma_f32_16x16x4_f32: A[i][k] = w_k r_k[ti*16+i], B[k][j] = r_k[tj*16+j].  Tile t (row-major over the upper
    // triangle) belongs to wave t % 4; the tile lists are compile-time (schur_wave<NT, W>).
    extern __shared__ __attribute__((aligned(16))) float sG[];      // [SC_SLAB][GSP]
    const int sp = bid - nSplitBase;
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

