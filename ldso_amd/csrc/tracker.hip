// tracker.hip — CoarseTracker direct image alignment on gfx950 (reference src/frontend/CoarseTracker.cc).
//
//   makeK                 CoarseTracker.cc:219-246   (host, float arithmetic as the reference)
//   makeCoarseDepthL0     CoarseTracker.cc:258-438   k_tr_scatter / k_tr_pool / k_tr_dilate / k_tr_count+scan+write
//   calcRes + calcGSSSE   CoarseTracker.cc:440-632   tr_eval(): ONE fused pass — projection, bilinear Vec3f gather,
//                                                    Huber, energy, flow indicators and the 9x9 weighted outer
//                                                    product; the warped buffers are never materialised
//   trackNewestCoarse     CoarseTracker.cc:61-217    k_tr_track: the whole coarse-to-fine LM loop runs inside one
//                                                    persistent workgroup per motion hypothesis (no host sync, no
//                                                    launch per iteration); the 8x8 LDLT, SE3::exp and the
//                                                    accept/reject logic run on the device in fp64.
// The per-level point clouds are tiny (10^3..10^4 points): the path is latency bound, so the design minimises
// dependent launches, and batches hypotheses across workgroups (FullSystem::trackNewCoarse tries up to 83).
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>
#include <string>
#include <cstring>
#include <cmath>
#include "../../include/ldso_hip.h"
#include "lie_dev.h"
#include "pyramid.h"
#include <cstdlib>

void ldso_set_error(const std::string &s);
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ldso_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return LDSO_E_HIP; } } while (0)
#define REQ(cond, msg) do { if (!(cond)) { ldso_set_error(msg); return LDSO_E_INVALID; } } while (0)
#define RUN(x) do { int r_ = (x); if (r_ != LDSO_OK) return r_; } while (0)

#define TR_NT 256          // 4 wavefronts, one per SIMD: 512 registers (VGPR + AGPR) per lane - tr_eval keeps 4 points per lane in flight without scratch spills
#define TR_MAXL LDSO_PYR_LEVELS

struct TrLevel {
    int w, h, n;
    float fx, fy, cx, cy;
    float Ki[9];
    const float *newImg;      // Vec3f AoS of the frame being tracked
    const float *refImg;      // Vec3f AoS of the reference keyframe
    float *pc_u, *pc_v, *pc_idepth, *pc_color;
    float *idepth, *wsum, *wsum_bak;
    int *blockCnt;            // compaction scratch
};

struct TrParams {
    TrLevel lv[TR_MAXL];
    int levels;
    float ref_a, ref_b, ref_exposure, new_exposure;
    float huberTH, coarseCutoffTH, affineOptModeA, affineOptModeB;
};

struct TrHyp {                 // one motion hypothesis in / result out
    double T[12];
    float a, b;
    int coarsestLvl;
    double minRes[5];
    double lastResiduals[5];
    double flow[3];
    int ok, iterations;
    int evals[5], pivotedSolves;          // calcRes evaluations per pyramid level (algorithmic bytes of a track = sum evals[l] * pc_n[l] * 64 B); LM solves that fell back to the pivoted factorisation
    double dbg[12];             // LDSO_STAMPS builds: time in tr_eval / serial LM sections / evals count
};

// ---------------------------------------------------------------------------------------------------------
// makeCoarseDepthL0
// ---------------------------------------------------------------------------------------------------------
// The reference adds the points to the level-0 maps one after the other (CoarseTracker.cc:268-283): float sums in point order.
// Deterministic and in that order here, without float atomics: a first kernel threads the points of every pixel into a list
// (integer atomics: the list order is arbitrary, its content is not), the second lets the lowest-indexed point of a pixel add all of
// the pixel's points in ascending index order (lists are short: a selection walk).  Points that round to a pixel outside the image
// (the reference would write out of bounds) are ignored.
__device__ __forceinline__ int tr_pt_pixel(const float *pts, int i, int w, int h) {
    const int u = (int) (pts[4 * i + 0] + 0.5f), v = (int) (pts[4 * i + 1] + 0.5f);
    return (u >= 0 && u < w && v >= 0 && v < h) ? u + w * v : -1;
}
__global__ void k_tr_scatter_link(const float *pts, int n, int *head /*w*h, -1*/, int *next /*n*/, int w, int h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = tr_pt_pixel(pts, i, w, h);
    if (px >= 0) next[i] = atomicExch(&head[px], i);
}
__global__ void k_tr_scatter(const float *pts, int n, float *idepth, float *wsum, const int *head, const int *next, int w, int h) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = tr_pt_pixel(pts, i, w, h);
    if (px < 0) return;
    for (int j = head[px]; j >= 0; j = next[j]) if (j < i) return;       // an earlier point owns this pixel
    float sid = 0.f, sw = 0.f;
    int cur = i;
    while (cur >= 0) {
        const float new_idepth = pts[4 * cur + 2];
        const float weight = sqrtf((float) (1e-3 / ((double) pts[4 * cur + 3] + 1e-12)));
        sid += new_idepth * weight;
        sw += weight;
        int nxt = -1;                                                     // the smallest index above cur
        for (int j = head[px]; j >= 0; j = next[j]) if (j > cur && (nxt < 0 || j < nxt)) nxt = j;
        cur = nxt;
    }
    idepth[px] = sid; wsum[px] = sw;
}

__global__ void k_tr_pool(const float *id_lm, const float *ws_lm, float *id_l, float *ws_l, int wl, int hl, int wlm1) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wl * hl) return;
    int x = i % wl, y = i / wl;
    int bidx = 2 * x + 2 * y * wlm1;
    id_l[i] = id_lm[bidx] + id_lm[bidx + 1] + id_lm[bidx + wlm1] + id_lm[bidx + wlm1 + 1];
    ws_l[i] = ws_lm[bidx] + ws_lm[bidx + 1] + ws_lm[bidx + wlm1] + ws_lm[bidx + wlm1 + 1];
}

// in-place dilation exactly as the reference: reads the weight backup and idepth of pixels with weight > 0,
// writes only pixels with weight <= 0, so a parallel sweep equals the sequential one.
__global__ void k_tr_dilate(float *idepth, float *wsum, const float *bak, int wl, int hl, int diagonal) {
    int i = blockIdx.x * blockDim.x + threadIdx.x + wl;
    int wh = wl * hl - wl;
    if (i >= wh) return;
    if (bak[i] <= 0) {
        float sum = 0, num = 0, numn = 0;
        int o0 = diagonal ? 1 + wl : 1, o1 = diagonal ? -1 - wl : -1, o2 = diagonal ? wl - 1 : wl, o3 = diagonal ? -wl + 1 : -wl;
        if (bak[i + o0] > 0) { sum += idepth[i + o0]; num += bak[i + o0]; numn++; }
        if (bak[i + o1] > 0) { sum += idepth[i + o1]; num += bak[i + o1]; numn++; }
        if (bak[i + o2] > 0) { sum += idepth[i + o2]; num += bak[i + o2]; numn++; }
        if (bak[i + o3] > 0) { sum += idepth[i + o3]; num += bak[i + o3]; numn++; }
        if (numn > 0) { idepth[i] = sum / numn; wsum[i] = num / numn; }
    }
}

// order-preserving compaction over the interior (2 <= x < w-2, 2 <= y < h-2), row-major like the reference
__device__ __forceinline__ bool tr_keep(const float *idepth, const float *wsum, const float *ref, int i, float &id, float &col) {
    float ws = wsum[i];
    if (!(ws > 0)) return false;
    id = idepth[i] / ws;
    col = ref[3 * i];
    return isfinite(col) && (id > 0);
}

__global__ void k_tr_count(TrLevel L) {
    __shared__ int sc[256 / 64];
    int wi = L.w - 4, hi = L.h - 4;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (e < wi * hi) { int x = 2 + e % wi, y = 2 + e / wi; float id, col; keep = tr_keep(L.idepth, L.wsum, L.refImg, x + y * L.w, id, col); }
    int c = __popcll(__ballot(keep));
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) L.blockCnt[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}

__global__ void k_tr_scan(int *cnt, int nb, int *total) {     // single block exclusive scan
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += blockDim.x) {
        int i = base + threadIdx.x;
        int v = (i < nb) ? cnt[i] : 0;
        int inc = v;
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if ((threadIdx.x & 63) >= o) inc += t; }
        __shared__ int ws[16];
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = inc;
        __syncthreads();
        int off = carry;
        for (int wv = 0; wv < (int) (threadIdx.x >> 6); wv++) off += ws[wv];
        if (i < nb) cnt[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ void k_tr_write(TrLevel L) {
    __shared__ int sc[256 / 64];
    int wi = L.w - 4, hi = L.h - 4;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    float id = 0, col = 0;
    int x = 0, y = 0;
    if (e < wi * hi) { x = 2 + e % wi; y = 2 + e / wi; keep = tr_keep(L.idepth, L.wsum, L.refImg, x + y * L.w, id, col); }
    unsigned long long m = __ballot(keep);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sc[wv] = __popcll(m);
    __syncthreads();
    int off = L.blockCnt[blockIdx.x];
    for (int q = 0; q < wv; q++) off += sc[q];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (keep) { L.pc_u[off] = (float) x; L.pc_v[off] = (float) y; L.pc_idepth[off] = id; L.pc_color[off] = col; }
}

// ---------------------------------------------------------------------------------------------------------
// fused calcRes + calcGSSSE over one level by one workgroup; results in LDS `out` (doubles):
//   [0] E  [1] numTermsInE  [2] sumSquaredShiftT  [3] sumSquaredShiftRT  [4] sumSquaredShiftNum  [5] numSaturated
//   [6] numTermsInWarped   [7..51] 45 upper-triangular entries of sum w J J^T (9x9)
// ---------------------------------------------------------------------------------------------------------
#ifdef LDSO_STAMPS
#define LD_STAMP_ON_TR 1
#else
#define LD_STAMP_ON_TR 0
#endif
#define TR_NACC 52
#define TR_U 4            // points per thread and pass of tr_eval

__device__ __forceinline__ float tr_readlane(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }

__device__ __forceinline__ void aff_from_to_f(float expF, float expT, float aF, float bF, float aT, float bT, float &a, float &b) {
    if (expF == 0 || expT == 0) { expT = expF = 1; }
    a = expf(aT - aF) * expT / expF;
    b = bT - a * bF;
}

// A workgroup's share of a level: nAct workgroups split the n points into contiguous chunks (multiples of 64); a workgroup
// works with only as many wavefronts as its share needs (TR_U points per lane), so the small levels pay the 52-value reduction
// on few wavefronts.  The points of the first pass stay in registers across the evaluations of a level (they do not depend on the
// candidate pose): an evaluation then costs ONE memory latency (the tap gather) instead of two.
struct TrPts { float id[TR_U], x[TR_U], y[TR_U], col[TR_U]; int lvl; };
// LDS scratch of tr_eval: per wavefront the staged vectors U | V ([16][TR_UVP] floats each), then one 16x16 result per wavefront
#define TR_UVP 68                                   // row pitch: lane l of an MFMA reads bank 4 (l & 15) + (l >> 4) (+ 4 m): conflict-free
#define TR_LDS_FLOATS ((TR_NT / 64) * (2 * 16 * TR_UVP + 256))
// accumulator q of the 52 sums <-> entry (row, col) of M = sum_p u_p v_p^T (see tr_eval): q < 7: (9 + q, 9); then the upper triangle of the 9x9
static __device__ const unsigned char TR_QROW[TR_NACC] = {9, 10, 11, 12, 13, 14, 15, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 7, 7, 8};
static __device__ const unsigned char TR_QCOL[TR_NACC] = {9, 9, 9, 9, 9, 9, 9, 0, 1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4, 5, 6, 7, 8, 2, 3, 4, 5, 6, 7, 8, 3, 4, 5, 6, 7, 8, 4, 5, 6, 7, 8, 5, 6, 7, 8, 6, 7, 8, 7, 8, 8};
typedef float __attribute__((ext_vector_type(4))) tr_f4;
// pointers read from TrParams are generic to the compiler (flat loads); they all point to device memory
typedef const __attribute__((address_space(1))) float *tr_gptr;

#if LD_STAMP_ON_TR
__device__ long long g_trPh[5][8];      // debug: per level, time in the phases of tr_eval (wave 0 of workgroup 0), and the evaluation count
#define TPH(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) { long long n_ = wall_clock64(); g_trPh[lvl][k] += n_ - tph_; tph_ = n_; } } while (0)
#else
#define TPH(k) do { } while (0)
#endif
__device__ __forceinline__ void tr_eval(const TrParams &P, int lvl, const float *Rt /*LDS: R row major (9), t (3)*/, float aff_a, float aff_b, float cutoffTH,
                        double *out /*LDS TR_NACC*/, float *red /*LDS TR_LDS_FLOATS*/, int g, int nAct, TrPts &pc) {
    const TrLevel &L = P.lv[lvl];
#if LD_STAMP_ON_TR
    long long tph_ = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nW = blockDim.x >> 6;
    const int wl = L.w, hl = L.h;
    const float fxl = L.fx, fyl = L.fy, cxl = L.cx, cyl = L.cy;
    const tr_gptr gImg = (tr_gptr) L.newImg, gId = (tr_gptr) L.pc_idepth, gU = (tr_gptr) L.pc_u, gV = (tr_gptr) L.pc_v, gCol = (tr_gptr) L.pc_color;
    const int share = (((L.n + nAct - 1) / nAct) + 63) & ~63;
    const int lo = g * share, hi = min(L.n, lo + share);
    // latency first: the share is spread over all wavefronts (one per SIMD) before a lane takes a second point
    const int nWact = max(1, min(nW, (hi - lo + 63) / 64)), nth = nWact * 64;
    const int nU = min(TR_U, max(1, (hi - lo + nth - 1) / nth));          // points per lane in the first pass (uniform)
    // RKi = R.cast<float>() * Ki ; t.cast<float>()
    float RKi[9], t[3];
    for (int r = 0; r < 3; r++) t[r] = Rt[9 + r];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) RKi[r * 3 + c] = (Rt[r * 3 + 0] * L.Ki[0 * 3 + c] + Rt[r * 3 + 1] * L.Ki[1 * 3 + c]) + Rt[r * 3 + 2] * L.Ki[2 * 3 + c];
    float affA, affB;
    aff_from_to_f(P.ref_exposure, P.new_exposure, P.ref_a, P.ref_b, aff_a, aff_b, affA, affB);
    const float maxEnergy = 2 * P.huberTH * cutoffTH - P.huberTH * P.huberTH;
    const float b0 = P.ref_b;

    // The 52 sums are entries of ONE matrix M = sum_p u_p v_p^T with u = [hw J (9) | E, terms, 3 flow sums, saturated, warped] and
    // v = [J (9) | 1 | 0 ...]: the fp32 matrix cores do the 45 products per point AND the reduction over the points
    // (v_mfma_f32_16x16x4_f32: 4 points per instruction) - no per-lane accumulators, no 52-value cross-lane reduction.  Every lane
    // stages its point's u, v in the wave's LDS scratch ([component][point], conflict-free pitch) and reads them back as operands
    // (same wave: no barrier); M accumulates in 2 x 4 registers per lane.
    float *sU = red + wave * (2 * 16 * TR_UVP), *sV = sU + 16 * TR_UVP;
    tr_f4 Dm0 = {0.f, 0.f, 0.f, 0.f}, Dm1 = {0.f, 0.f, 0.f, 0.f};
    if (tid < nth) {
#pragma unroll
        for (int c = 10; c < 16; c++) sV[c * TR_UVP + lane] = 0.f;            // components 10..15 of v stay zero
    }
    TPH(0);

    // one pass: TR_U points per lane; all 12-float tap gathers of a pass are issued together
    auto pass = [&](const float (&id)[TR_U], const float (&x)[TR_U], const float (&y)[TR_U], const float (&col)[TR_U], int ib, int nu) {
        float uu[TR_U], vv[TR_U], Ku[TR_U], Kv[TR_U], nid[TR_U], tap[TR_U][12];
        bool ok[TR_U], in[TR_U];
#pragma unroll
        for (int u_ = 0; u_ < TR_U; u_++) {
            if (u_ >= nu) { ok[u_] = false; in[u_] = false; continue; }          // uniform
            in[u_] = ib + u_ * nth < hi;
            float p0 = ((RKi[0] * x[u_] + RKi[1] * y[u_]) + RKi[2] * 1.0f) + t[0] * id[u_];
            float p1 = ((RKi[3] * x[u_] + RKi[4] * y[u_]) + RKi[5] * 1.0f) + t[1] * id[u_];
            float p2 = ((RKi[6] * x[u_] + RKi[7] * y[u_]) + RKi[8] * 1.0f) + t[2] * id[u_];
            uu[u_] = p0 / p2; vv[u_] = p1 / p2;
            Ku[u_] = fxl * uu[u_] + cxl; Kv[u_] = fyl * vv[u_] + cyl;
            nid[u_] = id[u_] / p2;
            ok[u_] = in[u_] && (Ku[u_] > 2 && Kv[u_] > 2 && Ku[u_] < wl - 3 && Kv[u_] < hl - 3 && nid[u_] > 0);
            const int ix = ok[u_] ? (int) Ku[u_] : 0, iy = ok[u_] ? (int) Kv[u_] : 0;
            const tr_gptr bp = gImg + 3 * (ix + iy * wl);
            const tr_gptr bq = bp + 3 * wl;
#pragma unroll
            for (int c = 0; c < 6; c++) { tap[u_][c] = bp[c]; tap[u_][6 + c] = bq[c]; }
        }
#pragma unroll
        for (int u_ = 0; u_ < TR_U; u_++) {
            if (u_ >= nu) break;                                                   // uniform
            const int i = ib + u_ * nth;
            float uv[16], vj[9];
#pragma unroll
            for (int c = 0; c < 16; c++) uv[c] = 0.f;
#pragma unroll
            for (int c = 0; c < 9; c++) vj[c] = 0.f;
            if (in[u_] && lvl == 0 && i % 32 == 0) {
                const float *Ki = L.Ki;
                const float xx = x[u_], yy = y[u_], idd = id[u_];
                float k0 = (Ki[0] * xx + Ki[1] * yy) + Ki[2] * 1.0f, k1 = (Ki[3] * xx + Ki[4] * yy) + Ki[5] * 1.0f, k2 = (Ki[6] * xx + Ki[7] * yy) + Ki[8] * 1.0f;
                float a0 = k0 + t[0] * idd, a1 = k1 + t[1] * idd, a2 = k2 + t[2] * idd;
                float KuT = fxl * (a0 / a2) + cxl, KvT = fyl * (a1 / a2) + cyl;
                float c0 = k0 - t[0] * idd, c1 = k1 - t[1] * idd, c2 = k2 - t[2] * idd;
                float KuT2 = fxl * (c0 / c2) + cxl, KvT2 = fyl * (c1 / c2) + cyl;
                float r0 = (((RKi[0] * xx + RKi[1] * yy) + RKi[2] * 1.0f)) - t[0] * idd, r1 = (((RKi[3] * xx + RKi[4] * yy) + RKi[5] * 1.0f)) - t[1] * idd,
                      r2 = (((RKi[6] * xx + RKi[7] * yy) + RKi[8] * 1.0f)) - t[2] * idd;
                float Ku3 = fxl * (r0 / r2) + cxl, Kv3 = fyl * (r1 / r2) + cyl;
                uv[11] = ((KuT - xx) * (KuT - xx) + (KvT - yy) * (KvT - yy)) + ((KuT2 - xx) * (KuT2 - xx) + (KvT2 - yy) * (KvT2 - yy));
                uv[12] = ((Ku[u_] - xx) * (Ku[u_] - xx) + (Kv[u_] - yy) * (Kv[u_] - yy)) + ((Ku3 - xx) * (Ku3 - xx) + (Kv3 - yy) * (Kv3 - yy));
                uv[13] = 2;
            }
            if (ok[u_]) {
                const float refColor = col[u_], u = uu[u_], v = vv[u_], new_idepth = nid[u_];
                const int ix = (int) Ku[u_], iy = (int) Kv[u_];
                float dx = Ku[u_] - ix, dy = Kv[u_] - iy, dxdy = dx * dy;
                float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
                const float *tp = tap[u_];
                float h0 = ((w11 * tp[9] + w01 * tp[6]) + w10 * tp[3]) + w00 * tp[0];
                float h1 = ((w11 * tp[10] + w01 * tp[7]) + w10 * tp[4]) + w00 * tp[1];
                float h2 = ((w11 * tp[11] + w01 * tp[8]) + w10 * tp[5]) + w00 * tp[2];
                if (isfinite(h0)) {
                    float residual = h0 - (float) (affA * refColor + affB);
                    float hw = fabsf(residual) < P.huberTH ? 1 : P.huberTH / fabsf(residual);
                    uv[10] = 1;
                    if (fabsf(residual) > cutoffTH) {
                        uv[9] = maxEnergy; uv[14] = 1;
                    } else {
                        uv[9] = hw * residual * residual * (2 - hw); uv[15] = 1;
                        // calcGSSSE row (CoarseTracker.cc:592-615)
                        float ddx = h1 * fxl, ddy = h2 * fyl;
                        vj[0] = new_idepth * ddx;
                        vj[1] = new_idepth * ddy;
                        vj[2] = 0 - new_idepth * (u * ddx + v * ddy);
                        vj[3] = 0 - ((u * v) * ddx + ddy * (1 + v * v));
                        vj[4] = (u * v) * ddy + ddx * (1 + u * u);
                        vj[5] = u * ddy - v * ddx;
                        vj[6] = affA * (b0 - refColor);
                        vj[7] = -1;
                        vj[8] = residual;
#pragma unroll
                        for (int r = 0; r < 9; r++) uv[r] = vj[r] * hw;
                    }
                }
            }
            // stage (all lanes of the wave: a lane without a valid point contributes u = 0), then 16 MFMAs over the 64 points
#pragma unroll
            for (int c = 0; c < 16; c++) sU[c * TR_UVP + lane] = uv[c];
#pragma unroll
            for (int c = 0; c < 9; c++) sV[c * TR_UVP + lane] = vj[c];
            sV[9 * TR_UVP + lane] = 1.f;
#pragma unroll
            for (int m = 0; m < 16; m += 2) {
                const int o0 = (lane & 15) * TR_UVP + 4 * m + (lane >> 4);
                Dm0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sU[o0], sV[o0], Dm0, 0, 0, 0);
                Dm1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sU[o0 + 4], sV[o0 + 4], Dm1, 0, 0, 0);
            }
        }
    };

    if (tid < nth) {
        if (pc.lvl != lvl) {                     // uniform: first evaluation of this level
#pragma unroll
            for (int u_ = 0; u_ < TR_U; u_++) {
                const int i = lo + tid + u_ * nth;
                const int ic = i < hi ? i : 0;
                pc.id[u_] = gId[ic]; pc.x[u_] = gU[ic]; pc.y[u_] = gV[ic]; pc.col[u_] = gCol[ic];
            }
        }
        pass(pc.id, pc.x, pc.y, pc.col, lo + tid, nU);
        for (int ib = lo + tid + nth * TR_U; ib < hi; ib += nth * TR_U) {
            float id[TR_U], x[TR_U], y[TR_U], col[TR_U];
#pragma unroll
            for (int u_ = 0; u_ < TR_U; u_++) {
                const int i = ib + u_ * nth;
                const int ic = i < hi ? i : 0;
                id[u_] = gId[ic]; x[u_] = gU[ic]; y[u_] = gV[ic]; col[u_] = gCol[ic];
            }
            pass(id, x, y, col, ib, TR_U);
        }
    }
    pc.lvl = lvl;
    TPH(1);
    // one 16x16 result per wavefront: lane l, register r holds M[(l >> 4) * 4 + r][l & 15]
    float *sM = red + (TR_NT / 64) * (2 * 16 * TR_UVP);
    if (wave < nWact) {
        const tr_f4 Dm = Dm0 + Dm1;
#pragma unroll
        for (int r = 0; r < 4; r++) sM[wave * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = Dm[r];
    }
    TPH(2);
    __syncthreads();
    TPH(3);
    if (tid < TR_NACC) {
        const int o = TR_QROW[tid] * 16 + TR_QCOL[tid];
        float rv[TR_NT / 64];
#pragma unroll
        for (int w_ = 0; w_ < TR_NT / 64; w_++) rv[w_] = (w_ < nWact) ? sM[w_ * 256 + o] : 0.f;
        double s = 0;
#pragma unroll
        for (int w_ = 0; w_ < TR_NT / 64; w_++) if (w_ < nWact) s += (double) rv[w_];
        out[tid] = s;
    }
    __syncthreads();
    TPH(4);
#if LD_STAMP_ON_TR
    if (threadIdx.x == 0 && blockIdx.x == 0) g_trPh[lvl][7]++;
#endif
}

// H (8x8), b (8) from the accumulated sums: divide by the padded n, apply the reference's scale swap.
// One thread per entry (threads 0..71 of the block): no local arrays (they would live in scratch memory).
__device__ __forceinline__ void tr_hb(const double *acc, double *H, double *b) {
    const int tid = threadIdx.x;
    if (tid >= 72) return;
    const int r = (tid < 64) ? (tid >> 3) : (tid - 64), c = (tid < 64) ? (tid & 7) : 8;
    const int nw = (int) acc[6];
    const int npad = (nw + 3) / 4 * 4;
    const double inv = (double) (1.0f / (float) npad);
    // cols 0-2 x SCALE_XI_ROT, 3-5 x SCALE_XI_TRANS (sic)
    auto cs = [](int q) -> double { return (q < 3) ? 1.0 : (q < 6) ? 0.5 : (q == 6) ? 10.0 : 1000.0; };
    const int lo = min(r, c), hi = max(r, c);
    const double m = (double) (float) acc[7 + lo * 9 - (lo * (lo - 1)) / 2 + (hi - lo)];
    if (c < 8) H[r * 8 + c] = m * inv * cs(r) * cs(c);
    else b[r] = m * inv * cs(r);
}

// Eigen-style pivoted LDLT solve of an n x n (n <= 8) system, single thread
__device__ void small_ldlt_solve(const double *Ain, const double *rhs, double *x, int n) {
    double A[64];
    int tr[8];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int k = 0; k < n; k++) {
        int idx = k; double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[i * n + i]) > best) { best = fabs(A[i * n + i]); idx = i; }
        tr[k] = idx;
        if (idx != k) {
            for (int j = 0; j < k; j++) { double a = A[k * n + j]; A[k * n + j] = A[idx * n + j]; A[idx * n + j] = a; }
            for (int j = idx + 1; j < n; j++) { double a = A[j * n + k]; A[j * n + k] = A[j * n + idx]; A[j * n + idx] = a; }
            { double a = A[k * n + k]; A[k * n + k] = A[idx * n + idx]; A[idx * n + idx] = a; }
            for (int j = k + 1; j < idx; j++) { double a = A[j * n + k]; A[j * n + k] = A[idx * n + j]; A[idx * n + j] = a; }
        }
        double temp[8];
        for (int j = 0; j < k; j++) temp[j] = A[j * n + j] * A[k * n + j];
        double s = 0;
        for (int j = 0; j < k; j++) s += A[k * n + j] * temp[j];
        A[k * n + k] -= s;
        for (int i = k + 1; i < n; i++) { double tt = 0; for (int j = 0; j < k; j++) tt += A[i * n + j] * temp[j]; A[i * n + k] -= tt; }
        double akk = A[k * n + k];
        if (fabs(akk) > 0.0) for (int i = k + 1; i < n; i++) A[i * n + k] /= akk;
    }
    for (int i = 0; i < n; i++) x[i] = rhs[i];
    for (int k = 0; k < n; k++) if (tr[k] != k) { double a = x[k]; x[k] = x[tr[k]]; x[tr[k]] = a; }
    for (int i = 0; i < n; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= A[i * n + j] * x[j]; x[i] = s; }
    for (int i = 0; i < n; i++) x[i] = (fabs(A[i * n + i]) > 2.2250738585072014e-308) ? x[i] / A[i * n + i] : 0.0;
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < n; j++) s -= A[j * n + i] * x[j]; x[i] = s; }
    for (int k = n - 1; k >= 0; k--) if (tr[k] != k) { double a = x[k]; x[k] = x[tr[k]]; x[tr[k]] = a; }
}

// H (1 + lambda on the diagonal) x = rhsSign rhs, 8 x 8, by ONE lane with everything in registers: unpivoted LDL^T (the matrix is symmetric positive definite
// after the scaling, so Eigen's diagonal pivoting - which the reference runs, CoarseTracker.cc:120-128 - gives the same solution up to rounding; zero pivots
// as there: the column stays as it is, the component is dropped by D^+), one reciprocal per column (v_rcp_f64 + two Newton steps: 1.1e-16) instead of the
// ~12-instruction IEEE division sequence, the solution scaled by the same reciprocals.  On the critical path of every LM iteration: until round 4 a
// wavefront ran Eigen's pivoted factorisation lane-parallel (pivot search by DPP maxima + ballot, rows and columns exchanged by ds_bpermute: eight rounds of
// ~250 ns, 2.1 us per solve); one lane's ~250 fp64 instructions take 1.0 us (round 5, A/B on one box: kernel 311 -> 2xx us per track, same 28 iterations).
__device__ __forceinline__ double tr_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}
// Returns false when a pivot has lost six digits against the diagonal entry it started from (|d_k| < 1e-6 |H_kk|: column k is, to float precision, a combination of the
// columns in front of it - a rank-deficient system: fewer than eight reference points on a level; the entries of H are float sums, so "zero" pivots come out at ~1e-7):
// Eigen's diagonal pivoting defers such a pivot to the end, where the unpivoted elimination inverts it early - the caller then runs the pivoted factorisation
// (tr_solve_pivoted8, the reference's algorithm) instead.
__device__ __forceinline__ bool ldlt8_lane(const double *H /*LDS 64, row major*/, const double *rhs /*LDS 8*/, double rhsSign, double diagScale, double *x /*8 registers*/) {
    double A[8][8], y[8], inv[8];
    bool wellConditioned = true;
#pragma unroll
    for (int i = 0; i < 8; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) A[i][j] = H[i * 8 + j];
        y[i] = rhsSign * rhs[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) A[i][i] *= diagScale;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double d = A[k][k];
        wellConditioned = wellConditioned && (fabs(d) >= 1e-6 * fabs(H[k * 8 + k] * diagScale));          // (the entry it started from: re-read, not kept in a register)
        const bool valid = fabs(d) > 2.2250738585072014e-308;
        inv[k] = valid ? tr_rcp(d) : 0.0;
        const double sc = valid ? inv[k] : 1.0;
#pragma unroll
        for (int j = k + 1; j < 8; j++) {
            const double L = A[j][k] * sc;          // L[j][k] (zero pivot: the column as it is)
            if (valid) {
#pragma unroll
                for (int i = j; i < 8; i++) A[i][j] = __builtin_fma(-A[i][k], L, A[i][j]);
            }
            y[j] = __builtin_fma(-y[k], L, y[j]);
            A[j][k] = L;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = y[j] * inv[j];
#pragma unroll
    for (int c = 7; c >= 1; c--)
#pragma unroll
        for (int r = 0; r < c; r++) x[r] = __builtin_fma(-A[c][r], x[c], x[r]);
    return wellConditioned;
}

__device__ __forceinline__ void tr_vec6(const double *acc, double *rs) {
    const double a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3], a4 = acc[4], a5 = acc[5];          // all loads in front of the first store (both may be LDS)
    rs[0] = (double) (float) a0;
    rs[1] = (double) (int) a1;
    rs[2] = (double) ((float) a2 / ((float) a4 + 0.1f));
    rs[3] = 0;
    rs[4] = (double) ((float) a3 / ((float) a4 + 0.1f));
    rs[5] = (double) ((float) (int) a5 / (float) (int) a1);
}

// The 8 x 8 LM system by the reference's own algorithm (Eigen's diagonally pivoted LDL^T, CoarseTracker.cc:120-128): the fallback of ldlt8_lane for ill-conditioned
// systems.  Rare, single thread, out of line (its arrays live in scratch memory).
__device__ __attribute__((noinline)) void tr_solve_pivoted8(const double *sH, const double *sB, double rhsSign, double diagScale, double *sInc) {
    double Hl[64], nb[8], xs[8];
    for (int i = 0; i < 64; i++) Hl[i] = sH[i];
    for (int i = 0; i < 8; i++) { Hl[i * 8 + i] *= diagScale; nb[i] = rhsSign * sB[i]; }
    small_ldlt_solve(Hl, nb, xs, 8);
    for (int i = 0; i < 8; i++) sInc[i] = xs[i];
}

// The step when an affine parameter is fixed (setting_affineOptModeA/B < 0; CoarseTracker.cc:129-166): rare, single thread, kept out
// of line so that its 8x8 temporaries (scratch memory) stay out of the tracking kernel's register allocation.
__device__ __attribute__((noinline)) void tr_solve_fixed_affine(const double *sH, const double *sB, const double *sNb, float lambda, bool fixA, bool fixB, double *sInc) {
    double Hl[64], inc[8];
    for (int i = 0; i < 64; i++) Hl[i] = sH[i];
    for (int i = 0; i < 8; i++) { Hl[i * 8 + i] *= (1 + lambda); inc[i] = 0; }
    if (fixA && fixB) {
        double H6[36], x6[6];
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) H6[i * 6 + j] = Hl[i * 8 + j];
        small_ldlt_solve(H6, sNb, x6, 6);
        for (int i = 0; i < 6; i++) inc[i] = x6[i];
    } else if (fixB) {
        double H7[49], x7[7];
        for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) H7[i * 7 + j] = Hl[i * 8 + j];
        small_ldlt_solve(H7, sNb, x7, 7);
        for (int i = 0; i < 7; i++) inc[i] = x7[i];
    } else {
        double bs[8], H7[49], x7[7], nb7[7];
        for (int i = 0; i < 8; i++) bs[i] = sB[i];
        for (int i = 0; i < 8; i++) Hl[i * 8 + 6] = Hl[i * 8 + 7];
        for (int j = 0; j < 8; j++) Hl[6 * 8 + j] = Hl[7 * 8 + j];
        bs[6] = bs[7];
        for (int i = 0; i < 7; i++) { nb7[i] = -bs[i]; for (int j = 0; j < 7; j++) H7[i * 7 + j] = Hl[i * 8 + j]; }
        small_ldlt_solve(H7, nb7, x7, 7);
        for (int i = 0; i < 6; i++) inc[i] = x7[i];
        inc[7] = x7[6];
    }
    for (int i = 0; i < 8; i++) sInc[i] = inc[i];
}

// ---------------------------------------------------------------------------------------------------------
// Cooperative evaluation: G workgroups share ONE hypothesis.  Workgroup 0 of the group runs the LM loop (it is the only one that
// decides anything); for every calcRes / calcGSSSE evaluation it publishes the candidate (pose, affine, level, cut-off) in device
// memory under a new sequence number, every workgroup of the group evaluates a contiguous 1/nAct share of the level's points
// (nAct grows with the level size: the coarse levels stay on the leader alone), the helpers store their 52 partial sums, the
// leader adds them in workgroup order (deterministic).  The hand-over is made of self-validating 64-bit words (payload | sequence
// number) written and read with device-scope relaxed atomics (performed at the memory side, visible across the XCDs' L2s): one
// memory round trip per direction, no fences, no counters.  All workgroups of a launch must be resident (the host bounds
// nhyp * G by the CU count); sequence numbers grow over the launches of a handle, so nothing has to be cleared in between.
// ---------------------------------------------------------------------------------------------------------
#define TR_GMAX 16
#ifndef TR_COOP_MIN
#define TR_COOP_MIN 512        // smaller levels stay on the leader: less than the ~1.4 us of a hand-over to win
#endif
#ifndef TR_COOP_PER
#define TR_COOP_PER 64         // finest share: one wavefront with one point per lane
#endif
#define TR_COOP_SLOTS 128       // = the maximum number of hypotheses of ldso_tr_track_batch: coop[] is indexed by hypothesis
#define TR_SPIN_LIMIT 500000000ll   // bail-out of the hand-over polls: 5 s of the 100 MHz wall clock (a track takes < 1 ms)
struct TrCoop {
    // every 64-bit word carries (payload << 32 | sequence number): a word is valid by itself, no fence / second round trip needed
    unsigned long long cmd[16];                          // R (9), t (3) as float, affine a, b, cut-off, level (-1: the track is over)
    unsigned long long part[TR_GMAX][TR_NACC][2];        // a helper's partial sums: low / high half of the double
};
__device__ __forceinline__ unsigned long long tr_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tr_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int tr_nact(int n, int G) { return (G == 1 || n < TR_COOP_MIN) ? 1 : min(G, (n + TR_COOP_PER - 1) / TR_COOP_PER); }

// leader side of one evaluation (all threads of the leader workgroup)
template <int G>
__device__ __forceinline__ void tr_eval_lead(const TrParams &P, int lvl, const double *T, float a, float b, float cutoff, double *sAcc, float *sRed, float *sRt,
                                             TrCoop *co, int &seq, TrPts &pc, int *sAbort) {
    const int tid = threadIdx.x, nAct = tr_nact(P.lv[lvl].n, G);
    if (tid < 12) sRt[tid] = (float) T[tid < 9 ? (tid / 3) * 4 + tid % 3 : (tid - 9) * 4 + 3];
    __syncthreads();
    if (G > 1 && nAct > 1) {
        ++seq;
        if (tid < 16) {
            const unsigned pl = tid < 12 ? __builtin_bit_cast(unsigned, sRt[tid]) : tid == 12 ? __builtin_bit_cast(unsigned, a) : tid == 13 ? __builtin_bit_cast(unsigned, b)
                              : tid == 14 ? __builtin_bit_cast(unsigned, cutoff) : (unsigned) lvl;
            tr_st(&co->cmd[tid], ((unsigned long long) pl << 32) | (unsigned) seq);
        }
    }
    tr_eval(P, lvl, sRt, a, b, cutoff, sAcc, sRed, 0, nAct, pc);
    if (G > 1 && nAct > 1) {
        if (tid < TR_NACC) {
            // all helpers' words are requested together (one memory round trip once they are there), added in workgroup order
            unsigned long long w0[G > 1 ? G - 1 : 1], w1[G > 1 ? G - 1 : 1];
            bool ok;
            unsigned spins = 0; long long t0 = 0;
            do {
                ok = true;
#pragma unroll
                for (int g = 1; g < G; g++) if (g < nAct) {
                    w0[g - 1] = tr_ld(&co->part[g][tid][0]); w1[g - 1] = tr_ld(&co->part[g][tid][1]);
                }
#pragma unroll
                for (int g = 1; g < G; g++) if (g < nAct) ok = ok && ((unsigned) w0[g - 1] == (unsigned) seq) && ((unsigned) w1[g - 1] == (unsigned) seq);
                // a helper workgroup that never becomes resident (CUs held by another process / a masked device) must not hang the
                // stream: give up after TR_SPIN_LIMIT, the track then reports an error instead of a pose
                if (!ok && (++spins & 1023u) == 0) { const long long now = wall_clock64(); if (t0 == 0) t0 = now; else if (now - t0 > TR_SPIN_LIMIT) { *sAbort = 1; break; } }
            } while (!ok);
            double s_ = sAcc[tid];
#pragma unroll
            for (int g = 1; g < G; g++) if (g < nAct) s_ += __builtin_bit_cast(double, (w1[g - 1] & 0xFFFFFFFF00000000ull) | (w0[g - 1] >> 32));
            sAcc[tid] = s_;
        }
        __syncthreads();
    }
}

// helper workgroups: evaluate shares until the leader says the track is over
template <int G>
__device__ void tr_helper_loop(const TrParams &P, TrCoop *co, int g, int seq, double *sAcc, float *sRed, float *sRt, float *sF, int *sI) {
    const int tid = threadIdx.x;
    TrPts pc; pc.lvl = -1;
    for (;;) {
        if (tid < 64) {
            // the 16 command words are read by one instruction; they belong to one command when their sequence numbers agree (a
            // command this workgroup takes part in is stable until it has answered; torn reads only happen on commands it sits out)
            unsigned long long w; unsigned tag, t0; bool ok;
            unsigned spins = 0; long long c0 = 0; bool dead = false;
            do {
                w = tr_ld(&co->cmd[tid & 15]); tag = (unsigned) w;
                t0 = (unsigned) __builtin_amdgcn_readfirstlane((int) tag);
                ok = (__builtin_amdgcn_ballot_w64(tag != t0) == 0ull) && ((int) (t0 - (unsigned) seq) > 0);
                if (!ok) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 1023u) == 0) { const long long now = wall_clock64(); if (c0 == 0) c0 = now; else if (now - c0 > 2 * TR_SPIN_LIMIT) { dead = true; break; } }   // wave-uniform: the leader is gone
                }
            } while (!ok);
            const unsigned pl = (unsigned) (w >> 32);
            if (tid < 12) sRt[tid] = __builtin_bit_cast(float, pl);
            else if (tid < 15) sF[tid - 12] = __builtin_bit_cast(float, pl);
            else if (tid == 15) { sI[0] = dead ? -1 : (int) pl; sI[1] = (int) t0; }
        }
        __syncthreads();
        const int lvl = sI[0];
        seq = sI[1];
        if (lvl < 0) return;
        const int nAct = tr_nact(P.lv[lvl].n, G);
        if (g < nAct) {
            tr_eval(P, lvl, sRt, sF[0], sF[1], sF[2], sAcc, sRed, g, nAct, pc);
            if (tid < TR_NACC) {
                const unsigned long long u = __builtin_bit_cast(unsigned long long, sAcc[tid]);
                tr_st(&co->part[g][tid][0], (u << 32) | (unsigned) seq);
                tr_st(&co->part[g][tid][1], (u & 0xFFFFFFFF00000000ull) | (unsigned) seq);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// trackNewestCoarse: G workgroups per hypothesis (G = 1: one workgroup runs everything)
// ---------------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(TR_NT) void k_tr_track(const TrParams *__restrict__ Pp, TrHyp *hyps, TrCoop *coop, int seq0) {
    const TrParams &P = *Pp;      // device memory, not a by-value argument: a dynamically indexed kernel argument would be copied to scratch memory
    __shared__ double sAcc[TR_NACC];
    __shared__ __attribute__((aligned(16))) float sRed[TR_LDS_FLOATS];
    __shared__ float sRt[12];
    __shared__ double sT[12], sTnew[12];
    __shared__ float sAff[2], sAffNew[2];
    __shared__ double sH[64], sB[8], sNb[8], sInc[8], sResOld[6];
    __shared__ int sCtl[5];          // 0: continue LM loop, 1: accept, 2: abort (return false), 3: iterations, 4: hand-over timed out (error)
#if LD_STAMP_ON_TR
    // debug builds: device timing of tr_eval per level (dynamically indexed -> scratch memory: never in a product build)
    long long tEval = 0, tS1 = 0, tS2 = 0, tS3 = 0, tQ = 0, tTot0 = wall_clock64(), tLv[5] = {0, 0, 0, 0, 0}; int nEval = 0, nLv[5] = {0, 0, 0, 0, 0};
#define TEV(call) do { long long t_ = wall_clock64(); call; t_ = wall_clock64() - t_; tEval += t_; nEval++; if (lvl < 5) { tLv[lvl] += t_; nLv[lvl]++; } if (threadIdx.x == 0) sEv[lvl]++; } while (0)
#else
#define TEV(call) do { call; if (threadIdx.x == 0) sEv[lvl]++; } while (0)
#endif
    __shared__ float sLambda, sCutRep;
    __shared__ int sEv[5];
    __shared__ float sCoF[4];
    if (threadIdx.x < 5) sEv[threadIdx.x] = 0;
    TrHyp &hy = hyps[blockIdx.x / G];
    TrCoop *co = (G > 1) ? coop + blockIdx.x / G : nullptr;
    const int tid = threadIdx.x;
    int coSeq = seq0;
    if (G > 1 && (blockIdx.x % G) != 0) { tr_helper_loop<G>(P, co, (int) (blockIdx.x % G), seq0, sAcc, sRed, sRt, sCoF, sCtl); return; }
    TrPts pc; pc.lvl = -1;
    if (tid < 12) sT[tid] = hy.T[tid];
    if (tid == 0) { sAff[0] = hy.a; sAff[1] = hy.b; sCtl[3] = 0; sCtl[2] = 0; sCtl[4] = 0; for (int i = 0; i < 5; i++) hy.lastResiduals[i] = NAN; for (int i = 0; i < 3; i++) hy.flow[i] = 1000; }
    __syncthreads();
    const int maxIterations[5] = {10, 20, 50, 50, 50};
    const float lambdaExtrapolationLimit = 0.001f;
    bool haveRepeated = false;

    for (int lvl = hy.coarsestLvl; lvl >= 0; lvl--) {
        float levelCutoffRepeat = 1;
        TEV(tr_eval_lead<G>(P, lvl, sT, sAff[0], sAff[1], P.coarseCutoffTH * levelCutoffRepeat, sAcc, sRed, sRt, co, coSeq, pc, &sCtl[4]););
        if (tid == 0) tr_vec6(sAcc, sResOld);
        __syncthreads();
        while (sResOld[5] > 0.6 && levelCutoffRepeat < 50 && !sCtl[4]) {
            levelCutoffRepeat *= 2;
            TEV(tr_eval_lead<G>(P, lvl, sT, sAff[0], sAff[1], P.coarseCutoffTH * levelCutoffRepeat, sAcc, sRed, sRt, co, coSeq, pc, &sCtl[4]););
            if (tid == 0) tr_vec6(sAcc, sResOld);
            __syncthreads();
        }
        tr_hb(sAcc, sH, sB);
        if (tid == 0) sLambda = 0.01f;
        __syncthreads();

        for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
            // the accepted energy ratio, read HERE: thread 0 overwrites sResOld in the accept block below while other wavefronts may still
            // be deciding; the closing barrier of the previous iteration orders this read behind that write, the barrier after the step
            // orders it in front of the next one
            const double oldRatio = sResOld[0] / sResOld[1];
            // H (1 + lambda on the diagonal) x = -b (CoarseTracker.cc:120-128)
#if LD_STAMP_ON_TR
            tQ = wall_clock64();
#endif
            // lane 0 alone: the solve (right-hand side -b, ldlt8_lane), then the step from the increment it has just stored
            if (tid == 0) {
                double xs[8];
                const bool wellConditioned = ldlt8_lane(sH, sB, -1.0, (double) (1 + sLambda), xs);
                for (int i = 0; i < 8; i++) sInc[i] = xs[i];
                if (!wellConditioned) { tr_solve_pivoted8(sH, sB, -1.0, (double) (1 + sLambda), sInc); hy.pivotedSolves++; }
            }
#if LD_STAMP_ON_TR
            tS1 += wall_clock64() - tQ; tQ = wall_clock64();
#endif
            if (tid == 0) {
                sCtl[3]++;
                const float lambda = sLambda;
                double inc[8];
                const bool fixA = P.affineOptModeA < 0, fixB = P.affineOptModeB < 0;
                if (fixA || fixB) { for (int i = 0; i < 8; i++) sNb[i] = -sB[i]; tr_solve_fixed_affine(sH, sB, sNb, lambda, fixA, fixB, sInc); }
                for (int i = 0; i < 8; i++) inc[i] = sInc[i];
                float extrapFac = 1;
                if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf((float) sqrt((double) (lambdaExtrapolationLimit / lambda)));
                double incScaled[8], nrm = 0, sum = 0;
                const double sc[8] = {1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 10.0, 1000.0};
                for (int i = 0; i < 8; i++) { inc[i] *= extrapFac; nrm += inc[i] * inc[i]; incScaled[i] = inc[i] * sc[i]; sum += incScaled[i]; }
                if (!isfinite(sum)) for (int i = 0; i < 8; i++) incScaled[i] = 0;
                // operands into registers first, results stored at the end: a product written straight into LDS orders every load behind each of its
                // stores (same type, possibly the same memory) - twelve dependent LDS round trips for one 3 x 4 product on the LM loop's critical lane
                double E[12], Told[12], Tn[12];
#pragma unroll
                for (int i = 0; i < 12; i++) Told[i] = sT[i];
                const float a0 = sAff[0], b0 = sAff[1];
                ld::se3_exp(incScaled, E);
                ld::se3_mul(E, Told, Tn);
                float an = a0, bn = b0;
                an += incScaled[6]; bn += incScaled[7];
#pragma unroll
                for (int i = 0; i < 12; i++) sTnew[i] = Tn[i];
                sAffNew[0] = an; sAffNew[1] = bn;
                sCtl[0] = (sqrt(nrm) > 1e-3) ? 1 : 0;          // continue after this iteration?
            }
            __syncthreads();
#if LD_STAMP_ON_TR
            tS2 += wall_clock64() - tQ;
#endif
            TEV(tr_eval_lead<G>(P, lvl, sTnew, sAffNew[0], sAffNew[1], P.coarseCutoffTH * levelCutoffRepeat, sAcc, sRed, sRt, co, coSeq, pc, &sCtl[4]););
#if LD_STAMP_ON_TR
            tQ = wall_clock64();
#endif
            if (sCtl[4]) break;
            // accept? (CoarseTracker.cc:168-170) - every thread takes the decision itself from the sums the evaluation left in LDS
            // (the two operands of tr_vec6's entries 0 and 1), instead of one thread publishing it behind a barrier
            const bool accept = (((double) (float) sAcc[0]) / ((double) (int) sAcc[1])) < oldRatio;
            if (accept) tr_hb(sAcc, sH, sB);
            if (tid == 0) {
                if (accept) {
                    // every load in front of the first store (see the step above)
                    double acc6[6], rn[6], Tn[12];
#pragma unroll
                    for (int i = 0; i < 6; i++) acc6[i] = sAcc[i];
#pragma unroll
                    for (int i = 0; i < 12; i++) Tn[i] = sTnew[i];
                    const float an = sAffNew[0], bn = sAffNew[1], lam = sLambda;
                    tr_vec6(acc6, rn);
#pragma unroll
                    for (int i = 0; i < 6; i++) sResOld[i] = rn[i];
                    sAff[0] = an; sAff[1] = bn;
#pragma unroll
                    for (int i = 0; i < 12; i++) sT[i] = Tn[i];
                    sLambda = lam * 0.5f;
                } else {
                    float lam = sLambda * 4;
                    if (lam < lambdaExtrapolationLimit) lam = lambdaExtrapolationLimit;
                    sLambda = lam;
                }
            }
            __syncthreads();
#if LD_STAMP_ON_TR
            tS3 += wall_clock64() - tQ;
#endif
            if (!sCtl[0]) break;
        }
        if (tid == 0) {
            float lr = sqrtf((float) (sResOld[0] / sResOld[1]));
            hy.lastResiduals[lvl] = lr;
            for (int i = 0; i < 3; i++) hy.flow[i] = sResOld[2 + i];
            if (lr > 1.5 * hy.minRes[lvl]) sCtl[2] = 1;
        }
        __syncthreads();
        if (sCtl[2] || sCtl[4]) break;
        if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    }
#if LD_STAMP_ON_TR
    if (tid == 0) { hy.dbg[0] = (double) tEval; hy.dbg[1] = (double) (wall_clock64() - tTot0); hy.dbg[2] = nEval; for (int q = 0; q < 5; q++) hy.dbg[3 + q] = nLv[q] ? (double) tLv[q] / nLv[q] : 0.0; hy.dbg[8] = (double) tS1; hy.dbg[9] = (double) tS2; hy.dbg[10] = (double) tS3; }
#endif
    if (G > 1) {      // release the helper workgroups
        ++coSeq;
        if (tid < 16) tr_st(&co->cmd[tid], ((unsigned long long) 0xFFFFFFFFu << 32) | (unsigned) coSeq);
    }
    if (tid == 0) {
        hy.iterations = sCtl[3];
        for (int q = 0; q < 5; q++) hy.evals[q] = sEv[q];
        if (sCtl[4]) { hy.ok = -2; }      // the host turns this into LDSO_E_HIP
        else if (sCtl[2]) { hy.ok = 0; }
        else {
            for (int i = 0; i < 12; i++) hy.T[i] = sT[i];
            float a = sAff[0], b = sAff[1];
            bool ok = true;
            if ((P.affineOptModeA != 0 && (fabsf(a) > 1.2)) || (P.affineOptModeB != 0 && (fabsf(b) > 200))) ok = false;
            float ra, rb;
            aff_from_to_f(P.ref_exposure, P.new_exposure, P.ref_a, P.ref_b, a, b, ra, rb);
            if ((P.affineOptModeA == 0 && (fabsf(logf(ra)) > 1.5)) || (P.affineOptModeB == 0 && (fabsf(rb) > 200))) ok = false;
            if (ok) { if (P.affineOptModeA < 0) a = 0; if (P.affineOptModeB < 0) b = 0; }
            hy.a = a; hy.b = b;
            hy.ok = ok ? 1 : 0;
        }
    }
}

// stand-alone calcRes / calcGSSSE for the step-wise API (one workgroup)
__global__ __launch_bounds__(TR_NT) void k_tr_calc(const TrParams *__restrict__ Pp, int lvl, const double *Tdev, float a, float b, float cutoffTH, double *outAcc) {
    const TrParams &P = *Pp;
    __shared__ double sAcc[TR_NACC];
    __shared__ __attribute__((aligned(16))) float sRed[TR_LDS_FLOATS];
    __shared__ float sRt[12];
    const int tid = threadIdx.x;
    if (tid < 12) sRt[tid] = (float) Tdev[tid < 9 ? (tid / 3) * 4 + tid % 3 : (tid - 9) * 4 + 3];
    __syncthreads();
    TrPts pc; pc.lvl = -1;
    tr_eval(P, lvl, sRt, a, b, cutoffTH, sAcc, sRed, 0, 1, pc);
    if (threadIdx.x < TR_NACC) outAcc[threadIdx.x] = sAcc[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct ldso_tracker {
    int device = 0, w = 0, h = 0, levels = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    ldso_settings_t settings;
    TrParams P;
    std::vector<void *> allocs;
    float *d_newImg[TR_MAXL] = {nullptr}, *d_refImg[TR_MAXL] = {nullptr};
    float *d_pts = nullptr;
    int *d_next = nullptr;            // per-point list links of the level-0 scatter
    float *d_color = nullptr;          // level-0 irradiance staging of ldso_tr_set_new_frame_image
    int ptsCap = 0;
    int *d_total = nullptr;
    double *d_T = nullptr, *d_acc = nullptr;
    TrHyp *d_hyp = nullptr, *h_hyp = nullptr;      // device records, pinned staging copy (no pageable-memory detour on the per-track round trip)
    TrParams *d_P = nullptr, *h_P = nullptr, Pdev;   // device copy of P (what the kernels read), pinned staging buffer, what the device copy holds
    TrCoop *d_coop = nullptr;         // cooperative evaluation: one record per hypothesis
    int numCU = 256, coopSeq = 1;     // sequence numbers already used by earlier launches on d_coop
    double lastAcc[TR_NACC];
    bool haveAcc = false;
    int lastEvals[5] = {0, 0, 0, 0, 0};      // of hypothesis 0 of the last track call
    int lastPivotedSolves = 0;               // LM solves of the last track call (all hypotheses) that fell back to the pivoted factorisation (ldlt8_lane)
};

template <class T> static int tr_alloc(ldso_tracker *H, T **p, size_t n) {
    void *q = nullptr;
    CHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    // hipMemset on device memory is asynchronous (legacy null stream) and the handles work on NON-BLOCKING streams, which do not order themselves
    // behind it: without the wait a zero-fill that is still queued (the null stream busy with another library's work, e.g. torch's) could land on top
    // of data the handle's first uploads / kernels have already written
    CHK(hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T)));
    CHK(hipStreamSynchronize(nullptr));
    H->allocs.push_back(q);
    *p = (T *) q;
    return LDSO_OK;
}
#define TA(ptr, n) do { int r_ = tr_alloc(H, &(ptr), (n)); if (r_ != LDSO_OK) return r_; } while (0)

// the kernels read TrParams from device memory: upload it when the host copy changed (stream ordered; the callers synchronise the
// stream before they return, so the pinned staging buffer is free again)
static int tr_sync_params(ldso_tracker *H) {
    if (memcmp(&H->P, &H->Pdev, sizeof(TrParams)) == 0) return LDSO_OK;
    CHK(hipStreamSynchronize(H->stream));
    memcpy(H->h_P, &H->P, sizeof(TrParams));
    CHK(hipMemcpyAsync(H->d_P, H->h_P, sizeof(TrParams), hipMemcpyHostToDevice, H->stream));
    H->Pdev = H->P;
    return LDSO_OK;
}

extern "C" {

int ldso_tr_create(int device, int w, int h, int levels, ldso_tracker_t **out) {
    REQ(out && w > 16 && h > 16 && levels >= 1 && levels <= TR_MAXL && (w >> (levels - 1)) >= 8, "ldso_tr_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { ldso_set_error("no HIP device visible"); return LDSO_E_NODEVICE; }
    REQ(device >= 0 && device < ndev, "ldso_tr_create: device index out of range");
    CHK(hipSetDevice(device));
    ldso_tracker *H = new ldso_tracker();
    H->device = device; H->w = w; H->h = h; H->levels = levels;
    ldso_settings_default(&H->settings);
    CHK(hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking));
    H->ownStream = true;
    memset(&H->P, 0, sizeof(H->P));
    H->P.levels = levels;
    for (int l = 0; l < levels; l++) {
        TrLevel &L = H->P.lv[l];
        L.w = w >> l; L.h = h >> l; L.n = 0;
        size_t n = (size_t) L.w * L.h;
        TA(H->d_newImg[l], n * 3); TA(H->d_refImg[l], n * 3);
        L.newImg = H->d_newImg[l]; L.refImg = H->d_refImg[l];
        TA(L.pc_u, n); TA(L.pc_v, n); TA(L.pc_idepth, n); TA(L.pc_color, n);
        // 64 floats of zeroed padding on both sides: the reference's dilation reads one element before / after
        // the image (CoarseTracker.cc:331-345, index i-1-w at i==w) out of its over-allocated buffers
        TA(L.idepth, n + 128); TA(L.wsum, n + 128); TA(L.wsum_bak, n + 128);
        L.idepth += 64; L.wsum += 64; L.wsum_bak += 64;
        TA(L.blockCnt, n / 256 + 2);
    }
    TA(H->d_total, TR_MAXL); TA(H->d_T, 12); TA(H->d_acc, TR_NACC); TA(H->d_hyp, 128); TA(H->d_coop, TR_COOP_SLOTS); TA(H->d_P, 1);
    CHK(hipHostMalloc((void **) &H->h_P, sizeof(TrParams)));
    CHK(hipHostMalloc((void **) &H->h_hyp, 128 * sizeof(TrHyp)));
    memset(&H->Pdev, 0xFF, sizeof(TrParams));
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) H->numCU = pr.multiProcessorCount; }
    *out = H;
    return LDSO_OK;
}

int ldso_tr_destroy(ldso_tracker_t *H) {
    if (!H) return LDSO_OK;
    hipSetDevice(H->device);
    hipDeviceSynchronize();
    for (void *p : H->allocs) hipFree(p);
    if (H->d_pts) hipFree(H->d_pts);
    if (H->d_next) hipFree(H->d_next);
    if (H->d_color) hipFree(H->d_color);
    if (H->h_P) hipHostFree(H->h_P);
    if (H->h_hyp) hipHostFree(H->h_hyp);
    if (H->ownStream && H->stream) hipStreamDestroy(H->stream);
    delete H;
    return LDSO_OK;
}

int ldso_tr_set_stream(ldso_tracker_t *H, void *s) {
    REQ(H, "null handle");
    if (H->ownStream && H->stream) { hipStreamSynchronize(H->stream); if (s) { hipStreamDestroy(H->stream); H->ownStream = false; } }
    if (s) { H->stream = (hipStream_t) s; H->ownStream = false; }
    else if (!H->ownStream) { CHK(hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking)); H->ownStream = true; }
    return LDSO_OK;
}

int ldso_tr_set_settings(ldso_tracker_t *H, const ldso_settings_t *s) {
    REQ(H && s, "null argument");
    H->settings = *s;
    H->P.huberTH = s->huberTH; H->P.coarseCutoffTH = s->coarseCutoffTH; H->P.affineOptModeA = s->affineOptModeA; H->P.affineOptModeB = s->affineOptModeB;
    return LDSO_OK;
}

// CoarseTracker::makeK, float arithmetic as the reference (CoarseTracker.cc:219-246)
int ldso_tr_make_k(ldso_tracker_t *H, const ldso_calib_t *calib) {
    REQ(H && calib, "null argument");
    ldso_tr_set_settings(H, &H->settings);
    float fx[TR_MAXL], fy[TR_MAXL], cx[TR_MAXL], cy[TR_MAXL];
    fx[0] = (float) (50.0 * calib->value[0]); fy[0] = (float) (50.0 * calib->value[1]); cx[0] = (float) (50.0 * calib->value[2]); cy[0] = (float) (50.0 * calib->value[3]);
    for (int l = 1; l < H->levels; l++) {
        fx[l] = (float) (fx[l - 1] * 0.5); fy[l] = (float) (fy[l - 1] * 0.5);
        cx[l] = (float) ((cx[0] + 0.5) / ((int) 1 << l) - 0.5); cy[l] = (float) ((cy[0] + 0.5) / ((int) 1 << l) - 0.5);
    }
    for (int l = 0; l < H->levels; l++) {
        TrLevel &L = H->P.lv[l];
        L.fx = fx[l]; L.fy = fy[l]; L.cx = cx[l]; L.cy = cy[l];
        float K[9] = {fx[l], 0, cx[l], 0, fy[l], cy[l], 0, 0, 1};
        auto cof = [&](int a, int b) { int a1 = (a + 1) % 3, a2 = (a + 2) % 3, b1 = (b + 1) % 3, b2 = (b + 2) % 3; return K[a1 * 3 + b1] * K[a2 * 3 + b2] - K[a1 * 3 + b2] * K[a2 * 3 + b1]; };
        float k0 = cof(0, 0), k1 = cof(1, 0), k2 = cof(2, 0);
        float det = (k0 * K[0] + k1 * K[3]) + k2 * K[6];
        float invdet = 1.0f / det;
        L.Ki[0] = k0 * invdet; L.Ki[1] = k1 * invdet; L.Ki[2] = k2 * invdet;
        L.Ki[3] = cof(0, 1) * invdet; L.Ki[4] = cof(1, 1) * invdet; L.Ki[5] = cof(2, 1) * invdet;
        L.Ki[6] = cof(0, 2) * invdet; L.Ki[7] = cof(1, 2) * invdet; L.Ki[8] = cof(2, 2) * invdet;
    }
    return LDSO_OK;
}

static int tr_set_ref_common(ldso_tracker_t *H, float ref_a, float ref_b, float ref_exposure, const float *pts, int n);

int ldso_tr_set_ref(ldso_tracker_t *H, const float *const *ref_dIp, float ref_a, float ref_b, float ref_exposure, const float *pts, int n) {
    REQ(H && ref_dIp && (pts || n == 0) && n >= 0, "ldso_tr_set_ref: bad arguments");
    CHK(hipSetDevice(H->device));
    for (int l = 0; l < H->levels; l++) {
        REQ(ref_dIp[l], "ldso_tr_set_ref: missing pyramid level");
        size_t bytes = (size_t) H->P.lv[l].w * H->P.lv[l].h * 3 * sizeof(float);
        CHK(hipMemcpyAsync(H->d_refImg[l], ref_dIp[l], bytes, hipMemcpyHostToDevice, H->stream));
        H->P.lv[l].refImg = H->d_refImg[l];
    }
    return tr_set_ref_common(H, ref_a, ref_b, ref_exposure, pts, n);
}

// the reference keyframe's pyramid already resident (ldso_pyramid_t, zero-copy: the tracker reads the pyramid's levels until the next
// ldso_tr_set_ref*; the caller keeps the pyramid alive that long)
int ldso_tr_set_ref_pyramid(ldso_tracker_t *H, ldso_pyramid_t *pyr, float ref_a, float ref_b, float ref_exposure, const float *pts, int n) {
    REQ(H && pyr && (pts || n == 0) && n >= 0, "ldso_tr_set_ref_pyramid: bad arguments");
    REQ(pyr->built && pyr->device == H->device && pyr->w == H->w && pyr->h == H->h && pyr->levels >= H->levels, "ldso_tr_set_ref_pyramid: pyramid does not match the tracker (device, size, levels) or holds no image");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamWaitEvent(H->stream, pyr->ready, 0));
    for (int l = 0; l < H->levels; l++) H->P.lv[l].refImg = pyr->lv[l];
    return tr_set_ref_common(H, ref_a, ref_b, ref_exposure, pts, n);
}

static int tr_set_ref_common(ldso_tracker_t *H, float ref_a, float ref_b, float ref_exposure, const float *pts, int n) {
    H->P.ref_a = ref_a; H->P.ref_b = ref_b; H->P.ref_exposure = ref_exposure;
    if (n > H->ptsCap) {
        if (H->d_pts) hipFree(H->d_pts);
        if (H->d_next) hipFree(H->d_next);
        H->d_pts = nullptr; H->d_next = nullptr; H->ptsCap = 0;
        void *q; CHK(hipMalloc(&q, (size_t) n * 16)); H->d_pts = (float *) q;
        CHK(hipMalloc(&q, (size_t) n * 4)); H->d_next = (int *) q;
        H->ptsCap = n;
    }
    if (n) CHK(hipMemcpyAsync(H->d_pts, pts, (size_t) n * 16, hipMemcpyHostToDevice, H->stream));
    // makeCoarseDepthL0
    TrLevel *lv = H->P.lv;
    CHK(hipMemsetAsync(lv[0].idepth, 0, (size_t) lv[0].w * lv[0].h * 4, H->stream));
    CHK(hipMemsetAsync(lv[0].wsum, 0, (size_t) lv[0].w * lv[0].h * 4, H->stream));
    if (n) {
        int *head = reinterpret_cast<int *>(lv[0].wsum_bak);        // free until the dilation below
        CHK(hipMemsetAsync(head, 0xFF, (size_t) lv[0].w * lv[0].h * 4, H->stream));
        hipLaunchKernelGGL(k_tr_scatter_link, dim3((n + 255) / 256), dim3(256), 0, H->stream, H->d_pts, n, head, H->d_next, lv[0].w, lv[0].h);
        hipLaunchKernelGGL(k_tr_scatter, dim3((n + 255) / 256), dim3(256), 0, H->stream, H->d_pts, n, lv[0].idepth, lv[0].wsum, head, H->d_next, lv[0].w, lv[0].h);
    }
    for (int l = 1; l < H->levels; l++) {
        int npx = lv[l].w * lv[l].h;
        hipLaunchKernelGGL(k_tr_pool, dim3((npx + 255) / 256), dim3(256), 0, H->stream, lv[l - 1].idepth, lv[l - 1].wsum, lv[l].idepth, lv[l].wsum, lv[l].w, lv[l].h, lv[l - 1].w);
    }
    for (int l = 0; l < H->levels; l++) {
        int npx = lv[l].w * lv[l].h;
        CHK(hipMemcpyAsync(lv[l].wsum_bak, lv[l].wsum, (size_t) npx * 4, hipMemcpyDeviceToDevice, H->stream));
        hipLaunchKernelGGL(k_tr_dilate, dim3((npx + 255) / 256), dim3(256), 0, H->stream, lv[l].idepth, lv[l].wsum, lv[l].wsum_bak, lv[l].w, lv[l].h, l < 2 ? 1 : 0);
    }
    for (int l = 0; l < H->levels; l++) {
        int ni = (lv[l].w - 4) * (lv[l].h - 4);
        int nb = (ni + 255) / 256;
        hipLaunchKernelGGL(k_tr_count, dim3(nb), dim3(256), 0, H->stream, lv[l]);
        hipLaunchKernelGGL(k_tr_scan, dim3(1), dim3(1024), 0, H->stream, lv[l].blockCnt, nb, H->d_total + l);
        hipLaunchKernelGGL(k_tr_write, dim3(nb), dim3(256), 0, H->stream, lv[l]);
    }
    int tot[TR_MAXL] = {0};
    CHK(hipMemcpyAsync(tot, H->d_total, (size_t) H->levels * 4, hipMemcpyDeviceToHost, H->stream));      // one read-back for all levels
    CHK(hipStreamSynchronize(H->stream));
    for (int l = 0; l < H->levels; l++) lv[l].n = tot[l];
    CHK(hipGetLastError());
    return LDSO_OK;
}

int ldso_tr_set_new_frame(ldso_tracker_t *H, const float *const *new_dIp, float exposure) {
    REQ(H && new_dIp, "ldso_tr_set_new_frame: bad arguments");
    CHK(hipSetDevice(H->device));
    for (int l = 0; l < H->levels; l++) {
        REQ(new_dIp[l], "ldso_tr_set_new_frame: missing pyramid level");
        size_t bytes = (size_t) H->P.lv[l].w * H->P.lv[l].h * 3 * sizeof(float);
        CHK(hipMemcpyAsync(H->d_newImg[l], new_dIp[l], bytes, hipMemcpyHostToDevice, H->stream));
        H->P.lv[l].newImg = H->d_newImg[l];
    }
    H->P.new_exposure = exposure;
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

hipError_t img_launch_make_images(const float *d_color, int w, int h, int levels, float *const *d_levels, hipStream_t st);

// CoarseTracker's new frame from the raw level-0 irradiance: FrameHessian::makeImages runs on the device (images.hip), one
// w*h float upload instead of the 12-byte AoS pyramid
int ldso_tr_set_new_frame_image(ldso_tracker_t *H, const float *irradiance, float exposure) {
    REQ(H && irradiance, "ldso_tr_set_new_frame_image: bad arguments");
    CHK(hipSetDevice(H->device));
    const size_t n = (size_t) H->w * H->h;
    if (!H->d_color) CHK(hipMalloc(&H->d_color, n * sizeof(float)));
    CHK(hipMemcpyAsync(H->d_color, irradiance, n * sizeof(float), hipMemcpyHostToDevice, H->stream));
    CHK(img_launch_make_images(H->d_color, H->w, H->h, H->levels, H->d_newImg, H->stream));
    for (int l = 0; l < H->levels; l++) H->P.lv[l].newImg = H->d_newImg[l];
    H->P.new_exposure = exposure;
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

// the frame to be tracked as a resident ldso_pyramid_t (zero-copy; stream-ordered after the pyramid's build, no host synchronisation)
int ldso_tr_set_new_frame_pyramid(ldso_tracker_t *H, ldso_pyramid_t *pyr, float exposure) {
    REQ(H && pyr, "ldso_tr_set_new_frame_pyramid: bad arguments");
    REQ(pyr->built && pyr->device == H->device && pyr->w == H->w && pyr->h == H->h && pyr->levels >= H->levels, "ldso_tr_set_new_frame_pyramid: pyramid does not match the tracker (device, size, levels) or holds no image");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamWaitEvent(H->stream, pyr->ready, 0));
    for (int l = 0; l < H->levels; l++) H->P.lv[l].newImg = pyr->lv[l];
    H->P.new_exposure = exposure;
    return LDSO_OK;
}

// debug / test fetch of a level of the new frame's pyramid ((w>>lvl)*(h>>lvl)*3 floats)
int ldso_tr_get_new_frame_level(ldso_tracker_t *H, int lvl, float *out) {
    REQ(H && out && lvl >= 0 && lvl < H->levels, "ldso_tr_get_new_frame_level: bad arguments");
    CHK(hipSetDevice(H->device));
    CHK(hipMemcpyAsync(out, H->P.lv[lvl].newImg, (size_t) H->P.lv[lvl].w * H->P.lv[lvl].h * 3 * sizeof(float), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

static int tr_calc(ldso_tracker *H, int lvl, const double *T, float a, float b, float cutoff) {
    CHK(hipMemcpyAsync(H->d_T, T, 12 * 8, hipMemcpyHostToDevice, H->stream));
    { const int r_ = tr_sync_params(H); if (r_ != LDSO_OK) return r_; }
    hipLaunchKernelGGL(k_tr_calc, dim3(1), dim3(TR_NT), 0, H->stream, H->d_P, lvl, H->d_T, a, b, cutoff, H->d_acc);
    CHK(hipGetLastError());
    CHK(hipMemcpyAsync(H->lastAcc, H->d_acc, TR_NACC * 8, hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    H->haveAcc = true;
    return LDSO_OK;
}

int ldso_tr_calc_res(ldso_tracker_t *H, int lvl, const double T[12], float a, float b, float cutoffTH, double rs[6], int *n_warped) {
    REQ(H && T && rs && lvl >= 0 && lvl < H->levels, "ldso_tr_calc_res: bad arguments");
    CHK(hipSetDevice(H->device));
    int r = tr_calc(H, lvl, T, a, b, cutoffTH);
    if (r != LDSO_OK) return r;
    const double *acc = H->lastAcc;
    rs[0] = (double) (float) acc[0]; rs[1] = (double) (int) acc[1];
    rs[2] = (double) ((float) acc[2] / ((float) acc[4] + 0.1f)); rs[3] = 0; rs[4] = (double) ((float) acc[3] / ((float) acc[4] + 0.1f));
    rs[5] = (double) ((float) (int) acc[5] / (float) (int) acc[1]);
    if (n_warped) *n_warped = ((int) acc[6] + 3) / 4 * 4;
    return LDSO_OK;
}

int ldso_tr_calc_gs(ldso_tracker_t *H, int lvl, const double T[12], float a, float b, double Hout[64], double bout[8]) {
    REQ(H && T && Hout && bout && lvl >= 0 && lvl < H->levels, "ldso_tr_calc_gs: bad arguments");
    REQ(H->haveAcc, "ldso_tr_calc_gs: call ldso_tr_calc_res first (the reference reuses the warped buffers of the last calcRes)");
    const double *acc = H->lastAcc;
    int nw = (int) acc[6];
    int npad = (nw + 3) / 4 * 4;
    double inv = (double) (1.0f / (float) npad);
    const double cs[8] = {1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 10.0, 1000.0};
    double M[81];
    int q = 7;
    for (int r = 0; r < 9; r++) for (int c = r; c < 9; c++) { double v = (double) (float) acc[q++]; M[r * 9 + c] = v; M[c * 9 + r] = v; }
    for (int r = 0; r < 8; r++) { for (int c = 0; c < 8; c++) Hout[r * 8 + c] = M[r * 9 + c] * inv * cs[r] * cs[c]; bout[r] = M[r * 9 + 8] * inv * cs[r]; }
    return LDSO_OK;
}

// Cooperative launches (G > 1) spin-wait across workgroups: forward progress needs every workgroup of the launch resident.  One launch
// alone is (nhyp * G <= #CUs, one 256-thread workgroup per CU whatever else runs: other kernels finish and free their CUs); two such
// launches from different handles could each hold CUs with spinning leaders while the other's helpers wait for a CU.  Within a process
// they are therefore chained per device through an event; across processes (or under a CU mask) the bounded spins of the kernel turn a
// would-be hang into LDSO_E_HIP.
static std::mutex g_coopMutex;
static hipEvent_t g_coopLast[64] = {nullptr};
static int tr_coop_chain_begin(ldso_tracker_t *H) {
    if (H->device < 0 || H->device >= 64) return LDSO_OK;
    hipEvent_t &e = g_coopLast[H->device];
    if (!e) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    else CHK(hipStreamWaitEvent(H->stream, e, 0));
    return LDSO_OK;
}

int ldso_tr_track_batch(ldso_tracker_t *H, int nhyp, double *T_inout /*nhyp*12*/, float *aff_inout /*nhyp*2*/, int coarsestLvl, const double minRes[5],
                        double *lastResiduals /*nhyp*5*/, double *flow /*nhyp*3*/, int *ok /*nhyp*/, int *iterations /*nhyp*/) {
    REQ(H && nhyp >= 1 && nhyp <= 128 && T_inout && aff_inout && coarsestLvl >= 0 && coarsestLvl < 5 && coarsestLvl < H->levels, "ldso_tr_track: bad arguments");
    CHK(hipSetDevice(H->device));
    TrHyp *hy = H->h_hyp;             // the stream is synchronised before this function returns: the buffer is free again
    for (int i = 0; i < nhyp; i++) {
        memset(&hy[i], 0, sizeof(TrHyp));
        memcpy(hy[i].T, T_inout + i * 12, 96);
        hy[i].a = aff_inout[2 * i]; hy[i].b = aff_inout[2 * i + 1]; hy[i].coarsestLvl = coarsestLvl;
        for (int k = 0; k < 5; k++) hy[i].minRes[k] = minRes ? minRes[k] : NAN;
    }
    CHK(hipMemcpyAsync(H->d_hyp, hy, nhyp * sizeof(TrHyp), hipMemcpyHostToDevice, H->stream));
    // few hypotheses: TR_GMAX workgroups share each of them on the large levels (all workgroups resident: nhyp * G <= CUs);
    // many hypotheses fill the chip by themselves
    { const int r_ = tr_sync_params(H); if (r_ != LDSO_OK) return r_; }
    const int G = getenv("LDSO_TR_NO_COOP") ? 1 : nhyp * 16 <= H->numCU ? 16 : nhyp * 12 <= H->numCU ? 12 : nhyp * 8 <= H->numCU ? 8 : nhyp * 4 <= H->numCU ? 4 : 1;
    if (G > 1) {
        std::lock_guard<std::mutex> lk(g_coopMutex);
        { const int r_ = tr_coop_chain_begin(H); if (r_ != LDSO_OK) return r_; }
        if (H->coopSeq > (1 << 30)) { CHK(hipMemsetAsync(H->d_coop, 0, TR_COOP_SLOTS * sizeof(TrCoop), H->stream)); H->coopSeq = 1; }
        if (G == 16) hipLaunchKernelGGL(k_tr_track<16>, dim3(nhyp * 16), dim3(TR_NT), 0, H->stream, H->d_P, H->d_hyp, H->d_coop, H->coopSeq);
        else if (G == 12) hipLaunchKernelGGL(k_tr_track<12>, dim3(nhyp * 12), dim3(TR_NT), 0, H->stream, H->d_P, H->d_hyp, H->d_coop, H->coopSeq);
        else if (G == 8) hipLaunchKernelGGL(k_tr_track<8>, dim3(nhyp * 8), dim3(TR_NT), 0, H->stream, H->d_P, H->d_hyp, H->d_coop, H->coopSeq);
        else hipLaunchKernelGGL(k_tr_track<4>, dim3(nhyp * 4), dim3(TR_NT), 0, H->stream, H->d_P, H->d_hyp, H->d_coop, H->coopSeq);
        H->coopSeq += 1024;           // more than the evaluations of one track (5 levels x (50 iterations + 7 cut-off repeats) + 1)
        if (H->device >= 0 && H->device < 64) CHK(hipEventRecord(g_coopLast[H->device], H->stream));
    } else {
        hipLaunchKernelGGL(k_tr_track<1>, dim3(nhyp), dim3(TR_NT), 0, H->stream, H->d_P, H->d_hyp, (TrCoop *) nullptr, 0);
    }
    CHK(hipGetLastError());
    CHK(hipMemcpyAsync(hy, H->d_hyp, nhyp * sizeof(TrHyp), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    for (int i = 0; i < nhyp; i++) if (hy[i].ok == -2) {
        ldso_set_error("ldso_tr_track: the cooperating workgroups of a hypothesis did not become co-resident (device shared with another process or CU-masked?); set LDSO_TR_NO_COOP=1");
        return LDSO_E_HIP;
    }
    for (int i = 0; i < nhyp; i++) {
        memcpy(T_inout + i * 12, hy[i].T, 96);
        aff_inout[2 * i] = hy[i].a; aff_inout[2 * i + 1] = hy[i].b;
        if (lastResiduals) memcpy(lastResiduals + i * 5, hy[i].lastResiduals, 40);
        if (flow) memcpy(flow + i * 3, hy[i].flow, 24);
        if (ok) ok[i] = hy[i].ok;
        if (iterations) iterations[i] = hy[i].iterations;
        if (i == 0) { memcpy(H->lastEvals, hy[i].evals, sizeof(H->lastEvals)); H->lastPivotedSolves = 0; }
        H->lastPivotedSolves += hy[i].pivotedSolves;
#if LD_STAMP_ON_TR
        if (i == 0) { long long ph[5][8]; hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_trPh), sizeof(ph)); long long z[5][8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_trPh), z, sizeof(z));
            for (int l = 0; l < 5; l++) if (ph[l][7]) fprintf(stderr, "[tr phases] lvl %d: setup %.2f pass %.2f dpp %.2f barrier %.2f sum %.2f us per eval (%d evals)\n", l, ph[l][0] / 100.0 / ph[l][7], ph[l][1] / 100.0 / ph[l][7], ph[l][2] / 100.0 / ph[l][7], ph[l][3] / 100.0 / ph[l][7], ph[l][4] / 100.0 / ph[l][7], (int) ph[l][7]); }
#endif
        if (LD_STAMP_ON_TR && i == 0) fprintf(stderr, "[tr stamps] evals %d: %.1f us in tr_eval of %.1f us kernel; per-eval us by level 0..4: %.1f %.1f %.1f %.1f %.1f; solve %.1f step %.1f post %.1f us\n", (int) hy[i].dbg[2], hy[i].dbg[0] / 100.0, hy[i].dbg[1] / 100.0, hy[i].dbg[3] / 100, hy[i].dbg[4] / 100, hy[i].dbg[5] / 100, hy[i].dbg[6] / 100, hy[i].dbg[7] / 100, hy[i].dbg[8] / 100, hy[i].dbg[9] / 100, hy[i].dbg[10] / 100);
    }
    return LDSO_OK;
}

// The hypothesis loop of FullSystem::trackNewCoarse (FullSystem.cc:319-356) replayed on the results of ONE
// ldso_tr_track_batch call that ran every try to the end (minRes = NaN): try i is accepted / aborted exactly as the
// sequential loop would have done with the `achievedRes` of the tries before it - a try whose residual on some level exceeds
// 1.5 x achievedRes there counts as aborted at that level (finer levels NaN, trackingIsGood = false), `achievedRes` is taken
// over "always" once one try was good, and the loop stops at the first try with achievedRes[0] < lastCoarseRMSE0 *
// reTrackThreshold.  Pure host function (no device work).  best = -1: tracking failed entirely.
int ldso_tr_select_hypothesis(int nhyp, int coarsestLvl, const double *lastResiduals /*nhyp*5*/, const int *ok /*nhyp*/, double lastCoarseRMSE0,
                              double reTrackThreshold, int *best, int *tries_consumed, double achievedRes_out[5]) {
    if (nhyp < 0 || coarsestLvl < 0 || coarsestLvl > 4 || (nhyp > 0 && (!lastResiduals || !ok)) || !best) { ldso_set_error("ldso_tr_select_hypothesis: bad arguments"); return LDSO_E_INVALID; }
    double achieved[5] = {NAN, NAN, NAN, NAN, NAN};
    bool haveOneGood = false;
    int tries = 0, win = -1;
    for (int i = 0; i < nhyp; i++) {
        double lr[5] = {NAN, NAN, NAN, NAN, NAN};
        bool good = ok[i] != 0;
        for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
            lr[lvl] = lastResiduals[i * 5 + lvl];
            if (lr[lvl] > 1.5 * achieved[lvl]) { good = false; break; }          // CoarseTracker.cc:193-200 (false with a NaN threshold)
        }
        tries++;
        if (good && std::isfinite((float) lr[0]) && !(lr[0] >= achieved[0])) { win = i; haveOneGood = true; }
        if (haveOneGood)
            for (int l = 0; l < 5; l++) if (!std::isfinite((float) achieved[l]) || achieved[l] > lr[l]) achieved[l] = lr[l];
        if (haveOneGood && achieved[0] < lastCoarseRMSE0 * reTrackThreshold) break;
    }
    *best = win;
    if (tries_consumed) *tries_consumed = tries;
    if (achievedRes_out) for (int l = 0; l < 5; l++) achievedRes_out[l] = achieved[l];
    return LDSO_OK;
}

// The motion-hypothesis list of FullSystem::trackNewCoarse (FullSystem.cc:189-309) from the worldToCam poses (Frame::getPose()) of the two
// frames before the new one in allFrameHistory (sprelast, slast) and of the tracker's reference key frame (lastF): constant / double / half /
// zero motion, identity, and 3 x 26 small rotations about the constant-motion guess (rotDelta = 0.02, 0.03, 0.04 as the reference's float
// counter produces them; Sophus' SO3 constructor normalises the quaternion (1, +-d, +-d, +-d)).  Pure host function.
static void tr_quat_to_pose(double w, double x, double y, double z, double *T) {
    const double n = sqrt(w * w + x * x + y * y + z * z);
    w /= n; x /= n; y /= n; z /= n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    T[0] = 1 - (tyy + tzz); T[1] = txy - twz; T[2] = txz + twy; T[3] = 0;
    T[4] = txy + twz; T[5] = 1 - (txx + tzz); T[6] = tyz - twx; T[7] = 0;
    T[8] = txz - twy; T[9] = tyz + twx; T[10] = 1 - (txx + tyy); T[11] = 0;
}
int ldso_tr_motion_hypotheses(const double sprelast[12], const double slast[12], const double lastF[12], int poses_valid, double *out, int *n_out) {
    REQ(sprelast && slast && lastF && out && n_out, "ldso_tr_motion_hypotheses: null argument");
    const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (!poses_valid) { memcpy(out, I, sizeof(I)); *n_out = 1; return LDSO_OK; }       // FullSystem.cc:306-309
    double inv[12], fh_2_slast[12], lastF_2_slast[12], fi[12], cm[12], tmp[12], xi[6], h[12];
    ld::se3_inv(slast, inv); ld::se3_mul(sprelast, inv, fh_2_slast);                    // slast_2_sprelast, "assumed to be the same as fh_2_slast"
    ld::se3_inv(lastF, inv); ld::se3_mul(slast, inv, lastF_2_slast);
    ld::se3_inv(fh_2_slast, fi);
    int n = 0;
    ld::se3_mul(fi, lastF_2_slast, cm); memcpy(out + 12 * n++, cm, 96);                 // constant motion
    ld::se3_mul(fi, cm, tmp); memcpy(out + 12 * n++, tmp, 96);                          // double motion (a frame was skipped)
    ld::se3_log(fh_2_slast, xi); for (int i = 0; i < 6; i++) xi[i] *= 0.5;
    ld::se3_exp(xi, h); ld::se3_inv(h, tmp); ld::se3_mul(tmp, lastF_2_slast, h); memcpy(out + 12 * n++, h, 96);      // half motion
    memcpy(out + 12 * n++, lastF_2_slast, 96);                                          // zero motion
    memcpy(out + 12 * n++, I, 96);                                                      // zero motion from the key frame
    static const int sgn[26][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}, {-1, 1, 0}, {0, -1, 1}, {-1, 0, 1},
                                      {1, -1, 0}, {0, 1, -1}, {1, 0, -1}, {-1, -1, 0}, {0, -1, -1}, {-1, 0, -1}, {-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1},
                                      {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};
    for (float rotDelta = 0.02; rotDelta < 0.05; rotDelta += 0.01)
        for (int k = 0; k < 26; k++) {
            double q[12];
            tr_quat_to_pose(1, sgn[k][0] * rotDelta, sgn[k][1] * rotDelta, sgn[k][2] * rotDelta, q);
            ld::se3_mul(cm, q, out + 12 * n++);
        }
    *n_out = n;
    return LDSO_OK;
}

// Vec4 FullSystem::trackNewCoarse (FullSystem.cc:179-386) on a tracker whose reference and new frame are set.  The reference runs its tries
// one after the other and stops at the first one that is good enough (:355) - almost always the first.  Here: try 0 alone (cooperative
// single-hypothesis launch); only if the loop would go on, ALL remaining tries in one batched launch, and ldso_tr_select_hypothesis replays
// the sequential accept / abort / early-exit decisions on the residuals (a try the sequential loop would have aborted on a coarse level
// counts as aborted).  lastCoarseRMSE: in / out (FullSystem::lastCoarseRMSE); result4 = (achievedRes[0], flow[0..2]); new_w2c = the pose
// handed to the new frame (:376-377), aff_out its aff_g2l; good = 0 is the reference's "tracking failed entirely" branch (:359-365).
int ldso_tr_track_new_coarse(ldso_tracker_t *H, const double sprelast[12], const double slast[12], const double lastF[12], int poses_valid, const float aff_last[2],
                             double lastCoarseRMSE[5], double reTrackThreshold, double result4[4], double new_w2c[12], float aff_out[2], int *tries_consumed, int *good) {
    REQ(H && aff_last && lastCoarseRMSE && result4 && new_w2c && aff_out, "ldso_tr_track_new_coarse: null argument");
    std::vector<double> T(83 * 12), T0(83 * 12), lr(83 * 5), flow(83 * 3);
    std::vector<float> aff(83 * 2);
    std::vector<int> ok(83);
    int n = 0;
    RUN(ldso_tr_motion_hypotheses(sprelast, slast, lastF, poses_valid, T0.data(), &n));
    T = T0;
    for (int i = 0; i < n; i++) { aff[2 * i] = aff_last[0]; aff[2 * i + 1] = aff_last[1]; }
    const int coarsest = H->levels - 1;
    int best = -1, used = 0;
    double achieved[5];
    RUN(ldso_tr_track_batch(H, 1, T.data(), aff.data(), coarsest, nullptr, lr.data(), flow.data(), ok.data(), nullptr));
    RUN(ldso_tr_select_hypothesis(1, coarsest, lr.data(), ok.data(), lastCoarseRMSE[0], reTrackThreshold, &best, &used, achieved));
    const bool done = best == 0 && achieved[0] < lastCoarseRMSE[0] * reTrackThreshold;
    if (!done && n > 1) {
        RUN(ldso_tr_track_batch(H, n - 1, T.data() + 12, aff.data() + 2, coarsest, nullptr, lr.data() + 5, flow.data() + 3, ok.data() + 1, nullptr));
        RUN(ldso_tr_select_hypothesis(n, coarsest, lr.data(), ok.data(), lastCoarseRMSE[0], reTrackThreshold, &best, &used, achieved));
    }
    double lastF_2_fh[12];
    if (best >= 0) {
        memcpy(lastF_2_fh, T.data() + 12 * best, 96);
        aff_out[0] = aff[2 * best]; aff_out[1] = aff[2 * best + 1];
        for (int i = 0; i < 3; i++) result4[1 + i] = flow[3 * best + i];
    } else {
        memcpy(lastF_2_fh, T0.data(), 96);
        aff_out[0] = aff_last[0]; aff_out[1] = aff_last[1];
        result4[1] = result4[2] = result4[3] = 0;
    }
    result4[0] = achieved[0];
    for (int l = 0; l < 5; l++) lastCoarseRMSE[l] = achieved[l];
    // camToWorld = lastF^-1 * lastF_2_fh^-1, the frame's pose is its inverse (:376-377) = lastF_2_fh * lastF
    ld::se3_mul(lastF_2_fh, lastF, new_w2c);
    if (tries_consumed) *tries_consumed = used;
    if (good) *good = best >= 0 ? 1 : 0;
    return LDSO_OK;
}

// calcRes evaluations per level of the last ldso_tr_track / hypothesis 0 of the last batch, and the reference point counts pc_n:
// the algorithmic bytes of that track are sum_l evals[l] * pc_n[l] * 64 (16 B point + 48 B taps per evaluation)
int ldso_tr_last_track_evals(ldso_tracker_t *H, int evals[5], int pc_n[5]) {
    REQ(H && evals && pc_n, "null argument");
    for (int l = 0; l < 5; l++) { evals[l] = H->lastEvals[l]; pc_n[l] = (l < H->levels) ? H->P.lv[l].n : 0; }
    return LDSO_OK;
}

// LM solves of the last ldso_tr_track / ldso_tr_track_batch call whose 8 x 8 system was rank-deficient to float precision (a pivot below 1e-6 of the diagonal entry it started from) and went
// through the reference's pivoted LDL^T instead of the unpivoted register version (0 on any normal track)
int ldso_tr_last_track_pivoted_solves(ldso_tracker_t *H, int *n) {
    REQ(H && n, "null argument");
    *n = H->lastPivotedSolves;
    return LDSO_OK;
}

// The LM solve of k_tr_track on its own (debug / test entry): H (1 + lambda on the diagonal = diag_scale) x = -b by one lane, exactly the code path of the kernel -
// the unpivoted register factorisation and, when it reports a rank-deficient system, the pivoted one.
__global__ void k_tr_solve8(const double *H, const double *b, double diagScale, double *x, int *pivoted) {
    __shared__ double sH[64], sB[8], sInc[8];
    if (threadIdx.x < 64) sH[threadIdx.x] = H[threadIdx.x];
    if (threadIdx.x < 8) sB[threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        double xs[8];
        const bool wellConditioned = ldlt8_lane(sH, sB, -1.0, diagScale, xs);
        for (int i = 0; i < 8; i++) sInc[i] = xs[i];
        if (!wellConditioned) tr_solve_pivoted8(sH, sB, -1.0, diagScale, sInc);
        *pivoted = wellConditioned ? 0 : 1;
        for (int i = 0; i < 8; i++) x[i] = sInc[i];
    }
}
int ldso_tr_debug_solve8(const double H[64], const double b[8], double diag_scale, double x[8], int *pivoted) {
    REQ(H && b && x && pivoted, "ldso_tr_debug_solve8: null argument");
    double *d = nullptr;
    CHK(hipMalloc((void **) &d, (64 + 8 + 8 + 1) * sizeof(double)));
    CHK(hipMemcpy(d, H, 64 * sizeof(double), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d + 64, b, 8 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_tr_solve8, dim3(1), dim3(64), 0, 0, d, d + 64, diag_scale, d + 72, (int *) (d + 80));
    CHK(hipGetLastError());
    CHK(hipMemcpy(x, d + 72, 8 * sizeof(double), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(pivoted, d + 80, sizeof(int), hipMemcpyDeviceToHost));
    CHK(hipFree(d));
    return LDSO_OK;
}

int ldso_tr_track(ldso_tracker_t *H, double T[12], float aff[2], int coarsestLvl, const double minRes[5], double lastResiduals[5], double flow[3], int *ok, int *iterations) {
    return ldso_tr_track_batch(H, 1, T, aff, coarsestLvl, minRes, lastResiduals, flow, ok, iterations);
}

int ldso_tr_get_pc(ldso_tracker_t *H, int lvl, float *u, float *v, float *idepth, float *color, int *n) {
    REQ(H && lvl >= 0 && lvl < H->levels, "ldso_tr_get_pc: bad arguments");
    CHK(hipSetDevice(H->device));
    const TrLevel &L = H->P.lv[lvl];
    if (n) *n = L.n;
    size_t bytes = (size_t) L.n * 4;
    if (u) CHK(hipMemcpyAsync(u, L.pc_u, bytes, hipMemcpyDeviceToHost, H->stream));
    if (v) CHK(hipMemcpyAsync(v, L.pc_v, bytes, hipMemcpyDeviceToHost, H->stream));
    if (idepth) CHK(hipMemcpyAsync(idepth, L.pc_idepth, bytes, hipMemcpyDeviceToHost, H->stream));
    if (color) CHK(hipMemcpyAsync(color, L.pc_color, bytes, hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

}  // extern "C"
