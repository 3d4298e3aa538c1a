// initializer.hip — the monocular initialiser CoarseInitializer (reference src/frontend/CoarseInitializer.cc) for gfx950.
//
// One trackFrame (:40-178) is a Levenberg-Marquardt loop per pyramid level over (pose, affine, one inverse depth per point)
// with the depths eliminated point-wise (JbBuffer, :287-297,:363-386).  Device design:
//   k_ini_eval   grid kernel, 8 lanes per point (one lane per pattern pixel, as the BA linearise kernel): the lazily applied
//                applyStep (:673-687) of the previously accepted step, resetPoints' per-point part (:625-626) on the first
//                evaluation of a level, doStep (:645-671), then calcResAndGS (:181-405) and the calcEC sums (:412-428).
//                Per-lane fp32 accumulators (45 entries of the 9x9 Hessian, 8x9+1 of the Schur block), one partial row per block.
//   k_ini_ctl    ONE workgroup: sums the partial rows in double, takes the accept/reject decision of :120-146, keeps the LM
//                state (lambda, fails, iteration, level) in device memory, solves the damped 6x6 / 8x8 system with a
//                lane-parallel pivoted LDL^T, and runs everything that is sequential in the reference: optReg (:430-459),
//                propagateDown/Up (:462-522) and the neighbour average of resetPoints (:629-641).
//   k_ini_prep   grid kernel between the two: the per-lane inputs of the optReg sweep that follows if the step is accepted.
// The host enqueues BEGIN + the maximal number of (eval, prep, ctl) triples once; finished levels / a finished frame make the
// remaining launches return at once (device-side `done`), so there is no host round trip inside trackFrame.
//
// optReg and the resetPoints average update points IN PLACE in index order, reading neighbours that may already have been
// updated.  That order is kept exactly: the host builds, per level, a schedule of passes such that every neighbour with a
// lower index sits in an earlier pass and every reader with a lower index in an earlier-or-equal pass; one wavefront executes
// the passes with the working values in LDS (reads before writes within a pass by lock-step execution).  The dependency
// depth of that order (a diagonal front through the raster: 200-590 passes per level at 640 x 480, 20-40 points wide) is what
// a sweep costs, so a pass is made as short as it can be: optReg with two lanes per point and a fixed selection instead of a
// count-dependent one (ini_sw2_point: 28 instructions, 0.11 us per pass; round 4: one lane, a 10-key sorting network, 0.38 us).
#include <hip/hip_runtime.h>
#include <vector>
#include <string>
#include <cstring>
#include <cmath>
#include <algorithm>
#include "../../include/ldso_hip.h"
#include "lie_dev.h"

void ldso_set_error(const std::string &s);
extern "C" hipError_t img_launch_make_images(const float *d_color, int w, int h, int levels, float *const *d_levels, hipStream_t st);
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ldso_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return LDSO_E_HIP; } } while (0)
#define REQ(cond, msg) do { if (!(cond)) { ldso_set_error(msg); return LDSO_E_INVALID; } } while (0)

#define INI_MAXL 5            // maxIterations[] has five entries (CoarseInitializer.cc:43)
#define INI_NT 256            // threads of an eval block: 32 points x 8 pattern pixels
#define INI_MAXBLK 256        // eval blocks (= partial rows) per launch
#define INI_NPART 128         // floats per partial row
#define INI_CT 1024           // threads of the control block (its per-point passes are latency bound: many loads in flight)
#define INI_NB 12             // neighbour row pitch (10 used)
// partial row layout
#define PR_SC 45              // 8 x 9 Schur block
#define PR_SC88 117
#define PR_E 118
#define PR_ECO 119
#define PR_ECN 120
#define PR_ECC 121
#define PR_N 122

struct IniLevel {
    int n, w, h, nPass;
    float fx, fy, cx, cy;
    double Ki[9];
    const float *first, *cur;           // dIp[lvl] of the first and of the new frame
    float *u, *v, *idepth, *idepth_new, *iR, *iRSumNum, *lastHessian, *lastHessian_new, *maxstep, *energy0, *energy1, *energy_new0, *energy_new1, *outlierTH;
    int *isGood, *isGood_new, *parent, *nb;
    float *jb[2];                       // [n][10]
    // resetPoints sweep (top level only; one lane per point):
    const int *sched;                   // [nPass][64] point index or -1
    const int *schedOff;                // [nPass][64] LDS byte offset of the point's key (dummy slot n*4 for idle lanes)
    const int *schedNb;                 // [nPass][64][INI_NB] LDS byte offsets of the neighbours' keys in schedule order (dummy slot: none)
    // optReg sweep (two lanes per point): schedule of passes of <= 32 points and the per-lane inputs ini_prep writes before every sweep
    int nPass2;
    int nIdle;
    const int *slotOf;                  // [n] place of the point in the schedule: pass * 32 + position
    const int *idleSlot;                // [nIdle] places (of nPass2 + INI_SWPAD passes) that hold no point
    int4 *swRec;                        // [nPass2 + INI_SWPAD][64][2]: see SwRec
    const int *childOff, *childIdx;     // children (points of level - 1) of every point of this level, ascending
};

struct IniCtl {
    double Tcur[12], Tnew[12];
    float aCur, bCur, aNew, bNew;
    float inc[8];
    float lambda;
    float H[64], b[8], Hsc[64], bsc[8], resOld[3];
    float Hn[64], bn[8], Hscn[64], bscn[8], resNew[3], ec[3];
    int lvl, mode, iteration, fails, done, snapped, snappedAt, frameID, jbSel, applyPending, evals, ready;
    int idleWritten;                    // the records of the schedule places that hold no point are written (ini_prep_idle)
    int ldsBase;                        // LDS address of the control block's key array (the sweep records hold LDS addresses)
    int sweepDue, upGoing;              // level + 1 whose optReg sweep (view: good points, applied depths) is the next control step's; in the propagateUp chain
    int steps;                          // control steps of this frame so far (INI_STEP launches that did something)
    int prepReady;                      // k_ini_prep has written the records of the level for the step this evaluation tried; consumed by the control step
    long long dbgSweepTicks, dbgSweepPasses, dbgCtlTicks, dbgSweeps, dbgPrepTicks, dbgFrontTicks, dbgTailTicks, dbgSpare;     // LDSO_STAMPS builds only (100 MHz wall clock)
};

struct IniParams {
    IniLevel L[INI_MAXL];
    int levels, fixAffine;
    float huberTH, firstExposure, newExposure;
    IniCtl *ctl;
    float *part;                        // [INI_MAXBLK][INI_NPART]
};

__device__ __forceinline__ float ini_dpp_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float ini_dpp_xor2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float ini_dpp_hmir(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float ini_dpp_rmir(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true)); }
__device__ __forceinline__ float ini_dpp_ror8(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true)); }
__device__ __forceinline__ float ini_sum8(float x) { x += ini_dpp_xor1(x); x += ini_dpp_xor2(x); x += ini_dpp_hmir(x); return x; }
__device__ __forceinline__ float ini_min8(float x) { x = fminf(x, ini_dpp_xor1(x)); x = fminf(x, ini_dpp_xor2(x)); x = fminf(x, ini_dpp_hmir(x)); return x; }
__device__ __forceinline__ float ini_sum16(float x) { x = ini_sum8(x); x += ini_dpp_rmir(x); return x; }

__device__ __forceinline__ int ini_nblocks(int n, int gridCap) { const int nb = (n + 31) / 32; return nb < gridCap ? (nb < 1 ? 1 : nb) : gridCap; }

// ---------------------------------------------------------------------------------------------------------
// eval: (lazy applyStep) + (resetPoints | doStep) + calcResAndGS + calcEC sums
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(INI_NT) void k_ini_eval(IniParams P, int stage) {
    const IniCtl *ctl = P.ctl;
    if (!stage && (ctl->done || ctl->sweepDue)) return;         // frame finished / a level transition's control step comes first
    const int lvl = ctl->lvl;
    const IniLevel &L = P.L[lvl];
    const int n = L.n;
    const int nblk = ini_nblocks(n, INI_MAXBLK);
    if ((int) blockIdx.x >= nblk) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = tid & 7;
    __shared__ float sRow[16][52];
    __shared__ float sSC[16][8][10];

    // ---- uniform prologue ----
    const int mode = stage ? 2 : ctl->mode, pend = stage ? 0 : ctl->applyPending, jbSel = ctl->jbSel, snapped = ctl->snapped;
    const float lambda = ctl->lambda;
    float inc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) inc[i] = ctl->inc[i];
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = ctl->Tnew[i];
    float RKi[9], t[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) RKi[r * 3 + c] = (float) (T[r * 4 + 0] * L.Ki[0 * 3 + c] + T[r * 4 + 1] * L.Ki[1 * 3 + c] + T[r * 4 + 2] * L.Ki[2 * 3 + c]);
        t[r] = (float) T[r * 4 + 3];
    }
    const float affA = expf(ctl->aNew), affB = ctl->bNew;
    const float fxl = L.fx, fyl = L.fy, cxl = L.cx, cyl = L.cy;
    const int wl = L.w, hl = L.h;
    const float alphaK = 2.5f * 2.5f, alphaW = 150.0f * 150.0f, couplingWeight = 1.0f;
    float alphaOpt;
    {
        const double tsq = T[3] * T[3] + T[7] * T[7] + T[11] * T[11];
        const float alphaEnergy = (float) ((double) alphaW * (0.0 + tsq * (double) n));
        alphaOpt = (alphaEnergy > alphaK * (float) n) ? 0.0f : alphaW;
    }
    const float huberTH = P.huberTH;
    const int pdx = (k == 0) ? 0 : (k == 1) ? -1 : (k == 2) ? 1 : (k == 3) ? -2 : (k == 4) ? 0 : (k == 5) ? 2 : (k == 6) ? -1 : 0;
    const int pdy = (k == 0) ? -2 : (k == 1) ? -1 : (k == 2) ? -1 : (k == 3) ? 0 : (k == 4) ? 0 : (k == 5) ? 0 : (k == 6) ? 1 : 2;
    const float *jbOld = L.jb[jbSel];
    float *jbNew = L.jb[jbSel ^ 1];

    float acc[45];
#pragma unroll
    for (int i = 0; i < 45; i++) acc[i] = 0.0f;
    float accSC[10];
#pragma unroll
    for (int i = 0; i < 10; i++) accSC[i] = 0.0f;
    float accE = 0.0f, accEO = 0.0f, accEN = 0.0f, accEC = 0.0f;

    for (int i = blockIdx.x * 32 + (tid >> 3); i < ((n + 31) & ~31); i += nblk * 32) {
        const bool valid = i < n;
        const int ii = valid ? i : n - 1;
        float pu = L.u[ii], pv = L.v[ii], idp = L.idepth[ii], idn = L.idepth_new[ii], iR = L.iR[ii];
        int good = L.isGood[ii];
        float e0 = L.energy0[ii], e1 = L.energy1[ii];
        const float oTH = L.outlierTH[ii];
        if (mode != 2) {
            if (pend) {                                   // applyStep of the accepted evaluation (:673-687), this point's share
                if (!good) { idp = iR; idn = iR; if (valid && k == 0) L.idepth[ii] = idp; }
                else {
                    e0 = L.energy_new0[ii]; e1 = L.energy_new1[ii]; good = L.isGood_new[ii]; idp = idn;
                    if (valid && k == 0) {
                        L.energy0[ii] = e0; L.energy1[ii] = e1; L.isGood[ii] = good; L.idepth[ii] = idp;
                        L.lastHessian[ii] = L.lastHessian_new[ii];
                    }
                }
            }
            if (mode == 0) { e0 = 0.0f; e1 = 0.0f; idn = idp; if (valid && k == 0) { L.energy0[ii] = 0.0f; L.energy1[ii] = 0.0f; } }   // resetPoints :625-626
            else if (good) {                              // doStep (:645-671)
                const float *J = jbOld + (size_t) ii * 10;
                float p4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) p4[q] = J[q] * inc[q] + J[q + 4] * inc[q + 4];
                const float b = J[8] + ((p4[0] + p4[2]) + (p4[1] + p4[3]));
                float step = -b * J[9] / (1 + lambda);
                float maxstep = 0.25f * L.maxstep[ii];
                if (maxstep > 1e10f) maxstep = 1e10f;
                if (step > maxstep) step = maxstep;
                if (step < -maxstep) step = -maxstep;
                float newIdepth = idp + step;
                if (newIdepth < 1e-3f) newIdepth = 1e-3f;
                if (newIdepth > 50.0f) newIdepth = 50.0f;
                idn = newIdepth;
            }
        }
        // ---- calcResAndGS for this point (:206-334) ----
        bool bad = true;
        float dp[9], dd = 0.0f, e = 0.0f, ms = 1e10f;
#pragma unroll
        for (int q = 0; q < 9; q++) dp[q] = 0.0f;
        if (valid && good) {
            const float px = pu + (float) pdx, py = pv + (float) pdy;
            const float pt0 = ((RKi[0] * px + RKi[1] * py) + RKi[2] * 1.0f) + t[0] * idn;
            const float pt1 = ((RKi[3] * px + RKi[4] * py) + RKi[5] * 1.0f) + t[1] * idn;
            const float pt2 = ((RKi[6] * px + RKi[7] * py) + RKi[8] * 1.0f) + t[2] * idn;
            const float uu = pt0 / pt2, vv = pt1 / pt2;
            const float Ku = fxl * uu + cxl, Kv = fyl * vv + cyl;
            const float new_idepth = idn / pt2;
            if (Ku > 1 && Kv > 1 && Ku < wl - 2 && Kv < hl - 2 && new_idepth > 0) {
                float hit[3], rlR;
                {
                    const int ix = (int) Ku, iy = (int) Kv;
                    const float dx = Ku - ix, dy = Kv - iy, dxdy = dx * dy;
                    const float *bp = L.cur + 3 * ((size_t) ix + (size_t) iy * wl);
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        hit[c] = dxdy * bp[3 + 3 * wl + c] + (dy - dxdy) * bp[3 * wl + c] + (dx - dxdy) * bp[3 + c] + (1 - dx - dy + dxdy) * bp[c];
                }
                {
                    const int ix = (int) px, iy = (int) py;
                    const float dx = px - ix, dy = py - iy, dxdy = dx * dy;
                    const float *bp = L.first + 3 * ((size_t) ix + (size_t) iy * wl);
                    rlR = dxdy * bp[3 + 3 * wl] + (dy - dxdy) * bp[3 * wl] + (dx - dxdy) * bp[3] + (1 - dx - dy + dxdy) * bp[0];
                }
                if (isfinite(rlR) && isfinite(hit[0])) {
                    bad = false;
                    const float residual = hit[0] - affA * rlR - affB;
                    float hw = fabsf(residual) < huberTH ? 1 : huberTH / fabsf(residual);
                    e = hw * residual * residual * (2 - hw);
                    const float dxdd = (t[0] - t[2] * uu) / pt2;
                    const float dydd = (t[1] - t[2] * vv) / pt2;
                    if (hw < 1) hw = sqrtf(hw);
                    const float dxI = hw * hit[1] * fxl, dyI = hw * hit[2] * fyl;
                    dp[0] = new_idepth * dxI;
                    dp[1] = new_idepth * dyI;
                    dp[2] = -new_idepth * (uu * dxI + vv * dyI);
                    dp[3] = -uu * vv * dxI - (1 + vv * vv) * dyI;
                    dp[4] = (1 + uu * uu) * dxI + uu * vv * dyI;
                    dp[5] = -vv * dxI + uu * dyI;
                    dp[6] = -hw * affA * rlR;
                    dp[7] = -hw * 1;
                    dd = dxI * dxdd + dyI * dydd;
                    dp[8] = hw * residual;
                    const float nx = dxdd * fxl, ny = dydd * fyl;
                    ms = 1.0f / sqrtf(nx * nx + ny * ny);
                }
            }
        }
        // first bad pattern pixel of the 8-lane group: the reference breaks there (:245-259), pixels before it have updated maxstep
        const unsigned long long bm = __ballot(bad);
        const unsigned grp = (unsigned) (bm >> (lane & ~7)) & 0xFFu;
        const int firstBad = grp ? (__ffs(grp) - 1) : 8;
        const float maxstepNew = ini_min8((k < firstBad) ? ms : 1e10f);
        const float energy = ini_sum8(e);
        const bool goodNew = valid && good && firstBad == 8 && !(energy > oTH * 20);
        float Jb[10];
#pragma unroll
        for (int q = 0; q < 9; q++) Jb[q] = ini_sum8(dp[q] * dd);
        Jb[9] = ini_sum8(dd * dd);
        float en0, en1;
        if (goodNew) {
#pragma unroll
            for (int r = 0; r < 9; r++)
#pragma unroll
                for (int c = r; c < 9; c++) acc[r * 9 - (r * (r - 1)) / 2 + (c - r)] += dp[r] * dp[c];
            en0 = energy;
            en1 = (idn - 1) * (idn - 1);
            const float lastHn = Jb[9];
            Jb[8] += alphaOpt * (idn - 1);
            Jb[9] += alphaOpt;
            if (alphaOpt == 0) { Jb[8] += couplingWeight * (idn - iR); Jb[9] += couplingWeight; }
            Jb[9] = 1 / (1 + Jb[9]);
            const float w = Jb[9];
            // Schur block row k (updateSingleWeighted, MatrixAccumulators.h:1489-1614): diagonal J_r*J_r*w, off-diagonal J_c*(J_r*w)
            const float rv = (k == 0) ? Jb[0] : (k == 1) ? Jb[1] : (k == 2) ? Jb[2] : (k == 3) ? Jb[3] : (k == 4) ? Jb[4] : (k == 5) ? Jb[5] : (k == 6) ? Jb[6] : Jb[7];
            const float rw = rv * w;
#pragma unroll
            for (int q = 0; q < 9; q++) accSC[q] += (q == k) ? (rv * rv * w) : (Jb[q] * rw);
            if (k == 0) {
                accSC[9] += Jb[8] * Jb[8] * w;
                accE += energy;
                if (snapped) { const float ro = idp - iR, rn = idn - iR; accEO += ro * ro; accEN += rn * rn; accEC += 1.0f; }
                L.lastHessian_new[ii] = lastHn;
            }
            float *Jo = jbNew + (size_t) ii * 10;
            Jo[k] = rv;
            if (k < 2) Jo[8 + k] = (k == 0) ? Jb[8] : Jb[9];
        } else {
            en0 = e0; en1 = e1;                            // energy_new = energy (:213, :303)
            if (valid && k == 0) accE += e0;
        }
        if (valid && k == 0) {
            L.idepth_new[ii] = idn;
            L.isGood_new[ii] = goodNew ? 1 : 0;
            L.energy_new0[ii] = en0; L.energy_new1[ii] = en1;
            L.maxstep[ii] = maxstepNew;
        }
    }

    // ---- block reduction -> one partial row ----
#pragma unroll
    for (int i = 0; i < 45; i++) acc[i] = ini_sum16(acc[i]);
    accE = ini_sum16(accE); accEO = ini_sum16(accEO); accEN = ini_sum16(accEN); accEC = ini_sum16(accEC);
#pragma unroll
    for (int q = 0; q < 10; q++) accSC[q] += ini_dpp_ror8(accSC[q]);
    const int row = wave * 4 + (lane >> 4);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 45; i++) sRow[row][i] = acc[i];
        sRow[row][45] = accE; sRow[row][46] = accEO; sRow[row][47] = accEN; sRow[row][48] = accEC;
    }
    if ((lane & 15) < 8) {
#pragma unroll
        for (int q = 0; q < 10; q++) sSC[row][lane & 7][q] = accSC[q];
    }
    __syncthreads();
    float *out = P.part + (size_t) blockIdx.x * INI_NPART;
    if (tid < 49) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) s += sRow[r][tid];
        out[tid < 45 ? tid : (PR_E + tid - 45)] = s;
    } else if (tid >= 64 && tid < 64 + 80) {
        const int kk = (tid - 64) / 10, q = (tid - 64) % 10;
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) s += sSC[r][kk][q];
        if (q < 9) out[PR_SC + kk * 9 + q] = s;
        else if (kk == 0) out[PR_SC88] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// control block
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ini_shfl(float v, int src) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v))); }
__device__ __forceinline__ float ini_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// Eigen-style pivoted LDL^T solve (float) of the leading nn x nn block by one wavefront, lane i*8+j holds A[i][j]; x by lanes 0..7
__device__ void ini_ldlt_wave(float a, float rhsLane, int nn, float *x /*LDS 8*/) {
    const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
    if (i >= nn || j >= nn) a = 0.0f;
    int tr[8];
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        tr[kk] = kk;
        if (kk < nn) {
            int idx = kk;
            float best = fabsf(ini_bcast(a, kk * 9));
#pragma unroll
            for (int q = kk + 1; q < 8; q++) { const float d = fabsf(ini_bcast(a, q * 9)); if (q < nn && d > best) { best = d; idx = q; } }
            tr[kk] = idx;
            if (idx != kk) {
                const int si = (i == kk) ? idx : (i == idx) ? kk : i, sj = (j == kk) ? idx : (j == idx) ? kk : j;
                a = ini_shfl(a, si * 8 + sj);
            }
            const float d = ini_bcast(a, kk * 9);
            const float ci = ini_shfl(a, i * 8 + kk), cj = ini_shfl(a, j * 8 + kk);
            if (fabsf(d) > 0.0f) {
                if (i > kk && j > kk) a -= ci * (cj / d);
                else if (j == kk && i > kk) a = ci / d;
                else if (i == kk && j > kk) a = cj / d;
            }
        }
    }
    float xr = (lane < nn) ? rhsLane : 0.0f;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) { const int t_ = tr[kk]; if (t_ != kk) { const int src = (lane == kk) ? t_ : (lane == t_) ? kk : lane; xr = ini_shfl(xr, src & 63); } }
#pragma unroll
    for (int c = 0; c < 7; c++) { const float xc = ini_bcast(xr, c); const float l = ini_shfl(a, (lane & 7) * 8 + c); if (lane < 8 && lane > c) xr -= l * xc; }
    { const float dr = ini_shfl(a, (lane & 7) * 9); xr = (fabsf(dr) > 1.17549435e-38f) ? xr / dr : 0.0f; }
#pragma unroll
    for (int c = 7; c >= 1; c--) { const float xc = ini_bcast(xr, c); const float l = ini_shfl(a, c * 8 + (lane & 7)); if (lane < c) xr -= l * xc; }
#pragma unroll
    for (int kk = 7; kk >= 0; kk--) { const int t_ = tr[kk]; if (t_ != kk) { const int src = (lane == kk) ? t_ : (lane == t_) ? kk : lane; xr = ini_shfl(xr, src & 63); } }
    if (lane < 8) x[lane] = (lane < nn) ? xr : 0.0f;
}

// The in-place sequential sweeps (see file header).  sIR[j] = key of iR of point j if it is good, INI_NOKEY otherwise; filled by the caller.
// Behind the keys: sIR[n] = INI_NOKEY (dummy of absent / not-good neighbours and of idle lanes), sIR[n + 1] = INI_MINKEY, sIR[n + 2] = a slot nobody reads.
// Order-preserving integer key of a (non-NaN) float: integer min/max need no NaN canonicalisation.  INI_NOKEY = point not good.
#define INI_NOKEY 0x7fffffff
#define INI_MINKEY ((int) 0x80000000)
#define INI_LDS_EXTRA 3
__device__ __forceinline__ int ini_key(float f) { const int b = __builtin_bit_cast(int, f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float ini_unkey(int k) { return __builtin_bit_cast(float, k ^ ((k >> 31) & 0x7fffffff)); }
// unconditional prefetch: the schedule arrays carry INI_SWPAD passes of padding (idle lanes) behind the last pass
#define INI_SWPAD 16

// ---- resetPoints' neighbour average (:629-641), one lane per point: a point that gets an average BECOMES good for the points behind it, and the float sum runs in
// neighbour order - nothing of it can be prepared or split.  Top level only, once per frame.
// Sweep inputs of one scheduled point: LDS BYTE offsets of its own key and of its 10 neighbours' keys (an absent neighbour and an
// idle lane point at the dummy slot sK[n], which always holds INI_NOKEY - no clamps or selects in the loop).
struct SwIn { int self; int4 a, b, c; };
__device__ __forceinline__ SwIn ini_sw_load(const IniLevel &L, int p, int lane) {
    SwIn r;
    const size_t q = (size_t) p * 64 + lane;
    r.self = L.schedOff[q];
    const int4 *row = (const int4 *) (L.schedNb + q * INI_NB);
    r.a = row[0]; r.b = row[1]; r.c = row[2];
    return r;
}
__device__ __forceinline__ void ini_sw_point(const SwIn &r, int *sK, int dummyOff) {
    char *base = (char *) sK;
#define GK(off) (*(const int *) (base + (off)))
    int v[10] = {GK(r.a.x), GK(r.a.y), GK(r.a.z), GK(r.a.w), GK(r.b.x), GK(r.b.y), GK(r.b.z), GK(r.b.w), GK(r.c.x), GK(r.c.y)};
    const int self = GK(r.self);
#undef GK
    float snd = 0, sn = 0;
#pragma unroll
    for (int q = 0; q < 10; q++) if (v[q] != INI_NOKEY) { snd += ini_unkey(v[q]); sn += 1; }
    if (r.self != dummyOff && self == INI_NOKEY && sn > 0) *(int *) (base + r.self) = ini_key(snd / sn);
}
// Executed by wave 0.  The pass inputs are stored in schedule order, so every pass needs one level of coalesced global loads,
// issued four passes ahead; the dependent chain of a pass is LDS gather -> sum -> LDS write.
__device__ void ini_sweep_reset(const IniLevel &L, int *sIR) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    if (L.nPass == 0) return;
    const int dummyOff = L.n * 4;
    SwIn r0 = ini_sw_load(L, 0, lane), r1 = ini_sw_load(L, 1, lane), r2 = ini_sw_load(L, 2, lane), r3 = ini_sw_load(L, 3, lane);
    for (int p = 0; p < L.nPass; p += 4) {
        ini_sw_point(r0, sIR, dummyOff); r0 = ini_sw_load(L, p + 4, lane);
        ini_sw_point(r1, sIR, dummyOff); r1 = ini_sw_load(L, p + 5, lane);
        ini_sw_point(r2, sIR, dummyOff); r2 = ini_sw_load(L, p + 6, lane);
        ini_sw_point(r3, sIR, dummyOff); r3 = ini_sw_load(L, p + 7, lane);
    }
}

// ---- optReg (:430-459), TWO lanes per point.  Which points are good does not change during the sweep, so everything but the median itself is prepared in parallel
// (ini_prep) right before it: the reference takes nth_element(nnn / 2) of the nnn good neighbours' iR.  Of the 10 - nnn others, 5 - nnn / 2 are pointed at a slot
// holding the smallest key and the rest at one holding the largest: the wanted value is then ALWAYS the 6th smallest of ten keys - a fixed selection, no count,
// no select chain.  Lane 2p sorts neighbours 0-4 of its point, lane 2p + 1 neighbours 5-9 (3-input min / med / max: 15 instructions); with a_1 <= .. <= a_5 and
// b_1 <= .. <= b_5 the 6th smallest of the union is min_i max(a_i, b_(6-i)): five maxima against the partner lane's registers (DPP) and two 3-input minima.
// A point that is not written (not good, or nnn <= 2) has its result directed at a slot nobody reads; so have the odd lanes and the idle ones.
// Per lane: a = {LDS addresses of keys 0..3 of its half}, b = {the 5th, LDS address the result goes to, idepth (float bits), -}; addresses, not offsets into
// the key array: a ds_read takes them as they are
struct SwRec { int4 a, b; };
typedef __attribute__((address_space(3))) int ini_lds_int;
__device__ __forceinline__ SwRec ini_sw2_load(const IniLevel &L, int p, int lane) {
    const int4 *r = L.swRec + ((size_t) p * 64 + lane) * 2;
    SwRec o; o.a = r[0]; o.b = r[1];
    return o;
}
__device__ __forceinline__ int ini_min3(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int ini_med3(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int ini_max3(int a, int b, int c) { int r; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int ini_pair(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); }      // the other lane of the pair
__device__ __forceinline__ void ini_sw2_point(const SwRec &r) {
    const float regWeight = 0.8f;
#define GK(adr) (*(const ini_lds_int *) (unsigned long long) (unsigned) (adr))
    const int v0 = GK(r.a.x), v1 = GK(r.a.y), v2 = GK(r.a.z), v3 = GK(r.a.w), v4 = GK(r.b.x);
#undef GK
    // sort five: s = sorted (v0, v1, v2), t = sorted (v3, v4), merged rank k = min over i + j = k of max(s_i, t_j)
    const int s1 = ini_min3(v0, v1, v2), s2 = ini_med3(v0, v1, v2), s3 = ini_max3(v0, v1, v2);
    const int t1 = min(v3, v4), t2 = max(v3, v4);
    const int m1 = min(s1, t1);
    const int m2 = ini_min3(s2, max(s1, t1), t2);
    const int m3 = ini_min3(s3, max(s2, t1), max(s1, t2));
    const int m4 = min(max(s3, t1), max(s2, t2));
    const int m5 = max(s3, t2);
    // 6th smallest of the pair's ten
    const int x1 = max(m1, ini_pair(m5)), x2 = max(m2, ini_pair(m4)), x3 = max(m3, ini_pair(m3)), x4 = max(m4, ini_pair(m2)), x5 = max(m5, ini_pair(m1));
    const int mk = ini_min3(ini_min3(x1, x2, x3), x4, x5);
    *(ini_lds_int *) (unsigned long long) (unsigned) r.b.y = ini_key((1 - regWeight) * __builtin_bit_cast(float, r.b.z) + regWeight * ini_unkey(mk));
}
// Wavefront 0 sweeps, with the inputs of INI_SW_DEPTH passes in flight.  remote: the records were written by other compute units (k_ini_prep) and lie in
// memory, not in this XCD's L2 - the other wavefronts of the block pull them in, in sweep order, ahead of wavefront 0 (one dword per 128-byte line).
#ifndef INI_SW_DEPTH
#define INI_SW_DEPTH 8
#endif
#ifndef INI_SW_TOUCH
#define INI_SW_TOUCH 1
#endif
__device__ void ini_sweep_reg(const IniLevel &L, int *sIR, bool remote) {
    if (L.nPass2 == 0) return;
    if (threadIdx.x >= 64) {
        if (!(INI_SW_TOUCH && remote)) return;
        const int *rec = (const int *) L.swRec;
        const int nLines = (L.nPass2 + INI_SWPAD) * 16;                  // 64 lanes x 32 bytes per pass
        int acc = 0;
        for (int l = threadIdx.x - 64; l < nLines; l += blockDim.x - 64) acc ^= rec[(size_t) l * 32];
        asm volatile("" :: "v"(acc));
        return;
    }
    const int lane = threadIdx.x;
    SwRec r[INI_SW_DEPTH];
#pragma unroll
    for (int u = 0; u < INI_SW_DEPTH; u++) r[u] = ini_sw2_load(L, u, lane);
    for (int p = 0; p < L.nPass2; p += INI_SW_DEPTH) {
#pragma unroll
        for (int u = 0; u < INI_SW_DEPTH; u++) { ini_sw2_point(r[u]); r[u] = ini_sw2_load(L, p + INI_SW_DEPTH + u, lane); }
    }
}

__device__ void ini_fill(const IniLevel &L, int *sIR, int pending) {
    const int nt = blockDim.x;
    if (threadIdx.x == 0) { sIR[L.n] = INI_NOKEY; sIR[L.n + 1] = INI_MINKEY; sIR[L.n + 2] = INI_NOKEY; }
    for (int j0 = threadIdx.x; j0 < L.n; j0 += 4 * nt) {
        int g[4], gn[4]; float r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = j0 + u * nt; const bool in = j < L.n; g[u] = in ? L.isGood[j] : 0; gn[u] = (in && pending) ? L.isGood_new[j] : 1; r[u] = in ? L.iR[j] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = j0 + u * nt; if (j < L.n) sIR[j] = (g[u] && gn[u]) ? ini_key(r[u]) : INI_NOKEY; }
    }
}
// The inputs of an optReg sweep of one point (its record pair goes to the point's place in the schedule).  good(j): point j is good in the view the sweep works on.
template <class Good>
__device__ __forceinline__ void ini_prep_point(const IniLevel &L, int i, int place, const int4 &na, const int4 &nb_, const int4 &nc, float id, int base, Good good) {
    const int offMax = base + L.n * 4, offMin = offMax + 4, offNone = offMax + 8;
    const int j[10] = {na.x, na.y, na.z, na.w, nb_.x, nb_.y, nb_.z, nb_.w, nc.x, nc.y};
    int off[10], nnn = 0;
#pragma unroll
    for (int e = 0; e < 10; e++) { const bool g = j[e] >= 0 && good(max(j[e], 0)); off[e] = g ? base + j[e] * 4 : -1; nnn += g ? 1 : 0; }
    int low = 5 - (nnn >> 1);
#pragma unroll
    for (int e = 0; e < 10; e++) if (off[e] < 0) { off[e] = (low > 0) ? offMin : offMax; low--; }
    const int selfOff = (good(i) && nnn > 2) ? base + i * 4 : offNone;
    const int idb = __builtin_bit_cast(int, id);
    int4 *dst = L.swRec + (size_t) place * 4;                           // two lanes x two int4
    dst[0] = make_int4(off[0], off[1], off[2], off[3]); dst[1] = make_int4(off[4], selfOff, idb, 0);
    dst[2] = make_int4(off[5], off[6], off[7], off[8]); dst[3] = make_int4(off[9], offNone, idb, 0);
}
// ... by the control block itself (sIR holds the keys): 2.4 MB through ONE compute unit at the two big levels, ~50 us.  Only where the sweep follows a change of
// the level made by the control step (propagateDown / propagateUp, a step that snaps); the sweep behind an accepted step is prepared by k_ini_prep.
__device__ void ini_prep(const IniLevel &L, const int *sIR, const float *idv) {
    const int nt = blockDim.x;
    const int base = (int) (unsigned long long) (const ini_lds_int *) sIR;
    for (int j0 = threadIdx.x; j0 < L.n; j0 += 4 * nt) {
        int q[4]; int4 na[4], nb_[4], nc[4]; float id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(j0 + u * nt, L.n - 1);
            const int4 *row = (const int4 *) (L.nb + (size_t) i * INI_NB);
            q[u] = L.slotOf[i]; na[u] = row[0]; nb_[u] = row[1]; nc[u] = row[2]; id[u] = idv[i];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = j0 + u * nt;
            if (i < L.n) ini_prep_point(L, i, q[u], na[u], nb_[u], nc[u], id[u], base, [&](int j) { return sIR[j] != INI_NOKEY; });
        }
    }
}
// the schedule places no point has: once per handle (BEGIN), every level
__device__ void ini_prep_idle(const IniLevel &L, const int *sIR) {
    const int base = (int) (unsigned long long) (const ini_lds_int *) sIR;
    const int offMax = base + L.n * 4, offNone = offMax + 8;
    const int4 i1 = make_int4(offMax, offMax, offMax, offMax), i2 = make_int4(offMax, offNone, 0, 0);
    for (int k = threadIdx.x; k < L.nIdle; k += blockDim.x) { int4 *dst = L.swRec + (size_t) L.idleSlot[k] * 4; dst[0] = i1; dst[1] = i2; dst[2] = i1; dst[3] = i2; }
}

// optReg(lvl) (:430-459).  pending: an accepted step has not been applied to the arrays yet (lazy applyStep) - use its view.
__device__ void ini_opt_reg(const IniLevel &L, int *sIR, int snapped, int pending, bool prepared, IniCtl *ctl) {
    if (!snapped) { for (int j = threadIdx.x; j < L.n; j += blockDim.x) L.iR[j] = 1.0f; __syncthreads(); return; }
#ifdef LDSO_STAMPS
    const long long tp_ = wall_clock64();
#endif
    ini_fill(L, sIR, pending);
    __syncthreads();
    const bool remote = prepared && ctl->prepReady;                                                 // k_ini_prep has written them (for this very sweep: prepared)
    if (!remote) ini_prep(L, sIR, pending ? L.idepth_new : L.idepth);
    __syncthreads();
#ifdef LDSO_STAMPS
    const long long t0_ = wall_clock64();
#endif
    ini_sweep_reg(L, sIR, remote);
#ifdef LDSO_STAMPS
    if (threadIdx.x == 0) { const long long t1_ = wall_clock64(); ctl->dbgSweepTicks += t1_ - t0_; ctl->dbgPrepTicks += t0_ - tp_; ctl->dbgSweepPasses += L.nPass2; ctl->dbgSweeps++; }
#endif
    __syncthreads();
    for (int j = threadIdx.x; j < L.n; j += blockDim.x) { const int kk = sIR[j]; if (kk != INI_NOKEY) L.iR[j] = ini_unkey(kk); }
    __syncthreads();
}

__device__ void ini_flush_apply(const IniLevel &L) {   // applyStep (:673-687) for every point of the level
    const int nt = blockDim.x;
    for (int j0 = threadIdx.x; j0 < L.n; j0 += 4 * nt) {
        int g[4], gn[4]; float r[4], e0[4], e1[4], idn[4], lh[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = min(j0 + u * nt, L.n - 1);
            g[u] = L.isGood[j]; gn[u] = L.isGood_new[j]; r[u] = L.iR[j]; e0[u] = L.energy_new0[j]; e1[u] = L.energy_new1[j]; idn[u] = L.idepth_new[j]; lh[u] = L.lastHessian_new[j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * nt;
            if (j >= L.n) continue;
            if (!g[u]) { L.idepth[j] = r[u]; L.idepth_new[j] = r[u]; continue; }
            L.energy0[j] = e0[u]; L.energy1[j] = e1[u]; L.isGood[j] = gn[u]; L.idepth[j] = idn[u]; L.lastHessian[j] = lh[u];
        }
    }
    __syncthreads();
}

#define INI_BEGIN 0
#define INI_STEP 1
#define INI_STAGE 2

// propagateDown(lvl) (:498-522) into level lvl - 1
__device__ void ini_propagate_down(const IniParams &P, int lvl) {
    const IniLevel &L = P.L[lvl];
    const int tid = threadIdx.x;
    const IniLevel &F = P.L[lvl - 1];
    const int nt = blockDim.x;
    for (int j0 = tid; j0 < F.n; j0 += 4 * nt) {
        int pa[4], pg[4], fg[4]; float plh[4], pir[4], fir[4], flh[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = min(j0 + u * nt, F.n - 1); pa[u] = F.parent[j]; fg[u] = F.isGood[j]; fir[u] = F.iR[j]; flh[u] = F.lastHessian[j]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { pg[u] = L.isGood[pa[u]]; plh[u] = L.lastHessian[pa[u]]; pir[u] = L.iR[pa[u]]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * nt;
            if (j >= F.n) continue;
            if (!pg[u] || plh[u] < 0.1f) continue;
            if (!fg[u]) {
                const float r = pir[u];
                F.iR[j] = r; F.idepth[j] = r; F.idepth_new[j] = r; F.isGood[j] = 1; F.lastHessian[j] = 0;
            } else {
                const float newiR = (fir[u] * flh[u] * 2 + pir[u] * plh[u]) / (flh[u] * 2 + plh[u]);
                F.iR[j] = newiR; F.idepth[j] = newiR; F.idepth_new[j] = newiR;
            }
        }
    }
}
// propagateUp(s) (:462-496) into level s + 1
__device__ void ini_propagate_up(const IniParams &P, int s) {
    const IniLevel &S = P.L[s], &D = P.L[s + 1];
    for (int p = threadIdx.x; p < D.n; p += blockDim.x) {
        float a = 0, sum = 0;
        for (int q = D.childOff[p]; q < D.childOff[p + 1]; q++) {
            const int c = D.childIdx[q];
            if (!S.isGood[c]) continue;
            a += S.iR[c] * S.lastHessian[c];
            sum += S.lastHessian[c];
        }
        D.iRSumNum[p] = sum;
        if (sum > 0) { const float r = a / sum; D.iR[p] = r; D.idepth[p] = r; D.isGood[p] = 1; }
        else D.iR[p] = a;
    }
}
// the end of trackFrame (:165-177), one lane
__device__ __forceinline__ void ini_frame_end(IniCtl *ctl) {
    ctl->frameID++;
    if (!ctl->snapped) ctl->snappedAt = 0;
    if (ctl->snapped && ctl->snappedAt == 0) ctl->snappedAt = ctl->frameID;
    ctl->ready = ctl->snapped && ctl->frameID > ctl->snappedAt + 5;
    ctl->done = 1;
}

// The control step proper.  ctl is the block's LDS copy of the control record (k_ini_ctl): the decision, the bookkeeping and the pose update are one lane's
// chains of dependent reads and writes of it.
__device__ __forceinline__ void ini_ctl_body(const IniParams &P, int phase, IniCtl *ctl, int *sIR) {
    __shared__ double sSum[INI_NPART];
    __shared__ double sPart[INI_CT / 128][128];
    __shared__ float sX[8];
    __shared__ int sFlag[4];
    const int tid = threadIdx.x;
    const int top = P.levels - 1;

    if (phase == INI_BEGIN) {                                   // trackFrame :42-69 + resetPoints' neighbour average at the top level
        const int snapped = ctl->snapped;
        if (!snapped) {
            for (int l = 0; l < P.levels; l++) {
                const IniLevel &L = P.L[l];
                for (int j = tid; j < L.n; j += blockDim.x) { L.iR[j] = 1.0f; L.idepth_new[j] = 1.0f; L.lastHessian[j] = 0.0f; }
            }
        }
        if (tid == 0) {
            if (!snapped) { ctl->Tcur[3] = 0; ctl->Tcur[7] = 0; ctl->Tcur[11] = 0; }
            if (P.firstExposure > 0 && P.newExposure > 0) { ctl->aCur = logf(P.newExposure / P.firstExposure); ctl->bCur = 0; }
            for (int q = 0; q < 12; q++) ctl->Tnew[q] = ctl->Tcur[q];
            ctl->aNew = ctl->aCur; ctl->bNew = ctl->bCur;
            ctl->lvl = top; ctl->mode = 0; ctl->done = 0; ctl->evals = 0; ctl->applyPending = 0; ctl->iteration = 0; ctl->fails = 0; ctl->lambda = 0.1f;
            for (int q = 0; q < 8; q++) ctl->inc[q] = 0;
        }
        if (!ctl->idleWritten) for (int l = 0; l < P.levels; l++) ini_prep_idle(P.L[l], sIR);
        __syncthreads();
        if (tid == 0) { ctl->idleWritten = 1; ctl->ldsBase = (int) (unsigned long long) (const ini_lds_int *) sIR; ctl->prepReady = 0; ctl->steps = 0; ctl->sweepDue = 0; ctl->upGoing = 0; }
        const IniLevel &L = P.L[top];
        ini_fill(L, sIR, 0);
        __syncthreads();
        ini_sweep_reset(L, sIR);
        __syncthreads();
        for (int j = tid; j < L.n; j += blockDim.x) {
            const int kk = sIR[j];
            if (!L.isGood[j] && kk != INI_NOKEY) { const float v = ini_unkey(kk); L.isGood[j] = 1; L.iR[j] = v; L.idepth[j] = v; L.idepth_new[j] = v; }
        }
        return;
    }

#ifdef LDSO_STAMPS
    const long long tk0_ = wall_clock64();
#endif
    if (phase == INI_STEP && ctl->sweepDue) {                   // a level transition of a snapped frame: the sweep behind propagateDown / propagateUp, then what follows it
        const int sl = ctl->sweepDue - 1, up = ctl->upGoing;
        ini_opt_reg(P.L[sl], sIR, 1, 0, true, ctl);
        if (up && sl + 1 < P.levels) {
            ini_propagate_up(P, sl);
            __syncthreads();
            if (tid == 0) ctl->sweepDue = sl + 2;
        } else if (tid == 0) {
            ctl->sweepDue = 0;
            if (up) { ctl->upGoing = 0; ini_frame_end(ctl); }
        }
        return;
    }
    const int lvl = ctl->lvl;
    const IniLevel &L = P.L[lvl];
    const int n = L.n;
    // ---- sum the partial rows ----
    {
        const int nblk = ini_nblocks(n, INI_MAXBLK);
        const int c = tid & 127, part = tid >> 7;              // 8 row slices
        double s = 0.0;
        if (c < PR_N) {
#pragma unroll 8
            for (int r = part; r < nblk; r += INI_CT / 128) s += (double) P.part[(size_t) r * INI_NPART + c];
        }
        sPart[part][c] = s;
        __syncthreads();
        if (tid < 128) { double a = 0.0; for (int q = 0; q < INI_CT / 128; q++) a += sPart[q][tid]; sSum[tid] = a; }
        __syncthreads();
    }
    const float alphaK = 2.5f * 2.5f, alphaW = 150.0f * 150.0f;
    if (tid < 64) {                                             // H_new, Hsc_new (:389-401)
        const int r = tid >> 3, c = tid & 7, lo = min(r, c), hi = max(r, c);
        ctl->Hn[tid] = (float) sSum[lo * 9 - (lo * (lo - 1)) / 2 + (hi - lo)];
        ctl->Hscn[tid] = (float) sSum[PR_SC + lo * 9 + hi];
    } else if (tid < 72) {
        const int r = tid - 64;
        ctl->bn[r] = (float) sSum[r * 9 - (r * (r - 1)) / 2 + (8 - r)];
        ctl->bscn[r] = (float) sSum[PR_SC + r * 9 + 8];
    }
    __syncthreads();
    if (tid == 0) {
        const double *T = ctl->Tnew;
        const double tsq = T[3] * T[3] + T[7] * T[7] + T[11] * T[11];
        float alphaEnergy = (float) ((double) alphaW * (0.0 + tsq * (double) n));
        float alphaOpt;
        if (alphaEnergy > alphaK * (float) n) { alphaOpt = 0; alphaEnergy = alphaK * (float) n; } else alphaOpt = alphaW;
        ctl->Hn[0] += alphaOpt * n; ctl->Hn[9] += alphaOpt * n; ctl->Hn[18] += alphaOpt * n;
        double lg[6];
        ld::se3_log(T, lg);
        for (int q = 0; q < 3; q++) ctl->bn[q] += (float) lg[q] * alphaOpt * n;
        ctl->resNew[0] = (float) sSum[PR_E]; ctl->resNew[1] = alphaEnergy; ctl->resNew[2] = (float) (2 * n);
        if (ctl->snapped) { ctl->ec[0] = 1.0f * (float) sSum[PR_ECO]; ctl->ec[1] = 1.0f * (float) sSum[PR_ECN]; ctl->ec[2] = (float) sSum[PR_ECC]; }
        else { ctl->ec[0] = 0; ctl->ec[1] = 0; ctl->ec[2] = (float) n; }
    }
    __syncthreads();
    if (phase == INI_STAGE) return;

    // ---- the decision of :120-158 ----
    if (tid == 0) {
        const int maxIterations[5] = {5, 5, 10, 30, 50};
        int accept, quit = 0, doOpt = 0;
        ctl->evals++;
        if (ctl->mode == 0) {                                   // first evaluation of the level (:81-88)
            accept = 1;
            ctl->lambda = 0.1f; ctl->fails = 0; ctl->iteration = 0; ctl->mode = 1;
        } else {
            const float eTotalNew = (ctl->resNew[0] + ctl->resNew[1] + ctl->ec[1]);
            const float eTotalOld = (ctl->resOld[0] + ctl->resOld[1] + ctl->ec[0]);
            accept = eTotalOld > eTotalNew;
            if (accept) {
                if (ctl->resNew[1] == alphaK * (float) n) ctl->snapped = 1;
                for (int q = 0; q < 12; q++) ctl->Tcur[q] = ctl->Tnew[q];
                ctl->aCur = ctl->aNew; ctl->bCur = ctl->bNew;
                doOpt = 1;
                ctl->lambda *= 0.5f; ctl->fails = 0;
                if (ctl->lambda < 0.0001f) ctl->lambda = 0.0001f;
            } else {
                ctl->fails++;
                ctl->lambda *= 4;
                if (ctl->lambda > 10000) ctl->lambda = 10000;
            }
            float nrm = 0;
            for (int q = 0; q < 8; q++) nrm += ctl->inc[q] * ctl->inc[q];
            nrm = sqrtf(nrm);
            if (!(nrm > 1e-4f) || ctl->iteration >= maxIterations[lvl] || ctl->fails >= 2) quit = 1;
            else ctl->iteration++;
        }
        if (accept) {
            for (int q = 0; q < 64; q++) { ctl->H[q] = ctl->Hn[q]; ctl->Hsc[q] = ctl->Hscn[q]; }
            for (int q = 0; q < 8; q++) { ctl->b[q] = ctl->bn[q]; ctl->bsc[q] = ctl->bscn[q]; }
            for (int q = 0; q < 3; q++) ctl->resOld[q] = ctl->resNew[q];
            ctl->applyPending = 1; ctl->jbSel ^= 1;
        } else ctl->applyPending = 0;
        sFlag[0] = accept; sFlag[1] = quit; sFlag[2] = doOpt; sFlag[3] = ctl->snapped;
    }
    __syncthreads();
    const int accept = sFlag[0], quit = sFlag[1], doOpt = sFlag[2], snapped = sFlag[3];
#ifdef LDSO_STAMPS
    if (tid == 0) ctl->dbgFrontTicks += wall_clock64() - tk0_;
#endif
    if (doOpt) ini_opt_reg(L, sIR, snapped, 1, true, ctl);                 // applyStep (pending) + optReg (:137-138)

    if (quit) {
        if (accept) ini_flush_apply(L);
        if (tid == 0) ctl->applyPending = 0;
        // Once snapped, the optReg sweep that follows a propagation is the next control step's (ini_transition_step): k_ini_prep prepares it in between, on the
        // whole chip, instead of this block streaming the level's records through one compute unit (~50 us at the big levels)
        if (lvl > 0) {
            ini_propagate_down(P, lvl);
            __syncthreads();
            if (snapped) { if (tid == 0) ctl->sweepDue = lvl; }          // level lvl - 1, stored + 1
            else ini_opt_reg(P.L[lvl - 1], sIR, 0, 0, false, ctl);
            if (tid == 0) {
                ctl->lvl = lvl - 1; ctl->mode = 0;
                for (int q = 0; q < 12; q++) ctl->Tnew[q] = ctl->Tcur[q];
                ctl->aNew = ctl->aCur; ctl->bNew = ctl->bCur;
            }
        } else if (snapped && P.levels > 1) {                   // :165-177, continued by the transition steps
            ini_propagate_up(P, 0);
            __syncthreads();
            if (tid == 0) { ctl->sweepDue = 2; ctl->upGoing = 1; }
        } else {
            for (int s = 0; s + 1 < P.levels; s++) {
                ini_propagate_up(P, s);
                __syncthreads();
                ini_opt_reg(P.L[s + 1], sIR, snapped, 0, false, ctl);
            }
            if (tid == 0) ini_frame_end(ctl);
        }
        return;
    }

    // ---- next increment (:90-111) ----
#ifdef LDSO_STAMPS
    const long long tt0_ = wall_clock64();
#endif
    if (tid < 64) {
        const int i = tid >> 3, j = tid & 7;
        const float lambda = ctl->lambda;
        const float wMi = (i < 3) ? 1.0f : (i < 6) ? 0.5f : (i == 6) ? 10.0f : 1000.0f, wMj = (j < 3) ? 1.0f : (j < 6) ? 0.5f : (j == 6) ? 10.0f : 1000.0f;
        float Hl = ctl->H[tid];
        if (i == j) Hl *= (1 + lambda);
        Hl -= ctl->Hsc[tid] * (1 / (1 + lambda));
        const float sc = (0.01f / (L.w * L.h));
        Hl = ((wMi * Hl) * wMj) * sc;
        float bl = 0.0f;
        if (tid < 8) {
            const float wMr = (tid < 3) ? 1.0f : (tid < 6) ? 0.5f : (tid == 6) ? 10.0f : 1000.0f;
            bl = ctl->b[tid] - ctl->bsc[tid] * (1 / (1 + lambda));
            bl = (wMr * bl) * sc;
        }
        ini_ldlt_wave(Hl, bl, P.fixAffine ? 6 : 8, sX);
    }
    __syncthreads();
    if (tid == 0) {
        double xi[6], E[12], Tn[12];
        for (int q = 0; q < 8; q++) {
            const float wMr = (q < 3) ? 1.0f : (q < 6) ? 0.5f : (q == 6) ? 10.0f : 1000.0f;
            ctl->inc[q] = (P.fixAffine && q >= 6) ? 0.0f : -(wMr * sX[q]);
        }
        for (int q = 0; q < 6; q++) xi[q] = (double) ctl->inc[q];
        ld::se3_exp(xi, E);
        ld::se3_mul(E, ctl->Tcur, Tn);
        for (int q = 0; q < 12; q++) ctl->Tnew[q] = Tn[q];
        ctl->aNew = ctl->aCur + ctl->inc[6];
        ctl->bNew = ctl->bCur + ctl->inc[7];
#ifdef LDSO_STAMPS
        ctl->dbgTailTicks += wall_clock64() - tt0_;
#endif
    }
}

__global__ __launch_bounds__(INI_CT) void k_ini_ctl(IniParams P, int phase) {
    extern __shared__ int sIR[];
    __shared__ IniCtl sCtl;
    static_assert(sizeof(IniCtl) % 4 == 0, "IniCtl is copied in dwords");
    if (phase == INI_STEP && P.ctl->done) return;           // the frame is finished: the rest of the enqueued launches return at once
#ifdef LDSO_STAMPS
    const long long tk0_ = wall_clock64();
#endif
    for (int q = threadIdx.x; q < (int) (sizeof(IniCtl) / 4); q += blockDim.x) ((int *) &sCtl)[q] = ((const int *) P.ctl)[q];
    __syncthreads();
    ini_ctl_body(P, phase, &sCtl, sIR);
    __syncthreads();
    if (threadIdx.x == 0) {
        sCtl.prepReady = 0;
        if (phase == INI_STEP) sCtl.steps++;
#ifdef LDSO_STAMPS
        if (phase != INI_BEGIN) sCtl.dbgCtlTicks += wall_clock64() - tk0_;
#endif
    }
    __syncthreads();
    for (int q = threadIdx.x; q < (int) (sizeof(IniCtl) / 4); q += blockDim.x) ((int *) P.ctl)[q] = ((const int *) &sCtl)[q];
}

// The sweep inputs of the step an evaluation has just tried, by the whole chip instead of the control block (launched between k_ini_eval and k_ini_ctl): if the
// control step accepts it, optReg (:137-138) sweeps the level in the view "good and still good after the step" with the new inverse depths - both final when
// k_ini_eval ends.  Wasted when the step is rejected (a few us beside the evaluation).
#define INI_PT 256
__global__ __launch_bounds__(INI_PT) void k_ini_prep(IniParams P) {
    const IniCtl *ctl = P.ctl;
    // nothing to prepare: frame finished; optReg only resets iR before the snap (a step that snaps is prepared by the control step itself)
    if (ctl->done || !ctl->snapped) return;
    const int due = ctl->sweepDue;
    if (!due && ctl->mode == 0) return;                         // first evaluation of a level: no step to accept
    // the sweep behind a propagation (level due - 1: good points, applied depths) or the one behind the step just evaluated (good and still good, new depths)
    const IniLevel &L = P.L[due ? due - 1 : ctl->lvl];
    const float *idv = due ? L.idepth : L.idepth_new;
    const int base = ctl->ldsBase;
    for (int i = blockIdx.x * INI_PT + threadIdx.x; i < L.n; i += gridDim.x * INI_PT) {
        const int4 *row = (const int4 *) (L.nb + (size_t) i * INI_NB);
        const int4 na = row[0], nb_ = row[1], nc = row[2];
        if (due) ini_prep_point(L, i, L.slotOf[i], na, nb_, nc, idv[i], base, [&](int j) { return L.isGood[j] != 0; });
        else ini_prep_point(L, i, L.slotOf[i], na, nb_, nc, idv[i], base, [&](int j) { return L.isGood[j] && L.isGood_new[j]; });
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) P.ctl->prepReady = 1;
}

// ---------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------
struct ldso_initializer {
    int device = 0, w = 0, h = 0, levels = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    IniParams P;
    std::vector<void *> allocs, levelAllocs;
    float *d_first[INI_MAXL] = {nullptr}, *d_new[INI_MAXL] = {nullptr};
    float *d_color = nullptr;
    int n[INI_MAXL] = {0};
    size_t ldsBytes = 0;
    int prepBlocks = 1;                 // k_ini_prep: one thread per point of the largest level
    int lastSteps = 0, stepsTaken = 0;  // control steps of the previous frame / of the frame just read back (get_state)
    int firstSteps = 0;                 // ldso_init_set_schedule: control steps of the first batch (0 = by the previous frame)
    bool prepareOnGrid = true;          // ldso_init_set_schedule: k_ini_prep launches (false: the control block prepares every sweep)
    bool frameDone = false;
    bool snappedAtFrameStart = false;   // host copy of the state's snapped (get_state / set_state / set_first): before the snap optReg does not sweep and the
                                        // k_ini_prep launches would be empty (the frame that snaps prepares its sweeps in the control block)
    bool haveFirst = false, haveNew = false;
};

template <class T> static int ini_alloc(std::vector<void *> &v, T **p, size_t n) {
    void *q = nullptr;
    CHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    // hipMemset on device memory is asynchronous (legacy null stream) and the handles work on NON-BLOCKING streams, which do not order themselves
    // behind it: without the wait a zero-fill that is still queued (the null stream busy with another library's work, e.g. torch's) could land on top
    // of data the handle's first uploads / kernels have already written
    CHK(hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T)));
    CHK(hipStreamSynchronize(nullptr));
    v.push_back(q);
    *p = (T *) q;
    return LDSO_OK;
}
#define IA(vec, ptr, n) do { int r_ = ini_alloc(vec, &(ptr), (n)); if (r_ != LDSO_OK) return r_; } while (0)

// upload helper: host vector -> new device array
// The schedule of the optReg sweep: pass[i] for every point such that (a) every neighbour j < i of i sits in an EARLIER pass (i reads j's new value), (b) every
// neighbour j > i of i sits in the SAME or a later pass (i reads j's old value; reads come before writes within a pass) and (c) no pass holds more than `width`
// points.  The reference's in-place loop over i (CoarseInitializer.cc:430-459) is pass 0, 1, 2, ... of this schedule executed in order.
// Two constructions, the shorter one wins: first fit in index order (both conditions only look at lower indices), and list scheduling by the length of the
// chain that still hangs on a point (the dependency depth is a diagonal front through the raster; where it is wider than a pass, the points with the longest
// tails go first).  640 x 480 (8.4k / 18.4k / 17.2k / 3.8k points, depth 341 / 586 / 412 / 196): first fit 341 / 719 / 654 / 196 passes, list 341 / 642 / 606 / 196.
static bool ini_schedule_valid(int n, const int *nb, int width, const std::vector<int> &pass, int nPass) {
    std::vector<int> cnt(std::max(nPass, 1), 0);
    for (int i = 0; i < n; i++) {
        if (pass[i] < 0 || pass[i] >= nPass || ++cnt[pass[i]] > width) return false;
        for (int q = 0; q < 10; q++) {
            const int j = nb[(size_t) i * 10 + q];
            if (j < 0 || j == i) continue;
            if (j < i ? !(pass[j] < pass[i]) : !(pass[j] >= pass[i])) return false;
        }
    }
    return true;
}
static int ini_sweep_schedule(int n, const int *nb, int width, int *passOut) {
    if (n <= 0) return 0;
    // first fit
    std::vector<int> ff(n, 0), rd(n, 0), fillOf;
    for (int i = 0; i < n; i++) {
        int e = rd[i];
        for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * 10 + q]; if (j >= 0 && j < i) e = std::max(e, ff[j] + 1); }
        while (e < (int) fillOf.size() && fillOf[e] >= width) e++;
        if (e >= (int) fillOf.size()) fillOf.resize(e + 1, 0);
        ff[i] = e;
        for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * 10 + q]; if (j > i && j < n) rd[j] = std::max(rd[j], e); }
        fillOf[e]++;
    }
    const int nFF = (int) fillOf.size();
    // list scheduling.  after[j]: the points i > j that have j as a neighbour (i waits for j's pass to be over); before[j]: the readers i < j of j (j must not
    // come before them); tail[i]: passes that must still follow the pass of i
    std::vector<int> offA(n + 1, 0), offB(n + 1, 0), waits(n, 0);
    for (int i = 0; i < n; i++) for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * 10 + q]; if (j < 0 || j >= n || j == i) continue; if (j < i) { offA[j + 1]++; waits[i]++; } else offB[j + 1]++; }
    for (int i = 0; i < n; i++) { offA[i + 1] += offA[i]; offB[i + 1] += offB[i]; }
    std::vector<int> after(offA[n]), before(offB[n]), curA(offA.begin(), offA.end() - 1), curB(offB.begin(), offB.end() - 1);
    for (int i = 0; i < n; i++) for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * 10 + q]; if (j < 0 || j >= n || j == i) continue; if (j < i) after[curA[j]++] = i; else before[curB[j]++] = i; }
    std::vector<int> tail(n, 0);
    for (int i = n - 1; i >= 0; i--) {
        int t = 0;
        for (int a = offA[i]; a < offA[i + 1]; a++) t = std::max(t, tail[after[a]] + 1);
        for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * 10 + q]; if (j > i && j < n) t = std::max(t, tail[j]); }
        tail[i] = t;
    }
    std::vector<int> ls(n, -1), ready, chosen, deferred;
    std::vector<char> inPass(n, 0);
    for (int i = 0; i < n; i++) if (waits[i] == 0) ready.push_back(i);
    int left = n, t = 0;
    bool stuck = false;
    while (left > 0 && !stuck) {
        std::sort(ready.begin(), ready.end(), [&](int a, int b) { return tail[a] != tail[b] ? tail[a] > tail[b] : a < b; });
        chosen.clear(); deferred.clear();
        auto free_ = [&](int i) { for (int a = offB[i]; a < offB[i + 1]; a++) { const int k = before[a]; if (ls[k] < 0 && !inPass[k]) return false; } return true; };
        for (int i : ready) { if ((int) chosen.size() < width && free_(i)) { chosen.push_back(i); inPass[i] = 1; } else deferred.push_back(i); }
        for (bool again = true; again && (int) chosen.size() < width;) {          // readers chosen later in the priority order free the points they held back
            again = false;
            for (size_t d = 0; d < deferred.size(); d++) {
                const int i = deferred[d];
                if (i >= 0 && (int) chosen.size() < width && free_(i)) { chosen.push_back(i); inPass[i] = 1; deferred[d] = -1; again = true; }
            }
        }
        if (chosen.empty()) { stuck = true; break; }
        ready.clear();
        for (int i : deferred) if (i >= 0) ready.push_back(i);
        for (int i : chosen) { ls[i] = t; inPass[i] = 0; left--; }
        for (int i : chosen) for (int a = offA[i]; a < offA[i + 1]; a++) if (--waits[after[a]] == 0) ready.push_back(after[a]);
        t++;
    }
    const bool useList = !stuck && t < nFF && ini_schedule_valid(n, nb, width, ls, t);
    for (int i = 0; i < n; i++) passOut[i] = useList ? ls[i] : ff[i];
    return useList ? t : nFF;
}

template <class T> static int ini_upload(ldso_initializer *H, T **dst, const std::vector<T> &src) {
    IA(H->levelAllocs, *dst, src.size());
    if (!src.empty()) CHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return LDSO_OK;
}

// SoA <-> record conversion
#define INI_FIELDS(X) X(u) X(v) X(idepth) X(idepth_new) X(iR) X(iRSumNum) X(lastHessian) X(lastHessian_new) X(maxstep) X(outlierTH)

static int ini_put_points(ldso_initializer *H, int l, const ldso_init_point_t *pts) {
    IniLevel &L = H->P.L[l];
    const int n = H->n[l];
    std::vector<float> f(n);
    std::vector<int> g(n);
#define X(name) for (int i = 0; i < n; i++) f[i] = pts[i].name; if (n) CHK(hipMemcpy(L.name, f.data(), (size_t) n * 4, hipMemcpyHostToDevice));
    INI_FIELDS(X)
#undef X
#define XE(dst, expr) for (int i = 0; i < n; i++) f[i] = pts[i].expr; if (n) CHK(hipMemcpy(L.dst, f.data(), (size_t) n * 4, hipMemcpyHostToDevice));
    XE(energy0, energy[0]) XE(energy1, energy[1]) XE(energy_new0, energy_new[0]) XE(energy_new1, energy_new[1])
#undef XE
#define XI(dst, expr) for (int i = 0; i < n; i++) g[i] = pts[i].expr; if (n) CHK(hipMemcpy(L.dst, g.data(), (size_t) n * 4, hipMemcpyHostToDevice));
    XI(isGood, isGood) XI(isGood_new, isGood_new)
#undef XI
    return LDSO_OK;
}

extern "C" {

int ldso_init_create(int device, int w, int h, int levels, ldso_initializer_t **out) {
    REQ(out && w > 16 && h > 16 && levels >= 1 && levels <= INI_MAXL && (w >> (levels - 1)) >= 8, "ldso_init_create: bad arguments (at most 5 pyramid levels)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { ldso_set_error("no HIP device visible"); return LDSO_E_NODEVICE; }
    REQ(device >= 0 && device < ndev, "ldso_init_create: device index out of range");
    CHK(hipSetDevice(device));
    ldso_initializer *H = new ldso_initializer();
    H->device = device; H->w = w; H->h = h; H->levels = levels;
    CHK(hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking));
    H->ownStream = true;
    memset(&H->P, 0, sizeof(H->P));
    H->P.levels = levels; H->P.fixAffine = 1; H->P.huberTH = 9.0f; H->P.firstExposure = 1; H->P.newExposure = 1;
    for (int l = 0; l < levels; l++) {
        const size_t npx = (size_t) (w >> l) * (h >> l);
        IA(H->allocs, H->d_first[l], npx * 3); IA(H->allocs, H->d_new[l], npx * 3);
        H->P.L[l].first = H->d_first[l]; H->P.L[l].cur = H->d_new[l];
        H->P.L[l].w = w >> l; H->P.L[l].h = h >> l;
    }
    IA(H->allocs, H->d_color, (size_t) w * h);
    IA(H->allocs, H->P.ctl, 1);
    IA(H->allocs, H->P.part, (size_t) INI_MAXBLK * INI_NPART);
    IniCtl c; memset(&c, 0, sizeof(c));
    c.Tcur[0] = c.Tcur[5] = c.Tcur[10] = 1.0; c.Tnew[0] = c.Tnew[5] = c.Tnew[10] = 1.0; c.frameID = -1; c.done = 1;
    CHK(hipMemcpy(H->P.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
    *out = H;
    return LDSO_OK;
}

int ldso_init_destroy(ldso_initializer_t *H) {
    if (!H) return LDSO_OK;
    hipSetDevice(H->device);
    hipDeviceSynchronize();
    for (void *p : H->allocs) hipFree(p);
    for (void *p : H->levelAllocs) hipFree(p);
    if (H->ownStream && H->stream) hipStreamDestroy(H->stream);
    delete H;
    return LDSO_OK;
}

int ldso_init_set_stream(ldso_initializer_t *H, void *s) {
    REQ(H, "null handle");
    if (H->ownStream && H->stream) { hipStreamSynchronize(H->stream); if (s) { hipStreamDestroy(H->stream); H->ownStream = false; } }
    if (s) { H->stream = (hipStream_t) s; H->ownStream = false; }
    else if (!H->ownStream) { CHK(hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking)); H->ownStream = true; }
    return LDSO_OK;
}

static int ini_images(ldso_initializer *H, const float *irr, float *const *levels) {
    CHK(hipSetDevice(H->device));
    CHK(hipMemcpyAsync(H->d_color, irr, (size_t) H->w * H->h * 4, hipMemcpyHostToDevice, H->stream));
    CHK(img_launch_make_images(H->d_color, H->w, H->h, H->levels, levels, H->stream));
    CHK(hipStreamSynchronize(H->stream));      // the host buffer may be reused by the caller
    return LDSO_OK;
}

int ldso_init_set_first(ldso_initializer_t *H, const float calib[4], const float *irradiance, float ab_exposure,
                        const ldso_init_point_t *const *points, const int *n_points, float huberTH, int fixAffine) {
    REQ(H && calib && irradiance && points && n_points, "ldso_init_set_first: null argument");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamSynchronize(H->stream));
    for (void *p : H->levelAllocs) hipFree(p);
    H->levelAllocs.clear();
    H->P.huberTH = huberTH; H->P.fixAffine = fixAffine ? 1 : 0; H->P.firstExposure = ab_exposure;
    // makeK (:689-715): doubles from the float level-0 intrinsics
    double fx[INI_MAXL], fy[INI_MAXL], cx[INI_MAXL], cy[INI_MAXL];
    fx[0] = calib[0]; fy[0] = calib[1]; cx[0] = calib[2]; cy[0] = calib[3];
    for (int l = 1; l < H->levels; l++) {
        fx[l] = fx[l - 1] * 0.5; fy[l] = fy[l - 1] * 0.5;
        cx[l] = (cx[0] + 0.5) / ((int) 1 << l) - 0.5; cy[l] = (cy[0] + 0.5) / ((int) 1 << l) - 0.5;
    }
    size_t maxN = 64;
    for (int l = 0; l < H->levels; l++) {
        IniLevel &L = H->P.L[l];
        const int n = n_points[l];
        REQ(n >= 0 && n <= 36000, "ldso_init_set_first: more than 36000 points on one level (LDS working set of the sweeps)");
        REQ(n == 0 || points[l], "ldso_init_set_first: null point array");
        H->n[l] = n; L.n = n;
        maxN = std::max<size_t>(maxN, n);
        L.fx = (float) fx[l]; L.fy = (float) fy[l]; L.cx = (float) cx[l]; L.cy = (float) cy[l];
        // K^-1 of the upper-triangular K in double (Eigen's cofactor inverse gives the same entries up to 1 ulp of double)
        for (int q = 0; q < 9; q++) L.Ki[q] = 0;
        L.Ki[0] = 1.0 / fx[l]; L.Ki[2] = -cx[l] / fx[l]; L.Ki[4] = 1.0 / fy[l]; L.Ki[5] = -cy[l] / fy[l]; L.Ki[8] = 1.0;
#define X(name) IA(H->levelAllocs, L.name, n);
        INI_FIELDS(X)
        X(energy0) X(energy1) X(energy_new0) X(energy_new1) X(isGood) X(isGood_new)
#undef X
        IA(H->levelAllocs, L.jb[0], (size_t) n * 10); IA(H->levelAllocs, L.jb[1], (size_t) n * 10);
        const ldso_init_point_t *pts = points[l];
        const int nUp = (l + 1 < H->levels) ? n_points[l + 1] : 0, nDown = (l > 0) ? n_points[l - 1] : 0;
        std::vector<int> parent(n), nb((size_t) n * INI_NB, -1);
        for (int i = 0; i < n; i++) {
            parent[i] = pts[i].parent;
            REQ(l + 1 >= H->levels || (parent[i] >= 0 && parent[i] < nUp), "ldso_init_set_first: parent index out of range");
            for (int q = 0; q < 10; q++) {
                const int j = pts[i].neighbours[q];
                REQ(j >= -1 && j < n, "ldso_init_set_first: neighbour index out of range");
                nb[(size_t) i * INI_NB + q] = j;
            }
        }
        { int r_ = ini_upload(H, &L.parent, parent); if (r_ != LDSO_OK) return r_; }
        { int r_ = ini_upload(H, &L.nb, nb); if (r_ != LDSO_OK) return r_; }
        // optReg sweep schedule (all levels; two lanes per point: passes of <= 32 points): ini_sweep_schedule
        {
            std::vector<int> pass(n, 0), nb10((size_t) n * 10);
            for (int i = 0; i < n; i++) for (int q = 0; q < 10; q++) nb10[(size_t) i * 10 + q] = nb[(size_t) i * INI_NB + q];
            L.nPass2 = ini_sweep_schedule(n, nb10.data(), 32, pass.data());
            std::vector<int> slotOf(n), at(L.nPass2, 0), idleSlot;
            for (int i = 0; i < n; i++) slotOf[i] = pass[i] * 32 + at[pass[i]]++;
            for (int p = 0; p < L.nPass2 + INI_SWPAD; p++) for (int q = (p < L.nPass2 ? at[p] : 0); q < 32; q++) idleSlot.push_back(p * 32 + q);
            L.nIdle = (int) idleSlot.size();
            { int *p = nullptr; int r_ = ini_upload(H, &p, slotOf); if (r_ != LDSO_OK) return r_; L.slotOf = p; }
            { int *p = nullptr; int r_ = ini_upload(H, &p, idleSlot); if (r_ != LDSO_OK) return r_; L.idleSlot = p; }
            IA(H->levelAllocs, L.swRec, (size_t) (L.nPass2 + INI_SWPAD) * 64 * 2);      // per-lane inputs: ini_prep writes them before every sweep (the places without a point: once)
        }
        if (l + 1 < H->levels) { L.nPass = 0; L.sched = nullptr; L.schedOff = nullptr; L.schedNb = nullptr; }
        else {
        // resetPoints sweep schedule (top level): dep(i) = max(dep(j) + 1 over neighbours j < i, dep(k) over readers k < i of i)
        std::vector<int> dep(n, 0);
        {
            std::vector<int> rd(n, 0);      // max dep of the lower-indexed readers seen so far
            for (int i = 0; i < n; i++) {
                int d = rd[i];
                for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * INI_NB + q]; if (j >= 0 && j < i) d = std::max(d, dep[j] + 1); }
                dep[i] = d;
                for (int q = 0; q < 10; q++) { const int j = nb[(size_t) i * INI_NB + q]; if (j > i) rd[j] = std::max(rd[j], d); }
            }
        }
        int nDep = 0;
        for (int i = 0; i < n; i++) nDep = std::max(nDep, dep[i] + 1);
        std::vector<std::vector<int>> byDep(nDep);
        for (int i = 0; i < n; i++) byDep[dep[i]].push_back(i);
        std::vector<int> sched;
        for (int d = 0; d < nDep; d++)
            for (size_t o = 0; o < byDep[d].size(); o += 64) {
                for (size_t q = 0; q < 64; q++) sched.push_back(o + q < byDep[d].size() ? byDep[d][o + q] : -1);
            }
        L.nPass = (int) (sched.size() / 64);
        sched.resize(sched.size() + (size_t) INI_SWPAD * 64, -1);
        { int *p = nullptr; int r_ = ini_upload(H, &p, sched); if (r_ != LDSO_OK) return r_; L.sched = p; }
        {
            const int dummy = n * 4;
            std::vector<int> snb(sched.size() * INI_NB, dummy), soff(sched.size(), dummy);
            for (size_t q = 0; q < sched.size(); q++)
                if (sched[q] >= 0) {
                    soff[q] = sched[q] * 4;
                    for (int e = 0; e < 10; e++) { const int j = nb[(size_t) sched[q] * INI_NB + e]; snb[q * INI_NB + e] = (j >= 0) ? j * 4 : dummy; }
                }
            int *p = nullptr; int r_ = ini_upload(H, &p, snb); if (r_ != LDSO_OK) return r_; L.schedNb = p;
            p = nullptr; r_ = ini_upload(H, &p, soff); if (r_ != LDSO_OK) return r_; L.schedOff = p;
        }
        }
        // children lists (points of level l-1 whose parent is p), ascending child index
        std::vector<int> off(n + 1, 0), idx(nDown);
        if (l > 0) {
            const ldso_init_point_t *ch = points[l - 1];
            for (int c = 0; c < nDown; c++) { REQ(ch[c].parent >= 0 && ch[c].parent < n, "ldso_init_set_first: parent index out of range"); off[ch[c].parent + 1]++; }
            for (int p = 0; p < n; p++) off[p + 1] += off[p];
            std::vector<int> cur(off.begin(), off.end() - 1);
            for (int c = 0; c < nDown; c++) idx[cur[ch[c].parent]++] = c;
        }
        { int *p = nullptr; int r_ = ini_upload(H, &p, off); if (r_ != LDSO_OK) return r_; L.childOff = p; }
        { int *p = nullptr; int r_ = ini_upload(H, &p, idx); if (r_ != LDSO_OK) return r_; L.childIdx = p; }
        { int r_ = ini_put_points(H, l, pts); if (r_ != LDSO_OK) return r_; }
    }
    H->ldsBytes = (maxN + INI_LDS_EXTRA) * sizeof(float);
    H->prepBlocks = (int) std::max<size_t>(1, (maxN + INI_PT - 1) / INI_PT);
    CHK(hipFuncSetAttribute((const void *) k_ini_ctl, hipFuncAttributeMaxDynamicSharedMemorySize, (int) H->ldsBytes));
    H->snappedAtFrameStart = false;
    // state of setFirst (:612-614)
    IniCtl c; memset(&c, 0, sizeof(c));
    c.Tcur[0] = c.Tcur[5] = c.Tcur[10] = 1.0; c.Tnew[0] = c.Tnew[5] = c.Tnew[10] = 1.0; c.done = 1;
    CHK(hipMemcpy(H->P.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
    { int r_ = ini_images(H, irradiance, H->d_first); if (r_ != LDSO_OK) return r_; }
    H->haveFirst = true; H->haveNew = false;
    return LDSO_OK;
}

int ldso_init_set_new_frame(ldso_initializer_t *H, const float *irradiance, float ab_exposure) {
    REQ(H && irradiance, "ldso_init_set_new_frame: null argument");
    REQ(H->haveFirst, "ldso_init_set_new_frame: no first frame");
    H->P.newExposure = ab_exposure;
    int r_ = ini_images(H, irradiance, H->d_new);
    if (r_ != LDSO_OK) return r_;
    H->haveNew = true;
    return LDSO_OK;
}

int ldso_init_get_state(ldso_initializer_t *H, ldso_init_state_t *s) {
    REQ(H && s, "null argument");
    CHK(hipSetDevice(H->device));
    IniCtl c;
    CHK(hipMemcpyAsync(&c, H->P.ctl, sizeof(c), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    memcpy(s->thisToNext, c.Tcur, sizeof(c.Tcur));
    H->snappedAtFrameStart = c.snapped != 0; H->frameDone = c.done != 0; H->stepsTaken = c.steps;
    s->aff_a = c.aCur; s->aff_b = c.bCur; s->snapped = c.snapped; s->snappedAt = c.snappedAt; s->frameID = c.frameID;
    s->ready = c.snapped && c.frameID > c.snappedAt + 5; s->evals = c.evals; s->pad_ = 0;
    return LDSO_OK;
}

int ldso_init_set_state(ldso_initializer_t *H, const ldso_init_state_t *s) {
    REQ(H && s, "null argument");
    CHK(hipSetDevice(H->device));
    IniCtl c;
    CHK(hipMemcpyAsync(&c, H->P.ctl, sizeof(c), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    memcpy(c.Tcur, s->thisToNext, sizeof(c.Tcur)); memcpy(c.Tnew, s->thisToNext, sizeof(c.Tnew));
    c.aCur = (float) s->aff_a; c.bCur = (float) s->aff_b; c.aNew = c.aCur; c.bNew = c.bCur;
    c.snapped = s->snapped; c.snappedAt = s->snappedAt; c.frameID = s->frameID;
    H->snappedAtFrameStart = c.snapped != 0;
    CHK(hipMemcpyAsync(H->P.ctl, &c, sizeof(c), hipMemcpyHostToDevice, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

int ldso_init_track_frame(ldso_initializer_t *H, const float *irradiance, float ab_exposure, ldso_init_state_t *state_out) {
    REQ(H, "null handle");
    REQ(H->haveFirst, "ldso_init_track_frame: no first frame");
    CHK(hipSetDevice(H->device));
    if (irradiance) { int r_ = ldso_init_set_new_frame(H, irradiance, ab_exposure); if (r_ != LDSO_OK) return r_; }
    REQ(H->haveNew, "ldso_init_track_frame: no new frame");
    const int maxIterations[5] = {5, 5, 10, 30, 50};
    int steps = 0;                                                          // the most control steps a frame can take
    for (int l = 0; l < H->levels; l++) steps += maxIterations[l] + 2;
    steps += 2 * (H->levels - 1);                                           // the transition steps of a snapped frame (down and up)
    // Control steps behind the one that finishes the frame return at once, but each still costs a dispatch (2-3 us, three kernels per step): a frame takes 19-40 of
    // the 58 possible steps at four levels.  So: enqueue what the previous frame took plus a margin, read the state back (the call does that anyway), and enqueue
    // the rest only if the frame is not finished - one more round trip in the rare case, 10-30 empty steps fewer in the usual one.  (First frame: half of the maximum.)
    int first = std::min(steps, H->firstSteps > 0 ? H->firstSteps : H->lastSteps > 0 ? H->lastSteps + H->lastSteps / 4 + 4 : (steps + 1) / 2);
    const bool prep = H->snappedAtFrameStart && H->prepareOnGrid;
    hipLaunchKernelGGL(k_ini_ctl, dim3(1), dim3(INI_CT), H->ldsBytes, H->stream, H->P, INI_BEGIN);
    ldso_init_state_t st;
    for (int from = 0; from < steps;) {
        for (int p = from; p < first; p++) {
            hipLaunchKernelGGL(k_ini_eval, dim3(INI_MAXBLK), dim3(INI_NT), 0, H->stream, H->P, 0);
            if (prep) hipLaunchKernelGGL(k_ini_prep, dim3(H->prepBlocks), dim3(INI_PT), 0, H->stream, H->P);
            hipLaunchKernelGGL(k_ini_ctl, dim3(1), dim3(INI_CT), H->ldsBytes, H->stream, H->P, INI_STEP);
        }
        CHK(hipGetLastError());
        int r_ = ldso_init_get_state(H, &st);
        if (r_ != LDSO_OK) return r_;
        if (H->frameDone) break;
        from = first; first = steps;
    }
    REQ(H->frameDone, "ldso_init_track_frame: the frame did not finish within the maximal number of control steps (internal)");
    H->lastSteps = H->stepsTaken;
    bool fin = true;
    for (int q = 0; q < 12; q++) fin = fin && std::isfinite(st.thisToNext[q]);
    if (state_out) *state_out = st;
    if (!fin) { ldso_set_error("ldso_init_track_frame: non-finite pose"); return LDSO_E_NONFINITE; }
    return LDSO_OK;
}

int ldso_init_set_schedule(ldso_initializer_t *H, int first_steps, int prepare_on_grid) {
    REQ(H && first_steps >= 0, "ldso_init_set_schedule: bad argument");
    H->firstSteps = first_steps; H->prepareOnGrid = prepare_on_grid != 0;
    return LDSO_OK;
}

// host logic only (no device): the optReg sweep schedule ldso_init_set_first builds for a level
int ldso_init_sweep_schedule(int n, const int *neighbours, int width, int *pass_out) {
    REQ(n >= 0 && (n == 0 || (neighbours && pass_out)) && width >= 1, "ldso_init_sweep_schedule: bad argument");
    for (size_t q = 0; q < (size_t) n * 10; q++) REQ(neighbours[q] >= -1 && neighbours[q] < n, "ldso_init_sweep_schedule: neighbour index out of range");
    std::vector<int> pass(n);
    const int np = ini_sweep_schedule(n, neighbours, width, pass.data());
    REQ(ini_schedule_valid(n, neighbours, width, pass, np), "ldso_init_sweep_schedule: internal error (schedule violates the update order)");
    for (int i = 0; i < n; i++) pass_out[i] = pass[i];
    return np;
}

// debug (LDSO_STAMPS builds): accumulated device-side ticks (100 MHz): sweep ticks, sweep passes, control-kernel ticks, sweeps, fill + prepare ticks,
// ticks up to the accept decision, ticks of the next increment (solve, exp), -
int ldso_init_debug_counters(ldso_initializer_t *H, long long out[8]) {
    REQ(H && out, "null argument");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamSynchronize(H->stream));
    IniCtl c;
    CHK(hipMemcpy(&c, H->P.ctl, sizeof(c), hipMemcpyDeviceToHost));
    out[0] = c.dbgSweepTicks; out[1] = c.dbgSweepPasses; out[2] = c.dbgCtlTicks; out[3] = c.dbgSweeps; out[4] = c.dbgPrepTicks; out[5] = c.dbgFrontTicks; out[6] = c.dbgTailTicks; out[7] = c.dbgSpare;
    return LDSO_OK;
}

int ldso_init_get_points(ldso_initializer_t *H, int l, ldso_init_point_t *out) {
    REQ(H && out && l >= 0 && l < H->levels, "ldso_init_get_points: bad argument");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamSynchronize(H->stream));
    IniCtl c;
    CHK(hipMemcpy(&c, H->P.ctl, sizeof(c), hipMemcpyDeviceToHost));
    REQ(!c.applyPending, "ldso_init_get_points: a step is pending (internal)");
    const IniLevel &L = H->P.L[l];
    const int n = H->n[l];
    std::vector<float> f(n);
    std::vector<int> g(n), nb((size_t) n * INI_NB);
#define X(name) if (n) CHK(hipMemcpy(f.data(), L.name, n * 4, hipMemcpyDeviceToHost)); for (int i = 0; i < n; i++) out[i].name = f[i];
    INI_FIELDS(X)
#undef X
#define XE(src, expr) if (n) CHK(hipMemcpy(f.data(), L.src, n * 4, hipMemcpyDeviceToHost)); for (int i = 0; i < n; i++) out[i].expr = f[i];
    XE(energy0, energy[0]) XE(energy1, energy[1]) XE(energy_new0, energy_new[0]) XE(energy_new1, energy_new[1])
#undef XE
#define XI(src, expr) if (n) CHK(hipMemcpy(g.data(), L.src, n * 4, hipMemcpyDeviceToHost)); for (int i = 0; i < n; i++) out[i].expr = g[i];
    XI(isGood, isGood) XI(isGood_new, isGood_new) XI(parent, parent)
#undef XI
    if (n) CHK(hipMemcpy(nb.data(), L.nb, (size_t) n * INI_NB * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) for (int q = 0; q < 10; q++) out[i].neighbours[q] = nb[(size_t) i * INI_NB + q];
    return LDSO_OK;
}

int ldso_init_set_points(ldso_initializer_t *H, int l, const ldso_init_point_t *pts) {
    REQ(H && pts && l >= 0 && l < H->levels, "ldso_init_set_points: bad argument");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamSynchronize(H->stream));
    return ini_put_points(H, l, pts);
}

int ldso_init_calc_res_and_gs(ldso_initializer_t *H, int lvl, const double refToNew[12], double aff_a, double aff_b,
                              float *Hm, float *b, float *Hsc, float *bsc, float *res, float *ec) {
    REQ(H && refToNew && lvl >= 0 && lvl < H->levels, "ldso_init_calc_res_and_gs: bad argument");
    REQ(H->haveFirst && H->haveNew, "ldso_init_calc_res_and_gs: frames missing");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamSynchronize(H->stream));
    IniCtl c;
    CHK(hipMemcpy(&c, H->P.ctl, sizeof(c), hipMemcpyDeviceToHost));
    memcpy(c.Tnew, refToNew, sizeof(c.Tnew));
    c.aNew = (float) aff_a; c.bNew = (float) aff_b; c.lvl = lvl;
    CHK(hipMemcpy(H->P.ctl, &c, sizeof(c), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ini_eval, dim3(INI_MAXBLK), dim3(INI_NT), 0, H->stream, H->P, 1);
    hipLaunchKernelGGL(k_ini_ctl, dim3(1), dim3(INI_CT), H->ldsBytes, H->stream, H->P, INI_STAGE);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(H->stream));
    CHK(hipMemcpy(&c, H->P.ctl, sizeof(c), hipMemcpyDeviceToHost));
    if (Hm) memcpy(Hm, c.Hn, sizeof(c.Hn));
    if (b) memcpy(b, c.bn, sizeof(c.bn));
    if (Hsc) memcpy(Hsc, c.Hscn, sizeof(c.Hscn));
    if (bsc) memcpy(bsc, c.bscn, sizeof(c.bscn));
    if (res) memcpy(res, c.resNew, sizeof(c.resNew));
    if (ec) memcpy(ec, c.ec, sizeof(c.ec));
    return LDSO_OK;
}

}  // extern "C"
