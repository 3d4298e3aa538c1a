// ba_dev.h — device-side data layout of one bundle-adjustment handle (gfx950).
//
// HBM layout (sizes at the BASELINE C3 window: F=7, P=2000, 640x480):
//   img[f]      level-0 image of frame f exactly as FrameHessian::dIp[0]: 12-byte AoS (I,dx,dy), w*h px   (3.7 MB each)
//   frames[F]   DevFrame: eval point, state, backup, step, priors (double)                                  (<1 KB each)
//   pairs[F*F]  DevPair at [host*F + target]: FrameFramePrecalc floats + adHTdeltaF + max frameEnergyTH    (144 B each)
//   ad*[F*F]    adjoints at [host + target*F] (reference slot order), double and float copies
//   points      SoA arrays of P entries (geometry, inverse depth) + one 64-byte PtRec per point and set (Schur scalars)
//   residuals   dense slot table [P][FS], slot index == target frame idx, FS = F rounded up to 8: SlotTab (static, 16 B) and
//               one 64-byte SlotRec per slot and set
//   ResSet x2   ping-pong "applied" residual state + accumulator partials (linearize writes the other
//               set; applyRes is a pointer swap, so a rejected LM step costs nothing)
#pragma once
#include <stdint.h>
#include "../../include/ldso_window.h"

#define LD_MAXF LDSO_MAX_FRAMES
#define LD_TOPN 91            // unique entries of the 13x13 symmetric relative Hessian block
#ifndef LD_WAVES
#define LD_WAVES 8            // waves per linearize block
#endif
#define LD_GEXTRA 8           // per-point extras appended to a G row: Hcd[4], bdSum, HdiF, pad, pad

struct DevPair {
    float KRKi[9];
    float Kt[3];
    float R0[9];
    float t0[3];
    float aff[2];
    float b0;
    float thMax;     // max(host.frameEnergyTH, target.frameEnergyTH)
    float dp[8];     // adHTdeltaF[host + target*F]
};

struct DevFrame {
    double evalPT[12];
    double state[10], state_zero[10], state_backup[10], step[10];
    double prior[8], delta[8], delta_prior[8];
    double PRE_w2c[12], PRE_c2w[12];
    double ns_pose[36], ns_scale[6], ns_affine[8];
    float ab_exposure, frameEnergyTH;
    int32_t frameID, imgSlot;
};

struct DevCalib {
    double value[4], value_zero[4], value_backup[4], step[4];
    float sf[4];        // value_scaledf: fx fy cx cy
    float si[4];        // value_scaledi: 1/fx 1/fy -cx/fx -cy/fy
    float cDeltaF[4];
    float pad_[4];
};

// ---- records (round 4): what a wavefront reads of a point / a residual slot arrives in FEW WIDE loads instead of ~30 four-byte arrays ----
// One residual slot of a set, 64 bytes.  Interleaved so that lane k of the slot's 8-lane group loads (and stores) ONE dwordx2 holding
// exactly its own pair: e[k].jp = component k of JpJdF, e[k].m = the k-th of the slot's eight scalars below.
#define LD_SM_CEN0 0       // centerProjectedTo[0..2] in m of lanes 0..2 (the lanes that use them)
#define LD_SM_ENERGY 3     // state_energy
#define LD_SM_EWO 4        // state_NewEnergyWithOutlier of the pass that produced this set (-1: none)
#define LD_SM_STATE 5      // ResState (int)
#define LD_SM_ACTIVE 6     // isActiveAndIsGoodNEW (int)
#define LD_SM_REMOVE 7     // (fix mode) residual became inactive -> host drops it (int)
struct SlotRec {
    struct { float jp; union { float f; int32_t i; } m; } e[8];
};
static_assert(sizeof(SlotRec) == 64, "SlotRec is one 64-byte line");
// static part of a slot (set by ldso_ba_set_window), 16 bytes: one dwordx4 per 8-lane group
struct SlotTab { int32_t rflat, rlin, rnew, rlidx; };
// Per-point scalars of a set, 64 bytes: read with ONE scalar load (a wavefront works on one point: the address is wave-uniform), written by
// lanes 0..3 as four dwordx4 of one store instruction.
struct alignas(64) PtRec {
    float HdiF, bdSumF, idH; int32_t nActive;        // lane 0
    float HcdA[4];                                   // lane 1
    float HcdL[4];                                   // lane 2
    float maxRelBS; int32_t numGood; float pad_[2];  // lane 3
};
static_assert(sizeof(PtRec) == 64, "PtRec is one 64-byte line");
struct PtAcc { float HddA, bdA, HddL, bdL; };        // accumulator scalars only the fetch functions read
// Geometry and inverse depth of a point, 64 bytes.  Read with one scalar load (dwords 0..7); the fused point step rewrites dwords 4..11 with
// two dwordx4 of one store instruction (lanes 1, 2).
struct alignas(64) PtGeo {
    float u, v, priorF, pad0_;                               // static (ldso_ba_set_window)
    float idepth, idepth_zero, step, idepth_backup;          // PointHessian::{idepth, idepth_zero, step, idepth_backup}
    float lastHdiF, lastBdSumF, lastIdH, pad1_;              // PointHessian::{HdiF, bdSumF, idepth_hessian} as the LAST solveSystemF left them (AccumulatedSCHessian.cc:9-51):
                                                             // the PtRec copies belong to the NEXT solve (the fused linearize pass already holds the new linearisation's scalars)
    float pad2_[4];
};
static_assert(sizeof(PtGeo) == 64, "PtGeo is one 64-byte line");
struct PtCw { float color, weight; };
// the lane <-> dword maps the kernels rely on (ba_linearize.hip: load_point, the slot / point stores)
#include <stddef.h>
static_assert(offsetof(SlotRec, e[3].m) == 8 * 3 + 4 && offsetof(SlotRec, e[7].jp) == 56, "SlotRec: lane k owns dwords 2k (JpJdF[k]) and 2k+1 (scalar k)");
static_assert(sizeof(SlotTab) == 16 && offsetof(SlotTab, rlin) == 4 && offsetof(SlotTab, rnew) == 8 && offsetof(SlotTab, rlidx) == 12, "SlotTab: x = rflat, y = rlin, z = rnew, w = rlidx");
static_assert(offsetof(PtRec, HdiF) == 0 && offsetof(PtRec, bdSumF) == 4 && offsetof(PtRec, idH) == 8 && offsetof(PtRec, nActive) == 12, "PtRec: lane 0's dwordx4");
static_assert(offsetof(PtRec, HcdA) == 16 && offsetof(PtRec, HcdL) == 32 && offsetof(PtRec, maxRelBS) == 48 && offsetof(PtRec, numGood) == 52, "PtRec: lanes 1..3");
static_assert(sizeof(PtAcc) == 16, "PtAcc: one dwordx4");
static_assert(offsetof(PtGeo, u) == 0 && offsetof(PtGeo, v) == 4 && offsetof(PtGeo, priorF) == 8 && offsetof(PtGeo, idepth) == 16 && offsetof(PtGeo, idepth_zero) == 20
              && offsetof(PtGeo, step) == 24 && offsetof(PtGeo, idepth_backup) == 28 && offsetof(PtGeo, lastHdiF) == 32 && offsetof(PtGeo, lastBdSumF) == 36 && offsetof(PtGeo, lastIdH) == 40,
              "PtGeo: dwords 0..7 are the scalar load, lane 1 stores dwords 4..7, lane 2 dwords 8..11");
static_assert(sizeof(PtCw) == 8, "PtCw: one dwordx2");

// One of the two ping-pong sets (see header comment).
struct ResSet {
    // Field ORDER matters: the batched linearisation kernel fetches these pointers just in time with one s_load_dwordx16 (ba_linearize.hip:
    // LDG16): the first eight pointers are one group (64 bytes).
    SlotRec *slot;         // [P*FS]
    PtRec *pt;             // [P]
    PtAcc *acc;            // [P]
    float *candE;          // [P] newest-frame candidate energy for setNewFrameEnergyTH (-1: none)
    float *G;              // [P][GS]  lifted Schur rows  g_p (8*FS frame entries + LD_GEXTRA)
    float *topA;           // [nChunks][FS][91]
    float *topL;           // [nChunks][FS][91]
    double *chunkEnergy;   // [nChunks]
    // ---- behind the group ------------------------------------------------------------------------------------------------
    int32_t *chunkCnt;     // [nChunks*2]  nres A, nres L
    float *chunkNID;       // [nChunks*2]  sum |idepth|, count   (doStepFromBackup statistics)
};

struct BaDims {
    int32_t F, FS, P, R, n, GS, w, h, nChunks, nL, nsg;
    int32_t pBegin, pEnd;      // shard of points owned by this rank
    float wM3G, hM3G;
    int32_t ks;                // K-splits (workgroups) per 16 x 16 Schur tile of the reduction (LD_SCT_KS for a lone window: latency; fewer for the windows of a batch: throughput)
};

struct BaPtrs {
    const float *img[LD_MAXF];
    DevFrame *frames;
    DevCalib *calib;
    DevPair *pairs;
    float *pairRt;                   // [F*F][12] at [host*F + target]: PRE_RTll (9) and PRE_tTll (3) of the CURRENT state (point activation)
    double *adHost, *adTarget;
    float *adHostF, *adTargetF;
    double *nsProj;                  // n*n projector onto the gauge nullspaces (reference ordering)
    double *HM, *bM;
    // points and residual slots: ONE group of eight consecutive pointers (ba_linearize.hip: LDG16) - everything a wave reads / writes of a point
    PtGeo *pgeo;                     // [P]     geometry + inverse depth + the scalars of the last solve, 64 bytes per point          ---- group B0
    PtCw *pcw;                       // [P*8]   (colour, weight) of the 8 pattern pixels
    SlotTab *rtab;                   // [P*FS]  residual slots: flat residual index (-1: none), linearised?, new?, index into Jlin / rtz
    int32_t *phost;                  // [P]
    ldso_rawjac_t *Jlin;             // linearised store
    float *rtz;
    int32_t *chunk_p0, *chunk_n;                                                                    // (end of group B0)
    int32_t *chunk_host;
    // solve-side buffers
    double *pairC;      // [F*F][PAIRC] lifted top contributions (A then L)
    float *scPart;      // [SC_SPLITS][n*(n+1)] Schur partials
    double *sys;        // HA,bA,HL,bL,Hsc,bsc,HFinal,bFinal,x  (debug/fetch)
    double *acc;        // GN fast path: [n*n+n] HFinal (lower triangle) | bFinal, initialised by k_linearize, accumulated by k_reduce (atomics)
    double *x;          // n
    float *xAd;         // [F*F*8] at [h*F+t]
    float *xc;          // 4 (calib step as float, = x[0:4])
    double *scalars;    // misc: [0] energy, [1] resInA, [2] resInL, [3] canbreak, [4] nonfinite flag, [5] sumNID mean
    double *energyLog;  // [64]
    ldso_rawjac_t *dumpJ;   // optional [R]
};

// first chunk of every host frame (chunks are host-major), passed to k_reduce by value: one load level less than a device array
struct ChunkStarts {
    int32_t v[LD_MAXF + 1];
};

// What a workgroup of the single-window GN linearisation (k_linearize_one) needs to find its chunk WITHOUT a table in memory: the chunks of a window
// are host-major, every host's points cut into pieces of `CH` points (build_chunks), so chunk c of host h starts at hostP0[h] + (c - cs[h]) CH.
struct LinHead {
    int32_t cs[LD_MAXF + 1];         // first chunk of every host (cs[F] = number of chunks)
    int32_t hostP0[LD_MAXF + 1];     // first point of every host inside the shard (hostP0[F] = one past the last point)
    int32_t CH, F;
};

// One window of a batch (ldso_ba_batch_*): everything the kernels of a GN iteration take as arguments for a single window, in
// device memory.  linBlock0 / redBlock0 = first workgroup of this window in the batched k_linearize / k_reduce launches.
// one workgroup of a batched k_linearize: its window (index into the launch's BatchItem table) and its chunk, resolved on the host - the
// kernel reads ONE 16-byte record instead of searching the window table and then three chunk arrays (two dependent memory round trips)
struct BatchBlock { int32_t win, p0, np, host_chunk; };      // host_chunk = host frame | chunk-in-window << 8

struct BatchItem {
    BaPtrs B;
    BaDims D;
    ResSet set[2];
    ChunkStarts cs;
    int32_t hasPrior, GSP, linBlock0, redBlock0;
};

// GN fast path: what k_linearize needs to initialise B.acc for the solve that follows it
struct GnInit {
    int enable, hasPrior;
    float calibPrior;
    int itCheck;        // >= 0: un-forced optimize(), iteration index (see LD_SC_STOP); -1 otherwise
};

// Un-forced optimize() (FullSystem.cc:829: `if (canbreak && iteration >= setting_minOptIterations) break;`) without a host round trip
// per iteration: k_gn_solve of iteration i stores i in scalars[LD_SC_STOP] when the loop has to end after that iteration, and the
// kernels of every later iteration return at once (itCheck > scalars[LD_SC_STOP]).  itCheck < 0: forced iterations, no check.
#define LD_SC_STOP 11
#define LD_ITER_SKIPPED(B, itCheck) ((itCheck) >= 0 && (double) (itCheck) > (B).scalars[LD_SC_STOP])

// Device-side phase stamps (scripts/dbg_gn.py): build with -DLDSO_STAMPS; they use energyLog[8..60] and therefore corrupt the
// energy log of optimize() runs with more than 6 iterations - never enable them in a product build.
#ifdef LDSO_STAMPS
#define LD_STAMP_ON 1
#else
#define LD_STAMP_ON 0
#endif

#define LD_PAIRC 296        // doubles per pair contribution: hh 64, tt 64, ht 64, hc 32, tc 32, cc 16, bh 8, bt 8, bc 4 (=292, padded)
#define LD_SC_SPLITS 16
#ifndef LD_SCT_KS
#define LD_SCT_KS 8          // GN fast path: K-splits (workgroups) per 16x16 Schur tile
#endif
#define LD_SYS_MATS 4       // HA, HL, Hsc, HFinal

#ifdef __HIPCC__
// The kernels of a GN iteration are chains of dependent memory levels (a 5 MB window on a chip that moves that in 0.7 us), and the first link of
// every chain is the kernel-argument block: 0.6 - 1.1 KB that the compiler fetches lazily, a few words at a time, each fetch a fresh scalar-cache
// line nobody on this CU has touched yet.  ld_touch_kernarg<LINES>() requests one dword of each of the first LINES 64-byte lines back to back and
// waits once: the block is in the scalar cache before the first real use.  LINES must not reach past the kernel's kernarg segment
// (explicit + hidden arguments: llvm-readelf --notes, .kernarg_segment_size).
template <int LINES> __device__ __forceinline__ void ld_touch_kernarg();
template <> __device__ __forceinline__ void ld_touch_kernarg<8>() {
    const unsigned long long ka = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr();
    int t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\ts_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7) : "s"(ka) : "memory");
}
template <> __device__ __forceinline__ void ld_touch_kernarg<10>() {
    const unsigned long long ka = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr();
    int t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    asm volatile("s_load_dword %0, %10, 0x0\n\ts_load_dword %1, %10, 0x40\n\ts_load_dword %2, %10, 0x80\n\ts_load_dword %3, %10, 0xc0\n\ts_load_dword %4, %10, 0x100\n\ts_load_dword %5, %10, 0x140\n\ts_load_dword %6, %10, 0x180\n\ts_load_dword %7, %10, 0x1c0\n\ts_load_dword %8, %10, 0x200\n\ts_load_dword %9, %10, 0x240\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9) : "s"(ka) : "memory");
}
template <> __device__ __forceinline__ void ld_touch_kernarg<12>() {
    const unsigned long long ka = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr();
    int t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;
    asm volatile("s_load_dword %0, %12, 0x0\n\ts_load_dword %1, %12, 0x40\n\ts_load_dword %2, %12, 0x80\n\ts_load_dword %3, %12, 0xc0\n\ts_load_dword %4, %12, 0x100\n\ts_load_dword %5, %12, 0x140\n\ts_load_dword %6, %12, 0x180\n\ts_load_dword %7, %12, 0x1c0\n\ts_load_dword %8, %12, 0x200\n\ts_load_dword %9, %12, 0x240\n\ts_load_dword %10, %12, 0x280\n\ts_load_dword %11, %12, 0x2c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10), "=&s"(t11) : "s"(ka) : "memory");
}
template <> __device__ __forceinline__ void ld_touch_kernarg<14>() {
    const unsigned long long ka = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr();
    int t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13;
    asm volatile("s_load_dword %0, %14, 0x0\n\ts_load_dword %1, %14, 0x40\n\ts_load_dword %2, %14, 0x80\n\ts_load_dword %3, %14, 0xc0\n\ts_load_dword %4, %14, 0x100\n\ts_load_dword %5, %14, 0x140\n\ts_load_dword %6, %14, 0x180\n\ts_load_dword %7, %14, 0x1c0\n\ts_load_dword %8, %14, 0x200\n\ts_load_dword %9, %14, 0x240\n\ts_load_dword %10, %14, 0x280\n\ts_load_dword %11, %14, 0x2c0\n\ts_load_dword %12, %14, 0x300\n\ts_load_dword %13, %14, 0x340\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10), "=&s"(t11), "=&s"(t12), "=&s"(t13) : "s"(ka) : "memory");
}
template <> __device__ __forceinline__ void ld_touch_kernarg<16>() {
    const unsigned long long ka = (unsigned long long) __builtin_amdgcn_kernarg_segment_ptr();
    int t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15;
    asm volatile("s_load_dword %0, %16, 0x0\n\ts_load_dword %1, %16, 0x40\n\ts_load_dword %2, %16, 0x80\n\ts_load_dword %3, %16, 0xc0\n\ts_load_dword %4, %16, 0x100\n\ts_load_dword %5, %16, 0x140\n\ts_load_dword %6, %16, 0x180\n\ts_load_dword %7, %16, 0x1c0\n\ts_load_dword %8, %16, 0x200\n\ts_load_dword %9, %16, 0x240\n\ts_load_dword %10, %16, 0x280\n\ts_load_dword %11, %16, 0x2c0\n\ts_load_dword %12, %16, 0x300\n\ts_load_dword %13, %16, 0x340\n\ts_load_dword %14, %16, 0x380\n\ts_load_dword %15, %16, 0x3c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10), "=&s"(t11), "=&s"(t12), "=&s"(t13), "=&s"(t14), "=&s"(t15) : "s"(ka) : "memory");
}
#endif
