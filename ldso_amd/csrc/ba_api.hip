// ba_api.hip — C-ABI (include/ldso_hip.h) of the windowed bundle adjustment: device memory, the
// flattening of the window into slot tables / chunks, kernel sequencing, fetchers.
//
// Kernel sequences:
//   fast path (ldso_ba_optimize, ldso_ba_enqueue_gn):  k_reduce(atomic) -> k_gn_solve -> k_linearize(point step fused)   per iteration
//   multi-GPU fast path:  k_reduce(atomic into the caller's all-reduce buffer) -> k_gn_export -> [all-reduce] -> k_gn_solve -> k_linearize
//   step-wise entry points (solve_system, do_step, ...):  k_reduce -> k_gather -> k_solve(flags) -> k_point_step -> k_linearize
//   marginalisation:  k_linearize<MARG> -> k_reduce -> k_gather -> k_marg_update;  k_marg_frame
#include <hip/hip_runtime.h>
#include <vector>
#include <string>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: ncclAllReduce is resolved at run time (the process may already hold an RCCL, e.g. PyTorch's)
#include "../../include/ldso_hip.h"
#include "ba_dev.h"
#include "pyramid.h"
#include "ba_solve.h"
#include "lie_dev.h"

hipError_t ba_launch_linearize(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, bool hasL, bool fix, int stepMode, const GnInit &gi, hipStream_t st);
hipError_t ba_launch_reduce(const BaPtrs &B, const BaDims &D, const ResSet &S, const ChunkStarts &chunkStart, bool hasL, int GSP, int atomicMode, bool hasPrior, float calibPrior, double l1, double il, int itCheck, hipStream_t st);
hipError_t ba_launch_gather(const BaPtrs &B, const BaDims &D, const ResSet &S, bool hasL, bool hasPrior, int GSP, double lambda,
                            const ldso_settings_t &St, int mode, double *rbuf, hipStream_t st);
hipError_t ba_launch_solve(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, hipStream_t st);
hipError_t ba_launch_point_step(const BaPtrs &B, const BaDims &D, const ResSet &S, int mode, hipStream_t st);
hipError_t ba_launch_linearize_one(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, int stepMode, const GnInit &gi, const LinHead &hd,
                                   hipStream_t st);
hipError_t ba_launch_linearize_marg(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, const int32_t *margFlags, hipStream_t st);
hipError_t ba_launch_marg_frame(const BaPtrs &B, const BaDims &D, int idx, double *work, double *outH, double *outb, hipStream_t st);
hipError_t ba_launch_acc_init(const BaPtrs &B, const BaDims &D, const GnInit &gi, hipStream_t st);
hipError_t ba_launch_gn_export(const BaPtrs &B, const BaDims &D, const ResSet &S, double *tail, hipStream_t st);
hipError_t ba_launch_activate(const BaPtrs &B, const BaDims &D, const ldso_settings_t &S, const ldso_immature_t *d_pts, ldso_activation_t *d_out, int n, int minObs,
                              float minIdepthH_act, int GNIts, hipStream_t st);
hipError_t ba_launch_linearize_batch(const BatchItem *d_items, const BatchBlock *d_blocks, int totalChunks, const int32_t *d_wgStart, int nWG, int FS, int cur, const ldso_settings_t &S, int stepMode, float calibPrior, hipStream_t st, int itCheck = -1);
hipError_t ba_launch_reduce_batch(const BatchItem *d_items, int nWin, int totalBlocks, int cur, float calibPrior, double l1, double il, hipStream_t st);
hipError_t ba_launch_gn_solve_batch(const BatchItem *d_items, int nWin, const BaDims &Dmax, int cur, const ldso_settings_t &St, int iteration, double lambda, hipStream_t st);
hipError_t ba_launch_lm_energies(const BaPtrs &B, const BaDims &D, const ResSet &S, float calibPrior, bool hasPrior, hipStream_t st);
hipError_t ba_launch_marg_update(const BaPtrs &B, const BaDims &D, double w, hipStream_t st);
hipError_t ba_launch_gn_solve(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, hipStream_t st);
hipError_t ba_launch_reduce_solve(const BaPtrs &B, const BaDims &D, const ResSet &S, const ldso_settings_t &St, const SolveArgs &A, const ChunkStarts &chunkStart,
                                  int atomicMode, float calibPrior, double l1, double il, hipStream_t st);

static thread_local std::string g_err;
void ldso_set_error(const std::string &s) { g_err = s; }

#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ldso_set_error(std::string(#call) + ": " + hipGetErrorString(e_)); return LDSO_E_HIP; } } while (0)
#define REQ(cond, msg) do { if (!(cond)) { ldso_set_error(msg); return LDSO_E_INVALID; } } while (0)
#define RUN(x) do { int r_ = (x); if (r_ != LDSO_OK) return r_; } while (0)

struct Timer { hipEvent_t a, b; int which; };

struct ldso_ba {
    int device = 0, w = 0, h = 0, maxF = 0, maxP = 0, FSmax = 0, maxChunks = 0, numCU = 256;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    ldso_settings_t settings;
    BaDims D;
    BaPtrs B;
    ResSet sets[2];
    int cur = 0;
    bool pendingApply = false;
    bool hasL = false;
    bool hasPrior = false;
    int GSP = 0;
    int R = 0;
    float *imgSlots[LD_MAXF] = {nullptr};
    bool imgOwned[LD_MAXF] = {false};
    int32_t *d_chunkStart = nullptr;
    int *d_waitCtr = nullptr;          // k_reduce_solve: producer counter (zero between launches)
    int32_t *d_margFlags = nullptr;
    float *d_color = nullptr;          // irradiance staging of ldso_ba_set_image_raw
    void *d_act = nullptr;             // staging of ldso_ba_activate_points: n immature records + n results
    int actCap = 0;
    double *ownAcc = nullptr;          // the handle's own HFinal/bFinal accumulator (B.acc may point at a caller's all-reduce buffer)
    ChunkStarts chunkStarts;
    ldso_rawjac_t *d_dumpJ = nullptr;
    std::vector<int32_t> flat2slot;
    std::vector<int32_t> imageSlot;
    std::vector<void *> allocs;
    // host staging of the window (for shard rebuilds)
    std::vector<int32_t> h_phost;
    // window upload: ONE pinned staging arena -> ONE device arena -> one scatter kernel (k_win_scatter) instead of ~25 copies + ~20 fills
    // the window's descriptors in device memory (one BatchItem): the plain linearisation of the GN iteration reads them from there - passed
    // as kernel arguments, the ~150 pointers outgrow the scalar registers (400 SGPR spill moves in the kernel)
    BatchItem *d_item = nullptr, *h_item = nullptr;
    BatchBlock *d_blocks = nullptr;    // [maxChunks] the window's chunks as k_linearize_batch reads them (window index 0)
    std::vector<BatchBlock> h_blocks;
    bool appliedValid = false;         // the applied residual set holds a linearisation of the resident window (its per-chunk partials feed the next reduce)
    LinHead linHead;                   // chunk geometry of the current window by value (k_linearize_one)
    int reduceSplits = LD_SCT_KS;      // K-splits per 16 x 16 Schur tile (BaDims::ks; ldso_ba_set_reduce_splits)
    bool linHeadOk = false;            // the chunks are regular (every host cut into CH-point pieces): true for everything build_chunks produces
    const void *inBatch = nullptr;     // the ldso_ba_batch this handle belongs to (at most one; it must outlive the batch: ldso_ba_destroy refuses while set)
    int chunkPoints = 0;               // points per workgroup of k_linearize: 0 = as few as keep the grid within one wave of workgroups (one window alone on the chip)
    std::vector<int32_t> chunkCuts;    // explicit chunk ends (ldso_ba_set_chunk_cuts / ldso_ba_batch_create: uneven chunks, one workload per workgroup); empty: regular chunks of chunkPoints
    BatchItem itemShadow;
    bool itemValid = false;
    int *h_stop = nullptr, *d_stop = nullptr;      // host-mapped word (and its device address): which iteration ended an un-forced optimize() loop
    char *h_down = nullptr;             // pinned arena of the fetch functions (ldso_ba_get_residuals / _points / _frames): device -> pinned host at link speed, one wait
    size_t downCap = 0;
    bool stageBusy = false;            // an asynchronous copy out of h_stage may still be in flight (ldso_ba_set_prior): the next user of the arena waits first
    // an edit of the resident window being recorded (ldso_ba_window_begin .. ldso_ba_window_commit): frames and residual targets are named by their index in the
    // RESIDENT window (inserted frames: oF, oF + 1, ...), points by their resident row
    struct NewPoint { ldso_point_t p; int before; std::vector<ldso_residual_t> res; float mrb; int32_t ngr; };
    struct WindowEdit {
        bool active = false;
        int oF = 0, oP = 0;
        std::vector<char> frameGone, rowGone;
        std::vector<int32_t> insertedSlots;
        std::vector<uint32_t> mask;          // per resident row, bit = edit-time frame id
        std::vector<NewPoint> fresh;
    } edit;
    char *h_stage = nullptr, *d_stage = nullptr;
    size_t stageCap = 0;
    // profiling
    bool profile = false;
    std::vector<Timer> timers;
    double tsum[5] = {0, 0, 0, 0, 0};
    int tcnt[5] = {0, 0, 0, 0, 0};
    int lastIterations = 0;
    bool noFusedLaunch = false;        // debug: k_reduce and k_gn_solve as two launches even where the fused k_reduce_solve applies
    // ldso_ba_enqueue_gn replays a cached HIP graph when the same launch sequence was enqueued before: the key is EVERYTHING the launches take as
    // arguments (pointer tables, dimensions, both residual sets, settings, chunk geometry, flags, stream, first iteration, count, parity of the sets),
    // byte for byte (round 6: the 64-bit hash of those bytes only pre-selects - a collision must not replay another window's launches)
    struct GnGraph { unsigned long long sig; std::vector<unsigned char> key; hipGraphExec_t exec; hipGraph_t graph; };
    std::vector<GnGraph> gnGraphs;
    std::vector<unsigned char> gnKeyScratch;      // the key of the current call (kept to avoid an allocation per enqueue)
    bool gnUseGraphs = true;
    double *distBuf = nullptr;         // ldso_ba_enqueue_gn_rccl / _p2p: all-reduce buffer [HFinal | bFinal | scalars | candidates]
    unsigned p2pSeq = 0;               // ldso_ba_enqueue_gn_p2p: exchanges done (the tag of the hand-over words)
    int *d_p2pErr = nullptr;           // set by k_p2p_sum when a peer's words did not arrive in time
    double neverStop = 1e300;          // source of the LD_SC_STOP reset (outlives the asynchronous copy)
};

extern "C" {

int ldso_version(void) { return 100; }
const char *ldso_last_error(void) { return g_err.c_str(); }
int ldso_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int ldso_settings_default(ldso_settings_t *s) {
    if (!s) return LDSO_E_INVALID;
    memset(s, 0, sizeof(*s));
    s->huberTH = 9; s->outlierTHSumComponent = 50 * 50; s->affineOptModeA = 1e12f; s->affineOptModeB = 1e8f;
    s->frameEnergyTHN = 0.7f; s->frameEnergyTHFacMedian = 1.5f; s->frameEnergyTHConstWeight = 0.5f; s->overallEnergyTHWeight = 1;
    s->initialCalibHessian = 5e9f; s->margWeightFac = 0.5f * 0.5f; s->idepthFixPriorMargFac = 600 * 600; s->thOptIterations = 1.2f;
    s->coarseCutoffTH = 20; s->minOptIterations = 1; s->solverMode = LDSO_SOLVER_FIX_LAMBDA | LDSO_SOLVER_ORTHOGONALIZE_X_LATER;
    s->forceAcceptStep = 1; s->solverModeDelta = 0.00001;
    return LDSO_OK;
}

int ldso_pyr_levels_used(int w, int h) {
    int wl = w, hl = h, lv = 1;
    while (wl % 2 == 0 && hl % 2 == 0 && wl * hl > 5000 && lv < LDSO_PYR_LEVELS) { wl /= 2; hl /= 2; lv++; }
    return lv;
}

int ldso_frame_set_evalPT(ldso_frame_t *f, const double worldToCam[12], const double state[10]) {
    if (!f || !worldToCam || !state) return LDSO_E_INVALID;
    memcpy(f->worldToCam_evalPT, worldToCam, 12 * sizeof(double));
    memcpy(f->state, state, 10 * sizeof(double));
    memcpy(f->state_zero, state, 10 * sizeof(double));
    ld::frame_nullspaces(f->worldToCam_evalPT, f->state_zero[6], f->ab_exposure, f->nullspaces_pose, f->nullspaces_scale, f->nullspaces_affine);
    return LDSO_OK;
}

int ldso_frame_set_prior(ldso_frame_t *f, const ldso_settings_t *s) {
    if (!f || !s) return LDSO_E_INVALID;
    for (int i = 0; i < 8; i++) f->prior[i] = 0;
    // Setting.cc:18-21: `float` globals in the reference - the double prior vector receives the float values (1e11f = 99999997952, 1e14f = 100000000376832)
    const double initialRotPrior = (double) 1e11f, initialTransPrior = (double) 1e10f, initialAffBPrior = (double) 1e14f, initialAffAPrior = (double) 1e14f;
    if (f->frameID == 0) {
        for (int i = 0; i < 3; i++) { f->prior[i] = initialTransPrior; f->prior[3 + i] = initialRotPrior; }
        if (s->solverMode & LDSO_SOLVER_REMOVE_POSEPRIOR) for (int i = 0; i < 6; i++) f->prior[i] = 0;
        f->prior[6] = initialAffAPrior; f->prior[7] = initialAffBPrior;
    } else {
        f->prior[6] = (s->affineOptModeA < 0) ? initialAffAPrior : (double) s->affineOptModeA;
        f->prior[7] = (s->affineOptModeB < 0) ? initialAffBPrior : (double) s->affineOptModeB;
    }
    return LDSO_OK;
}

// ---------------------------------------------------------------------------------------------------------
}  // extern "C"
template <class T> static int dalloc(ldso_ba *H, T **p, size_t n) {
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
#define DA(ptr, n) do { int r_ = dalloc(H, &(ptr), (n)); if (r_ != LDSO_OK) return r_; } while (0)
// One entry of the upload table at the head of the staging arena: copy `words` 32-bit words from arena offset `src` (bytes) to `dst`,
// or fill `dst` with zeros (src == LD_XFER_ZERO).
struct WinXfer { void *dst; unsigned long long src; unsigned long long words; };
#define LD_XFER_ZERO 0xFFFFFFFFFFFFFFFFull
#define LD_XFER_MAX 96
__global__ __launch_bounds__(256) void k_win_scatter(const char *__restrict__ arena, int nEntries) {
    const WinXfer *tab = reinterpret_cast<const WinXfer *>(arena);
    for (int e = 0; e < nEntries; e++) {
        const WinXfer x = tab[e];
        unsigned *dst = static_cast<unsigned *>(x.dst);
        const size_t n = (size_t) x.words, stride = (size_t) gridDim.x * blockDim.x;
        if (x.src == LD_XFER_ZERO) { for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = 0u; }
        else {
            const unsigned *src = reinterpret_cast<const unsigned *>(arena + x.src);
            for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
        }
    }
}
__global__ __launch_bounds__(256) void k_point_stats(PtRec *__restrict__ a, PtRec *__restrict__ b, const float *__restrict__ mrb, const int32_t *__restrict__ ngr, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float m = mrb[i]; const int32_t g = ngr[i];
    a[i].maxRelBS = m; a[i].numGood = g; b[i].maxRelBS = m; b[i].numGood = g;
}
// ---------------------------------------------------------------------------------------------------------------------------------------------------
// ldso_ba_update_window: the staging image a full ldso_ba_set_window of the SAME window would have uploaded, built ON THE DEVICE from the resident window
// (the applied set) and a small delta - which frames stay (EnergyFunctional::marginalizeFrame / insertFrame), which points stay and in which order (removePoint,
// dropPointsF, makeIDX), which residuals a point has (insertResidual / dropResidual as one bit per target frame) and the records of the fresh points.  One thread
// per (point, slot).  What is carried over is exactly what the host objects carry between two optimize() calls: u, v, priorF, colour / weights, the inverse depth,
// maxRelBaseline / numGoodResiduals, and per residual state_state, state_energy, isActive, isNew - everything else starts as ldso_ba_set_window starts it, so
// that the resident window and a fresh upload of the same objects are the same bytes (tests/test_resident_gpu.py).
// ---------------------------------------------------------------------------------------------------------------------------------------------------
struct WinDelta {
    // the resident window (read)
    const PtGeo *oGeo; const PtCw *oPcw; const int32_t *oHost; const SlotTab *oTab; const SlotRec *oSlot; const PtRec *oPt;
    int oF, oFS, oP;
    // the delta (device copies inside the staging arena)
    const int32_t *frameFrom;      // [F]  old index of new frame f, -1 = inserted
    const int32_t *pointFrom;      // [P]  old row of new point i, -1 - k = the k-th fresh point
    const uint32_t *resMask;       // [P]  bit t: the point has a residual whose target is frame t
    const int32_t *resBegin;       // [P + 1] flat index of the point's first residual (flat order: point-major, target-ascending)
    const ldso_point_t *fresh; const ldso_residual_t *freshRes; const int32_t *freshResBegin;      // the fresh points, their residuals (target-ascending), first residual of fresh point k
    const float *freshMrb; const int32_t *freshNgr;
    int F, FS, P;
    // the image (written)
    PtGeo *geo; PtCw *pcw; int32_t *phost; SlotTab *tab; SlotRec *sr; float *mrb; int32_t *ngr;
};
__global__ __launch_bounds__(256) void k_win_rebuild(WinDelta W) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= W.P * W.FS) return;
    const int row = q / W.FS, col = q - row * W.FS;
    const int from = W.pointFrom[row];
    const uint32_t mask = W.resMask[row];
    SlotTab t{-1, 0, 0, -1};
    SlotRec r;
#pragma unroll
    for (int k = 0; k < 8; k++) { r.e[k].jp = 0.0f; r.e[k].m.i = 0; }
    r.e[LD_SM_STATE].m.i = LDSO_RES_OOB;
    if (col < W.F && ((mask >> col) & 1u)) {
        const int below = __popc(mask & ((1u << col) - 1u));
        t.rflat = W.resBegin[row] + below;
        const int oc = (from >= 0) ? W.frameFrom[col] : -1;
        if (from >= 0) {
            const size_t os = (size_t) from * W.oFS + (oc >= 0 ? oc : 0);
            if (oc >= 0 && W.oTab[os].rflat >= 0) {          // the residual was there: its state lives on
                const SlotRec o = W.oSlot[os];
                t.rnew = W.oTab[os].rnew;
                r.e[LD_SM_STATE].m.i = o.e[LD_SM_STATE].m.i; r.e[LD_SM_ACTIVE].m.i = o.e[LD_SM_ACTIVE].m.i; r.e[LD_SM_ENERGY].m.f = o.e[LD_SM_ENERGY].m.f;
            } else {                                          // insertResidual for a point of the window (FullSystem.cc:447-470: state IN, energy 0, not active yet)
                t.rnew = 1;
                r.e[LD_SM_STATE].m.i = LDSO_RES_IN; r.e[LD_SM_ACTIVE].m.i = 0; r.e[LD_SM_ENERGY].m.f = 0.0f;
            }
        } else {
            const ldso_residual_t fr = W.freshRes[W.freshResBegin[-1 - from] + below];
            t.rnew = fr.is_new ? 1 : 0;
            r.e[LD_SM_STATE].m.i = fr.state_state; r.e[LD_SM_ACTIVE].m.i = fr.is_active ? 1 : 0; r.e[LD_SM_ENERGY].m.f = fr.state_energy;
        }
    }
    W.tab[q] = t; W.sr[q] = r;
    if (col < 8) W.pcw[(size_t) row * 8 + col] = (from >= 0) ? W.oPcw[(size_t) from * 8 + col] : PtCw{W.fresh[-1 - from].color[col], W.fresh[-1 - from].weights[col]};
    if (col == 0) {
        PtGeo g;
        memset(&g, 0, sizeof(g));
        int host;
        if (from >= 0) {
            const PtGeo o = W.oGeo[from];
            g.u = o.u; g.v = o.v; g.priorF = o.priorF; g.idepth = o.idepth; g.idepth_zero = o.idepth; g.idepth_backup = o.idepth;          // setIdepthZero(idepth) after every optimize()
            const int oh = W.oHost[from];
            host = -1;
            for (int f = 0; f < W.F; f++) host = (W.frameFrom[f] == oh) ? f : host;
            W.mrb[row] = W.oPt[from].maxRelBS; W.ngr[row] = W.oPt[from].numGood;
        } else {
            const ldso_point_t &p = W.fresh[-1 - from];
            g.u = p.u; g.v = p.v; g.priorF = p.priorF; g.idepth = p.idepth; g.idepth_zero = p.idepth_zero; g.idepth_backup = p.idepth;
            host = p.host;
            W.mrb[row] = W.freshMrb[-1 - from]; W.ngr[row] = W.freshNgr[-1 - from];
        }
        W.geo[row] = g; W.phost[row] = host;
    }
}

// host side of the arena: reserve (16-byte aligned) room, remember where it goes
struct WinStage {
    char *base; size_t cap, used; WinXfer *tab; int n;
    template <class T> T *put(T *dst, size_t count) {          // room for `count` elements that will land at dst; returns where to write them
        const size_t bytes = (count * sizeof(T) + 15) & ~(size_t) 15;
        if (n >= LD_XFER_MAX || used + bytes > cap) return nullptr;
        T *p = reinterpret_cast<T *>(base + used);
        if (count) { tab[n].dst = dst; tab[n].src = used; tab[n].words = count * sizeof(T) / 4; n++; }
        used += bytes;
        return p;
    }
    template <class T> T *raw(size_t count) {          // room without a destination (operands of k_win_rebuild)
        const size_t bytes = (count * sizeof(T) + 15) & ~(size_t) 15;
        if (used + bytes > cap) return nullptr;
        T *p = reinterpret_cast<T *>(base + used);
        used += bytes;
        return p;
    }
    template <class T> bool again(T *dst, const T *staged, size_t count) {      // the same staged data to a second destination
        if (n >= LD_XFER_MAX) return false;
        if (count) { tab[n].dst = dst; tab[n].src = (size_t) (reinterpret_cast<const char *>(staged) - base); tab[n].words = count * sizeof(T) / 4; n++; }
        return true;
    }
    template <class T> bool zero(T *dst, size_t count) {
        if (n >= LD_XFER_MAX) return false;
        if (count) { tab[n].dst = dst; tab[n].src = LD_XFER_ZERO; tab[n].words = count * sizeof(T) / 4; n++; }
        return true;
    }
};

template <class T> static int h2d(ldso_ba *H, T *dst, const std::vector<T> &src) {
    if (src.empty()) return LDSO_OK;
    CHK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, H->stream));
    return LDSO_OK;
}
// the fetch functions read the device tables through ONE pinned arena: reserve what the call needs (growth invalidates nothing: it happens before
// the first copy is enqueued), enqueue the copies back to back, wait once, read in place.  (std::vector destinations are pageable memory: the
// runtime stages them in small pieces - ldso_ba_get_residuals of a C3 window took 0.35 ms for 1 MB.)
static int down_reserve(ldso_ba *H, size_t bytes) {
    if (bytes <= H->downCap) return LDSO_OK;
    CHK(hipStreamSynchronize(H->stream));
    if (H->h_down) (void) hipHostFree(H->h_down);
    H->h_down = nullptr; H->downCap = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    CHK(hipHostMalloc((void **) &H->h_down, cap));
    H->downCap = cap;
    return LDSO_OK;
}
template <class T> static const T *down_put(ldso_ba *H, size_t &used, const T *src, size_t n, int &rc) {
    const size_t bytes = (n * sizeof(T) + 63) & ~(size_t) 63;
    const T *p = reinterpret_cast<const T *>(H->h_down + used);
    if (rc == LDSO_OK && n) { if (hipMemcpyAsync(H->h_down + used, src, n * sizeof(T), hipMemcpyDeviceToHost, H->stream) != hipSuccess) { ldso_set_error("hipMemcpyAsync (fetch) failed"); rc = LDSO_E_HIP; } }
    used += bytes;
    return p;
}

template <class T> static int d2h(ldso_ba *H, std::vector<T> &dst, const T *src, size_t n) {
    dst.resize(n);
    if (n == 0) return LDSO_OK;
    CHK(hipMemcpyAsync(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, H->stream));
    return LDSO_OK;
}
extern "C" {

static int alloc_set(ldso_ba *H, ResSet &S) {
    const size_t P = H->maxP, FS = H->FSmax, PS = P * FS, C = H->maxChunks;
    DA(S.slot, PS); DA(S.pt, P); DA(S.acc, P); DA(S.candE, P);
    DA(S.G, P * (8 * FS + LD_GEXTRA)); DA(S.topA, C * FS * LD_TOPN); DA(S.topL, C * FS * LD_TOPN);
    DA(S.chunkEnergy, C); DA(S.chunkCnt, C * 2); DA(S.chunkNID, C * 2);
    return LDSO_OK;
}

static int create_body(ldso_ba *H, int device, int w, int h, int max_frames, int max_points);

int ldso_ba_create(int device, int w, int h, int max_frames, int max_points, ldso_ba_t **out) {
    REQ(out && w > 8 && h > 8 && max_frames >= 2 && max_frames <= LD_MAXF && max_points > 0, "ldso_ba_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { ldso_set_error("no HIP device visible"); return LDSO_E_NODEVICE; }
    REQ(device >= 0 && device < ndev, "ldso_ba_create: device index out of range");
    CHK(hipSetDevice(device));
    ldso_ba *H = new ldso_ba();
    { const char *e = getenv("LDSO_GN_GRAPHS"); if (e && e[0] == '0') H->gnUseGraphs = false; }
    const int r = create_body(H, device, w, h, max_frames, max_points);
    if (r != LDSO_OK) { const std::string keep = g_err; ldso_ba_destroy(H); g_err = keep; return r; }      // nothing of a half-built handle leaks
    *out = H;
    return LDSO_OK;
}

static int create_body(ldso_ba *H, int device, int w, int h, int max_frames, int max_points) {
    H->device = device; H->w = w; H->h = h; H->maxF = max_frames; H->maxP = max_points;
    H->FSmax = (max_frames + 7) / 8 * 8;
    H->maxChunks = max_points / 4 + max_frames + 4;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) H->numCU = pr.multiProcessorCount; }
    ldso_settings_default(&H->settings);
    CHK(hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking));
    H->ownStream = true;
    CHK(hipHostMalloc((void **) &H->h_item, sizeof(BatchItem)));
    { void *q_ = nullptr; CHK(hipMalloc(&q_, sizeof(BatchItem))); H->d_item = (BatchItem *) q_; }
    { void *q_ = nullptr; CHK(hipMalloc(&q_, (size_t) H->maxChunks * sizeof(BatchBlock))); H->d_blocks = (BatchBlock *) q_; }
    CHK(hipHostMalloc((void **) &H->h_stop, 4 * sizeof(int), hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void **) &H->d_stop, H->h_stop, 0));
    memset(&H->B, 0, sizeof(H->B));
    memset(&H->D, 0, sizeof(H->D));
    if (const char *e = getenv("LDSO_REDUCE_KS")) { const int v = atoi(e); if (v >= 1 && v <= 16) H->reduceSplits = v; }          // tuning knob: ldso_ba_set_reduce_splits for every handle of the process
    BaPtrs &B = H->B;
    const size_t F = max_frames, P = max_points, FS = H->FSmax, nmax = 8 * F + 4;
    DA(B.frames, F); DA(B.calib, 1); DA(B.pairs, F * F); DA(B.pairRt, F * F * 12);
    DA(B.adHost, F * F * 64); DA(B.adTarget, F * F * 64); DA(B.adHostF, F * F * 64); DA(B.adTargetF, F * F * 64);
    DA(B.nsProj, nmax * 7); DA(B.HM, nmax * nmax); DA(B.bM, nmax);
    DA(B.pgeo, P); DA(B.pcw, P * 8); DA(B.phost, P);
    DA(B.rtab, P * FS);
    DA(B.Jlin, P * FS); DA(B.rtz, P * FS * 8);
    DA(B.chunk_p0, H->maxChunks); DA(B.chunk_n, H->maxChunks); DA(B.chunk_host, H->maxChunks);
    DA(H->d_chunkStart, F + 2);
    DA(H->d_waitCtr, 4);
    DA(H->d_margFlags, P);
    DA(B.pairC, 2 * F * F * LD_PAIRC);
    const size_t GSPmax = (8 * FS + LD_GEXTRA + 15) / 16 * 16;
    DA(B.scPart, (size_t) LD_SC_SPLITS * GSPmax * GSPmax);
    DA(B.sys, 4 * (nmax * nmax + nmax));
    DA(B.acc, nmax * nmax + nmax); H->ownAcc = B.acc;
    DA(B.x, nmax); DA(B.xAd, F * F * 8); DA(B.xc, 4); DA(B.scalars, 16); DA(B.energyLog, 64);
    DA(H->d_dumpJ, P * FS);
    B.dumpJ = nullptr;
    int r;
    if ((r = alloc_set(H, H->sets[0])) != LDSO_OK) return r;
    if ((r = alloc_set(H, H->sets[1])) != LDSO_OK) return r;
    return LDSO_OK;
}

int ldso_ba_destroy(ldso_ba_t *H) {
    if (!H) return LDSO_OK;
    REQ(H->inBatch == nullptr, "ldso_ba_destroy: the handle is part of a live batch (ldso_ba_batch_destroy first: the batch dereferences its handles)");
    hipSetDevice(H->device);
    hipDeviceSynchronize();
    for (void *p : H->allocs) hipFree(p);
    for (int i = 0; i < LD_MAXF; i++) if (H->imgOwned[i] && H->imgSlots[i]) hipFree(H->imgSlots[i]);
    if (H->d_color) hipFree(H->d_color);
    if (H->h_item) hipHostFree(H->h_item);
    if (H->d_item) hipFree(H->d_item);
    if (H->d_blocks) hipFree(H->d_blocks);
    if (H->h_stop) hipHostFree(H->h_stop);
    if (H->h_down) hipHostFree(H->h_down);
    if (H->h_stage) hipHostFree(H->h_stage);
    if (H->d_stage) hipFree(H->d_stage);
    if (H->d_act) hipFree(H->d_act);
    if (H->distBuf) hipFree(H->distBuf);
    if (H->d_p2pErr) hipFree(H->d_p2pErr);
    for (auto &t : H->timers) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    for (ldso_ba::GnGraph &g : H->gnGraphs) { hipGraphExecDestroy(g.exec); hipGraphDestroy(g.graph); }
    if (H->ownStream && H->stream) hipStreamDestroy(H->stream);
    delete H;
    return LDSO_OK;
}

int ldso_ba_set_stream(ldso_ba_t *H, void *s) {
    REQ(H, "null handle");
    CHK(hipSetDevice(H->device));
    // work of this handle that is still in flight on the OLD stream is not ordered against the new one: wait for it here - in particular a staged copy out of the
    // pinned arena (stageBusy, ldso_ba_set_prior), whose later "wait for H->stream" would otherwise wait on the wrong stream
    if (H->stream && (H->ownStream || H->stageBusy)) { CHK(hipStreamSynchronize(H->stream)); H->stageBusy = false; }
    if (H->ownStream && H->stream && s) { hipStreamDestroy(H->stream); H->ownStream = false; }
    if (s) { H->stream = (hipStream_t) s; H->ownStream = false; }
    else if (!H->ownStream) { CHK(hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking)); H->ownStream = true; }
    return LDSO_OK;
}

int ldso_ba_set_settings(ldso_ba_t *H, const ldso_settings_t *s) {
    REQ(H && s, "null argument");
    if (s->solverMode & (LDSO_SOLVER_SVD | LDSO_SOLVER_ORTHOGONALIZE_SYSTEM | LDSO_SOLVER_ORTHOGONALIZE_POINTMARG | LDSO_SOLVER_ORTHOGONALIZE_FULL |
                         LDSO_SOLVER_MOMENTUM | LDSO_SOLVER_STEPMOMENTUM)) {
        ldso_set_error("solverMode bits SVD/ORTHOGONALIZE_SYSTEM/_POINTMARG/_FULL/MOMENTUM/STEPMOMENTUM are not provided by the device path");
        return LDSO_E_UNSUPPORTED;
    }
    H->settings = *s;
    return LDSO_OK;
}

// a slot changed its buffer: the resident window keeps reading the slot, so its image pointers follow
static void rebind_slot(ldso_ba *H, int slot) {
    for (int f = 0; f < H->D.F && f < (int) H->imageSlot.size(); f++) if (H->imageSlot[f] == slot) H->B.img[f] = H->imgSlots[slot];
}

int ldso_ba_set_image(ldso_ba_t *H, int slot, const float *src) {
    REQ(H && src && slot >= 0 && slot < H->maxF, "ldso_ba_set_image: bad arguments");
    CHK(hipSetDevice(H->device));
    size_t bytes = (size_t) H->w * H->h * 3 * sizeof(float);
    if (!H->imgOwned[slot]) { void *p; CHK(hipMalloc(&p, bytes)); H->imgSlots[slot] = (float *) p; H->imgOwned[slot] = true; rebind_slot(H, slot); }
    CHK(hipMemcpyAsync(H->imgSlots[slot], src, bytes, hipMemcpyHostToDevice, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

hipError_t img_launch_make_images(const float *d_color, int w, int h, int levels, float *const *d_levels, hipStream_t st);

// level-0 image of a key frame from its raw irradiance: FrameHessian::makeImages level 0 (FrameHessian.cc:44-113) on the device
int ldso_ba_set_image_raw(ldso_ba_t *H, int slot, const float *irradiance) {
    REQ(H && irradiance && slot >= 0 && slot < H->maxF, "ldso_ba_set_image_raw: bad arguments");
    CHK(hipSetDevice(H->device));
    const size_t n = (size_t) H->w * H->h;
    if (!H->imgOwned[slot]) { void *q; CHK(hipMalloc(&q, n * 12)); H->imgSlots[slot] = (float *) q; H->imgOwned[slot] = true; rebind_slot(H, slot); }
    if (!H->d_color) CHK(hipMalloc(&H->d_color, n * sizeof(float)));
    CHK(hipMemcpyAsync(H->d_color, irradiance, n * sizeof(float), hipMemcpyHostToDevice, H->stream));
    float *lv[1] = {H->imgSlots[slot]};
    CHK(img_launch_make_images(H->d_color, H->w, H->h, 1, lv, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

// debug / test fetch of a key-frame image slot (w*h*3 floats)
int ldso_ba_get_image(ldso_ba_t *H, int slot, float *out) {
    REQ(H && out && slot >= 0 && slot < H->maxF && H->imgSlots[slot], "ldso_ba_get_image: bad arguments");
    CHK(hipSetDevice(H->device));
    CHK(hipMemcpyAsync(out, H->imgSlots[slot], (size_t) H->w * H->h * 12, hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

int ldso_ba_set_image_device(ldso_ba_t *H, int slot, const void *dev) {
    REQ(H && dev && slot >= 0 && slot < H->maxF, "ldso_ba_set_image_device: bad arguments");
    CHK(hipSetDevice(H->device));
    float *old = H->imgSlots[slot];
    if (H->imgOwned[slot] && old) {
        CHK(hipStreamSynchronize(H->stream));      // a kernel of the resident window may still be reading the slot
        hipFree(old); H->imgOwned[slot] = false;
    }
    H->imgSlots[slot] = (float *) dev;
    // the resident window keeps reading this slot: re-resolve its image pointers (no dangling B.img)
    for (int f = 0; f < H->D.F && f < (int) H->imageSlot.size(); f++) if (H->imageSlot[f] == slot) H->B.img[f] = H->imgSlots[slot];
    return LDSO_OK;
}

// image slot from a resident ldso_pyramid_t (level 0 = FrameHessian::dI): zero-copy, ordered after the pyramid's build on this stream
int ldso_ba_set_image_pyramid(ldso_ba_t *H, int slot, ldso_pyramid_t *pyr) {
    REQ(H && pyr && slot >= 0 && slot < H->maxF, "ldso_ba_set_image_pyramid: bad arguments");
    REQ(pyr->built && pyr->device == H->device && pyr->w == H->w && pyr->h == H->h, "ldso_ba_set_image_pyramid: pyramid does not match the handle (device, size) or holds no image");
    CHK(hipSetDevice(H->device));
    CHK(hipStreamWaitEvent(H->stream, pyr->ready, 0));
    return ldso_ba_set_image_device(H, slot, pyr->lv[0]);
}

#define H2D(dst, vec) do { int r_ = h2d(H, (dst), (vec)); if (r_ != LDSO_OK) return r_; } while (0)

static int build_chunks(ldso_ba *H) {
    // host-major chunks over the local shard [pBegin,pEnd)
    BaDims &D = H->D;
    // smallest multiple of 4 points per chunk that keeps the grid within one wave of workgroups (one per CU)
    int CH = 4;
    if (!H->chunkCuts.empty() && H->chunkCuts.back() != D.pEnd) H->chunkCuts.clear();          // cuts made for another window: back to the regular policy
    const std::vector<int32_t> &cuts = H->chunkCuts;          // explicit ends (ascending): a chunk also ends at every host boundary
    if (!cuts.empty()) CH = 1 << 30;
    else if (H->chunkPoints > 0) CH = H->chunkPoints;      // ldso_ba_set_chunk_points / ldso_ba_batch_create: many windows share a launch, fewer and fatter workgroups
    else for (;; CH += 4) {
        int cnt = 0, run = 0, prev = -1;
        for (int q = D.pBegin; q < D.pEnd; q++) { int hq = H->h_phost[q]; if (hq != prev) { cnt += (run + CH - 1) / CH; run = 0; prev = hq; } run++; }
        cnt += (run + CH - 1) / CH;
        if (cnt <= H->numCU || CH >= 1024) break;
    }
    std::vector<int32_t> p0, cn, ch, cs(D.F + 1, 0);
    int p = D.pBegin;
    size_t ci = 0;
    for (int hst = 0; hst < D.F; hst++) {
        cs[hst] = (int) p0.size();
        while (p < D.pEnd && H->h_phost[p] == hst) {
            int e = p;
            while (ci < cuts.size() && cuts[ci] <= p) ci++;
            const int stop = ci < cuts.size() ? cuts[ci] : D.pEnd;
            while (e < D.pEnd && H->h_phost[e] == hst && e - p < CH && e < stop) e++;
            p0.push_back(p); cn.push_back(e - p); ch.push_back(hst);
            p = e;
        }
    }
    cs[D.F] = (int) p0.size();
    REQ(p == D.pEnd, "ldso_ba_set_window: points must be ordered by host frame (EnergyFunctional::allPoints order)");
    REQ((int) p0.size() <= H->maxChunks, "too many chunks");
    D.nChunks = (int) p0.size();
    H2D(H->B.chunk_p0, p0); H2D(H->B.chunk_n, cn); H2D(H->B.chunk_host, ch); H2D(H->d_chunkStart, cs);
    H->h_blocks.resize(p0.size());
    for (size_t i = 0; i < p0.size(); i++) H->h_blocks[i] = BatchBlock{0, p0[i], cn[i], ch[i] | ((int32_t) i << 8)};
    CHK(hipMemcpyAsync(H->d_blocks, H->h_blocks.data(), p0.size() * sizeof(BatchBlock), hipMemcpyHostToDevice, H->stream));
    for (int i = 0; i <= LD_MAXF; i++) H->chunkStarts.v[i] = (i <= D.F) ? cs[i] : cs[D.F];
    {
        LinHead &L = H->linHead;
        memset(&L, 0, sizeof(L));
        L.CH = CH; L.F = D.F;
        int q = D.pBegin;
        for (int hst = 0; hst <= LD_MAXF; hst++) {
            L.cs[hst] = (hst <= D.F) ? cs[hst] : cs[D.F];
            L.hostP0[hst] = q;
            while (hst < D.F && q < D.pEnd && H->h_phost[q] == hst) q++;
        }
        // the closed form must reproduce the table (it does for every chunking build_chunks makes; checked, not assumed)
        bool ok = true;
        for (size_t i = 0; i < p0.size() && ok; i++) {
            const int hst = ch[i];
            ok = p0[i] == L.hostP0[hst] + ((int) i - L.cs[hst]) * CH && cn[i] == std::min(CH, L.hostP0[hst + 1] - p0[i]);
        }
        H->linHeadOk = ok;
    }
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

int ldso_ba_set_window(ldso_ba_t *H, int F, const int32_t *image_slot, int P, const ldso_point_t *pts, int R, const ldso_residual_t *res,
                       const ldso_rawjac_t *linJ, const float *lin_rtz) {
    REQ(H && image_slot && pts && res, "ldso_ba_set_window: null argument");
    REQ(F >= 2 && F <= H->maxF && P >= 1 && P <= H->maxP && R >= 0, "ldso_ba_set_window: window exceeds the handle's capacity");
    CHK(hipSetDevice(H->device));
    BaDims &D = H->D;
    BaPtrs &B = H->B;
    // a call that fails half way (bad indices, allocation) must not leave the dimensions of the NEW window over the data of the old one:
    // the handle then holds no window (every entry point that needs one says so)
    struct WinGuard { ldso_ba *H; bool ok; ~WinGuard() { if (!ok) { H->D.P = 0; H->D.R = 0; H->R = 0; H->appliedValid = false; H->itemValid = false; } } } guard{H, false};
    D.F = F; D.FS = (F + 7) / 8 * 8; D.P = P; D.R = R; D.n = 8 * F + 4; D.GS = 8 * D.FS + LD_GEXTRA; D.w = H->w; D.h = H->h; D.nsg = D.FS / 8; D.ks = H->reduceSplits;
    D.pBegin = 0; D.pEnd = P; D.wM3G = (float) (H->w - 3); D.hM3G = (float) (H->h - 3);
    H->GSP = (D.GS + 15) / 16 * 16;
    H->R = R;
    H->imageSlot.assign(image_slot, image_slot + F);
    for (int f = 0; f < F; f++) {
        REQ(image_slot[f] >= 0 && image_slot[f] < H->maxF && H->imgSlots[image_slot[f]] != nullptr, "ldso_ba_set_window: image slot not set");
        B.img[f] = H->imgSlots[image_slot[f]];
    }
    const int FS = D.FS;
    const size_t PS = (size_t) P * FS;
    // ---- staging arena: everything the window needs goes over PCIe in ONE copy and is distributed (or zero-filled) by ONE kernel ----
    size_t nLin = 0;
    for (int i = 0; i < R; i++) nLin += res[i].is_linearized ? 1 : 0;
    REQ(nLin == 0 || (linJ && lin_rtz), "ldso_ba_set_window: linearised residual without linJ / lin_res_toZeroF");
    auto A16 = [](size_t b) { return (b + 15) & ~(size_t) 15; };
    const size_t tabBytes = A16(LD_XFER_MAX * sizeof(WinXfer));
    const size_t need = tabBytes + A16((size_t) P * sizeof(PtGeo)) + A16((size_t) P * 8 * sizeof(PtCw)) + A16((size_t) P * 4) + A16(PS * sizeof(SlotTab)) + A16(nLin * sizeof(ldso_rawjac_t)) + A16(nLin * 32)
                      + A16(PS * sizeof(SlotRec)) + 64;
    if (need > H->stageCap) {
        CHK(hipStreamSynchronize(H->stream));
        if (H->h_stage) hipHostFree(H->h_stage);
        if (H->d_stage) hipFree(H->d_stage);
        H->h_stage = nullptr; H->d_stage = nullptr; H->stageCap = 0;
        const size_t cap = need + need / 4;
        CHK(hipHostMalloc((void **) &H->h_stage, cap));
        CHK(hipMalloc((void **) &H->d_stage, cap));
        H->stageCap = cap;
    }
    if (H->stageBusy) { CHK(hipStreamSynchronize(H->stream)); H->stageBusy = false; }
    WinStage W{H->h_stage, H->stageCap, tabBytes, reinterpret_cast<WinXfer *>(H->h_stage), 0};
    PtGeo *geo = W.put(B.pgeo, P);
    PtCw *pcw = W.put(B.pcw, (size_t) P * 8);
    int32_t *phost = W.put(B.phost, P);
    SlotTab *tab = W.put(B.rtab, PS);
    ldso_rawjac_t *Jl = W.put(B.Jlin, nLin);
    float *rtz = W.put(B.rtz, nLin * 8);
    SlotRec *sr = W.put(H->sets[0].slot, PS);
    REQ(geo && pcw && phost && tab && Jl && rtz && sr, "ldso_ba_set_window: staging arena too small (internal)");
    H->h_phost.resize(P);
    for (int i = 0; i < P; i++) {
        PtGeo g_;
        memset(&g_, 0, sizeof(g_));           // step, the scalars of the last solve: zero
        g_.u = pts[i].u; g_.v = pts[i].v; g_.priorF = pts[i].priorF; g_.idepth = pts[i].idepth; g_.idepth_zero = pts[i].idepth_zero; g_.idepth_backup = pts[i].idepth;
        geo[i] = g_;
        REQ(pts[i].host >= 0 && pts[i].host < F, "ldso_ba_set_window: point host out of range");
        H->h_phost[i] = pts[i].host; phost[i] = pts[i].host;
        for (int k = 0; k < 8; k++) pcw[(size_t) i * 8 + k] = PtCw{pts[i].color[k], pts[i].weights[k]};
    }
    memset(sr, 0, PS * sizeof(SlotRec));          // JpJdF, centre, energies, activity, removal flag: zero
    for (size_t q = 0; q < PS; q++) { tab[q] = SlotTab{-1, 0, 0, -1}; sr[q].e[LD_SM_STATE].m.i = LDSO_RES_OOB; }
    H->flat2slot.assign(R, -1);
    size_t nl = 0;
    for (int i = 0; i < R; i++) {
        const ldso_residual_t &r = res[i];
        REQ(r.point >= 0 && r.point < P && r.target >= 0 && r.target < F && r.host == pts[r.point].host && r.target != r.host, "ldso_ba_set_window: bad residual indices");
        size_t slot = (size_t) r.point * FS + r.target;
        REQ(tab[slot].rflat < 0, "ldso_ba_set_window: two residuals of one point target the same frame");
        tab[slot].rflat = i; tab[slot].rlin = r.is_linearized ? 1 : 0; tab[slot].rnew = r.is_new ? 1 : 0;
        sr[slot].e[LD_SM_STATE].m.i = r.state_state; sr[slot].e[LD_SM_ACTIVE].m.i = r.is_active ? 1 : 0; sr[slot].e[LD_SM_ENERGY].m.f = r.state_energy;
        H->flat2slot[i] = (int32_t) slot;
        if (r.is_linearized) {
            tab[slot].rlidx = (int32_t) nl;
            Jl[nl] = linJ[i];
            for (int k = 0; k < 8; k++) rtz[nl * 8 + k] = lin_rtz[(size_t) i * 8 + k];
            nl++;
            // takeData (Residuals.h:123-128)
            const ldso_rawjac_t &J = linJ[i];
            float v0 = J.JIdx2[0] * J.Jpdd[0] + J.JIdx2[1] * J.Jpdd[1], v1 = J.JIdx2[2] * J.Jpdd[0] + J.JIdx2[3] * J.Jpdd[1];
            for (int k = 0; k < 6; k++) sr[slot].e[k].jp = J.Jpdxi[0][k] * v0 + J.Jpdxi[1][k] * v1;
            sr[slot].e[6].jp = J.JabJIdx[0] * J.Jpdd[0] + J.JabJIdx[1] * J.Jpdd[1];
            sr[slot].e[7].jp = J.JabJIdx[2] * J.Jpdd[0] + J.JabJIdx[3] * J.Jpdd[1];
        }
    }
    D.nL = (int) nLin;
    H->hasL = D.nL > 0;
    H->cur = 0; H->pendingApply = false; H->appliedValid = false;
    bool okT = W.again(H->sets[1].slot, sr, PS);
    for (int s_ = 0; s_ < 2; s_++) {
        ResSet &S = H->sets[s_];
        okT = okT && W.zero(S.pt, (size_t) P) && W.zero(S.acc, (size_t) P) && W.zero(S.G, (size_t) P * D.GS);
    }
    // a new window has a new dimension 8F+4: the marginalisation prior starts at zero (ldso_ba_set_prior follows when there is one)
    okT = okT && W.zero(B.HM, (size_t) D.n * D.n) && W.zero(B.bM, (size_t) D.n) && W.zero(B.scalars, (size_t) 16)
              && W.zero(B.scPart, (size_t) LD_SC_SPLITS * H->GSP * H->GSP);
    REQ(okT, "ldso_ba_set_window: upload table overflow (internal)");
    H->hasPrior = false;
    CHK(hipMemcpyAsync(H->d_stage, H->h_stage, W.used, hipMemcpyHostToDevice, H->stream));
    hipLaunchKernelGGL(k_win_scatter, dim3(256), dim3(256), 0, H->stream, (const char *) H->d_stage, W.n);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(H->stream));
    const int rc_ = build_chunks(H);
    guard.ok = (rc_ == LDSO_OK);
    return rc_;
}

// PointHessian::maxRelBaseline / numGoodResiduals live across optimize() calls in the reference (FullSystem.cc:1521-1536 updates them in the
// fixing pass, AccumulatedSCHessian.cc:14-21 zeroes maxRelBaseline of points without an active residual).  ldso_ba_set_window starts both at
// zero; a caller that keeps the reference's objects seeds them here so that ldso_ba_get_points returns the values to store back.
int ldso_ba_set_point_stats(ldso_ba_t *H, const float *maxRelBaseline, const int32_t *numGoodResiduals) {
    REQ(H && H->D.P > 0 && maxRelBaseline && numGoodResiduals, "ldso_ba_set_point_stats: bad arguments / no window");
    CHK(hipSetDevice(H->device));
    // one 4-byte field of every 64-byte PtRec of both sets: through the pinned staging arena of ldso_ba_set_window (free again: that call ends
    // synchronised) and one scatter kernel - four strided 2-D copies took 0.25 ms for 2000 points
    const size_t P = (size_t) H->D.P;
    REQ(H->h_stage && H->stageCap >= 8 * P, "ldso_ba_set_point_stats: staging arena missing (internal)");
    if (H->stageBusy) { CHK(hipStreamSynchronize(H->stream)); H->stageBusy = false; }
    memcpy(H->h_stage, maxRelBaseline, 4 * P); memcpy(H->h_stage + 4 * P, numGoodResiduals, 4 * P);
    CHK(hipMemcpyAsync(H->d_stage, H->h_stage, 8 * P, hipMemcpyHostToDevice, H->stream));
    hipLaunchKernelGGL(k_point_stats, dim3((unsigned) ((P + 255) / 256)), dim3(256), 0, H->stream, H->sets[0].pt, H->sets[1].pt, (const float *) H->d_stage, (const int32_t *) (H->d_stage + 4 * P), (int) P);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(H->stream));          // the arena is handed back to the next ldso_ba_set_window
    return LDSO_OK;
}

// The window of the next optimize() as a DELTA against the resident one - what the reference's own maintenance calls do to the window between two key frames
// (EnergyFunctional.cc: insertFrame :32, insertResidual :26, dropResidual :63, removePoint :153, dropPointsF :224, marginalizeFrame :72, makeIDX :380), expressed
// in one call on the order makeIDX produces:
//   frame_from[f]   the old index of new frame f (frames that appear nowhere were marginalised), -1 = insertFrame
//   point_from[i]   the old row of new point i (rows that appear nowhere were removed / dropped / marginalised), -1 - k = the k-th fresh point (insertPoint);
//                   surviving points keep their relative order (makeIDX walks frames, then the host frame's features: both orders are stable)
//   res_mask[i]     bit t set: point i has a residual with target frame t (NEW numbering).  Against the resident slots this says insertResidual (bit set, slot
//                   empty: the residual starts IN, energy 0, isNew) and dropResidual (slot occupied, bit clear)
//   fresh / fresh_res / fresh_mrb / fresh_ngr   the new points in the layout of ldso_ba_set_window (host = new frame index), their residuals point-major and
//                   target-ascending with .point = index into `fresh`, PointHessian::maxRelBaseline / numGoodResiduals
// The flat residual order of the new window (ldso_ba_get_residuals ...) is point-major, target-ascending.  The result is the window a fresh ldso_ba_set_window +
// ldso_ba_set_point_stats of the same objects produces, byte for byte; ldso_ba_set_frames / ldso_ba_set_prior follow as they do there.
int ldso_ba_update_window(ldso_ba_t *H, int F, const int32_t *image_slot, const int32_t *frame_from, int P, const int32_t *point_from, const uint32_t *res_mask,
                          int n_fresh, const ldso_point_t *fresh, int n_fresh_res, const ldso_residual_t *fresh_res, const float *fresh_mrb, const int32_t *fresh_ngr) {
    REQ(H && image_slot && frame_from && point_from && res_mask, "ldso_ba_update_window: null argument");
    REQ(H->D.P > 0 && !H->pendingApply, "ldso_ba_update_window: no resident window, or a linearisation is pending (ldso_ba_apply_res first)");
    REQ(!H->hasL, "ldso_ba_update_window: the resident window holds linearised residuals (use ldso_ba_set_window)");
    REQ(H->D.pBegin == 0 && H->D.pEnd == H->D.P, "ldso_ba_update_window: sharded window");
    REQ(F >= 2 && F <= H->maxF && P >= 1 && P <= H->maxP && n_fresh >= 0 && n_fresh_res >= 0, "ldso_ba_update_window: window exceeds the handle's capacity");
    REQ(n_fresh == 0 || (fresh && fresh_mrb && fresh_ngr), "ldso_ba_update_window: fresh points without records");
    REQ(n_fresh_res == 0 || fresh_res, "ldso_ba_update_window: fresh residuals without records");
    CHK(hipSetDevice(H->device));
    const BaDims oD = H->D;
    const int FS = (F + 7) / 8 * 8;
    const size_t PS = (size_t) P * FS;
    // ---- validate the delta on the host (indices only; nothing of the resident data is read back) ----
    {
        std::vector<char> seen(oD.F, 0);
        for (int f = 0; f < F; f++) {
            REQ(frame_from[f] >= -1 && frame_from[f] < oD.F, "ldso_ba_update_window: frame_from out of range");
            if (frame_from[f] >= 0) { REQ(!seen[frame_from[f]], "ldso_ba_update_window: an old frame appears twice"); seen[frame_from[f]] = 1; }
            REQ(f == 0 || frame_from[f] < 0 || frame_from[f - 1] < frame_from[f], "ldso_ba_update_window: surviving frames must keep their order, inserted frames come last");
            REQ(image_slot[f] >= 0 && image_slot[f] < H->maxF && H->imgSlots[image_slot[f]] != nullptr, "ldso_ba_update_window: image slot not set");
        }
    }
    std::vector<int32_t> resBegin((size_t) P + 1), freshResBegin((size_t) n_fresh + 1, 0), newHost((size_t) P);
    {
        std::vector<int32_t> oldToNew(oD.F, -1);
        for (int f = 0; f < F; f++) if (frame_from[f] >= 0) oldToNew[frame_from[f]] = f;
        int lastOld = -1, nextFresh = 0, acc = 0;
        const uint32_t fmask = (F >= 32) ? 0xFFFFFFFFu : ((1u << F) - 1u);
        for (int k = 0; k < n_fresh; k++) freshResBegin[k + 1] = 0;
        int fr = 0;
        for (int i = 0; i < P; i++) {
            const int from = point_from[i];
            int host;
            if (from >= 0) {
                REQ(from < oD.P && from > lastOld, "ldso_ba_update_window: surviving points must keep their order");
                lastOld = from;
                host = oldToNew[H->h_phost[from]];
                REQ(host >= 0, "ldso_ba_update_window: a surviving point is hosted by a frame that left the window");
            } else {
                REQ(-1 - from == nextFresh && nextFresh < n_fresh, "ldso_ba_update_window: fresh points must be numbered in window order");
                host = fresh[nextFresh].host;
                REQ(host >= 0 && host < F, "ldso_ba_update_window: fresh point host out of range");
                const int cnt = __builtin_popcount(res_mask[i]);
                freshResBegin[nextFresh] = fr;
                for (int c = 0; c < cnt; c++) {
                    REQ(fr < n_fresh_res && fresh_res[fr].target >= 0 && fresh_res[fr].target < F && fresh_res[fr].host == host,
                        "ldso_ba_update_window: fresh residual names a target outside the window or another host than its point's");
                    REQ(fresh_res[fr].point == nextFresh && !fresh_res[fr].is_linearized && ((res_mask[i] >> fresh_res[fr].target) & 1u)
                        && (c == 0 || fresh_res[fr - 1].target < fresh_res[fr].target), "ldso_ba_update_window: fresh residuals must be point-major, target-ascending and match res_mask");
                    fr++;
                }
                nextFresh++;
            }
            REQ((res_mask[i] & ~fmask) == 0 && !((res_mask[i] >> host) & 1u), "ldso_ba_update_window: res_mask names a frame outside the window or the host itself");
            REQ(i == 0 || newHost[i - 1] <= host, "ldso_ba_update_window: points must be ordered by host frame (EnergyFunctional::allPoints order)");
            newHost[i] = host;
            resBegin[i] = acc; acc += __builtin_popcount(res_mask[i]);
        }
        resBegin[P] = acc;
        freshResBegin[n_fresh] = fr;
        REQ(nextFresh == n_fresh && fr == n_fresh_res, "ldso_ba_update_window: unused fresh points / residuals");
    }
    const int R = resBegin[P];
    // ---- arena: [table | delta | image]; only table + delta cross PCIe ----
    auto A16 = [](size_t b) { return (b + 15) & ~(size_t) 15; };
    const size_t tabBytes = A16(LD_XFER_MAX * sizeof(WinXfer));
    const size_t deltaBytes = A16((size_t) F * 4) + 2 * A16((size_t) P * 4) + A16(((size_t) P + 1) * 4) + A16((size_t) n_fresh * sizeof(ldso_point_t)) + A16((size_t) n_fresh_res * sizeof(ldso_residual_t))
                              + A16(((size_t) n_fresh + 1) * 4) + 2 * A16((size_t) n_fresh * 4);
    const size_t need = tabBytes + deltaBytes + A16((size_t) P * sizeof(PtGeo)) + A16((size_t) P * 8 * sizeof(PtCw)) + A16((size_t) P * 4) + A16(PS * sizeof(SlotTab)) + A16(PS * sizeof(SlotRec))
                        + 2 * A16((size_t) P * 4) + 64;
    if (need > H->stageCap) {          // the arena only ever holds staging data: growing it loses nothing of the resident window
        CHK(hipStreamSynchronize(H->stream));
        if (H->h_stage) hipHostFree(H->h_stage);
        if (H->d_stage) hipFree(H->d_stage);
        H->h_stage = nullptr; H->d_stage = nullptr; H->stageCap = 0; H->stageBusy = false;
        const size_t cap = need + need / 4;
        CHK(hipHostMalloc((void **) &H->h_stage, cap));
        CHK(hipMalloc((void **) &H->d_stage, cap));
        H->stageCap = cap;
    }
    if (H->stageBusy) { CHK(hipStreamSynchronize(H->stream)); H->stageBusy = false; }
    WinStage W{H->h_stage, H->stageCap, tabBytes, reinterpret_cast<WinXfer *>(H->h_stage), 0};
    int32_t *hFrameFrom = W.raw<int32_t>(F), *hPointFrom = W.raw<int32_t>(P);
    uint32_t *hMask = W.raw<uint32_t>(P);
    int32_t *hResBegin = W.raw<int32_t>((size_t) P + 1);
    ldso_point_t *hFresh = W.raw<ldso_point_t>(n_fresh);
    ldso_residual_t *hFreshRes = W.raw<ldso_residual_t>(n_fresh_res);
    int32_t *hFreshResBegin = W.raw<int32_t>((size_t) n_fresh + 1);
    float *hMrb = W.raw<float>(n_fresh); int32_t *hNgr = W.raw<int32_t>(n_fresh);
    REQ(hFrameFrom && hPointFrom && hMask && hResBegin && hFresh && hFreshRes && hFreshResBegin && hMrb && hNgr, "ldso_ba_update_window: staging arena too small (internal)");
    const size_t upBytes = W.used;
    memcpy(hFrameFrom, frame_from, (size_t) F * 4); memcpy(hPointFrom, point_from, (size_t) P * 4); memcpy(hMask, res_mask, (size_t) P * 4);
    memcpy(hResBegin, resBegin.data(), ((size_t) P + 1) * 4); memcpy(hFreshResBegin, freshResBegin.data(), ((size_t) n_fresh + 1) * 4);
    if (n_fresh) { memcpy(hFresh, fresh, (size_t) n_fresh * sizeof(ldso_point_t)); memcpy(hMrb, fresh_mrb, (size_t) n_fresh * 4); memcpy(hNgr, fresh_ngr, (size_t) n_fresh * 4); }
    if (n_fresh_res) memcpy(hFreshRes, fresh_res, (size_t) n_fresh_res * sizeof(ldso_residual_t));
    BaPtrs &B = H->B;
    // the image region: same destinations, same order, same zero fills as ldso_ba_set_window
    PtGeo *geo = W.put(B.pgeo, P);
    PtCw *pcw = W.put(B.pcw, (size_t) P * 8);
    int32_t *phost = W.put(B.phost, P);
    SlotTab *tab = W.put(B.rtab, PS);
    SlotRec *sr = W.put(H->sets[0].slot, PS);
    REQ(geo && pcw && phost && tab && sr, "ldso_ba_update_window: staging arena too small (internal)");
    float *mrb = W.raw<float>(P); int32_t *ngr = W.raw<int32_t>(P);
    REQ(mrb && ngr, "ldso_ba_update_window: staging arena too small (internal)");
    auto dev = [&](const void *hp) { return H->d_stage + (reinterpret_cast<const char *>(hp) - H->h_stage); };
    WinDelta Wd;
    const ResSet &So = H->sets[H->cur];
    Wd.oGeo = B.pgeo; Wd.oPcw = B.pcw; Wd.oHost = B.phost; Wd.oTab = B.rtab; Wd.oSlot = So.slot; Wd.oPt = So.pt; Wd.oF = oD.F; Wd.oFS = oD.FS; Wd.oP = oD.P;
    Wd.frameFrom = (const int32_t *) dev(hFrameFrom); Wd.pointFrom = (const int32_t *) dev(hPointFrom); Wd.resMask = (const uint32_t *) dev(hMask); Wd.resBegin = (const int32_t *) dev(hResBegin);
    Wd.fresh = (const ldso_point_t *) dev(hFresh); Wd.freshRes = (const ldso_residual_t *) dev(hFreshRes); Wd.freshResBegin = (const int32_t *) dev(hFreshResBegin);
    Wd.freshMrb = (const float *) dev(hMrb); Wd.freshNgr = (const int32_t *) dev(hNgr);
    Wd.F = F; Wd.FS = FS; Wd.P = P;
    Wd.geo = (PtGeo *) dev(geo); Wd.pcw = (PtCw *) dev(pcw); Wd.phost = (int32_t *) dev(phost); Wd.tab = (SlotTab *) dev(tab); Wd.sr = (SlotRec *) dev(sr); Wd.mrb = (float *) dev(mrb); Wd.ngr = (int32_t *) dev(ngr);
    // from here on the handle describes the new window (a failure leaves it without one, as in ldso_ba_set_window)
    struct WinGuard { ldso_ba *H; bool ok; ~WinGuard() { if (!ok) { H->D.P = 0; H->D.R = 0; H->R = 0; H->appliedValid = false; H->itemValid = false; } } } guard{H, false};
    BaDims &D = H->D;
    D.F = F; D.FS = FS; D.P = P; D.R = R; D.n = 8 * F + 4; D.GS = 8 * D.FS + LD_GEXTRA; D.w = H->w; D.h = H->h; D.nsg = D.FS / 8; D.ks = H->reduceSplits;
    D.pBegin = 0; D.pEnd = P; D.wM3G = (float) (H->w - 3); D.hM3G = (float) (H->h - 3); D.nL = 0;
    H->GSP = (D.GS + 15) / 16 * 16;
    H->R = R;
    H->imageSlot.assign(image_slot, image_slot + F);
    for (int f = 0; f < F; f++) B.img[f] = H->imgSlots[image_slot[f]];
    H->h_phost.assign(newHost.begin(), newHost.end());
    H->flat2slot.assign(R, -1);
    for (int i = 0; i < P; i++) { int k = resBegin[i]; for (int t = 0; t < F; t++) if ((res_mask[i] >> t) & 1u) H->flat2slot[k++] = (int32_t) ((size_t) i * FS + t); }
    H->hasL = false; H->cur = 0; H->pendingApply = false; H->appliedValid = false; H->itemValid = false;
    bool okT = W.again(H->sets[1].slot, sr, PS);
    for (int s_ = 0; s_ < 2; s_++) {
        ResSet &S = H->sets[s_];
        okT = okT && W.zero(S.pt, (size_t) P) && W.zero(S.acc, (size_t) P) && W.zero(S.G, (size_t) P * D.GS);
    }
    okT = okT && W.zero(B.HM, (size_t) D.n * D.n) && W.zero(B.bM, (size_t) D.n) && W.zero(B.scalars, (size_t) 16) && W.zero(B.scPart, (size_t) LD_SC_SPLITS * H->GSP * H->GSP);
    REQ(okT, "ldso_ba_update_window: upload table overflow (internal)");
    H->hasPrior = false;
    CHK(hipMemcpyAsync(H->d_stage, H->h_stage, upBytes, hipMemcpyHostToDevice, H->stream));
    hipLaunchKernelGGL(k_win_rebuild, dim3((unsigned) ((PS + 255) / 256)), dim3(256), 0, H->stream, Wd);
    CHK(hipGetLastError());
    hipLaunchKernelGGL(k_win_scatter, dim3(256), dim3(256), 0, H->stream, (const char *) H->d_stage, W.n);
    CHK(hipGetLastError());
    hipLaunchKernelGGL(k_point_stats, dim3((unsigned) ((P + 255) / 256)), dim3(256), 0, H->stream, H->sets[0].pt, H->sets[1].pt, (const float *) Wd.mrb, (const int32_t *) Wd.ngr, P);
    CHK(hipGetLastError());
    CHK(hipStreamSynchronize(H->stream));
    const int rc_ = build_chunks(H);
    guard.ok = (rc_ == LDSO_OK);
    return rc_;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The same delta, recorded call by call as the reference edits its window (EnergyFunctional.cc): ldso_ba_window_begin, then any sequence of
//   ldso_ba_remove_frame    marginalizeFrame :72 (:138-150: the frame leaves; FullSystem::marginalizeFrame :607-632: so do the residuals that target it and the points it hosts)
//   ldso_ba_insert_frame    insertFrame :32 (the new frame's id for the calls below is returned: oF, oF + 1, ...)
//   ldso_ba_remove_points   removePoint :153 / dropPointsF :224 / the points marginalizePointsF :165 has absorbed
//   ldso_ba_drop_residuals  dropResidual :63
//   ldso_ba_add_residuals   insertResidual :26 for points of the window
//   ldso_ba_add_points      insertPoint + insertResidual for freshly activated points, each placed in front of a resident row (makeIDX :380 order)
// and ldso_ba_window_commit, which numbers the surviving frames (in order) and the inserted ones behind them and applies everything as ONE
// ldso_ba_update_window.  Host-side bookkeeping only until the commit; an invalid edit is rejected there and leaves the resident window as it was.
// ---------------------------------------------------------------------------------------------------------------------------------------------------
int ldso_ba_window_begin(ldso_ba_t *H) {
    REQ(H && H->D.P > 0 && !H->hasL && !H->pendingApply, "ldso_ba_window_begin: no resident window to edit (or linearised residuals / a pending linearisation)");
    ldso_ba::WindowEdit &E = H->edit;
    E = ldso_ba::WindowEdit();
    E.active = true; E.oF = H->D.F; E.oP = H->D.P;
    E.frameGone.assign(E.oF, 0); E.rowGone.assign(E.oP, 0); E.mask.assign(E.oP, 0u);
    for (int i = 0; i < H->R; i++) { const int sl = H->flat2slot[i]; E.mask[sl / H->D.FS] |= 1u << (sl % H->D.FS); }
    return LDSO_OK;
}
#define REQ_EDIT(name) REQ(H && H->edit.active, name ": no edit in progress (ldso_ba_window_begin first)")
int ldso_ba_remove_frame(ldso_ba_t *H, int frame_idx) {
    REQ_EDIT("ldso_ba_remove_frame");
    REQ(frame_idx >= 0 && frame_idx < H->edit.oF && !H->edit.frameGone[frame_idx], "ldso_ba_remove_frame: not a frame of the resident window");
    H->edit.frameGone[frame_idx] = 1;
    for (int r = 0; r < H->edit.oP; r++) { if (H->h_phost[r] == frame_idx) H->edit.rowGone[r] = 1; H->edit.mask[r] &= ~(1u << frame_idx); }
    return LDSO_OK;
}
int ldso_ba_insert_frame(ldso_ba_t *H, int image_slot, int *frame_id_out) {
    REQ_EDIT("ldso_ba_insert_frame");
    REQ(image_slot >= 0 && image_slot < H->maxF && H->imgSlots[image_slot] != nullptr, "ldso_ba_insert_frame: image slot not set");
    REQ(H->edit.oF + (int) H->edit.insertedSlots.size() < 32, "ldso_ba_insert_frame: too many frames in one edit");
    if (frame_id_out) *frame_id_out = H->edit.oF + (int) H->edit.insertedSlots.size();
    H->edit.insertedSlots.push_back(image_slot);
    return LDSO_OK;
}
int ldso_ba_remove_points(ldso_ba_t *H, int n, const int32_t *rows) {
    REQ_EDIT("ldso_ba_remove_points");
    REQ(n >= 0 && (n == 0 || rows), "ldso_ba_remove_points: bad arguments");
    for (int i = 0; i < n; i++) REQ(rows[i] >= 0 && rows[i] < H->edit.oP, "ldso_ba_remove_points: row out of range");
    for (int i = 0; i < n; i++) H->edit.rowGone[rows[i]] = 1;
    return LDSO_OK;
}
static int edit_residuals(ldso_ba *H, int n, const int32_t *rows, const int32_t *targets, bool add, const char *) {
    const int nF = H->edit.oF + (int) H->edit.insertedSlots.size();
    for (int i = 0; i < n; i++) {
        REQ(rows[i] >= 0 && rows[i] < H->edit.oP && targets[i] >= 0 && targets[i] < nF, "residual edit: row / target out of range");
        const bool has = (H->edit.mask[rows[i]] >> targets[i]) & 1u;
        REQ(add ? (!has && targets[i] != H->h_phost[rows[i]] && (targets[i] >= H->edit.oF || !H->edit.frameGone[targets[i]])) : has,
            add ? "ldso_ba_add_residuals: the point already has that residual, or the target is its host / a removed frame" : "ldso_ba_drop_residuals: the point has no such residual");
    }
    for (int i = 0; i < n; i++) { if (add) H->edit.mask[rows[i]] |= 1u << targets[i]; else H->edit.mask[rows[i]] &= ~(1u << targets[i]); }
    return LDSO_OK;
}
int ldso_ba_drop_residuals(ldso_ba_t *H, int n, const int32_t *rows, const int32_t *targets) {
    REQ_EDIT("ldso_ba_drop_residuals");
    REQ(n >= 0 && (n == 0 || (rows && targets)), "ldso_ba_drop_residuals: bad arguments");
    return edit_residuals(H, n, rows, targets, false, "");
}
int ldso_ba_add_residuals(ldso_ba_t *H, int n, const int32_t *rows, const int32_t *targets) {
    REQ_EDIT("ldso_ba_add_residuals");
    REQ(n >= 0 && (n == 0 || (rows && targets)), "ldso_ba_add_residuals: bad arguments");
    return edit_residuals(H, n, rows, targets, true, "");
}
int ldso_ba_add_points(ldso_ba_t *H, int n, const ldso_point_t *pts, const int32_t *before_row, int n_res, const ldso_residual_t *res, const float *mrb, const int32_t *ngr) {
    REQ_EDIT("ldso_ba_add_points");
    REQ(n >= 0 && n_res >= 0 && (n == 0 || (pts && before_row)) && (n_res == 0 || res), "ldso_ba_add_points: bad arguments");
    const int nF = H->edit.oF + (int) H->edit.insertedSlots.size();
    const size_t first = H->edit.fresh.size();
    for (int i = 0; i < n; i++) {
        REQ(before_row[i] >= 0 && before_row[i] <= H->edit.oP && pts[i].host >= 0 && pts[i].host < nF, "ldso_ba_add_points: before_row / host out of range");
        ldso_ba::NewPoint q; q.p = pts[i]; q.before = before_row[i]; q.mrb = mrb ? mrb[i] : 0.0f; q.ngr = ngr ? ngr[i] : 0;
        H->edit.fresh.push_back(q);
    }
    for (int i = 0; i < n_res; i++) {
        if (!(res[i].point >= 0 && res[i].point < n && res[i].target >= 0 && res[i].target < nF && !res[i].is_linearized)) {
            H->edit.fresh.resize(first); ldso_set_error("ldso_ba_add_points: residual names a point / frame outside the call, or is linearised"); return LDSO_E_INVALID;
        }
        H->edit.fresh[first + res[i].point].res.push_back(res[i]);
    }
    return LDSO_OK;
}
int ldso_ba_window_commit(ldso_ba_t *H) {
    REQ_EDIT("ldso_ba_window_commit");
    ldso_ba::WindowEdit &E = H->edit;
    const int nIns = (int) E.insertedSlots.size();
    std::vector<int32_t> idToNew(E.oF + nIns, -1), frameFrom, slots;
    for (int f = 0; f < E.oF; f++) if (!E.frameGone[f]) { idToNew[f] = (int) frameFrom.size(); frameFrom.push_back(f); slots.push_back(H->imageSlot[f]); }
    for (int k = 0; k < nIns; k++) { idToNew[E.oF + k] = (int) frameFrom.size(); frameFrom.push_back(-1); slots.push_back(E.insertedSlots[k]); }
    const int F = (int) frameFrom.size();
    auto translate = [&](uint32_t m) { uint32_t o = 0; for (int f = 0; f < E.oF + nIns; f++) if (((m >> f) & 1u) && idToNew[f] >= 0) o |= 1u << idToNew[f]; return o; };
    // fresh points in front of their rows, in call order (stable)
    std::vector<int> order(E.fresh.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int) i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return E.fresh[a].before < E.fresh[b].before; });
    std::vector<int32_t> pointFrom; std::vector<uint32_t> mask; std::vector<ldso_point_t> fp; std::vector<ldso_residual_t> fr; std::vector<float> fm; std::vector<int32_t> fg;
    size_t nx = 0;
    int rc = LDSO_OK;
    auto emitFresh = [&](int idx) {
        ldso_ba::NewPoint q = E.fresh[idx];
        const int k = (int) fp.size();
        if (idToNew[q.p.host] < 0) { rc = LDSO_E_INVALID; return; }
        q.p.host = idToNew[q.p.host];
        uint32_t m = 0;
        for (ldso_residual_t &r : q.res) { if (idToNew[r.target] < 0) { rc = LDSO_E_INVALID; return; } r.target = idToNew[r.target]; r.host = q.p.host; r.point = k; m |= 1u << r.target; }
        std::sort(q.res.begin(), q.res.end(), [](const ldso_residual_t &a, const ldso_residual_t &b) { return a.target < b.target; });
        fp.push_back(q.p); fm.push_back(q.mrb); fg.push_back(q.ngr);
        for (const ldso_residual_t &r : q.res) fr.push_back(r);
        pointFrom.push_back(-1 - k); mask.push_back(m);
    };
    for (int r = 0; r <= E.oP; r++) {
        while (nx < order.size() && E.fresh[order[nx]].before == r) emitFresh(order[nx++]);
        if (r < E.oP && !E.rowGone[r]) { pointFrom.push_back(r); mask.push_back(translate(E.mask[r])); }
    }
    E.active = false;
    if (rc != LDSO_OK) { ldso_set_error("ldso_ba_window_commit: a fresh point or residual names a removed frame"); return rc; }
    REQ(!pointFrom.empty() && F >= 2, "ldso_ba_window_commit: the edit leaves no window");
    return ldso_ba_update_window(H, F, slots.data(), frameFrom.data(), (int) pointFrom.size(), pointFrom.data(), mask.data(), (int) fp.size(), fp.data(), (int) fr.size(), fr.data(), fm.data(), fg.data());
}

// Points per workgroup of the fused linearisation.  0 (default): the smallest chunk that keeps ONE window's grid within one workgroup per CU
// (latency of a single window).  n > 0 (multiple of 4): fixed chunks of n points - what ldso_ba_batch_create applies to its windows, where
// the launch is filled by many windows and a workgroup's fixed costs (operand staging, block reduction) should be spread over more points.
// The fp32 partial sums of the top Hessian are formed per chunk: two handles agree bit for bit only under the same chunking.
static int launch_linearize(ldso_ba *H, bool fix, int stepMode = 0, int itCheck = -1);
// New chunks for the resident window.  The applied residual set carries per-chunk partial sums (top Hessian, energies) that the next
// reduce reads: under a new chunking they are re-formed by linearising the applied state once more with the decisions of the original pass
// kept (stepMode bit 2: same residual states, energies and activity - the Jacobians are a function of the state - only the partials are cut
// differently).
static int rechunk(ldso_ba *H) {
    H->itemValid = false;
    RUN(build_chunks(H));
    if (H->appliedValid && !H->pendingApply) { RUN(launch_linearize(H, false, 4, -1)); H->cur ^= 1; }
    return LDSO_OK;
}
int ldso_ba_set_chunk_points(ldso_ba_t *H, int points_per_workgroup) {
    REQ(H && points_per_workgroup >= 0 && points_per_workgroup % 4 == 0 && points_per_workgroup <= 1024, "ldso_ba_set_chunk_points: 0 or a multiple of 4 up to 1024");
    REQ(!H->pendingApply, "ldso_ba_set_chunk_points: a linearisation is pending (ldso_ba_apply_res first)");
    H->chunkPoints = points_per_workgroup;
    if (H->D.P > 0) { CHK(hipSetDevice(H->device)); return rechunk(H); }
    return LDSO_OK;
}
// K-splits (workgroups) per 16 x 16 tile of the Schur complement in the GN fast path of THIS handle: the fp32 partial sums of a tile are formed per split, so two runs
// agree bit for bit only under the same number (a batch uses ldso_ba_batch_reduce_splits; default LD_SCT_KS = 8).  Takes effect with the next reduction.
int ldso_ba_set_reduce_splits(ldso_ba_t *H, int splits) {
    REQ(H && splits >= 1 && splits <= 16, "ldso_ba_set_reduce_splits: bad arguments");
    REQ(!H->inBatch, "ldso_ba_set_reduce_splits: the handle is part of a batch");
    H->reduceSplits = splits; H->D.ks = splits;
    return LDSO_OK;
}

int ldso_ba_get_dims(ldso_ba_t *H, int *F, int *P, int *R) {
    REQ(H, "ldso_ba_get_dims: null handle");
    if (F) *F = H->D.F;
    if (P) *P = H->D.P;
    if (R) *R = H->R;
    return LDSO_OK;
}
int ldso_ba_get_chunk_cuts(ldso_ba_t *H, int32_t *ends, int cap, int *n_out) {
    REQ(H && n_out && H->D.P > 0, "ldso_ba_get_chunk_cuts: bad arguments / no window");
    const int n = (int) H->h_blocks.size();
    *n_out = n;
    if (ends) { REQ(cap >= n, "ldso_ba_get_chunk_cuts: buffer too small"); for (int i = 0; i < n; i++) ends[i] = H->h_blocks[i].p0 + H->h_blocks[i].np; }
    return LDSO_OK;
}
int ldso_ba_set_chunk_cuts(ldso_ba_t *H, const int32_t *ends, int n) {
    REQ(H && n >= 0 && (n == 0 || ends), "ldso_ba_set_chunk_cuts: bad arguments");
    REQ(!H->pendingApply, "ldso_ba_set_chunk_cuts: a linearisation is pending (ldso_ba_apply_res first)");
    REQ(H->inBatch == nullptr, "ldso_ba_set_chunk_cuts: the handle belongs to a batch (its chunks are the batch's)");
    for (int i = 0; i < n; i++) REQ(ends[i] > (i ? ends[i - 1] : 0) && ends[i] <= (H->D.P > 0 ? H->D.P : H->maxP), "ldso_ba_set_chunk_cuts: ends must ascend and stay inside the window");
    H->chunkCuts.assign(ends, ends + n);
    if (H->D.P > 0) { CHK(hipSetDevice(H->device)); return rechunk(H); }
    return LDSO_OK;
}
int ldso_ba_get_chunk_points(ldso_ba_t *H, int *points_per_workgroup, int *workgroups) {
    REQ(H && points_per_workgroup, "ldso_ba_get_chunk_points: null argument");
    *points_per_workgroup = H->chunkPoints;
    if (workgroups) *workgroups = H->D.nChunks;
    return LDSO_OK;
}

int ldso_ba_set_shard(ldso_ba_t *H, int pb, int pe) {
    REQ(H && pb >= 0 && pe >= pb && pe <= H->D.P, "ldso_ba_set_shard: bad range");
    H->D.pBegin = pb; H->D.pEnd = pe;
    return build_chunks(H);
}

size_t ldso_ba_reduce_doubles(ldso_ba_t *H) {
    if (!H) return 0;
    size_t n = H->D.n;
    return 3 * (n * n + n) + 8 + (size_t) H->D.P;
}

static int launch_solve(ldso_ba *H, const ResSet &S, unsigned flags, int iteration = 0, double lambda = 0, int logIdx = -1, double *rout = nullptr, const double *rin = nullptr);

int ldso_ba_set_frames(ldso_ba_t *H, const ldso_frame_t *fr, const ldso_calib_t *calib) {
    REQ(H && fr && calib && H->D.F > 0, "ldso_ba_set_frames: set the window first");
    CHK(hipSetDevice(H->device));
    const int F = H->D.F;
    std::vector<DevFrame> df(F);
    for (int f = 0; f < F; f++) {
        DevFrame &d = df[f];
        memset(&d, 0, sizeof(d));
        memcpy(d.evalPT, fr[f].worldToCam_evalPT, sizeof(d.evalPT));
        memcpy(d.state, fr[f].state, sizeof(d.state)); memcpy(d.state_zero, fr[f].state_zero, sizeof(d.state_zero));
        memcpy(d.state_backup, fr[f].state, sizeof(d.state));
        memcpy(d.prior, fr[f].prior, sizeof(d.prior));
        memcpy(d.ns_pose, fr[f].nullspaces_pose, sizeof(d.ns_pose)); memcpy(d.ns_scale, fr[f].nullspaces_scale, sizeof(d.ns_scale));
        memcpy(d.ns_affine, fr[f].nullspaces_affine, sizeof(d.ns_affine));
        d.ab_exposure = fr[f].ab_exposure; d.frameEnergyTH = fr[f].frameEnergyTH; d.frameID = fr[f].frameID; d.imgSlot = H->imageSlot[f];
    }
    DevCalib dc;
    memset(&dc, 0, sizeof(dc));
    for (int i = 0; i < 4; i++) { dc.value[i] = calib->value[i]; dc.value_zero[i] = calib->value_zero[i]; dc.value_backup[i] = calib->value[i]; }
    CHK(hipMemcpyAsync(H->B.frames, df.data(), F * sizeof(DevFrame), hipMemcpyHostToDevice, H->stream));
    CHK(hipMemcpyAsync(H->B.calib, &dc, sizeof(dc), hipMemcpyHostToDevice, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    // first precalc needs valid calib floats for the adjoint-independent parts; adjoints, then precalc
    return launch_solve(H, H->sets[H->cur], SK_ADJ | SK_PRECALC);
}

int ldso_ba_set_prior(ldso_ba_t *H, const double *HM, const double *bM) {
    REQ(H && H->D.n > 0, "ldso_ba_set_prior: set the window first");
    CHK(hipSetDevice(H->device));
    size_t n = H->D.n;
    H->hasPrior = (HM != nullptr) || (bM != nullptr);
    // through the pinned staging arena of ldso_ba_set_window (free: that call ends synchronised) when it is there - two copies from pageable memory + a wait
    // took 0.13 ms for 29 KB; no wait at the end: the arena is not reused before the next ldso_ba_set_window / set_point_stats, which synchronise first
    const size_t bytesH = n * n * 8, bytesB = n * 8;
    if (H->h_stage && H->stageCap >= bytesH + bytesB + 64) {
        if (H->stageBusy) { CHK(hipStreamSynchronize(H->stream)); H->stageBusy = false; }          // a previous ldso_ba_set_prior's copy
        if (HM) { memcpy(H->h_stage, HM, bytesH); CHK(hipMemcpyAsync(H->B.HM, H->h_stage, bytesH, hipMemcpyHostToDevice, H->stream)); } else CHK(hipMemsetAsync(H->B.HM, 0, bytesH, H->stream));
        if (bM) { memcpy(H->h_stage + bytesH, bM, bytesB); CHK(hipMemcpyAsync(H->B.bM, H->h_stage + bytesH, bytesB, hipMemcpyHostToDevice, H->stream)); } else CHK(hipMemsetAsync(H->B.bM, 0, bytesB, H->stream));
        H->stageBusy = true;
        return LDSO_OK;
    }
    if (HM) CHK(hipMemcpyAsync(H->B.HM, HM, bytesH, hipMemcpyHostToDevice, H->stream)); else CHK(hipMemsetAsync(H->B.HM, 0, bytesH, H->stream));
    if (bM) CHK(hipMemcpyAsync(H->B.bM, bM, bytesB, hipMemcpyHostToDevice, H->stream)); else CHK(hipMemsetAsync(H->B.bM, 0, bytesB, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// launch helpers with optional HIP-event timing
// ---------------------------------------------------------------------------------------------------------
static void t_begin(ldso_ba *H, int which) {
    if (!H->profile) return;
    Timer t; t.which = which;
    hipEventCreate(&t.a); hipEventCreate(&t.b);
    hipEventRecord(t.a, H->stream);
    H->timers.push_back(t);
}
static void t_end(ldso_ba *H) { if (H->profile) hipEventRecord(H->timers.back().b, H->stream); }

static int launch_solve(ldso_ba *H, const ResSet &S, unsigned flags, int iteration, double lambda, int logIdx, double *rout, const double *rin) {
    SolveArgs A;
    A.flags = flags; A.iteration = iteration; A.lambda = lambda; A.hasL = H->hasL ? 1 : 0; A.hasPrior = H->hasPrior ? 1 : 0; A.GSP = H->GSP; A.logIdx = logIdx; A.reduceOut = rout; A.reduceIn = rin; A.itCheck = -1; A.waitCtr = nullptr; A.waitTarget = 0; A.hostStop = nullptr; A.lastIt = -1;
    t_begin(H, 2);
    CHK(ba_launch_solve(H->B, H->D, S, H->settings, A, H->stream));
    t_end(H);
    return LDSO_OK;
}
// the handle's BatchItem in device memory, uploaded when it changed (window, image slots, accumulator lent to an all-reduce buffer, prior)
static int refresh_item(ldso_ba *H) {
    BatchItem it;
    memset(&it, 0, sizeof(it));
    it.B = H->B; it.D = H->D; it.set[0] = H->sets[0]; it.set[1] = H->sets[1]; it.cs = H->chunkStarts;
    it.hasPrior = H->hasPrior ? 1 : 0; it.GSP = H->GSP; it.linBlock0 = 0; it.redBlock0 = 0;
    if (H->itemValid && memcmp(&it, &H->itemShadow, sizeof(it)) == 0) return LDSO_OK;
    CHK(hipStreamSynchronize(H->stream));            // the pinned copy may still be in flight (rare: the descriptors change per key frame)
    memcpy(H->h_item, &it, sizeof(it));
    CHK(hipMemcpyAsync(H->d_item, H->h_item, sizeof(it), hipMemcpyHostToDevice, H->stream));
    H->itemShadow = it; H->itemValid = true;
    return LDSO_OK;
}

static int launch_linearize(ldso_ba *H, bool fix, int stepMode, int itCheck) {
    t_begin(H, 0);
    GnInit gi; gi.enable = (H->D.pBegin > 0) ? 2 : 1; gi.hasPrior = H->hasPrior ? 1 : 0; gi.calibPrior = H->settings.initialCalibHessian; gi.itCheck = itCheck;
    if (!fix && !H->hasL && gi.enable == 1 && H->linHeadOk && H->B.dumpJ == nullptr) {          // (the Jacobian dump of ldso_ba_set_debug_dump: the argument-based kernel)
        // the plain linearisation (GN iterations) of one window, one or two slot groups: descriptor and chunk geometry in the kernel arguments (k_linearize_one;
        // until round 3 k_linearize_batch with one window for F <= 8 and the argument-based kernel for F > 8)
        CHK(ba_launch_linearize_one(H->B, H->D, H->sets[H->cur], H->sets[H->cur ^ 1], H->settings, stepMode, gi, H->linHead, H->stream));
    } else          // fixing pass, linearised residuals, shards of a multi-GPU window: the argument-based kernels
    CHK(ba_launch_linearize(H->B, H->D, H->sets[H->cur], H->sets[H->cur ^ 1], H->settings, H->hasL, fix, stepMode, gi, H->stream));
    t_end(H);
    if (H->profile) { t_begin(H, 4); t_end(H); }      // empty event pair: calibrates the event overhead (which = 4)
    return LDSO_OK;
}
static int launch_reduce(ldso_ba *H, const ResSet &S, bool atomicMode = false, double lambda = 0.0, int itCheck = -1) {
    t_begin(H, 1);
    if (H->settings.solverMode & LDSO_SOLVER_USE_GN) lambda = 0;
    if (H->settings.solverMode & LDSO_SOLVER_FIX_LAMBDA) lambda = 1e-5;
    const double l1 = 1 + lambda, il = (double) (1.0f / (1 + lambda));
    CHK(ba_launch_reduce(H->B, H->D, S, H->chunkStarts, H->hasL, H->GSP, atomicMode ? ((H->D.pBegin > 0) ? 2 : 1) : 0, H->hasPrior, H->settings.initialCalibHessian, l1, il, itCheck, H->stream));
    t_end(H);
    return LDSO_OK;
}
static int launch_gather(ldso_ba *H, const ResSet &S, double lambda, int mode, double *rbuf) {
    t_begin(H, 1);
    CHK(ba_launch_gather(H->B, H->D, S, H->hasL, H->hasPrior, H->GSP, lambda, H->settings, mode, rbuf, H->stream));
    t_end(H);
    return LDSO_OK;
}
static int launch_pstep(ldso_ba *H, const ResSet &S, int mode) {
    t_begin(H, 3);
    CHK(ba_launch_point_step(H->B, H->D, S, mode, H->stream));
    t_end(H);
    return LDSO_OK;
}

static int read_scalars(ldso_ba *H, double *sc) {
    CHK(hipMemcpyAsync(sc, H->B.scalars, 16 * sizeof(double), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

int ldso_ba_profile(ldso_ba_t *H, int enable) {
    REQ(H, "null handle");
    H->profile = enable != 0;
    for (auto &t : H->timers) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    H->timers.clear();
    for (int i = 0; i < 5; i++) { H->tsum[i] = 0; H->tcnt[i] = 0; }
    return LDSO_OK;
}

int ldso_ba_kernel_time_ms(ldso_ba_t *H, int which, double *avg_ms, int *launches) {
    REQ(H && which >= 0 && which < 5, "bad arguments");
    CHK(hipStreamSynchronize(H->stream));
    for (auto &t : H->timers) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { H->tsum[t.which] += ms; H->tcnt[t.which]++; }
        hipEventDestroy(t.a); hipEventDestroy(t.b);
    }
    H->timers.clear();
    if (avg_ms) *avg_ms = H->tcnt[which] ? H->tsum[which] / H->tcnt[which] : 0.0;
    if (launches) *launches = H->tcnt[which];
    return LDSO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// the optimisation slice
// ---------------------------------------------------------------------------------------------------------
// Average duration of the dominant kernel for bench.py's roofline: `reps` back-to-back launches of k_linearize on the applied
// state (read set -> scratch set, no point step, nothing applied: idempotent) between ONE pair of HIP events on the handle's
// stream, so the event overhead is amortised over the launches; includes the ~1.5 us dependent-launch boundary per launch.
int ldso_ba_time_linearize(ldso_ba_t *H, int reps, double *avg_us) {
    REQ(H && H->D.P > 0 && reps > 0 && avg_us, "bad arguments");
    CHK(hipSetDevice(H->device));
    REQ(!H->pendingApply, "ldso_ba_time_linearize: a linearizeAll result is pending");
    const bool prof = H->profile;
    H->profile = false;
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    RUN(launch_linearize(H, false, 0));          // warm
    CHK(hipEventRecord(a, H->stream));
    for (int i = 0; i < reps; i++) RUN(launch_linearize(H, false, 0));
    CHK(hipEventRecord(b, H->stream));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    H->profile = prof;
    *avg_us = (double) ms * 1e3 / reps;
    return LDSO_OK;
}

int ldso_ba_collect_active(ldso_ba_t *H) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    H->pendingApply = false;
    return launch_solve(H, H->sets[H->cur], SK_COLLECT);
}

int ldso_ba_linearize_all(ldso_ba_t *H, int fix, double *energy_out) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    RUN(launch_linearize(H, fix != 0));
    RUN(launch_solve(H, H->sets[H->cur ^ 1], SK_POST | SK_THRESH));
    H->pendingApply = true;
    if (fix) { H->cur ^= 1; H->pendingApply = false; H->appliedValid = true; }     // applyRes happens inside the reductor when fixing
    double sc[16];
    RUN(read_scalars(H, sc));
    if (energy_out) *energy_out = sc[0];
    if (!std::isfinite(sc[0])) return LDSO_E_NONFINITE;
    return LDSO_OK;
}

int ldso_ba_apply_res(ldso_ba_t *H) {
    REQ(H, "null handle");
    if (H->pendingApply) { H->cur ^= 1; H->pendingApply = false; H->appliedValid = true; }
    return LDSO_OK;
}

int ldso_ba_backup_state(ldso_ba_t *H) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    RUN(launch_solve(H, H->sets[H->cur], SK_BACKUP));
    RUN(launch_pstep(H, H->sets[H->cur], PS_BACKUP));
    return LDSO_OK;
}

#define REQ_UNSHARDED(name) REQ(H->D.pBegin == 0 && H->D.pEnd == H->D.P, name ": not available on a sharded handle (ldso_ba_set_shard): " \
                                    "use ldso_ba_gn_reduce_local / ldso_ba_gn_solve_reduced or ldso_ba_reduce_local / ldso_ba_solve_reduced around the all-reduce")

int ldso_ba_solve_system(ldso_ba_t *H, int iteration, double lambda) {
    REQ(H && H->D.P > 0, "no window");
    REQ_UNSHARDED("ldso_ba_solve_system");
    CHK(hipSetDevice(H->device));
    const ResSet &S = H->sets[H->cur];
    RUN(launch_reduce(H, S));
    RUN(launch_gather(H, S, lambda, 0, nullptr));
    RUN(launch_solve(H, S, SK_SOLVE, iteration, lambda));
    RUN(launch_pstep(H, S, PS_RESUB));
    double sc[16];
    RUN(read_scalars(H, sc));
    if (sc[4] != 0.0) return LDSO_E_NONFINITE;
    return LDSO_OK;
}

int ldso_ba_do_step(ldso_ba_t *H, int *canbreak) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    RUN(launch_solve(H, H->sets[H->cur], SK_STEP | SK_PRECALC));
    RUN(launch_pstep(H, H->sets[H->cur], PS_STEP));
    double sc[16];
    RUN(read_scalars(H, sc));
    if (canbreak) *canbreak = sc[3] != 0.0;
    return LDSO_OK;
}

int ldso_ba_load_state_backup(ldso_ba_t *H) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    RUN(launch_solve(H, H->sets[H->cur], SK_LOADBK | SK_PRECALC));
    RUN(launch_pstep(H, H->sets[H->cur], PS_LOAD));
    H->pendingApply = false;
    return LDSO_OK;
}

// one GN iteration = solveSystem + doStepFromBackup + linearizeAll(false) + applyRes: 4 launches (k_reduce, k_gather,
// k_gn_solve, k_linearize with the point step fused in), no host sync
static int enqueue_iteration(ldso_ba *H, int iteration, double lambda, int logIdx, bool postOfPrev, int itCheck = -1, int lastIt = -1) {
    const ResSet &S = H->sets[H->cur];
    if (H->B.acc != H->ownAcc) {      // the accumulator was lent to an all-reduce buffer: take it back (and re-initialise)
        H->B.acc = H->ownAcc;
        GnInit gi; gi.enable = 1; gi.hasPrior = H->hasPrior ? 1 : 0; gi.calibPrior = H->settings.initialCalibHessian; gi.itCheck = -1;
        CHK(ba_launch_acc_init(H->B, H->D, gi, H->stream));
    }
    (void) postOfPrev;
    SolveArgs A;
    A.flags = 0; A.iteration = iteration; A.lambda = lambda; A.hasL = H->hasL ? 1 : 0; A.hasPrior = H->hasPrior ? 1 : 0; A.GSP = H->GSP; A.logIdx = logIdx;
    A.reduceOut = nullptr; A.reduceIn = nullptr; A.itCheck = itCheck; A.waitCtr = nullptr; A.waitTarget = 0; A.hostStop = (itCheck >= 0) ? H->d_stop : nullptr; A.lastIt = lastIt;
    const int nT = H->GSP / 16;
    const int nReduce = H->D.F * H->D.F * (H->hasL ? 2 : 1) + H->D.ks * nT * (nT + 1) / 2 + 1;      // grid of ba_launch_reduce in atomic mode
    if (nReduce + 2 <= H->numCU && !H->noFusedLaunch) {
        // k_reduce (fp64 atomics straight into B.acc, no k_gather on this path) and the control step in ONE launch: the control
        // workgroup waits on a device counter for the reduce workgroups (k_reduce_solve, ba_solve.hip).  Only while every workgroup
        // of the launch gets its own CU (F <= 8; from F = 9 the Schur part alone has 180 workgroups): the fused kernel's LDS footprint allows one workgroup per CU.
        A.waitCtr = H->d_waitCtr;
        double lam = lambda;
        if (H->settings.solverMode & LDSO_SOLVER_USE_GN) lam = 0;
        if (H->settings.solverMode & LDSO_SOLVER_FIX_LAMBDA) lam = 1e-5;
        const double l1 = 1 + lam, il = (double) (1.0f / (1 + lam));
        t_begin(H, 2);
        CHK(ba_launch_reduce_solve(H->B, H->D, S, H->settings, A, H->chunkStarts, (H->D.pBegin > 0) ? 2 : 1, H->settings.initialCalibHessian, l1, il, H->stream));
        t_end(H);
    } else {
        RUN(launch_reduce(H, S, true, lambda, itCheck));
        t_begin(H, 2);
        CHK(ba_launch_gn_solve(H->B, H->D, S, H->settings, A, H->stream));
        t_end(H);
    }
    RUN(launch_linearize(H, false, 1, itCheck));
    H->cur ^= 1; H->appliedValid = true;      // forceAcceptStep: applyRes
    return LDSO_OK;
}

static unsigned long long fnv1a(unsigned long long h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
// everything the launches of `iters` forced iterations take as arguments, as bytes (the cache key) ...
static void gn_key(const ldso_ba *H, int first_iteration, int iters, std::vector<unsigned char> &key) {
    key.clear();
    auto put = [&key](const void *p, size_t n) { const unsigned char *b = (const unsigned char *) p; key.insert(key.end(), b, b + n); };
    put(&H->B, sizeof(H->B)); put(&H->D, sizeof(H->D)); put(H->sets, sizeof(H->sets)); put(&H->settings, sizeof(H->settings));
    put(&H->chunkStarts, sizeof(H->chunkStarts)); put(&H->linHead, sizeof(H->linHead));
    const long long misc[12] = {first_iteration, iters, H->cur, H->hasL, H->hasPrior, H->GSP, H->linHeadOk, H->noFusedLaunch, H->numCU, (long long) (size_t) H->stream, (long long) (size_t) H->ownAcc,
                                (long long) (size_t) H->d_waitCtr};
    put(misc, sizeof(misc));
}
// ... and their 64-bit hash (pre-selection only: a hit is confirmed on the bytes)
static unsigned long long gn_signature(const std::vector<unsigned char> &key) { return fnv1a(1469598103934665603ull, key.data(), key.size()); }
static int enqueue_gn_plain(ldso_ba *H, int first_iteration, int iters) {
    CHK(hipMemsetAsync(H->d_waitCtr, 0, 4 * sizeof(int), H->stream));      // an aborted launch must not leave the producer counter armed
    for (int i = 0; i < iters; i++) RUN(enqueue_iteration(H, first_iteration + i, 1e-1, -1, true));
    return LDSO_OK;
}
// The iterations are 2-3 dependent launches each, the host runs far ahead of the device, and what is left to remove on the device side is the per-packet work of the
// command processor: the same sequence captured ONCE into a HIP graph and replayed is 35.55 against 36.04 us per iteration at C3 (scripts/r5/graph_gn.py).  The
// graph is keyed by a hash of every launch argument (gn_signature): anything that changes what the kernels are handed - a new window, other settings, a prior,
// another stream, another iteration index (the orthogonalisation starts at iteration 2) - captures anew; profiling runs (per-kernel events) and callers that are
// capturing themselves take the plain path.  LDSO_GN_GRAPHS=0 turns it off.
int ldso_ba_enqueue_gn(ldso_ba_t *H, int first_iteration, int iters) {
    REQ(H && H->D.P > 0 && iters >= 0, "bad arguments");
    REQ_UNSHARDED("ldso_ba_enqueue_gn");
    CHK(hipSetDevice(H->device));
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (!H->gnUseGraphs || H->profile || iters < 2 || H->B.acc != H->ownAcc || hipStreamIsCapturing(H->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
        return enqueue_gn_plain(H, first_iteration, iters);
    std::vector<unsigned char> &key = H->gnKeyScratch;
    gn_key(H, first_iteration, iters, key);
    const unsigned long long sig = gn_signature(key);
    for (ldso_ba::GnGraph &g : H->gnGraphs)
        if (g.sig == sig && g.key == key) {
            CHK(hipGraphLaunch(g.exec, H->stream));
            if (iters & 1) H->cur ^= 1;          // what the captured enqueue did to the handle's host state: the sets swap once per iteration
            H->appliedValid = true;
            return LDSO_OK;
        }
    // capture; the handle's host state advances as in a plain enqueue, the device work happens at the launch below
    const int cur0 = H->cur;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(H->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void) hipGetLastError(); return enqueue_gn_plain(H, first_iteration, iters); }
    const int rc = enqueue_gn_plain(H, first_iteration, iters);
    const hipError_t ec = hipStreamEndCapture(H->stream, &graph);
    hipGraphExec_t exec = nullptr;
    if (rc != LDSO_OK || ec != hipSuccess || graph == nullptr || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void) hipGetLastError();
        if (graph) hipGraphDestroy(graph);
        H->cur = cur0;                            // nothing ran: enqueue for real
        H->gnUseGraphs = false;                   // this runtime / stream does not capture the sequence: do not try again
        return enqueue_gn_plain(H, first_iteration, iters);
    }
    if (H->gnGraphs.size() >= 4) { hipGraphExecDestroy(H->gnGraphs.front().exec); hipGraphDestroy(H->gnGraphs.front().graph); H->gnGraphs.erase(H->gnGraphs.begin()); }
    H->gnGraphs.push_back(ldso_ba::GnGraph{sig, key, exec, graph});
    CHK(hipGraphLaunch(exec, H->stream));
    return LDSO_OK;
}

int ldso_ba_sync(ldso_ba_t *H) {
    REQ(H, "null handle");
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

// EnergyFunctional::calcMEnergyF / calcLEnergyF_MT (EnergyFunctional.cc:353-378, 627-682) at the current state: the two extra
// terms of the LM accept test (FullSystem.cc:805-826).  (With setting_forceAceptStep the reference skips them, FullSystem.cc:1694-1704.)
int ldso_ba_calc_lm_energies(ldso_ba_t *H, double *energy_M, double *energy_L) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    CHK(ba_launch_lm_energies(H->B, H->D, H->sets[H->cur], H->settings.initialCalibHessian, H->hasPrior, H->stream));
    double sc[16];
    RUN(read_scalars(H, sc));
    if (energy_M) *energy_M = sc[12];
    if (energy_L) *energy_L = sc[13];
    return LDSO_OK;
}

// FullSystem::optimize with setting_forceAceptStep = false (FullSystem.cc:777-831): every iteration is accepted or rejected on
// E_P + E_L + E_M; a rejected step restores the backup (loadSateBackup), re-linearises and multiplies lambda by 100.  One host
// round trip per stage - this is not the default schedule of the reference (Setting.cc:73) and not the timed path.
static int optimize_lm(ldso_ba *H, int mnumOptIts, int force_all, float *rmse_out, int *iters_out) {
    const int F = H->D.F;
    if (!force_all) { if (F < 3) mnumOptIts = 20; if (F < 4) mnumOptIts = 15; }
    REQ(mnumOptIts + 2 < 64, "too many iterations");
    std::vector<double> elog;
    RUN(ldso_ba_collect_active(H));
    double lastE = 0, lastL = 0, lastM = 0;
    RUN(ldso_ba_linearize_all(H, 0, &lastE));
    RUN(ldso_ba_calc_lm_energies(H, &lastM, &lastL));
    elog.push_back(lastE);
    RUN(ldso_ba_apply_res(H));
    double lambda = 1e-1;
    int done = 0;
    for (int it = 0; it < mnumOptIts; it++) {
        RUN(ldso_ba_backup_state(H));
        RUN(ldso_ba_solve_system(H, it, lambda));
        int canbreak = 0;
        RUN(ldso_ba_do_step(H, &canbreak));
        double newE = 0, newL = 0, newM = 0;
        RUN(ldso_ba_linearize_all(H, 0, &newE));
        RUN(ldso_ba_calc_lm_energies(H, &newM, &newL));
        elog.push_back(newE);
        done = it + 1;
        if (newE + newL + newM < lastE + lastL + lastM) {
            RUN(ldso_ba_apply_res(H));
            lastE = newE; lastL = newL; lastM = newM;
            lambda *= 0.25;
        } else {
            RUN(ldso_ba_load_state_backup(H));
            RUN(ldso_ba_linearize_all(H, 0, &lastE));
            H->pendingApply = false;                 // the re-linearisation at the restored state is not applied (FullSystem.cc:821-826)
            RUN(ldso_ba_calc_lm_energies(H, &lastM, &lastL));
            lambda *= 1e2;
        }
        if (canbreak && it >= H->settings.minOptIterations && !force_all) break;
    }
    RUN(launch_solve(H, H->sets[H->cur], SK_REANCHOR | SK_ADJ | SK_NONULLSPACE | SK_PRECALC));
    double Efix = 0;
    RUN(ldso_ba_linearize_all(H, 1, &Efix));
    elog.push_back(Efix);
    CHK(hipMemsetAsync(H->B.energyLog, 0, 64 * 8, H->stream));
    CHK(hipMemcpyAsync(H->B.energyLog, elog.data(), elog.size() * sizeof(double), hipMemcpyHostToDevice, H->stream));
    double sc[16];
    RUN(read_scalars(H, sc));
    H->lastIterations = (int) elog.size() - 2;
    if (iters_out) *iters_out = done;
    if (rmse_out) *rmse_out = sqrtf((float) (sc[0] / (8 * sc[9])));
    if (!std::isfinite(sc[0]) || sc[4] != 0.0) return LDSO_E_NONFINITE;
    return LDSO_OK;
}

int ldso_ba_optimize(ldso_ba_t *H, int mnumOptIts, int force_all, float *rmse_out, int *iters_out) {
    REQ(H && H->D.P > 0, "no window");
    REQ_UNSHARDED("ldso_ba_optimize");
    CHK(hipSetDevice(H->device));
    CHK(hipMemsetAsync(H->d_waitCtr, 0, 4 * sizeof(int), H->stream));      // an aborted launch must not leave the producer counter armed
    if (!H->settings.forceAcceptStep) return optimize_lm(H, mnumOptIts, force_all, rmse_out, iters_out);
    const int F = H->D.F;
    if (F < 2) { if (rmse_out) *rmse_out = 0; return LDSO_OK; }
    if (!force_all) { if (F < 3) mnumOptIts = 20; if (F < 4) mnumOptIts = 15; }
    REQ(mnumOptIts + 2 < 64, "too many iterations");
    CHK(hipMemsetAsync(H->B.energyLog, 0, 64 * 8, H->stream));
    H->pendingApply = false;
    RUN(launch_linearize(H, false, 2));            // stepMode bit 1: resetOOB of the optimize() preamble fused into the first linearizeAll
    H->cur ^= 1; H->appliedValid = true;           // applyRes
    int done = 0;
    double lambda = 1e-1;
    {   // no iteration has asked to stop yet
        CHK(hipMemcpyAsync(H->B.scalars + LD_SC_STOP, &H->neverStop, sizeof(double), hipMemcpyHostToDevice, H->stream));
    }
    volatile int *stopWord = H->h_stop;
    *stopWord = -1;
    for (int it = 0; it < mnumOptIts; it++) {
        // un-forced: the device decides (canbreak && it >= minOptIterations, FullSystem.cc:829); later iterations become no-ops
        RUN(enqueue_iteration(H, it, lambda, it, true, force_all ? -1 : it, mnumOptIts - 1));    // POST/THRESH/LOG of the previous linearize ride along
        lambda *= 0.25;
    }
    done = mnumOptIts;
    if (!force_all && mnumOptIts > 0) {
        // The control step of the iteration that ends the loop writes its index into a host-mapped word: the host learns `done` while the
        // GPU is still busy and enqueues the tail right behind the iterations that turned into no-ops (no stream synchronisation, which
        // cost a 40 us bubble).  The stream is polled as well so that a failed launch cannot hang the caller.
        int spins = 0;
        while (*stopWord < 0) {
            if ((++spins & 0x3FF) == 0) {
                const hipError_t q = hipStreamQuery(H->stream);
                if (q == hipSuccess) break;                         // everything ran: the word is final (or the loop never reported)
                if (q != hipErrorNotReady) CHK(q);
            }
        }
        int stopIt = *stopWord;
        if (stopIt < 0) {                                           // not reported (cannot happen on a healthy run): fall back to the device scalar
            double sc[16];
            RUN(read_scalars(H, sc));
            stopIt = (sc[LD_SC_STOP] < (double) mnumOptIts) ? (int) sc[LD_SC_STOP] : mnumOptIts - 1;
        }
        done = stopIt + 1;
        if ((mnumOptIts - done) & 1) H->cur ^= 1;   // the skipped iterations never wrote / applied a residual set
    }
    // tail: statistics of the last linearize, re-anchor the newest frame, adjoints, precalc, linearizeAll(true)
    RUN(launch_solve(H, H->sets[H->cur], SK_POST | SK_THRESH | SK_LOG | SK_REANCHOR | SK_ADJ | SK_NONULLSPACE | SK_PRECALC, 0, 0, done));
    RUN(launch_linearize(H, true));
    H->cur ^= 1;
    RUN(launch_solve(H, H->sets[H->cur], SK_POST | SK_THRESH | SK_LOG, 0, 0, done + 1));
    double sc[16];
    RUN(read_scalars(H, sc));
    H->lastIterations = done;
    if (iters_out) *iters_out = done;
    if (rmse_out) *rmse_out = sqrtf((float) (sc[0] / (8 * sc[9])));
    if (!std::isfinite(sc[0]) || sc[4] != 0.0) return LDSO_E_NONFINITE;
    return LDSO_OK;
}

// EnergyFunctional::marginalizePointsF (EnergyFunctional.cc:165-222) for the points with flags[p] != 0, including the
// re-linearise + fixLinearizationF pass FullSystem::flagPointsForRemoval ran on them (FullSystem.cc:1241-1250).
// The applied state of the window is not changed; the caller removes the points (next ldso_ba_set_window).
int ldso_ba_marginalize_points(ldso_ba_t *H, const int32_t *flags, double *HM_out, double *bM_out) {
    REQ(H && flags && H->D.P > 0, "ldso_ba_marginalize_points: bad arguments");
    CHK(hipSetDevice(H->device));
    REQ(!H->pendingApply, "ldso_ba_marginalize_points: a linearizeAll result is pending (apply or discard it first)");
    REQ(H->D.pBegin == 0 && H->D.pEnd == H->D.P, "ldso_ba_marginalize_points: not available on a sharded handle");
    const size_t n = H->D.n;
    if (!H->hasPrior) { CHK(hipMemsetAsync(H->B.HM, 0, n * n * 8, H->stream)); CHK(hipMemsetAsync(H->B.bM, 0, n * 8, H->stream)); }
    CHK(hipMemcpyAsync(H->d_margFlags, flags, (size_t) H->D.P * 4, hipMemcpyHostToDevice, H->stream));
    const ResSet &scratch = H->sets[H->cur ^ 1];
    CHK(ba_launch_linearize_marg(H->B, H->D, H->sets[H->cur], scratch, H->settings, H->d_margFlags, H->stream));
    CHK(ba_launch_reduce(H->B, H->D, scratch, H->chunkStarts, /*hasL*/ false, H->GSP, 0, false, 0.0f, 1.0, 1.0, -1, H->stream));
    CHK(ba_launch_gather(H->B, H->D, scratch, /*hasL*/ false, /*hasPrior*/ false, H->GSP, 0.0, H->settings, 0, nullptr, H->stream));
    CHK(ba_launch_marg_update(H->B, H->D, (double) H->settings.margWeightFac, H->stream));
    H->hasPrior = true;
    if (HM_out) CHK(hipMemcpyAsync(HM_out, H->B.HM, n * n * 8, hipMemcpyDeviceToHost, H->stream));
    if (bM_out) CHK(hipMemcpyAsync(bM_out, H->B.bM, n * 8, hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cc:72-151) applied to the device prior: returns the prior of the
// window without frame `frame_idx` ((8(F-1)+4)^2 row-major, 8(F-1)+4).  The handle keeps its window; the caller rebuilds
// it without the frame (ldso_ba_set_window / ldso_ba_set_prior with the returned matrices).
int ldso_ba_marginalize_frame(ldso_ba_t *H, int frame_idx, double *HM_out, double *bM_out) {
    REQ(H && HM_out && bM_out && H->D.F >= 2 && frame_idx >= 0 && frame_idx < H->D.F, "ldso_ba_marginalize_frame: bad arguments");
    CHK(hipSetDevice(H->device));
    const size_t n = H->D.n, nd = n - 8;
    if (!H->hasPrior) { CHK(hipMemsetAsync(H->B.HM, 0, n * n * 8, H->stream)); CHK(hipMemsetAsync(H->B.bM, 0, n * 8, H->stream)); }
    // scratch: B.sys holds 4 (n^2 + n) doubles: work = first n^2 + n, output after it
    double *work = H->B.sys, *oH = work + n * n + n, *ob = oH + nd * nd;
    CHK(ba_launch_marg_frame(H->B, H->D, frame_idx, work, oH, ob, H->stream));
    CHK(hipMemcpyAsync(HM_out, oH, nd * nd * 8, hipMemcpyDeviceToHost, H->stream));
    CHK(hipMemcpyAsync(bM_out, ob, nd * 8, hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

// ---- multi-GPU fast path: the all-reduce buffer IS the HFinal / bFinal accumulator -----------------------------------
//   layout: [HFinal lower triangle (n*n) | bFinal (n) | 8 scalars | P energy candidates], n = 8F+4
size_t ldso_ba_gn_reduce_doubles(ldso_ba_t *H) {
    if (!H) return 0;
    size_t n = H->D.n;
    return n * n + n + 8 + (size_t) H->D.P;
}

int ldso_ba_gn_reduce_local(ldso_ba_t *H, void *buf, double lambda) {
    REQ(H && buf && H->D.P > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    const ResSet &S = H->sets[H->cur];
    if (H->B.acc != (double *) buf) {
        // first use of this buffer: re-point the accumulator at it and give it what the last k_linearize put into the old one
        H->B.acc = (double *) buf;
        GnInit gi; gi.enable = (H->D.pBegin > 0) ? 2 : 1; gi.hasPrior = H->hasPrior ? 1 : 0; gi.calibPrior = H->settings.initialCalibHessian; gi.itCheck = -1;
        CHK(ba_launch_acc_init(H->B, H->D, gi, H->stream));
    }
    RUN(launch_reduce(H, S, true, lambda));
    const size_t n = H->D.n;
    CHK(ba_launch_gn_export(H->B, H->D, S, (double *) buf + n * n + n, H->stream));
    return LDSO_OK;
}

int ldso_ba_gn_solve_reduced(ldso_ba_t *H, const void *buf, int iteration, double lambda) {
    REQ(H && buf && H->D.P > 0 && H->B.acc == (const double *) buf, "ldso_ba_gn_solve_reduced: pass the buffer of ldso_ba_gn_reduce_local");
    CHK(hipSetDevice(H->device));
    const ResSet &S = H->sets[H->cur];
    const size_t n = H->D.n;
    SolveArgs A;
    A.flags = 0; A.iteration = iteration; A.lambda = lambda; A.hasL = H->hasL ? 1 : 0; A.hasPrior = H->hasPrior ? 1 : 0; A.GSP = H->GSP; A.logIdx = -1;
    A.reduceOut = nullptr; A.reduceIn = (const double *) buf + n * n + n; A.itCheck = -1; A.waitCtr = nullptr; A.waitTarget = 0; A.hostStop = nullptr; A.lastIt = -1;
    t_begin(H, 2);
    CHK(ba_launch_gn_solve(H->B, H->D, S, H->settings, A, H->stream));
    t_end(H);
    RUN(launch_linearize(H, false, 1));
    H->cur ^= 1; H->appliedValid = true;
    return LDSO_OK;
}

// ---- batched windows: B independent windows per launch ---------------------------------------------------------------------------
// One 7-keyframe window is tiny for an MI355X (SURVEY 7 hard part 1, 8e): a batch runs the Gauss-Newton iteration of several
// independent windows (several agents / sequences / hypotheses) with three launches per iteration for ALL of them: k_reduce_batch
// (every window's reduce workgroups) -> k_gn_solve_batch (two control workgroups per window) -> k_linearize_batch (every window's
// chunks).  Per-window arithmetic is exactly that of ldso_ba_enqueue_gn's split schedule; the windows only share the launches.
struct ldso_ba_batch {
    std::vector<ldso_ba *> h;
    BatchItem *d_items = nullptr;      // [n] numbered over the whole batch, then [n] numbered per half (see ldso_ba_batch_enqueue_gn)
    std::vector<BatchItem> items;
    BatchBlock *d_blocks = nullptr;    // [totalChunks] workgroups of the whole batch, then [halfChunks[0]] + [halfChunks[1]] per half
    std::vector<BatchBlock> blocks;
    size_t blocksCap = 0;
    int chunkPoints = 0;               // the chunking ldso_ba_batch_create gave its windows
    int totalChunks = 0, totalReduce = 0, FS = 0, cur = 0;
    int n0 = 0;                        // windows in the first half (= all of them for batches under 4 windows)
    int ks = LD_SCT_KS;                // K-splits per Schur tile of the batched reduction (ldso_ba_batch_create: 4 from 4 windows on, LDSO_BATCH_KS)
    int halfChunks[2] = {0, 0}, halfReduce[2] = {0, 0};
    // Balanced launches (round 6): workgroup w of a batched k_linearize works through the blocks [wgStart[w], wgStart[w + 1]) of its launch's table, cut by
    // ldso_ba_batch_create so that every workgroup carries the same load.  wg[0] = the whole batch, wg[1] / wg[2] = the halves; empty: one block per workgroup
    std::vector<int32_t> wg[3];
    int32_t *d_wg = nullptr; size_t wgCap = 0;
    std::vector<int32_t> wgHost;       // what d_wg holds (kept: the copy is asynchronous)
    int nWG[3] = {0, 0, 0}; size_t wgOff[3] = {0, 0, 0};
    bool balanced = false;
    hipStream_t aux = nullptr;         // second stream: the two halves run half an iteration apart
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evEnd = nullptr;
    BaDims Dmax;
};

static int batch_refresh(ldso_ba_batch *Bt) {
    ldso_ba *H0 = Bt->h[0];
    const size_t n = Bt->h.size();
    for (int pass = 0; pass < 2; pass++) {          // pass 0: blocks numbered over the whole batch; pass 1: per half
        int lin = 0, red = 0;
        for (size_t i = 0; i < n; i++) {
            ldso_ba *H = Bt->h[i];
            BatchItem &it = Bt->items[pass * n + i];
            if (pass == 1 && (int) i == Bt->n0) { Bt->halfChunks[0] = lin; Bt->halfReduce[0] = red; lin = 0; red = 0; }
            if (H->B.acc != H->ownAcc) H->B.acc = H->ownAcc;
            it.B = H->B; it.D = H->D; it.D.ks = Bt->ks; it.set[0] = H->sets[0]; it.set[1] = H->sets[1]; it.cs = H->chunkStarts;
            it.hasPrior = H->hasPrior ? 1 : 0; it.GSP = H->GSP; it.linBlock0 = lin; it.redBlock0 = red;
            const int nT = H->GSP / 16;
            lin += H->D.nChunks;
            red += H->D.F * H->D.F + Bt->ks * nT * (nT + 1) / 2 + 1;
        }
        if (pass == 0) { Bt->totalChunks = lin; Bt->totalReduce = red; }
        else if (Bt->n0 == (int) n) { Bt->halfChunks[0] = lin; Bt->halfReduce[0] = red; Bt->halfChunks[1] = 0; Bt->halfReduce[1] = 0; }
        else { Bt->halfChunks[1] = lin; Bt->halfReduce[1] = red; }
    }
    CHK(hipMemcpyAsync(Bt->d_items, Bt->items.data(), Bt->items.size() * sizeof(BatchItem), hipMemcpyHostToDevice, H0->stream));
    // the workgroup table: whole batch (window index into items[0..n)), then half A (index into items[n..n+n0)) and half B (items[n+n0..))
    Bt->blocks.clear();
    for (size_t i = 0; i < n; i++) for (const BatchBlock &b : Bt->h[i]->h_blocks) Bt->blocks.push_back(BatchBlock{(int32_t) i, b.p0, b.np, b.host_chunk});
    for (size_t i = 0; i < n; i++) { const int32_t w = (int) i < Bt->n0 ? (int32_t) i : (int32_t) i - Bt->n0; for (const BatchBlock &b : Bt->h[i]->h_blocks) Bt->blocks.push_back(BatchBlock{w, b.p0, b.np, b.host_chunk}); }
    if (Bt->blocks.size() > Bt->blocksCap) {
        CHK(hipStreamSynchronize(H0->stream));
        if (Bt->aux) CHK(hipStreamSynchronize(Bt->aux));
        if (Bt->d_blocks) hipFree(Bt->d_blocks);
    if (Bt->d_wg) hipFree(Bt->d_wg);
        Bt->d_blocks = nullptr; Bt->blocksCap = 0;
        void *q = nullptr;
        CHK(hipMalloc(&q, Bt->blocks.size() * sizeof(BatchBlock)));
        Bt->d_blocks = (BatchBlock *) q; Bt->blocksCap = Bt->blocks.size();
    }
    CHK(hipMemcpyAsync(Bt->d_blocks, Bt->blocks.data(), Bt->blocks.size() * sizeof(BatchBlock), hipMemcpyHostToDevice, H0->stream));
    if (Bt->balanced) {
        // the per-workgroup block ranges of the three launches (whole batch | half A | half B), valid while the windows keep the chunks ldso_ba_batch_create cut
        bool ok = (int) Bt->wg[0].size() >= 2 && Bt->wg[0].back() == Bt->totalChunks && Bt->wg[1].back() == Bt->halfChunks[0] && (Bt->halfChunks[1] == 0 || Bt->wg[2].back() == Bt->halfChunks[1]);
        REQ(ok, "ldso_ba_batch: the windows of the batch were re-chunked behind its back (ldso_ba_set_window / ldso_ba_set_chunk_points on a member): destroy and re-create the batch");
        std::vector<int32_t> all;
        for (int u = 0; u < 3; u++) { Bt->wgOff[u] = all.size(); Bt->nWG[u] = Bt->wg[u].empty() ? 0 : (int) Bt->wg[u].size() - 1; all.insert(all.end(), Bt->wg[u].begin(), Bt->wg[u].end()); }
        if (all.size() > Bt->wgCap) {
            CHK(hipStreamSynchronize(H0->stream));
            if (Bt->aux) CHK(hipStreamSynchronize(Bt->aux));
            if (Bt->d_wg) hipFree(Bt->d_wg);
            Bt->d_wg = nullptr; Bt->wgCap = 0;
            void *q = nullptr;
            CHK(hipMalloc(&q, all.size() * sizeof(int32_t)));
            Bt->d_wg = (int32_t *) q; Bt->wgCap = all.size();
        }
        Bt->wgHost.swap(all);
        CHK(hipMemcpyAsync(Bt->d_wg, Bt->wgHost.data(), Bt->wgHost.size() * sizeof(int32_t), hipMemcpyHostToDevice, H0->stream));
    }
    return LDSO_OK;
}

// Cut the windows [i0, i1) of a batch into chunks so that `nWG` workgroups, each working through a run of consecutive chunks, carry the same load.  A chunk
// (= one pass of linearize_body: operand staging, software-pipeline fill, block reduction) costs `c0` point-equivalents on top of its points, and never
// straddles a host frame.  The smallest per-workgroup budget that fits all points into nWG workgroups is found by bisection; cuts[i] receives the chunk ends
// of window i, wg the first chunk of every workgroup (nWG + 1 entries, counted over the windows [i0, i1) in order).
// The balancer proper, host logic without a device (C-ABI: ldso_ba_balance_chunks, tests/test_batch_balance_cpu.py): segments (runs of points that may share a chunk: one
// window's points of one host frame) in launch order -> chunk ends per segment (relative to the segment) and the first chunk of every workgroup.
struct BalSeg { int owner, p0, n; };
static long balance_segments(const std::vector<BalSeg> &segs, int nWG, int c0, std::vector<std::vector<int32_t>> *cutsByOwner, std::vector<int32_t> *wg, std::vector<int32_t> *flatEnds) {
    long total = 0;
    for (const BalSeg &sg : segs) total += sg.n;
    auto run = [&](long budget, bool emit) -> bool {
        size_t si = 0; int used = 0;          // points of segs[si] already handed out
        int blocks = 0;
        if (emit) { if (wg) wg->assign(1, 0); if (flatEnds) flatEnds->clear(); }
        for (int w = 0; w < nWG && si < segs.size(); w++) {
            long left = budget;
            while (si < segs.size()) {
                const int rem = segs[si].n - used;
                long can = left - c0;
                if (can < LD_WAVES && left != budget) break;          // not worth a chunk of its own here: the next workgroup takes it
                if (can < 1) can = 1;
                int take = (int) std::min<long>(rem, can);
                if (take < rem) { take = std::max(take / LD_WAVES * LD_WAVES, 1); if (rem - take < LD_WAVES) take = rem; }          // whole rounds of the workgroup's wavefronts, no crumbs left behind
                if (emit) {
                    if (cutsByOwner) (*cutsByOwner)[(size_t) segs[si].owner].push_back(segs[si].p0 + used + take);
                    if (flatEnds) flatEnds->push_back(segs[si].p0 + used + take);
                }
                blocks++; left -= c0 + take; used += take;
                if (used == segs[si].n) { si++; used = 0; }
                if (left <= 0) break;
            }
            if (emit && wg) wg->push_back(blocks);
        }
        if (emit && wg) while ((int) wg->size() < nWG + 1) wg->push_back(blocks);
        return si == segs.size();
    };
    long lo = std::max<long>(1, total / std::max(nWG, 1)), hi = total + (long) c0 * (long) segs.size() + 1;
    while (lo < hi) { const long mid = (lo + hi) / 2; if (run(mid, false)) hi = mid; else lo = mid + 1; }
    run(lo, true);
    return lo;
}
static void balance_batch(ldso_ba *const *handles, int i0, int i1, int nWG, int c0, std::vector<std::vector<int32_t>> &cuts, std::vector<int32_t> &wg) {
    std::vector<BalSeg> segs;
    for (int i = i0; i < i1; i++) {
        const ldso_ba *H = handles[i];
        cuts[i].clear();
        int p = 0;
        while (p < H->D.P) { int e = p; while (e < H->D.P && H->h_phost[e] == H->h_phost[p]) e++; segs.push_back(BalSeg{i, p, e - p}); p = e; }
    }
    balance_segments(segs, nWG, c0, &cuts, &wg, nullptr);
}
// host logic, no device: `n_seg` segments of seg_points[i] points each (in launch order; a chunk never spans two segments), `n_wg` workgroups, `chunk_cost` points of fixed
// cost per chunk -> chunk_end[] (cumulative over ALL points, ascending, the last one = the total), wg_first_chunk[n_wg + 1]; returns the number of chunks (< 0: error / cap too small)
int ldso_ba_balance_chunks(int n_seg, const int32_t *seg_points, int n_wg, int chunk_cost, int32_t *chunk_end, int cap, int32_t *wg_first_chunk, int64_t *budget_out) {
    REQ(n_seg >= 1 && seg_points && n_wg >= 1 && chunk_cost >= 0 && chunk_end && wg_first_chunk, "ldso_ba_balance_chunks: bad arguments");
    std::vector<BalSeg> segs;
    int p = 0;
    for (int i = 0; i < n_seg; i++) { REQ(seg_points[i] >= 1, "ldso_ba_balance_chunks: empty segment"); segs.push_back(BalSeg{0, p, seg_points[i]}); p += seg_points[i]; }
    std::vector<int32_t> wg, ends;
    const long budget = balance_segments(segs, n_wg, chunk_cost, nullptr, &wg, &ends);
    if ((int) ends.size() > cap) { ldso_set_error("ldso_ba_balance_chunks: chunk_end[] too small"); return LDSO_E_INVALID; }
    for (size_t i = 0; i < ends.size(); i++) chunk_end[i] = ends[i];
    for (int w = 0; w <= n_wg; w++) wg_first_chunk[w] = wg[(size_t) w];
    if (budget_out) *budget_out = budget;
    return (int) ends.size();
}

int ldso_ba_batch_create(ldso_ba_t *const *handles, int n, ldso_ba_batch_t **out) {
    REQ(handles && n >= 1 && out, "ldso_ba_batch_create: bad arguments");
    ldso_ba *H0 = handles[0];
    REQ(H0 && H0->D.P > 0, "ldso_ba_batch_create: window 0 is not set");
    for (int i = 0; i < n; i++) {
        ldso_ba *H = handles[i];
        REQ(H && H->D.P > 0, "ldso_ba_batch_create: every handle needs a resident window");
        REQ(H->device == H0->device && H->stream == H0->stream, "ldso_ba_batch_create: the handles of a batch share one device and one stream (ldso_ba_set_stream)");
        REQ(H->D.FS == H0->D.FS, "ldso_ba_batch_create: the windows of a batch use the same slot-table width (all F <= 8 or all 9 <= F <= 16)");
        REQ(H->inBatch == nullptr, "ldso_ba_batch_create: a handle belongs to at most one batch at a time");
        for (int k = 0; k < i; k++) REQ(handles[k] != H, "ldso_ba_batch_create: the same handle twice");
        REQ(!H->hasL, "ldso_ba_batch_create: windows with linearised residuals run on their own handle");
        REQ(H->D.pBegin == 0 && H->D.pEnd == H->D.P, "ldso_ba_batch_create: sharded handles cannot be batched");
        REQ(H->settings.forceAcceptStep && !H->pendingApply, "ldso_ba_batch_create: forced-accept schedule, no pending linearisation");
        REQ(memcmp(&H->settings, &H0->settings, sizeof(H0->settings)) == 0, "ldso_ba_batch_create: the batched kernels run with ONE ldso_settings_t: every handle of a batch must have been created with identical settings");
        // re-chunking re-forms a window's partial sums into its OTHER ping-pong set: the windows stay at one parity only if all of them are re-cut or none
        REQ(H->chunkPoints == H0->chunkPoints, "ldso_ba_batch_create: the handles of a batch share one chunking policy (ldso_ba_set_chunk_points: all automatic or all the same value)");
    }
    CHK(hipSetDevice(H0->device));
    // Chunking of a batch: the launch is filled by all windows together, so a workgroup takes several points per wavefront (its fixed
    // costs - operand staging, block reduction, ~4.5 us - are then a fraction of its life) while the grid still holds a few workgroups
    // per CU for balance.  Handles with an explicit ldso_ba_set_chunk_points keep theirs.
    int Bt_chunk = 0;
    bool balanced = false;
    std::vector<int32_t> wgTab[3];
    {
        long total = 0;
        for (int i = 0; i < n; i++) total += handles[i]->D.P;
        int ppw = (int) (total / ((long) H0->numCU * LD_WAVES));               // points per wavefront slot of the chip
        const char *e = getenv("LDSO_BATCH_PPW");                              // kernel experiments: the regular chunks of rounds 3-5 with this many points per wavefront
        bool every = true;
        for (int i = 0; i < n; i++) every = every && handles[i]->chunkPoints == 0 && handles[i]->chunkCuts.empty();
        if (ppw > 1 && every && !(e && *e)) {
            // Round 6: every workgroup of a launch gets the SAME load.  With regular chunks the batched launch ran as ceil(chunks / CUs) rounds of equal
            // workgroups - 1344 on 256 CUs: the last round a quarter full - and every chunk paid its fixed costs (staging, pipeline fill, block reduction:
            // about two points per wavefront) for six points per wavefront.  Now one workgroup per CU and launch works through a run of chunks cut to measure.
            int c0 = 2 * LD_WAVES;          // fixed cost of a chunk in points (two rounds of the workgroup's wavefronts)
            if (const char *ec = getenv("LDSO_BATCH_C0")) { if (*ec) c0 = std::max(0, atoi(ec)); }          // kernel experiments
            const int n0 = (n >= 4) ? n / 2 : n;
            // A half-batch launch does not take every CU: the other half's k_reduce_batch_dense / k_gn_solve_batch run beside it (two streams), and a workgroup that
            // owns its CU for the whole launch leaves them nothing to start on.  Measured (32 windows, MI355X): 128 / 192 / 208 / 224 / 240 / 256 workgroups per half
            // -> 141.6 / 159.2 / 162.2 / 168.3 / 167.7 / 149.5 k window-iterations/s (profiles/r06_batch_sweeps.log).
            int nWG = (n >= 4) ? std::max(1, H0->numCU * 7 / 8) : H0->numCU;
            if (const char *ew = getenv("LDSO_BATCH_NWG")) { if (*ew) nWG = std::max(1, atoi(ew)); }          // kernel experiments
            std::vector<std::vector<int32_t>> cuts((size_t) n);
            balance_batch(handles, 0, n0, nWG, c0, cuts, wgTab[1]);
            if (n0 < n) balance_batch(handles, n0, n, nWG, c0, cuts, wgTab[2]);
            // the whole-batch launch (ldso_ba_batch_time_linearize) runs the two halves' workgroups one after the other
            wgTab[0] = wgTab[1];
            if (n0 < n) for (size_t u = 1; u < wgTab[2].size(); u++) wgTab[0].push_back(wgTab[1].back() + wgTab[2][u]);
            for (int i = 0; i < n; i++) {
                handles[i]->chunkCuts = cuts[i];
                const int r_ = rechunk(handles[i]);
                if (r_ != LDSO_OK) { for (int k = 0; k <= i; k++) { handles[k]->chunkCuts.clear(); rechunk(handles[k]); } return r_; }      // leave nobody with the batch's chunks
            }
            long chunks = 0;
            for (int i = 0; i < n; i++) chunks += handles[i]->D.nChunks;
            Bt_chunk = (int) std::max<long>(1, (total + chunks / 2) / chunks);
            balanced = true;
        } else {
            if (e && *e) ppw = atoi(e);
            ppw = ppw < 1 ? 1 : ppw > 8 ? 8 : ppw;
            if (!(e && *e) && ppw > 6) ppw = 6;
            const int CH = ppw * LD_WAVES;
            Bt_chunk = ppw > 1 ? CH : 0;
            for (int i = 0; i < n; i++) if (handles[i]->chunkPoints == 0 && handles[i]->chunkCuts.empty() && ppw > 1) {      // ppw == 1: the single-window chunking already is the right one
                handles[i]->chunkPoints = CH;
                const int r_ = rechunk(handles[i]);
                handles[i]->chunkPoints = 0;                                         // the policy stays "automatic": the next ldso_ba_set_window re-chunks for a single window
                if (r_ != LDSO_OK) { for (int k = 0; k <= i; k++) rechunk(handles[k]); return r_; }      // leave nobody with the batch's chunks
            }
        }
    }
    ldso_ba_batch *Bt = new ldso_ba_batch();
    Bt->chunkPoints = Bt_chunk;
    Bt->balanced = balanced;
    for (int u = 0; u < 3; u++) Bt->wg[u] = wgTab[u];
    Bt->h.assign(handles, handles + n);
    Bt->items.resize(2 * (size_t) n);
    Bt->n0 = (n >= 4) ? n / 2 : n;
    // K-splits per Schur tile of the batched reduction: a lone window spreads every 16 x 16 tile of its Schur complement over LD_SCT_KS = 8 workgroups (latency); the
    // windows of a batch fill the chip anyway and halve the workgroups and the fp64 atomics (round 6, A/B on one box: 4 -> +3.3 % window-iterations/s at B = 32, 2 -> -11 %)
    Bt->ks = (n >= 4) ? 4 : LD_SCT_KS;
    if (const char *e = getenv("LDSO_BATCH_KS")) { const int v = atoi(e); if (v >= 1 && v <= 16) Bt->ks = v; }
    Bt->FS = H0->D.FS;
    Bt->Dmax = H0->D;
    for (int i = 0; i < n; i++) if (handles[i]->D.F > Bt->Dmax.F) Bt->Dmax = handles[i]->D;
    void *q = nullptr;
    for (int i = 0; i < n; i++) handles[i]->inBatch = Bt;
    // every failure from here on goes through ldso_ba_batch_destroy: it restores the single-window chunking and releases the handles
    if (hipMalloc(&q, 2 * (size_t) n * sizeof(BatchItem)) != hipSuccess) { (void) hipGetLastError(); ldso_ba_batch_destroy(Bt); ldso_set_error("ldso_ba_batch_create: hipMalloc failed"); return LDSO_E_HIP; }
    Bt->d_items = (BatchItem *) q;
    if (Bt->n0 < n) {
        if (hipStreamCreateWithFlags(&Bt->aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&Bt->ev0, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&Bt->ev1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&Bt->evEnd, hipEventDisableTiming) != hipSuccess) {
            ldso_ba_batch_destroy(Bt); ldso_set_error("ldso_ba_batch_create: stream / event creation failed"); return LDSO_E_HIP;
        }
    }
    *out = Bt;
    return LDSO_OK;
}

// points per workgroup ldso_ba_batch_create chose for the windows of this batch (0: it left their single-window chunking alone)
// K-splits per Schur tile the batch reduces its windows with (see ldso_ba_set_reduce_splits)
int ldso_ba_batch_reduce_splits(ldso_ba_batch_t *Bt, int *splits) {
    REQ(Bt && splits, "ldso_ba_batch_reduce_splits: bad arguments");
    *splits = Bt->ks;
    return LDSO_OK;
}

int ldso_ba_batch_chunk_points(ldso_ba_batch_t *Bt, int *points_per_workgroup) {
    REQ(Bt && points_per_workgroup, "ldso_ba_batch_chunk_points: null argument");
    *points_per_workgroup = Bt->chunkPoints;
    return LDSO_OK;
}

int ldso_ba_batch_destroy(ldso_ba_batch_t *Bt) {
    if (!Bt) return LDSO_OK;
    hipSetDevice(Bt->h[0]->device);
    hipStreamSynchronize(Bt->h[0]->stream);
    if (Bt->aux) { hipStreamSynchronize(Bt->aux); hipStreamDestroy(Bt->aux); }
    if (Bt->ev0) hipEventDestroy(Bt->ev0);
    if (Bt->ev1) hipEventDestroy(Bt->ev1);
    if (Bt->evEnd) hipEventDestroy(Bt->evEnd);
    if (Bt->d_items) hipFree(Bt->d_items);
    if (Bt->d_blocks) hipFree(Bt->d_blocks);
    if (Bt->d_wg) hipFree(Bt->d_wg);
    // back to the single-window chunking (handles that were re-chunked by ldso_ba_batch_create)
    if (Bt->balanced) { for (ldso_ba *H : Bt->h) { H->chunkCuts.clear(); if (H->D.P > 0) rechunk(H); } }
    else if (Bt->chunkPoints > 0) for (ldso_ba *H : Bt->h) if (H->chunkPoints == 0 && H->D.P > 0) rechunk(H);
    for (ldso_ba *H : Bt->h) if (H->inBatch == Bt) H->inBatch = nullptr;
    delete Bt;
    return LDSO_OK;
}

// `iters` forced Gauss-Newton iterations of every window of the batch: 3 launches per iteration for the whole batch, no host
// synchronisation.  Every window must hold an applied linearisation (ldso_ba_linearize_all + ldso_ba_apply_res, or a previous
// optimize / enqueue) and all of them must be at the same ping-pong parity (true after identical call sequences).
int ldso_ba_batch_enqueue_gn(ldso_ba_batch_t *Bt, int first_iteration, int iters) {
    REQ(Bt && iters >= 0, "ldso_ba_batch_enqueue_gn: bad arguments");
    ldso_ba *H0 = Bt->h[0];
    CHK(hipSetDevice(H0->device));
    for (ldso_ba *H : Bt->h) REQ(H->cur == H0->cur && !H->pendingApply, "ldso_ba_batch_enqueue_gn: the windows of a batch must be at the same stage");
    RUN(batch_refresh(Bt));
    double lam = 1e-1;
    if (H0->settings.solverMode & LDSO_SOLVER_USE_GN) lam = 0;
    if (H0->settings.solverMode & LDSO_SOLVER_FIX_LAMBDA) lam = 1e-5;
    const double l1 = 1 + lam, il = (double) (1.0f / (1 + lam));
    int cur = H0->cur;
    // The control step of a batch occupies two workgroups per window for ~30 us: the batch runs as two halves on two streams, the second
    // half an iteration behind the first, so that one half's reduce + control step overlap the other half's (chip-filling) linearisation.
    const int n = (int) Bt->h.size(), n0 = Bt->n0, n1 = n - n0;
    const BatchItem *itA = Bt->d_items + n, *itB = Bt->d_items + n + n0;
    if (n1 > 0 && iters > 0) { CHK(hipEventRecord(Bt->ev0, H0->stream)); CHK(hipStreamWaitEvent(Bt->aux, Bt->ev0, 0)); }
    for (int i = 0; i < iters; i++) {
        CHK(ba_launch_reduce_batch(itA, n0, Bt->halfReduce[0], cur, H0->settings.initialCalibHessian, l1, il, H0->stream));
        CHK(ba_launch_gn_solve_batch(itA, n0, Bt->Dmax, cur, H0->settings, first_iteration + i, 1e-1, H0->stream));
        if (n1 > 0 && i == 0) { CHK(hipEventRecord(Bt->ev1, H0->stream)); CHK(hipStreamWaitEvent(Bt->aux, Bt->ev1, 0)); }
        CHK(ba_launch_linearize_batch(itA, Bt->d_blocks + Bt->totalChunks, Bt->halfChunks[0], Bt->balanced ? Bt->d_wg + Bt->wgOff[1] : nullptr, Bt->nWG[1], Bt->FS, cur, H0->settings, 1, H0->settings.initialCalibHessian, H0->stream));
        if (n1 > 0) {
            CHK(ba_launch_reduce_batch(itB, n1, Bt->halfReduce[1], cur, H0->settings.initialCalibHessian, l1, il, Bt->aux));
            CHK(ba_launch_gn_solve_batch(itB, n1, Bt->Dmax, cur, H0->settings, first_iteration + i, 1e-1, Bt->aux));
            CHK(ba_launch_linearize_batch(itB, Bt->d_blocks + Bt->totalChunks + Bt->halfChunks[0], Bt->halfChunks[1], Bt->balanced ? Bt->d_wg + Bt->wgOff[2] : nullptr, Bt->nWG[2], Bt->FS, cur, H0->settings, 1, H0->settings.initialCalibHessian, Bt->aux));
        }
        cur ^= 1;
    }
    if (n1 > 0 && iters > 0) { CHK(hipEventRecord(Bt->evEnd, Bt->aux)); CHK(hipStreamWaitEvent(H0->stream, Bt->evEnd, 0)); }      // ldso_ba_sync(handle) covers both halves
    for (ldso_ba *H : Bt->h) H->cur = cur;
    return LDSO_OK;
}

// average duration of the batched k_linearize for bench.py's roofline: `reps` back-to-back launches on the applied state (read set ->
// scratch set, no point step: idempotent) between one pair of HIP events on the batch's stream
int ldso_ba_batch_time_linearize(ldso_ba_batch_t *Bt, int reps, double *avg_us) {
    REQ(Bt && reps > 0 && avg_us, "ldso_ba_batch_time_linearize: bad arguments");
    ldso_ba *H0 = Bt->h[0];
    CHK(hipSetDevice(H0->device));
    RUN(batch_refresh(Bt));
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(ba_launch_linearize_batch(Bt->d_items, Bt->d_blocks, Bt->totalChunks, Bt->balanced ? Bt->d_wg + Bt->wgOff[0] : nullptr, Bt->nWG[0], Bt->FS, H0->cur, H0->settings, 0, H0->settings.initialCalibHessian, H0->stream));
    CHK(hipEventRecord(a, H0->stream));
    for (int i = 0; i < reps; i++) CHK(ba_launch_linearize_batch(Bt->d_items, Bt->d_blocks, Bt->totalChunks, Bt->balanced ? Bt->d_wg + Bt->wgOff[0] : nullptr, Bt->nWG[0], Bt->FS, H0->cur, H0->settings, 0, H0->settings.initialCalibHessian, H0->stream));
    CHK(hipEventRecord(b, H0->stream));
    CHK(hipEventSynchronize(b));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    *avg_us = (double) ms * 1e3 / reps;
    return LDSO_OK;
}

// ---- the sharded iteration with the collective inside, for a C / C++ host (no torch): RCCL's ncclAllReduce on the handle's stream ----
typedef ncclResult_t (*allreduce_fn)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
static allreduce_fn find_allreduce() {
    static allreduce_fn fn = nullptr;
    if (fn) return fn;
    fn = (allreduce_fn) dlsym(RTLD_DEFAULT, "ncclAllReduce");          // an RCCL already in the process (the caller created `comm` with it)
    if (!fn) {
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (lib) fn = (allreduce_fn) dlsym(lib, "ncclAllReduce");
    }
    return fn;
}

int ldso_ba_enqueue_gn_rccl(ldso_ba_t *H, void *nccl_comm, int first_iteration, int iters) {
    REQ(H && nccl_comm && H->D.P > 0 && iters >= 0, "ldso_ba_enqueue_gn_rccl: bad arguments");
    CHK(hipSetDevice(H->device));
    allreduce_fn allreduce = find_allreduce();
    if (!allreduce) { ldso_set_error("ldso_ba_enqueue_gn_rccl: ncclAllReduce not found (librccl.so)"); return LDSO_E_UNSUPPORTED; }
    const size_t nd = ldso_ba_gn_reduce_doubles(H);
    if (!H->distBuf) { void *q = nullptr; CHK(hipMalloc(&q, ((size_t) (8 * H->maxF + 4) * (8 * H->maxF + 5) + 8 + H->maxP) * sizeof(double))); H->distBuf = (double *) q; }
    for (int i = 0; i < iters; i++) {
        RUN(ldso_ba_gn_reduce_local(H, H->distBuf, 1e-1));
        const ncclResult_t r = allreduce(H->distBuf, H->distBuf, nd, ncclDouble, ncclSum, (ncclComm_t) nccl_comm, H->stream);
        if (r != ncclSuccess) { ldso_set_error("ldso_ba_enqueue_gn_rccl: ncclAllReduce failed"); return LDSO_E_HIP; }
        RUN(ldso_ba_gn_solve_reduced(H, H->distBuf, first_iteration + i, 1e-1));
    }
    return LDSO_OK;
}


// ---- one-shot peer-write all-reduce --------------------------------------------------------------------------------
// The reduce buffer of a GN iteration is small (29 KB + 8 P bytes at F = 7): a ring all-reduce pays 2 (N - 1) latency-bound hops for it.
// Here every rank owns a RECEIVE WINDOW of 2 x N slots (two parities x one slot per source rank) that its peers can address (xGMI peer
// mapping / hipIpcOpenMemHandle).  Per iteration a rank (1) forms its partial (ldso_ba_gn_reduce_local), (2) k_p2p_push writes it into slot
// `rank` of EVERY rank's window as self-validating 64-bit words (32 payload bits | 32-bit exchange number - the hand-over of the cooperative
// tracker: a word is valid by itself, so no fence has to order data before a flag across the fabric), (3) k_p2p_sum polls the N slots of
// its own window and adds them in RANK ORDER (deterministic, unlike a ring whose order depends on the chunk), (4) the replicated solve.
// One fabric traversal per direction.  Parity: a rank can run at most one exchange ahead of a peer (it needs that peer's partial of the
// exchange to finish it), so two slots per source suffice.  The polls are bounded (2 s): a missing peer turns into LDSO_E_HIP at
// ldso_ba_p2p_check instead of a hung stream.
struct P2PWindows { unsigned long long *w[16]; };
// `cap` = slot stride in doubles = the CAPACITY of the handles (maxF / maxP: ldso_ba_p2p_window_bytes), not the current window's size: the
// parity regions then stay where they are when ldso_ba_set_window changes the window dimension between two exchanges (a rank that has
// moved on to the next window must not write over words a slower peer has not summed yet)
__global__ __launch_bounds__(256) void k_p2p_push(const double *__restrict__ src, int nd, size_t cap, P2PWindows W, int rank, int nRanks, int parity, unsigned seq) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += gridDim.x * blockDim.x) {
        const unsigned long long u = __builtin_bit_cast(unsigned long long, src[i]);
        const unsigned long long w0 = (u << 32) | seq, w1 = (u & 0xFFFFFFFF00000000ull) | seq;
        const size_t o = (((size_t) parity * nRanks + rank) * cap + i) * 2;
        for (int q = 0; q < nRanks; q++) {
            __hip_atomic_store(W.w[q] + o, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(W.w[q] + o + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(256) void k_p2p_sum(const unsigned long long *__restrict__ own, int nd, size_t cap, int nRanks, int parity, unsigned seq, double *__restrict__ out, int *err) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int q = 0; q < nRanks; q++) {
            const unsigned long long *p = own + (((size_t) parity * nRanks + q) * cap + i) * 2;
            unsigned long long w0, w1;
            unsigned spins = 0; long long t0 = 0; bool dead = false;
            for (;;) {
                w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned) w0 == seq && (unsigned) w1 == seq) break;
                if ((++spins & 255u) == 0) { const long long now = wall_clock64(); if (t0 == 0) t0 = now; else if (now - t0 > 200000000ll) { dead = true; break; } }
            }
            if (dead) { *err = 1; acc = __builtin_nan(""); break; }
            acc += __builtin_bit_cast(double, (w1 & 0xFFFFFFFF00000000ull) | (w0 >> 32));
        }
        out[i] = acc;
    }
}

static size_t p2p_slot_doubles(const ldso_ba *H) {          // one slot: the largest reduce buffer the handle can produce
    const size_t n = 8 * (size_t) H->maxF + 4;
    return n * n + n + 8 + (size_t) H->maxP;
}
size_t ldso_ba_p2p_window_bytes(ldso_ba_t *H, int n_ranks) {
    if (!H || n_ranks < 1 || n_ranks > 16) return 0;
    return 2 * (size_t) n_ranks * p2p_slot_doubles(H) * 16;
}
// This rank's receive window: uncached device memory (remote writes must be seen by a polling kernel), zeroed; ipc_handle_out (64 bytes,
// hipIpcMemHandle_t) lets another process map it with ldso_ba_p2p_window_open.  Ranks of ONE process pass the pointer itself.
int ldso_ba_p2p_window_alloc(ldso_ba_t *H, int n_ranks, void **window_out, void *ipc_handle_out) {
    REQ(H && window_out && n_ranks >= 1 && n_ranks <= 16, "ldso_ba_p2p_window_alloc: bad arguments (1..16 ranks)");
    CHK(hipSetDevice(H->device));
    void *p = nullptr;
    const size_t bytes = ldso_ba_p2p_window_bytes(H, n_ranks);
    // no fallback to cached memory: a polling k_p2p_sum may never see peer stores that sit in another L2, and every exchange would end in the
    // 2 s timeout without a hint of the cause
    {
        const hipError_t e_ = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
        if (e_ != hipSuccess) { (void) hipGetLastError(); ldso_set_error(std::string("ldso_ba_p2p_window_alloc: uncached device memory unavailable (hipExtMallocWithFlags: ") + hipGetErrorString(e_) + ")"); return LDSO_E_UNSUPPORTED; }
    }
    CHK(hipMemset(p, 0, bytes));
    CHK(hipStreamSynchronize(nullptr));          // (asynchronous zero-fill, see dalloc)
    if (ipc_handle_out) {
        hipIpcMemHandle_t hnd;
        const hipError_t e_ = hipIpcGetMemHandle(&hnd, p);
        if (e_ != hipSuccess) { hipFree(p); ldso_set_error(std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e_)); return LDSO_E_HIP; }
        memcpy(ipc_handle_out, &hnd, sizeof(hnd));
    }
    *window_out = p;
    return LDSO_OK;
}
int ldso_ba_p2p_window_open(ldso_ba_t *H, const void *ipc_handle, void **window_out) {
    REQ(H && ipc_handle && window_out, "ldso_ba_p2p_window_open: null argument");
    CHK(hipSetDevice(H->device));
    hipIpcMemHandle_t hnd; memcpy(&hnd, ipc_handle, sizeof(hnd));
    CHK(hipIpcOpenMemHandle(window_out, hnd, hipIpcMemLazyEnablePeerAccess));
    return LDSO_OK;
}
int ldso_ba_p2p_window_close(ldso_ba_t *H, void *window, int opened_from_handle) {
    REQ(H && window, "ldso_ba_p2p_window_close: null argument");
    CHK(hipSetDevice(H->device));
    if (opened_from_handle) CHK(hipIpcCloseMemHandle(window)); else CHK(hipFree(window));
    return LDSO_OK;
}
// `iters` forced Gauss-Newton iterations of this rank's shard with the one-shot exchange above instead of ncclAllReduce (same contract as
// ldso_ba_enqueue_gn_rccl: every rank calls it with the same iteration arguments; windows[q] = rank q's receive window as THIS process
// addresses it, windows[rank] = the own one).  n_ranks == 1 degenerates to the single-GPU iteration through the same kernels.
int ldso_ba_enqueue_gn_p2p(ldso_ba_t *H, int rank, int n_ranks, void *const *windows, int first_iteration, int iters) {
    REQ(H && windows && n_ranks >= 1 && n_ranks <= 16 && rank >= 0 && rank < n_ranks && H->D.P > 0 && iters >= 0, "ldso_ba_enqueue_gn_p2p: bad arguments");
    for (int q = 0; q < n_ranks; q++) REQ(windows[q] != nullptr, "ldso_ba_enqueue_gn_p2p: a window pointer is null");
    CHK(hipSetDevice(H->device));
    const size_t nd = ldso_ba_gn_reduce_doubles(H);
    if (!H->distBuf) { void *q = nullptr; CHK(hipMalloc(&q, ((size_t) (8 * H->maxF + 4) * (8 * H->maxF + 5) + 8 + H->maxP) * sizeof(double))); H->distBuf = (double *) q; }
    if (!H->d_p2pErr) { void *q = nullptr; CHK(hipMalloc(&q, sizeof(int))); H->d_p2pErr = (int *) q; CHK(hipMemsetAsync(H->d_p2pErr, 0, sizeof(int), H->stream)); }
    P2PWindows W;
    for (int q = 0; q < 16; q++) W.w[q] = (unsigned long long *) (q < n_ranks ? windows[q] : nullptr);
    const int grid = (int) ((nd + 255) / 256);
    // everything that may synchronise the stream from the host (the handle's descriptor follows the accumulator pointer: refresh_item) happens
    // BEFORE the first polling kernel is enqueued - a single host thread driving several ranks of one process must not wait for a kernel
    // that polls for a peer it has not launched yet
    if (H->B.acc != H->distBuf) {
        H->B.acc = H->distBuf;
        GnInit gi; gi.enable = (H->D.pBegin > 0) ? 2 : 1; gi.hasPrior = H->hasPrior ? 1 : 0; gi.calibPrior = H->settings.initialCalibHessian; gi.itCheck = -1;
        CHK(ba_launch_acc_init(H->B, H->D, gi, H->stream));
    }
    RUN(refresh_item(H));
    for (int i = 0; i < iters; i++) {
        RUN(ldso_ba_gn_reduce_local(H, H->distBuf, 1e-1));
        const unsigned seq = ++H->p2pSeq;
        if (seq == 0xFFFFFFFFu) { ldso_set_error("ldso_ba_enqueue_gn_p2p: exchange counter exhausted (re-create the windows)"); return LDSO_E_INVALID; }
        hipLaunchKernelGGL(k_p2p_push, dim3(grid), dim3(256), 0, H->stream, (const double *) H->distBuf, (int) nd, p2p_slot_doubles(H), W, rank, n_ranks, (int) (seq & 1), seq);
        hipLaunchKernelGGL(k_p2p_sum, dim3(grid), dim3(256), 0, H->stream, (const unsigned long long *) windows[rank], (int) nd, p2p_slot_doubles(H), n_ranks, (int) (seq & 1), seq, H->distBuf, H->d_p2pErr);
        CHK(hipGetLastError());
        RUN(ldso_ba_gn_solve_reduced(H, H->distBuf, first_iteration + i, 1e-1));
    }
    return LDSO_OK;
}
// after ldso_ba_sync: LDSO_E_HIP if a peer's words did not arrive within the poll limit of some exchange since the last check
int ldso_ba_p2p_check(ldso_ba_t *H) {
    REQ(H, "null handle");
    if (!H->d_p2pErr) return LDSO_OK;
    CHK(hipSetDevice(H->device));
    int e = 0;
    CHK(hipMemcpyAsync(&e, H->d_p2pErr, sizeof(int), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    if (e) { CHK(hipMemsetAsync(H->d_p2pErr, 0, sizeof(int), H->stream)); ldso_set_error("ldso_ba_enqueue_gn_p2p: a peer's partial did not arrive within 2 s (peer not running / window not mapped)"); return LDSO_E_HIP; }
    return LDSO_OK;
}

// FullSystem::optimizeImmaturePoint (FullSystem.cc:892-1010) for n immature points against the key frames of the window that is
// resident in the handle (ldso_ba_set_image*, ldso_ba_set_window, ldso_ba_set_frames: images, calibration, current poses).
int ldso_ba_activate_points(ldso_ba_t *H, int n, const ldso_immature_t *pts, int min_obs, float min_idepth_hessian, int gn_iterations, ldso_activation_t *out) {
    REQ(H && n >= 0 && (n == 0 || (pts && out)) && gn_iterations >= 0, "ldso_ba_activate_points: bad arguments");
    REQ(H->D.F >= 2, "ldso_ba_activate_points: set the window and the frames first");
    if (n == 0) return LDSO_OK;
    CHK(hipSetDevice(H->device));
    for (int f = 0; f < H->D.F; f++) REQ(H->B.img[f] != nullptr, "ldso_ba_activate_points: a key-frame image is missing");
    if (n > H->actCap) {
        if (H->d_act) hipFree(H->d_act);
        H->d_act = nullptr; H->actCap = 0;
        CHK(hipMalloc(&H->d_act, (size_t) n * (sizeof(ldso_immature_t) + sizeof(ldso_activation_t))));
        H->actCap = n;
    }
    ldso_immature_t *dp = (ldso_immature_t *) H->d_act;
    ldso_activation_t *dout = (ldso_activation_t *) ((char *) H->d_act + (size_t) H->actCap * sizeof(ldso_immature_t));
    CHK(hipMemcpyAsync(dp, pts, (size_t) n * sizeof(ldso_immature_t), hipMemcpyHostToDevice, H->stream));
    CHK(ba_launch_activate(H->B, H->D, H->settings, dp, dout, n, min_obs, min_idepth_hessian, gn_iterations, H->stream));
    CHK(hipMemcpyAsync(out, dout, (size_t) n * sizeof(ldso_activation_t), hipMemcpyDeviceToHost, H->stream));
    CHK(hipStreamSynchronize(H->stream));
    return LDSO_OK;
}

int ldso_ba_reduce_local(ldso_ba_t *H, void *buf) {
    REQ(H && buf && H->D.P > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    const ResSet &S = H->sets[H->cur];
    RUN(launch_reduce(H, S));
    RUN(launch_solve(H, S, SK_POST | SK_EXPORT, 0, 0, -1, (double *) buf, nullptr));
    RUN(launch_gather(H, S, 0.0, 1, (double *) buf));
    return LDSO_OK;
}

int ldso_ba_solve_reduced(ldso_ba_t *H, const void *buf, int iteration, double lambda, int do_step) {
    REQ(H && buf && H->D.P > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    const ResSet &S = H->sets[H->cur];
    unsigned fl = SK_FROMREDUCED | SK_THRESH | SK_SOLVE;
    if (do_step) fl |= SK_BACKUP | SK_STEP | SK_PRECALC;
    RUN(launch_gather(H, S, lambda, 2, (double *) buf));
    RUN(launch_solve(H, S, fl, iteration, lambda, -1, nullptr, (const double *) buf));
    RUN(launch_pstep(H, S, do_step ? (PS_RESUB | PS_BACKUP | PS_STEP) : PS_RESUB));
    if (do_step) { RUN(launch_linearize(H, false)); H->cur ^= 1; H->appliedValid = true; }
    return LDSO_OK;
}

// ---------------------------------------------------------------------------------------------------------
// fetchers
// ---------------------------------------------------------------------------------------------------------
#define D2H(vec, src, n) do { int r_ = d2h(H, (vec), (src), (n)); if (r_ != LDSO_OK) return r_; } while (0)
// The three result getters share their halves: enqueue the copies into the pinned download arena (fetch_*_put), ONE stream synchronisation, unpack (fetch_*_unpack).
// ldso_ba_get_results does all three behind a single synchronisation (the drop-in's write-back: two round trips of ~20 us fewer per optimize()).
struct FetchRes { const SlotRec *rn, *rc; };
struct FetchPts { const PtGeo *geo; const PtRec *pt; const PtAcc *acc; };
struct FetchFrm { const DevFrame *df; const DevCalib *dc; };
static size_t fetch_res_bytes(const ldso_ba *H) { return 2 * ((size_t) H->D.P * H->D.FS * sizeof(SlotRec) + 64); }
static size_t fetch_pts_bytes(const ldso_ba *H) { return (size_t) H->D.P * (sizeof(PtGeo) + sizeof(PtRec) + sizeof(PtAcc)) + 3 * 64; }
static size_t fetch_frm_bytes(const ldso_ba *H) { return (size_t) H->D.F * sizeof(DevFrame) + sizeof(DevCalib) + 2 * 64; }
static FetchRes fetch_res_put(ldso_ba *H, size_t &used, int &rc) {
    const size_t PS = (size_t) H->D.P * H->D.FS;
    // "new" values come from the set written by the last linearize, applied values from the current set
    const ResSet &Sn = H->pendingApply ? H->sets[H->cur ^ 1] : H->sets[H->cur];
    const ResSet &Sc = H->sets[H->cur];
    FetchRes f;
    f.rn = down_put(H, used, Sn.slot, PS, rc);
    f.rc = (&Sn != &Sc) ? down_put(H, used, Sc.slot, PS, rc) : f.rn;
    return f;
}
static void fetch_res_unpack(const ldso_ba *H, const FetchRes &f, ldso_res_out_t *out, int32_t *state_state, int32_t *is_active, int32_t *to_remove) {
    for (int i = 0; i < H->R; i++) {
        size_t s = H->flat2slot[i];
        const SlotRec &n_ = f.rn[s], &c_ = f.rc[s];
        if (out) {
            ldso_res_out_t &o = out[i];
            o.state_NewEnergy = n_.e[LD_SM_ENERGY].m.f; o.state_NewEnergyWithOutlier = n_.e[LD_SM_EWO].m.f; o.state_NewState = n_.e[LD_SM_STATE].m.i;
            for (int k = 0; k < 3; k++) o.centerProjectedTo[k] = n_.e[LD_SM_CEN0 + k].m.f;
            for (int k = 0; k < 8; k++) o.JpJdF[k] = n_.e[k].jp;
        }
        if (state_state) state_state[i] = c_.e[LD_SM_STATE].m.i;
        if (is_active) is_active[i] = c_.e[LD_SM_ACTIVE].m.i;
        if (to_remove) to_remove[i] = n_.e[LD_SM_REMOVE].m.i;
    }
}
static FetchPts fetch_pts_put(ldso_ba *H, size_t &used, int &rc) {
    const size_t P = H->D.P;
    const ResSet &S = H->sets[H->cur];
    FetchPts f;
    f.geo = down_put(H, used, H->B.pgeo, P, rc);
    f.pt = down_put(H, used, S.pt, P, rc);
    f.acc = down_put(H, used, S.acc, P, rc);
    return f;
}
static void fetch_pts_unpack(const ldso_ba *H, const FetchPts &f, ldso_point_out_t *out) {
    const PtGeo *geo = f.geo; const PtRec *pt = f.pt; const PtAcc *acc = f.acc;
    for (size_t i = 0; i < (size_t) H->D.P; i++) {
        ldso_point_out_t &o = out[i];
        o.step = geo[i].step; o.HdiF = geo[i].lastHdiF; o.bdSumF = geo[i].lastBdSumF; o.idepth_hessian = geo[i].lastIdH; o.Hdd_accAF = acc[i].HddA; o.bd_accAF = acc[i].bdA;
        o.Hdd_accLF = acc[i].HddL; o.bd_accLF = acc[i].bdL;
        for (int k = 0; k < 4; k++) { o.Hcd_accAF[k] = pt[i].HcdA[k]; o.Hcd_accLF[k] = pt[i].HcdL[k]; }
        o.idepth = geo[i].idepth; o.maxRelBaseline = pt[i].maxRelBS; o.numGoodResiduals = pt[i].numGood;
    }
}
static FetchFrm fetch_frm_put(ldso_ba *H, size_t &used, int &rc) {
    FetchFrm f;
    f.df = down_put(H, used, H->B.frames, (size_t) H->D.F, rc);
    f.dc = down_put(H, used, H->B.calib, (size_t) 1, rc);
    return f;
}
static void fetch_frm_unpack(const ldso_ba *H, const FetchFrm &f, ldso_frame_t *fr, double *step, double *cv, double *cs, double *pre) {
    const DevFrame *df = f.df;
    const DevCalib &dc = *f.dc;
    for (int q = 0; q < H->D.F; q++) {
        if (fr) {
            ldso_frame_t &o = fr[q];
            memcpy(o.worldToCam_evalPT, df[q].evalPT, sizeof(o.worldToCam_evalPT));
            memcpy(o.state, df[q].state, sizeof(o.state)); memcpy(o.state_zero, df[q].state_zero, sizeof(o.state_zero));
            memcpy(o.prior, df[q].prior, sizeof(o.prior));
            memcpy(o.nullspaces_pose, df[q].ns_pose, sizeof(o.nullspaces_pose)); memcpy(o.nullspaces_scale, df[q].ns_scale, sizeof(o.nullspaces_scale));
            memcpy(o.nullspaces_affine, df[q].ns_affine, sizeof(o.nullspaces_affine));
            o.ab_exposure = df[q].ab_exposure; o.frameEnergyTH = df[q].frameEnergyTH; o.frameID = df[q].frameID; o.pad_ = 0;
        }
        if (step) memcpy(step + q * 10, df[q].step, 10 * sizeof(double));
        if (pre) memcpy(pre + q * 12, df[q].PRE_w2c, 12 * sizeof(double));
    }
    if (cv) memcpy(cv, dc.value, 4 * sizeof(double));
    if (cs) memcpy(cs, dc.step, 4 * sizeof(double));
}

int ldso_ba_get_residuals(ldso_ba_t *H, ldso_res_out_t *out, int32_t *state_state, int32_t *is_active, int32_t *to_remove) {
    REQ(H && H->D.P > 0, "no window");
    CHK(hipSetDevice(H->device));
    RUN(down_reserve(H, fetch_res_bytes(H)));
    size_t used = 0; int rcD = LDSO_OK;
    const FetchRes f = fetch_res_put(H, used, rcD);
    if (rcD != LDSO_OK) return rcD;
    CHK(hipStreamSynchronize(H->stream));
    fetch_res_unpack(H, f, out, state_state, is_active, to_remove);
    return LDSO_OK;
}

int ldso_ba_get_points(ldso_ba_t *H, ldso_point_out_t *out) {
    REQ(H && out && H->D.P > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    RUN(down_reserve(H, fetch_pts_bytes(H)));
    size_t used = 0; int rcD = LDSO_OK;
    const FetchPts f = fetch_pts_put(H, used, rcD);
    if (rcD != LDSO_OK) return rcD;
    CHK(hipStreamSynchronize(H->stream));
    fetch_pts_unpack(H, f, out);
    return LDSO_OK;
}

int ldso_ba_get_frames(ldso_ba_t *H, ldso_frame_t *fr, double *step, double *cv, double *cs, double *pre) {
    REQ(H && H->D.F > 0, "no window");
    CHK(hipSetDevice(H->device));
    RUN(down_reserve(H, fetch_frm_bytes(H)));
    size_t used = 0; int rcD = LDSO_OK;
    const FetchFrm f = fetch_frm_put(H, used, rcD);
    if (rcD != LDSO_OK) return rcD;
    CHK(hipStreamSynchronize(H->stream));
    fetch_frm_unpack(H, f, fr, step, cv, cs, pre);
    return LDSO_OK;
}

int ldso_ba_get_results(ldso_ba_t *H, ldso_res_out_t *res, int32_t *state_state, int32_t *is_active, int32_t *to_remove, ldso_point_out_t *points,
                        ldso_frame_t *frames, double *step, double *calib_value, double *calib_step) {
    REQ(H && H->D.P > 0 && H->D.F > 0 && points, "ldso_ba_get_results: bad arguments");
    CHK(hipSetDevice(H->device));
    RUN(down_reserve(H, fetch_res_bytes(H) + fetch_pts_bytes(H) + fetch_frm_bytes(H)));
    size_t used = 0; int rcD = LDSO_OK;
    // the small blocks first: their unpacking could start while the residual records are still on their way (it does not: one synchronisation keeps the call simple)
    const FetchFrm ff = fetch_frm_put(H, used, rcD);
    const FetchPts fp = fetch_pts_put(H, used, rcD);
    const FetchRes fr_ = fetch_res_put(H, used, rcD);
    if (rcD != LDSO_OK) return rcD;
    CHK(hipStreamSynchronize(H->stream));
    fetch_frm_unpack(H, ff, frames, step, calib_value, calib_step, nullptr);
    fetch_pts_unpack(H, fp, points);
    fetch_res_unpack(H, fr_, res, state_state, is_active, to_remove);
    return LDSO_OK;
}

int ldso_ba_get_system(ldso_ba_t *H, double *HA, double *bA, double *HL, double *bL, double *Hsc, double *bsc, double *HF, double *bF, double *x) {
    REQ(H && H->D.n > 0, "no window");
    CHK(hipSetDevice(H->device));
    const size_t n = H->D.n, blk = n * n + n;
    std::vector<double> sys, xx;
    D2H(sys, H->B.sys, 4 * blk); D2H(xx, H->B.x, n);
    CHK(hipStreamSynchronize(H->stream));
    double *Hs[4] = {HA, HL, Hsc, HF}, *bs[4] = {bA, bL, bsc, bF};
    for (int m = 0; m < 4; m++) {
        if (Hs[m]) memcpy(Hs[m], sys.data() + m * blk, n * n * 8);
        if (bs[m]) memcpy(bs[m], sys.data() + m * blk + n * n, n * 8);
    }
    if (x) memcpy(x, xx.data(), n * 8);
    return LDSO_OK;
}

int ldso_ba_set_debug_split_launch(ldso_ba_t *H, int enable) {
    REQ(H, "null handle");
    H->noFusedLaunch = enable != 0;
    return LDSO_OK;
}

int ldso_ba_set_debug_dump(ldso_ba_t *H, int enable) {
    REQ(H, "null handle");
    H->B.dumpJ = enable ? H->d_dumpJ : nullptr;
    return LDSO_OK;
}

int ldso_ba_get_jacobians(ldso_ba_t *H, const int32_t *ids, int n, ldso_rawjac_t *out) {
    REQ(H && ids && out && n >= 0 && H->D.P > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    std::vector<ldso_rawjac_t> all;
    if (H->B.dumpJ == nullptr) {
        // recompute on demand: run the linearize kernel at the current state with the J dump enabled, into the
        // spare set; neither the applied set nor the frame-energy thresholds are touched.
        REQ(!H->pendingApply, "ldso_ba_get_jacobians: a linearisation is pending (call ldso_ba_apply_res first, or enable ldso_ba_set_debug_dump)");
        BaPtrs Bd = H->B;
        Bd.dumpJ = H->d_dumpJ;
        CHK(ba_launch_linearize(Bd, H->D, H->sets[H->cur], H->sets[H->cur ^ 1], H->settings, H->hasL, false, 0, GnInit{0, 0, 0.0f, -1}, H->stream));
    }
    D2H(all, H->d_dumpJ, (size_t) H->R);
    CHK(hipStreamSynchronize(H->stream));
    for (int i = 0; i < n; i++) { REQ(ids[i] >= 0 && ids[i] < H->R, "residual id out of range"); out[i] = all[ids[i]]; }
    return LDSO_OK;
}

int ldso_ba_get_precalc(ldso_ba_t *H, float *out) {
    REQ(H && out && H->D.F > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    const int F = H->D.F;
    std::vector<DevPair> dp;
    D2H(dp, H->B.pairs, (size_t) F * F);
    CHK(hipStreamSynchronize(H->stream));
    for (int i = 0; i < F * F; i++) {
        float *o = out + i * 27;
        memcpy(o, dp[i].KRKi, 36); memcpy(o + 9, dp[i].Kt, 12); memcpy(o + 12, dp[i].R0, 36); memcpy(o + 21, dp[i].t0, 12);
        o[24] = dp[i].aff[0]; o[25] = dp[i].aff[1]; o[26] = dp[i].b0;
    }
    return LDSO_OK;
}

// [h*F+t][14]: PRE_RTll 9, PRE_tTll 3 (current state), PRE_aff_mode 2 - what point activation reads (debug / test fetch)
int ldso_ba_get_pair_rt(ldso_ba_t *H, float *out) {
    REQ(H && out && H->D.F > 0, "bad arguments");
    CHK(hipSetDevice(H->device));
    const int F = H->D.F;
    std::vector<DevPair> dp;
    std::vector<float> rt;
    D2H(dp, H->B.pairs, (size_t) F * F);
    D2H(rt, H->B.pairRt, (size_t) F * F * 12);
    CHK(hipStreamSynchronize(H->stream));
    for (int i = 0; i < F * F; i++) { memcpy(out + i * 14, rt.data() + i * 12, 48); out[i * 14 + 12] = dp[i].aff[0]; out[i * 14 + 13] = dp[i].aff[1]; }
    return LDSO_OK;
}

int ldso_ba_get_counts(ldso_ba_t *H, int *a, int *l) {
    REQ(H, "null handle");
    double sc[16];
    RUN(read_scalars(H, sc));
    if (a) *a = (int) sc[9];
    if (l) *l = (int) sc[10];
    return LDSO_OK;
}

int ldso_ba_get_energy_log(ldso_ba_t *H, double *out, int cap) {
    REQ(H && out, "bad arguments");
    std::vector<double> e;
    D2H(e, H->B.energyLog, (size_t) 64);
    CHK(hipStreamSynchronize(H->stream));
    int n = H->lastIterations + 2;
    if (cap >= 64) { for (int i = 0; i < 64; i++) out[i] = e[i]; return n; }
    for (int i = 0; i < std::min(n, cap); i++) out[i] = e[i];
    return n;
}

}  // extern "C"
