// ba_linearize.hip — fused per-point kernel of the windowed photometric BA (gfx950, wave64).
//
// Replaces, for every active point of the window, the reference's
//   PointFrameResidual::linearize        src/internal/Residuals.cc:13-214
//   PointFrameResidual::applyRes/takeData include/internal/Residuals.h:70-87,123-128
//   AccumulatedTopHessianSSE::addPoint<0|1>  src/internal/OptimizationBackend/AccumulatedTopHessian.cc:8-118
//   AccumulatedSCHessianSSE::addPoint (per-point part)  .../AccumulatedSCHessian.cc:9-31
// Mapping: one wavefront per point; lane = slot*8 + k where slot = target frame (8 per pass) and
// k = pattern pixel.  All residuals of a point share its host, and points arrive host-major
// (EnergyFunctional::allPoints order), so a block's register accumulators belong to fixed
// (host, target=slot) pairs: the 13x13 relative Hessian block of a pair is accumulated as the outer
// product of each pixel row, in registers, with no atomics; a block writes ONE partial per slot.
// The Schur complement is kept in lifted form: per point the row g_p = [Ad^T JpJdF | Hcd] is stored
// (G matrix) and reduced later as G diag(HdiF) G^T by ba_reduce.hip.
//
// Roofline: HBM/gather bound — per residual 8 px x 4 taps x 12 B of the target image.
// Arithmetic is IEEE (compile with -ffp-contract=off): every elementwise expression below follows
// the reference's operation order so that energies and residual states are bit-identical to the
// CPU path; fused multiply-adds are used only (explicitly) in the accumulators.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ba_dev.h"

#ifndef LD_PAIR
#define LD_PAIR 1          // windows of 9..12 key frames: two points share the pass of the second slot group (k_linearize_one<2, true>)
#endif
#ifndef LD_MFMA_SUMS
#define LD_MFMA_SUMS 0
#endif
#ifndef LD_OPAQUE_K
#define LD_OPAQUE_K 0          // experiment (round 6): masks derived from the lane index recomputed per point instead of kept in (spilled) SGPR pairs - 37 fewer vector instructions in the kernel and SLOWER (B = 32: 173 against 160 us, C5 44.2 against 43.1, one box): not used
#endif
#ifndef LD_LDG_PLAIN
#define LD_LDG_PLAIN 0
#endif
#ifndef LD_STASH
#define LD_STASH 1          // the records of the point in work wait in LDS (see STASH in linearize_body)
#endif
#ifndef LD_STASH_LM
#define LD_STASH_LM 1          // ... and its uniform fields are read from there (ds_read_b32) instead of by v_readlane / DPP broadcasts
#endif
#ifndef LD_PEEL
#define LD_PEEL 1
#endif
#ifndef LD_PIPE_ARGS
#define LD_PIPE_ARGS 1          // the argument-based kernels (fix / linearised / marginalisation / dump passes) up to 8 key frames run the software pipeline too
#endif
#define RES_IN 0
#define RES_OOB 1
#define RES_OUTLIER 2

// value of lane (i-1) within the 16-lane row; callers zero it for k==0
__device__ __forceinline__ float dpp_row_shr1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor1(float x) {   // quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float x) {   // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor3(float x) {   // quad_perm [3,2,1,0]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x1B, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_half_mirror(float x) {   // lane i <-> 7-i within each 8-lane half row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
}
// sum over the 8 lanes of a slot group, result in all 8 lanes (tree order)
__device__ __forceinline__ float sum8(float x) {
    x += dpp_quad_xor1(x);
    x += dpp_quad_xor2(x);
    x += dpp_half_mirror(x);
    return x;
}
// lane i receives lane i-J of its 16-lane row (0 when that leaves the row)
template <int J> __device__ __forceinline__ float dpp_row_shr(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + J, 0xF, 0xF, true));
}
// lane j (0..3) of each quad broadcast to its quad
template <int J> __device__ __forceinline__ float dpp_quad_bcast(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), J * 0x55, 0xF, 0xF, true));
}
// element J and element 4 + J of every 8-lane group, in all 8 lanes of the group (two DPP moves and two selects instead of two ds_bpermute)
template <int J> __device__ __forceinline__ void group_bcast_pair(float x, int k, float &lo, float &hi) {
    const float a = dpp_quad_bcast<J>(x);           // lanes 0-3: x[J], lanes 4-7: x[4 + J]
    const float b = dpp_half_mirror(a);             // lanes 0-3: x[4 + J], lanes 4-7: x[J]
    lo = (k < 4) ? a : b; hi = (k < 4) ? b : a;
}

// lane-indexed choice among wave-uniform-per-slot values (k = pattern lane 0..7) as a binary tree over the bits of k: 7 (6: 5, 4: 3) v_cndmask with THREE lane masks.
// Written with array elements in a chain of ternaries the choice became control flow (clang evaluates an array element of a conditional operator under a
// branch; the chain of k == c branches is folded into a switch and lowered, for a divergent k, as a tree of exec-mask branches: ~14 vector + ~25 scalar
// instructions per choice, round 6 ISA count) - scalar arguments of a function are selected.
static __device__ __forceinline__ float sel2(const bool c, const float a, const float b) { return c ? a : b; }
static __device__ __forceinline__ float selk8(const int k, const float a0, const float a1, const float a2, const float a3, const float a4, const float a5, const float a6, const float a7) {
    const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0, b2 = (k & 4) != 0;
    const float l0 = sel2(b0, a1, a0), l1 = sel2(b0, a3, a2), l2 = sel2(b0, a5, a4), l3 = sel2(b0, a7, a6);
    const float m0 = sel2(b1, l1, l0), m1 = sel2(b1, l3, l2);
    return sel2(b2, m1, m0);
}
static __device__ __forceinline__ float selk4(const int k, const float a0, const float a1, const float a2, const float a3) {
    const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0;
    return sel2(b1, sel2(b0, a3, a2), sel2(b0, a1, a0));
}
// k in 0..5 (lanes 6, 7 receive a4 / a5: their callers overwrite the result)
static __device__ __forceinline__ float selk6(const int k, const float a0, const float a1, const float a2, const float a3, const float a4, const float a5) {
    const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0, b2 = (k & 4) != 0;
    const float l0 = sel2(b0, a1, a0), l1 = sel2(b0, a3, a2), l2 = sel2(b0, a5, a4);
    const float m0 = sel2(b1, l1, l0);
    return sel2(b2, l2, m0);
}

// sum over the 8 lanes in the reference's sequential order ((((x0+x1)+x2)+...)+x7), result in all 8 lanes.
// The running sum lives in lane 7 of each group (lanes 7 and 15 of a row): step j adds x_j fetched with row_shr:(7-j) - one
// v_add_f32_dpp per step whose DPP operand (x) is old, so no hazard padding; other lanes compute values nobody reads.
__device__ __forceinline__ float seq8(float x, int k, int lane) {
    float t = dpp_row_shr<7>(x);
    t = t + dpp_row_shr<6>(x);
    t = t + dpp_row_shr<5>(x);
    t = t + dpp_row_shr<4>(x);
    t = t + dpp_row_shr<3>(x);
    t = t + dpp_row_shr<2>(x);
    t = t + dpp_row_shr<1>(x);
    t = t + x;
    (void) lane;
    float lo, hi;
    group_bcast_pair<3>(t, k, lo, hi);          // hi = element 7 of the group
    return hi;
}

// sum over the 8 slots of a wave (lanes with equal k), result in every lane: slot pairs by row_ror:8, then the four rows by two ds_bpermute
// butterflies (tree order; the sequential-order sums that decide residual states use seq8).  gfx950's v_permlane16_swap / v_permlane32_swap
// butterflies were measured against this in rounds 3 and 4: bit-identical, 2 % slower in every configuration - removed.
__device__ __forceinline__ float sum_slots(float x, int a16, int a32) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));
#if LD_MFMA_SUMS
    // experiment (round 6): the sum over the four 16-lane rows on the fp32 matrix core instead of two ds_bpermute butterflies - v_mfma_f32_16x16x4_f32 with A = 1:
    // D[i][j] = sum_k B[k][j], B[k][j] = the value of lane 16 k + j, every lane l reads column j = l & 15 back (all four result registers hold the same sum)
    (void) a16; (void) a32;
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ z = {0.0f, 0.0f, 0.0f, 0.0f};
    const v4f_ d = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, x, z, 0, 0, 0);
    return d[0];
#endif
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a16, __builtin_bit_cast(int, x)));
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a32, __builtin_bit_cast(int, x)));
    return x;
}
// sum over the 4 slots of a half-wave (lanes with equal k inside lanes 0..31 / 32..63), result in every lane of the half: the first two steps of sum_slots
__device__ __forceinline__ float sum_half(float x, int a16) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a16, __builtin_bit_cast(int, x)));
    return x;
}
// ---- just-in-time pointer groups ------------------------------------------------------------------------------------------------
// The kernel touches ~55 device arrays per point.  Held in scalar registers for the whole kernel (what the compiler does with a
// descriptor it may load once) they do not fit the 102 SGPRs of a wave: 105 of them were spilled into VGPR lanes, and every use paid
// two v_readlane plus a 64-bit VGPR address (v_lshl_add_u64, flat_load / flat_store) - 260 of the 1234 vector instructions of the point
// loop.  Where the descriptor lives in device memory (k_linearize_batch; DESC = true) a group of eight pointers (64 consecutive bytes of
// BaPtrs / ResSet, see ba_dev.h) is fetched with ONE s_load_dwordx16 right where it is used (scalar cache hit, no vector instruction)
// and is dead a few instructions later; `asm volatile` keeps the compiler from hoisting the load back to the top of the kernel.
typedef int v16i_t __attribute__((ext_vector_type(16)));
template <bool DESC, int OFF> static __device__ __forceinline__ v16i_t ldg16(const void *base) {
    v16i_t t;
#if LD_LDG_PLAIN
    if (DESC) return *(const __attribute__((address_space(4))) v16i_t *) ((unsigned long long) base + OFF);          // experiment: the compiler's own scalar load, free to be hoisted / merged
#endif
    if (DESC) asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(base), "n"(OFF) : "memory");
    else __builtin_memcpy(&t, (const char *) base + OFF, 64);          // kernel arguments: the compiler's own scalar loads
    return t;
}
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <bool DESC, int OFF> static __device__ __forceinline__ v4i_t ldg4(const void *base) {          // two pointers
    v4i_t t;
    if (DESC) asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(base), "n"(OFF) : "memory");
    else __builtin_memcpy(&t, (const char *) base + OFF, 16);
    return t;
}
// two groups with one wait
template <bool DESC, int OFFA, int OFFB> static __device__ __forceinline__ void ldg16x2(v16i_t &a, const void *baseA, v16i_t &b, const void *baseB) {
    if (DESC) asm volatile("s_load_dwordx16 %0, %2, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(baseA), "s"(baseB), "n"(OFFA), "n"(OFFB) : "memory");
    else { __builtin_memcpy(&a, (const char *) baseA + OFFA, 64); __builtin_memcpy(&b, (const char *) baseB + OFFB, 64); }
}
#define GP(T, t16, i) ((T *) ((((unsigned long long) (unsigned) (t16)[2 * (i) + 1]) << 32) | (unsigned long long) (unsigned) (t16)[2 * (i)]))
#define OFF_B0 ((int) offsetof(BaPtrs, pgeo))
enum { BP_GEO = 0, BP_CW = 1, BP_RTAB = 2, BP_PHOST = 3, BP_JLIN = 4, BP_RTZ = 5 };      // indices of the BaPtrs pointers inside their group
#define OFF_S0 ((int) offsetof(ResSet, slot))
static_assert(offsetof(BaPtrs, chunk_n) == offsetof(BaPtrs, pgeo) + 56, "BaPtrs pointer group (ba_dev.h)");
static_assert(offsetof(ResSet, slot) == 0 && offsetof(ResSet, chunkEnergy) == 56 && offsetof(ResSet, chunkCnt) == 64, "ResSet pointer group (ba_dev.h)");
// indices of the ResSet pointers inside their group
enum { RS_SLOT = 0, RS_PT = 1, RS_ACC = 2, RS_CAND = 3, RS_G = 4, RS_TOPA = 5, RS_TOPL = 6, RS_CHUNKE = 7 };

// floats of LDS behind sXa: sE [LD_WAVES] doubles | sC [4 LD_WAVES] ints | sN [LD_WAVES] floats, rounded to 16 bytes; the image pointers follow
#define LD_TAIL_FLOATS ((2 * LD_WAVES + 4 * LD_WAVES + LD_WAVES + 3) & ~3)
// flat index of entry (r,c), r<=c, in the packed upper triangle of a 13x13 matrix
__host__ __device__ constexpr int tri13(int r, int c) { return r * 13 - (r * (r - 1)) / 2 + (c - r); }

// Everything a wave reads from HBM for one point besides the image taps.  Round 6: the records of a point are loaded TWO points ahead of the
// arithmetic that consumes them and its image taps ONE point ahead (software pipeline of linearize_body), so nothing of a point may sit in scalar
// registers while it waits (an outstanding s_load makes every LDS wait of the point in front of it a full lgkmcnt(0)): the 64-byte PtGeo and PtRec
// of the point arrive as ONE VGPR each (lane L holds dword L & 15 - the 16 lanes of a row read one 64-byte line, the four rows the same line) and
// are read out with v_readlane where they are used.
template <int NSG>
struct PtIn {
    float rgeo, rrec;             // dword (lane & 15) of the point's PtGeo / of its PtRec in the applied set
    float color, wgt;
    int rflat[NSG], rlin[NSG], rnew[NSG], rlidx[NSG];     // SlotTab of this lane's slot(s)
    float jp[NSG], m[NSG];        // this lane's pair of the slot record: JpJdF[k] and scalar k (LD_SM_*; integers as raw bits)
    const float *ls;              // where the point's records are parked in the wavefront's LDS (STASH, linearize_body): [64 lanes][rgeo, rrec, colour, weight] | [64 lanes][rflat, rlin, jp, m]
};
// dword i of a record held one-dword-per-lane (wave-uniform result: a scalar register)
#define RLF(v, i) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (v)), (i)))
#define RLI(v, i) __builtin_amdgcn_readlane(__builtin_bit_cast(int, (v)), (i))
// PtGeo dwords (ba_dev.h)
#define GEO_U 0
#define GEO_V 1
#define GEO_PRIOR 2
#define GEO_IDP 4
#define GEO_IDZ 5
#define GEO_STEP 6
// PtRec dwords (ba_dev.h)
#define REC_HDI 0
#define REC_BDSUM 1
#define REC_IDH 2
#define REC_NACT 3
#define REC_HCDA 4
#define REC_HCDL 8
#define REC_MAXRELBS 12
#define REC_NUMGOOD 13

// element i of a device array with the BYTE offset computed in 32 bits: base pointer (SGPR pair) + zero-extended
// lane offset is the addressing mode the hardware has (saddr + voffset); a 64-bit per-lane address costs two registers and a
// 64-bit shift-add per access.  Every table of a window is far below 4 GB.
// The pointers of a window are generic (flat) pointers to the compiler: accessed as such they become flat_load / flat_store, which have no
// scalar-base addressing mode (two v_mov + v_lshl_add_u64 per access to build a 64-bit VGPR address) and count against lgkmcnt as well as
// vmcnt (every s_waitcnt lgkmcnt(0) of a scalar load then waits for all vector memory traffic in flight).  Every table of a window is
// hipMalloc'ed device memory: the access is made through an address_space(1) (global) pointer -> global_load/store v, v_off, s[base].
template <class T> using gptr_t = __attribute__((address_space(1))) T *;
template <class T> static __device__ __forceinline__ __attribute__((address_space(1))) T &at_g(T *p, unsigned i) { return *(gptr_t<T>) ((gptr_t<char>) p + (size_t) (i * (unsigned) sizeof(T))); }
#define AT(ptr, i) (at_g((ptr), (unsigned) (i)))

typedef int v4i32_t __attribute__((ext_vector_type(4)));
typedef float v8f_t __attribute__((ext_vector_type(8)));
typedef int v2i32_t __attribute__((ext_vector_type(2)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

// The record of one point as a wavefront needs it: one dword of its PtGeo and of its PtRec per lane, this lane's (colour, weight) pair (one
// dwordx2), and per slot group one dwordx2 / dwordx4 (SlotTab, the same 16 bytes for the 8 lanes of a slot) and one dwordx2 (this lane's pair of
// the 64-byte SlotRec): 3 + 2 NSG vector loads, no scalar load.
template <int NSG, bool HAS_L, bool FIX, bool DESC>
static __device__ __forceinline__ void load_point(PtIn<NSG> &q, const BaPtrs &B, const ResSet &cur, unsigned FS, unsigned p, unsigned s, unsigned k, unsigned lane) {
    const unsigned pU = (unsigned) __builtin_amdgcn_readfirstlane((int) p);
    {
        const v16i_t b0 = ldg16<DESC, OFF_B0>(&B);
        const float *geo = GP(const float, b0, BP_GEO);
        const v2f_t *pcw = GP(const v2f_t, b0, BP_CW);
        const v4i32_t *rtab = GP(const v4i32_t, b0, BP_RTAB);
        // addresses = (uniform base advanced to the point, on the scalar side) + (a lane offset that never changes)
        q.rgeo = AT(geo + (size_t) pU * (sizeof(PtGeo) / 4), lane & 15u);
        const v2f_t cw = AT(pcw + (size_t) pU * 8, k);
        q.color = cw.x; q.wgt = cw.y;
#pragma unroll
        for (int g = 0; g < NSG; g++) {
            // slot tables are dense [P][FS]: every index is readable.  Only the half of the entry this variant uses is loaded
            if constexpr (FIX || HAS_L) {
                const v4i32_t t4 = AT(rtab + (size_t) pU * FS, g * 8 + s);
                q.rflat[g] = t4.x; q.rlin[g] = t4.y; q.rnew[g] = FIX ? t4.z : 0; q.rlidx[g] = HAS_L ? t4.w : 0;
            } else {
                const v2i32_t t2 = AT((const v2i32_t *) (rtab + (size_t) pU * FS), (g * 8 + s) * 2);
                q.rflat[g] = t2.x; q.rlin[g] = t2.y; q.rnew[g] = 0; q.rlidx[g] = 0;
            }
        }
    }
    {
        const v16i_t s0 = ldg16<DESC, OFF_S0>(&cur);
        const v2f_t *slots = GP(const v2f_t, s0, RS_SLOT);
        const float *pts = GP(const float, s0, RS_PT);
        q.rrec = AT(pts + (size_t) pU * (sizeof(PtRec) / 4), lane & 15u);
#pragma unroll
        for (int g = 0; g < NSG; g++) {
            const v2f_t e = AT(slots + (size_t) pU * FS * 8, (g * 8 + s) * 8 + k);
            q.jp[g] = e.x; q.m[g] = e.y;
        }
    }
}

// Pair mode (round 6, windows of 9..12 key frames): the second slot group of such a window uses 4 of its 8 slots, so TWO points share its pass - lanes 0..31 work on the
// targets 8..11 of point A, lanes 32..63 on the targets 8..11 of point B (3 passes per 2 points instead of 4).  The records of a pair: PtGeo / PtRec with A in lanes
// 0..31 and B in lanes 32..63 (dword lane & 15), both points' colours / weights, the first group's slot records of A and of B, the shared second group's.
struct PairIn {
    float rgeo, rrec, colA, wgtA, colB, wgtB;
    int rfA0, rfB0, rf1;
    float jpA0, mA0, jpB0, mB0, jp1, m1;
};
template <bool DESC>
static __device__ __forceinline__ void load_pair(PairIn &q, const BaPtrs &B, const ResSet &cur, unsigned FS, unsigned pA, unsigned dAB, unsigned s, unsigned k, unsigned lane) {
    const unsigned pU = (unsigned) __builtin_amdgcn_readfirstlane((int) pA), dU = (unsigned) __builtin_amdgcn_readfirstlane((int) dAB);
    const unsigned hoff = (lane & 32u) ? dU : 0u;          // this lane's point, relative to A
    const unsigned s1 = 8u + (s & 3u);                     // this lane's slot of the shared second group
    {
        const v16i_t b0 = ldg16<DESC, OFF_B0>(&B);
        const float *geo = GP(const float, b0, BP_GEO);
        const v2f_t *pcw = GP(const v2f_t, b0, BP_CW);
        const v2i32_t *rtab = (const v2i32_t *) GP(const v4i32_t, b0, BP_RTAB);          // two dwordx2 per 16-byte entry: {rflat, rlin} first
        q.rgeo = AT(geo + (size_t) pU * (sizeof(PtGeo) / 4), hoff * (unsigned) (sizeof(PtGeo) / 4) + (lane & 15u));
        const v2f_t ca = AT(pcw + (size_t) pU * 8, k), cb = AT(pcw + (size_t) pU * 8, dU * 8u + k);
        q.colA = ca.x; q.wgtA = ca.y; q.colB = cb.x; q.wgtB = cb.y;
        q.rfA0 = AT(rtab + (size_t) pU * FS * 2, s * 2u).x;
        q.rfB0 = AT(rtab + (size_t) pU * FS * 2, (dU * FS + s) * 2u).x;
        q.rf1 = AT(rtab + (size_t) pU * FS * 2, (hoff * FS + s1) * 2u).x;
    }
    {
        const v16i_t s0 = ldg16<DESC, OFF_S0>(&cur);
        const v2f_t *slots = GP(const v2f_t, s0, RS_SLOT);
        const float *pts = GP(const float, s0, RS_PT);
        q.rrec = AT(pts + (size_t) pU * (sizeof(PtRec) / 4), hoff * (unsigned) (sizeof(PtRec) / 4) + (lane & 15u));
        const v2f_t a0 = AT(slots + (size_t) pU * FS * 8, s * 8u + k), b0_ = AT(slots + (size_t) pU * FS * 8, (dU * FS + s) * 8u + k);
        const v2f_t p1 = AT(slots + (size_t) pU * FS * 8, (hoff * FS + s1) * 8u + k);
        q.jpA0 = a0.x; q.mA0 = a0.y; q.jpB0 = b0_.x; q.mB0 = b0_.y; q.jp1 = p1.x; q.m1 = p1.y;
    }
}

// What the front half of a slot group (pattern projection -> tap loads) hands to its back half (everything behind the image taps), one group
// later: the 12 tap dwords (in flight) and the projected pixel.  PtStep: the inverse depth of the point after the fused point step.
struct TapsG { float t[12]; float Ku, Kv; };
// the bilinear sample of the target image at the projected pixel (I, dI/dx, dI/dy) - what is left of the taps once they have arrived
struct Hit { float h0, h1, h2, Ku, Kv; };
struct PtStep { float idp, idz; };

// stepMode != 0 fuses the point part of resubstituteF_MT + backupState + doStepFromBackup (EnergyFunctional.cc:518-547,
// FullSystem.cc:1585-1602, 1625-1673) in front of the linearisation of each point (what k_point_step does with
// PS_RESUB|PS_BACKUP|PS_STEP), so a forced-accept GN iteration needs no separate point pass.
// MARG = true: the accumulate of EnergyFunctional::marginalizePointsF (EnergyFunctional.cc:165-222) for the points flagged in
// margFlags, fused with what FullSystem::flagPointsForRemoval did to them just before (FullSystem.cc:1241-1250): every residual
// of a flagged point is reset (resetOOB), re-linearised at the current state, applied, and - if active - fixed
// (fixLinearizationF, Residuals.cc:216-242: res_toZeroF = resF - [JIdx (Jp delta) + JabF delta_ab]); the top / Schur accumulators
// then take addPoint<2> (resApprox = res_toZeroF) with priorF * idepthFixPriorMargFac and no prior shift.  Unflagged points
// contribute nothing.  The output set is scratch (never applied).
// The body is shared by k_linearize (one window: everything arrives as kernel arguments, i.e. in scalar registers) and
// k_linearize_batch (many independent windows per launch: the descriptors live in device memory).  chunk = index of the
// workgroup's chunk inside ITS window, gridBlocks = workgroups of that window (partition of the accumulator initialisation).
template <int NSG, bool HAS_L, bool FIX, bool MARG, bool DESC, bool ONE = false, bool PAIR = false>
static __device__ __forceinline__ void linearize_body(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, const int stepMode,
                                                      const GnInit &gi, const int32_t *__restrict__ margFlags, const int chunk, const int gridBlocks,
                                                      const int p0, const int np, const int h) {
    if (LD_ITER_SKIPPED(B, gi.itCheck)) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int FS = D.FS, F = D.F;
    // (opaque to the compiler: inside the batched kernel's loop over the blocks of a workgroup everything derived from the lane index is loop-invariant, and hoisted
    // out of that loop it stays live through the point loop - 24 VGPRs, measured: 184 -> 208, and with them the CU sharing with the other half-batch's reduce kernel)
    int tid_ = (int) threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, wave = tid >> 6, lane = tid & 63, s = lane >> 3, k = lane & 7, k_ = k, lane_ = lane;
    (void) k_; (void) lane_;
    const long long t0_ = wall_clock64();
#define LSTAMP(i) do { if (LD_STAMP_ON && chunk == 0 && tid == 0) B.energyLog[8 + (i)] = (double) (wall_clock64() - t0_); } while (0)

    // ---- LDS carve: pair structs of this host, float adjoints of this host, reduction scratch --------
    DevPair *sPair = (DevPair *) smem;                                  // [FS]
    float *sAd = smem + FS * (sizeof(DevPair) / 4);                     // [FS][8][8][2]: the float adjoints of (host, target) pairs, see the staging below
    float *sRed = sAd + 2 * FS * 64;                                       // [LD_WAVES][FS][91] (+ topL [FS][91] when HAS_L)
    float *sTopL = sRed + LD_WAVES * FS * LD_TOPN;                        // [LD_WAVES][FS][91] when HAS_L: each (wave, slot) cell has ONE writer lane
    float *sXa = sTopL + (HAS_L ? LD_WAVES * FS * LD_TOPN : 0);                      // [FS][8] xAd of this host (stepMode)
    // [FS] level-0 image of every target frame: the per-lane pointer comes from LDS (64 cycles) instead of a dependent global load of the
    // descriptor's img[] table in front of every tap gather
    const float **sImg = (const float **) (sXa + FS * 8 + LD_TAIL_FLOATS);      // behind sE | sC | sN of the block reduction (see below)
    // [LD_WAVES][2][2][64] x 16 bytes (one slot group per point, descriptor-based kernels): the records of the point in work wait here, written by the wavefront one point
    // ahead (the software pipeline below) - each wavefront its own 4 KB, no barrier
    float *sRec = (float *) (sImg + FS);

    if (gi.enable) {
        // initialise the HFinal / bFinal accumulator of the k_reduce that follows (lower triangle) with H_M and the diagonal
        // priors (EnergyFunctional.cc:257-291; the lambda scaling of these terms is added by k_reduce); b starts at zero.
        const int n = D.n, N1 = n * n + n, per = (N1 + gridBlocks - 1) / gridBlocks;
        const int z0 = chunk * per, z1 = min(N1, z0 + per);
        for (int e = z0 + tid; e < z1; e += blockDim.x) {
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

    // ---- software pipeline over the points of this wavefront (round 6) ---------------------------------------------------------------
    // point i: back half (arithmetic behind the taps) | point i + 1: front half (taps in flight) | point i + 2: records in flight.
    const int waveU = __builtin_amdgcn_readfirstlane(wave);
    PtIn<NSG> qa = {}, qb = {}, qc = {};
    int pi = waveU;
    if (!PAIR && pi < np) load_point<NSG, HAS_L, FIX, DESC>(qa, B, cur, FS, p0 + pi, s, k, lane);

    // ---- staging: all global loads first (one latency level), then the LDS stores --------------------------------
    {
        constexpr int PW = (int) (sizeof(DevPair) / 4), TH_OFF = (int) (offsetof(DevPair, thMax) / 4);
        constexpr int NPB = (LD_MAXF * PW + 64 * LD_WAVES - 1) / (64 * LD_WAVES), NAB = (LD_MAXF * 64) / (64 * LD_WAVES);
        const float thHost = B.frames[h].frameEnergyTH;
        float pv[NPB], ahv[NAB], atv[NAB], xav = 0.0f;
#pragma unroll
        for (int u = 0; u < NPB; u++) {
            const int i = tid + u * 64 * LD_WAVES, t = i / PW, o = i % PW;
            float v = 0.0f;
            if (i < FS * PW && t < F) {
                // the energy threshold of a pair is max(host, target) of the frames' CURRENT thresholds (Residuals.cc:191)
                v = (o == TH_OFF) ? fmaxf(thHost, B.frames[t].frameEnergyTH) : ((const float *) &B.pairs[h * F + t])[o];
            }
            pv[u] = v;
        }
#pragma unroll
        for (int u = 0; u < NAB; u++) {
            const int i = tid + u * 64 * LD_WAVES, t = i >> 6, o = i & 63;
            const bool in = (i < FS * 64) && (t < F);
            ahv[u] = in ? B.adHostF[(h + t * F) * 64 + o] : 0.0f;
            atv[u] = in ? B.adTargetF[(h + t * F) * 64 + o] : 0.0f;
        }
        if ((stepMode & 1) && tid < FS * 8) xav = ((tid >> 3) < F) ? B.xAd[(size_t) (h * F + (tid >> 3)) * 8 + (tid & 7)] : 0.0f;
#pragma unroll
        for (int u = 0; u < NPB; u++) { const int i = tid + u * 64 * LD_WAVES; if (i < FS * PW) ((float *) sPair)[i] = pv[u]; }
#pragma unroll
        for (int u = 0; u < NAB; u++) {
            // element (kk, j) of a pair's 8 x 8 adjoints at [t][kk][j ^ kk][target, host]: lane kk of a slot reads ITS row as four ds_read_b128 and meets the entry of
            // column kk ^ r next to the r-th xor-permuted copy of the slot's JpJdF (the lifted Schur row below)
            const int i = tid + u * 64 * LD_WAVES;
            if (i < FS * 64) { const int kk = (i >> 3) & 7, e = (i & ~7) | ((i ^ kk) & 7); sAd[2 * e] = atv[u]; sAd[2 * e + 1] = ahv[u]; }
        }
        if ((stepMode & 1) && tid < FS * 8) sXa[tid] = xav;
        if (tid < FS) sImg[tid] = B.img[(tid < F) ? tid : 0];          // slots behind F: a readable image (their taps are loaded and never used)
    }
    if (HAS_L) for (int i = tid; i < LD_WAVES * FS * LD_TOPN; i += blockDim.x) sTopL[i] = 0.0f;
    __syncthreads();
    LSTAMP(1);

    const int ox = (k == 1 || k == 6) ? -1 : (k == 2) ? 1 : (k == 3) ? -2 : (k == 5) ? 2 : 0;     // staticPattern[8], Setting.cc:221
    const int oy = (k == 0) ? -2 : (k == 1 || k == 2) ? -1 : (k == 6) ? 1 : (k == 7) ? 2 : 0;
    const int W = D.w;
    // loop invariants of the descriptor as values: a store through a pointer the compiler cannot tell apart would otherwise force re-loads
    const float wM3G = D.wM3G, hM3G = D.hM3G;
    const unsigned GSu = (unsigned) D.GS;
    ldso_rawjac_t *const dumpJ = B.dumpJ;
    const int a16 = (lane ^ 16) << 2, a32 = (lane ^ 32) << 2;      // ds_bpermute addresses of sum_slots

    // top accumulators (13x13 symmetric block per slot), distributed over the 8 pattern lanes of the slot: lane k owns row k
    // (13 columns) and, for k < 5, row k + 8 (columns 8..12) - 18 registers per lane and slot group instead of 91, updated from
    // the per-residual 2x2 sums exactly as AccumulatorApprox::update / updateTopRight / updateBotRight do (MatrixAccumulators.h:893-1045)
    float accR[NSG][18];
#pragma unroll
    for (int g = 0; g < NSG; g++)
#pragma unroll
        for (int i = 0; i < 18; i++) accR[g][i] = 0.0f;
    double energySum = 0.0;     // sum of linearize() return values (slot leaders only)
    int nresA = 0, nresL = 0;
    float nidSum = 0.0f;
    int nidCnt = 0;

    // ================= FRONT half, per point: the fused point step ===============================================================
    // Dword i of the point's PtGeo / PtRec and the uniform scalars of a slot record (the m of lane j of its 8-lane group).  From registers: v_readlane (PtGeo / PtRec
    // arrive one dword per lane) and a DPP broadcast + select.  LM = true (the pipelined loop of the one-slot-group descriptor kernels, whose records are parked in LDS
    // anyway): ONE ds_read_b32 each, every lane the same address (or its slot's) - the loop is bound by vector-instruction issue and the LDS port is idle
    // (37 v_readlane + ~25 DPP moves / selects per point, round 6)
    using LM0 = std::false_type;
    auto qgeo = [&](auto lc, const PtIn<NSG> &q, const int i) -> float { if constexpr (decltype(lc)::value) return q.ls[i * 4]; else return RLF(q.rgeo, i); };
    auto qrec = [&](auto lc, const PtIn<NSG> &q, const int i) -> float { if constexpr (decltype(lc)::value) return q.ls[i * 4 + 1]; else return RLF(q.rrec, i); };
    auto qreci = [&](auto lc, const PtIn<NSG> &q, const int i) -> int { const float f_ = qrec(lc, q, i); return __builtin_bit_cast(int, f_); };
    auto qmlane = [&](auto lc, auto jc, const PtIn<NSG> &q, const int g) -> float {          // m of lane j (3: energy, 5: state, 6: activity) of this lane's slot
        constexpr int j = decltype(jc)::value;
        if constexpr (decltype(lc)::value) return q.ls[256 + (s * 8 + j) * 4 + 3];
        else { float lo, hi; group_bcast_pair<(j & 3)>(q.m[g], k, lo, hi); return (j < 4) ? lo : hi; }
    };
    using J3 = std::integral_constant<int, 3>; using J5 = std::integral_constant<int, 5>; using J6 = std::integral_constant<int, 6>;
    auto pstepT = [&](auto lc, const unsigned p, const PtIn<NSG> &q) -> PtStep {
        float idp = qgeo(lc, q, GEO_IDP), idz = qgeo(lc, q, GEO_IDZ);
        if (stepMode & 1) {
            // ---- resubstituteFPt for this point, then backupState + doStepFromBackup (stepfacD = 1) ------------
            const float rHdi = qrec(lc, q, REC_HDI), rBd = qrec(lc, q, REC_BDSUM), rIdH = qrec(lc, q, REC_IDH);
            float step = 0.0f;
            if (__builtin_amdgcn_readfirstlane(qreci(lc, q, REC_NACT)) > 0) {
                float b = rBd;
                float dot = 0;
                dot += xc0 * (qrec(lc, q, REC_HCDA + 0) + qrec(lc, q, REC_HCDL + 0)); dot += xc1 * (qrec(lc, q, REC_HCDA + 1) + qrec(lc, q, REC_HCDL + 1));
                dot += xc2 * (qrec(lc, q, REC_HCDA + 2) + qrec(lc, q, REC_HCDL + 2)); dot += xc3 * (qrec(lc, q, REC_HCDA + 3) + qrec(lc, q, REC_HCDL + 3));
                b -= dot;
#pragma unroll
                for (int g = 0; g < NSG; g++) {
                    const int t = g * 8 + s;
                    // the activity flag of the slot record sits in the m of lane 6 of its 8-lane group
                    const float h2 = qmlane(lc, J6{}, q, g);
                    const bool act = (t < F) && (q.rflat[g] >= 0) && (__builtin_bit_cast(int, h2) != 0);
                    float sres = seq8(sXa[t * 8 + k] * q.jp[g], k, lane);
                    sres = act ? sres : 0.0f;
                    b -= sum_slots(sres, a16, a32);
                }
                if (isfinite(b)) step = -b * rHdi; else { step = qgeo(lc, q, GEO_STEP); if (lane == 0) *(gptr_t<double>) (unsigned long long) (B.scalars + 4) = 1.0; }
            }
            const float ni = idp + 1.0f * step;
            {
                // dwords 4..11 of the point's PtGeo: lane 1 {idepth, idepth_zero, step, idepth_backup}, lane 2 PointHessian::{HdiF, bdSumF,
                // idepth_hessian} as the solve that produced this step left them - two dwordx4 of ONE store instruction (round 3: seven stores)
                const v16i_t b0 = ldg16<DESC, OFF_B0>(&B);
                v4f_t *w_geo = GP(v4f_t, b0, BP_GEO);
                {
                    // (every lane stores: odd lanes the dwords 4..7, even lanes the dwords 8..11 - 32 copies each, no exec-mask branch around the store)
                    const bool l1 = (lane & 1) != 0;
                    v4f_t v;
                    v.x = l1 ? ni : rHdi; v.y = l1 ? ni : rBd; v.z = l1 ? step : rIdH; v.w = l1 ? idp : 0.0f;
                    AT(w_geo, p * 4 + (l1 ? 1u : 2u)) = v;
                }
            }
            idp = ni; idz = ni;
        }
        PtStep r; r.idp = idp; r.idz = idz;
        return r;
    };
    auto pstep = [&](const unsigned p, const PtIn<NSG> &q) -> PtStep { return pstepT(LM0{}, p, q); };
    // ================= FRONT half, per slot group: pattern projection, tap loads ====================================================
    // MODE 0: the lanes' slots are the targets g * 8 + s of ONE point.  MODE 3 (PAIR, second slot group of windows of 9..12 key frames): lanes 0..31 hold the targets
    // 8..11 of point A, lanes 32..63 the targets 8..11 of point B - pu / pv / idp are per-lane values then
    auto front_xT = [&](auto lc, auto gc, auto mc, const PtIn<NSG> &q, const float pu, const float pv, const float idp, TapsG &T) {
        constexpr int g = decltype(gc)::value, MODE = decltype(mc)::value;
#if LD_OPAQUE_K
        int k = k_; asm volatile("" : "+v"(k));
#endif
        const float h1 = qmlane(lc, J5{}, q, g);          // the state of the slot record: m of lane 5 of the group
        const int qState = __builtin_bit_cast(int, h1);
        const int t = (MODE == 3) ? 8 + (s & 3) : g * 8 + s;
        const bool exists = (t < F) && (q.rflat[g] >= 0);           // MARG: the flag of the point is read by the back half; unflagged points load taps nobody uses
        const bool isLin = MARG ? false : (exists && (q.rlin[g] != 0));
        const bool reset = MARG || ((stepMode & 2) && !isLin);
        const int st = exists ? (reset ? RES_IN : qState) : RES_OOB;
        const DevPair &pr = sPair[t];
        // ---- pattern pixel projection at the current state (ResidualProjections.h:24-33) ---------
        float px_ = pu + (float) ox, py_ = pv + (float) oy;
        float q0 = ((pr.KRKi[0] * px_ + pr.KRKi[1] * py_) + pr.KRKi[2] * 1.0f) + pr.Kt[0] * idp;
        float q1 = ((pr.KRKi[3] * px_ + pr.KRKi[4] * py_) + pr.KRKi[5] * 1.0f) + pr.Kt[1] * idp;
        float q2 = ((pr.KRKi[6] * px_ + pr.KRKi[7] * py_) + pr.KRKi[8] * 1.0f) + pr.Kt[2] * idp;
        float Ku = q0 / q2, Kv = q1 / q2;
        const bool pixOK = Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
        T.Ku = Ku; T.Kv = Kv;
        // ---- the 4 x 12 bytes under the projected pixel (GlobalFuncs.h:89-103): every lane loads - lanes without a sample read pixel 0 of a
        // readable image (one line for all of them), so that the loads sit in straight-line code one group ahead of their use
        const bool want = exists && !isLin && st != RES_OOB && pixOK;
        const int ix = (int) Ku, iy = (int) Kv;
        const unsigned off = want ? (unsigned) (3 * (ix + iy * W)) : 0u;
        const float *ib = DESC ? sImg[t] : B.img[(t < F) ? t : 0];
        const gptr_t<const float> bp = (gptr_t<const float>) (unsigned long long) ib + off;
        const gptr_t<const float> bq = bp + 3 * W;
        const v4f_t r0a = *(gptr_t<const v4f_t>) bp; const v2f_t r0b = *(gptr_t<const v2f_t>) (bp + 4);
        const v4f_t r1a = *(gptr_t<const v4f_t>) bq; const v2f_t r1b = *(gptr_t<const v2f_t>) (bq + 4);
        T.t[0] = r0a.x; T.t[1] = r0a.y; T.t[2] = r0a.z; T.t[3] = r0a.w; T.t[4] = r0b.x; T.t[5] = r0b.y;
        T.t[6] = r1a.x; T.t[7] = r1a.y; T.t[8] = r1a.z; T.t[9] = r1a.w; T.t[10] = r1b.x; T.t[11] = r1b.y;
    };
    using M0 = std::integral_constant<int, 0>;
    auto front_x = [&](auto gc, auto mc, const PtIn<NSG> &q, const float pu, const float pv, const float idp, TapsG &T) { front_xT(LM0{}, gc, mc, q, pu, pv, idp, T); };
    auto front_gT = [&](auto lc, auto gc, const PtIn<NSG> &q, const PtStep &ps, TapsG &T) { front_xT(lc, gc, M0{}, q, qgeo(lc, q, GEO_U), qgeo(lc, q, GEO_V), ps.idp, T); };
    auto front_g = [&](auto gc, const PtIn<NSG> &q, const PtStep &ps, TapsG &T) { front_gT(LM0{}, gc, q, ps, T); };

    // ================= the bilinear Vec3f sample of the target image (GlobalFuncs.h:89-103) from the taps the front half loaded ====================================
    // (its own step since round 6: in the pipelined loop of one slot group per point it runs BEFORE the next point's front half, which then loads into the same
    // registers - one set of 14 tap registers and no copy per point)
    auto interp = [&](const TapsG &T) -> Hit {
        Hit H;
        const float Ku = T.Ku, Kv = T.Kv;
        const int ix = (int) Ku, iy = (int) Kv;
        const float dx = Ku - ix, dy = Kv - iy, dxdy = dx * dy;
        const float a0 = T.t[0], a1 = T.t[1], a2 = T.t[2], b0_ = T.t[3], b1 = T.t[4], b2 = T.t[5];
        const float c0_ = T.t[6], c1_ = T.t[7], c2_ = T.t[8], d0 = T.t[9], d1 = T.t[10], d2 = T.t[11];
        const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
        H.h0 = ((w11 * d0 + w01 * c0_) + w10 * b0_) + w00 * a0;
        H.h1 = ((w11 * d1 + w01 * c1_) + w10 * b1) + w00 * a1;
        H.h2 = ((w11 * d2 + w01 * c2_) + w10 * b2) + w00 * a2;
        H.Ku = Ku; H.Kv = Kv;
        return H;
    };

    // ================= BACK half of a point: everything behind the image taps ================================================================
    // per-point state of the back half (set by back_begin, updated by back_g of every slot group, consumed by back_end)
    float pu = 0, pv = 0, priorF = 0, color = 0, wgt = 0, idp = 0, idz = 0, deltaF = 0;
    bool flagged = true;
    int recNActive = 0;
    float HddA = 0, bdA = 0, HcdA0 = 0, HcdA1 = 0, HcdA2 = 0, HcdA3 = 0;
    float HddL = 0, bdL = 0, HcdL0 = 0, HcdL1 = 0, HcdL2 = 0, HcdL3 = 0;
    float hostPart = 0.0f;       // this lane's partial of the host block of g_p (component k)
    float maxRelBS = 0.0f;
    int numGood = 0, nActive = 0;
    float gT[NSG];
    auto back_beginT = [&](auto lc, const unsigned p, const PtIn<NSG> &q, const PtStep &ps) {
        pu = qgeo(lc, q, GEO_U); pv = qgeo(lc, q, GEO_V);
        priorF = MARG ? qgeo(lc, q, GEO_PRIOR) * S.idepthFixPriorMargFac : qgeo(lc, q, GEO_PRIOR);
        flagged = MARG ? (margFlags[p] != 0) : true;
        color = q.color; wgt = q.wgt;
        idp = ps.idp; idz = ps.idz;
        recNActive = qreci(lc, q, REC_NACT);
        deltaF = idp - idz;
        HddA = 0; bdA = 0; HcdA0 = 0; HcdA1 = 0; HcdA2 = 0; HcdA3 = 0;
        HddL = 0; bdL = 0; HcdL0 = 0; HcdL1 = 0; HcdL2 = 0; HcdL3 = 0;
        hostPart = 0.0f;
        // AccumulatedSCHessian.cc:14-21: the SOLVE that finds a point without an active residual zeroes its maxRelBaseline - i.e. the solve whose
        // point step is fused in front of this pass (PtRec.nActive = active residuals of the linearisation that solve used).  A pass that is followed
        // by no solve (the last linearizeAll(false) of optimize(), the fixing pass) zeroes nothing.
        maxRelBS = ((stepMode & 1) && recNActive <= 0) ? 0.0f : qrec(lc, q, REC_MAXRELBS);
        numGood = qreci(lc, q, REC_NUMGOOD);
        nActive = 0;
    };
    auto back_begin = [&](const unsigned p, const PtIn<NSG> &q, const PtStep &ps) { back_beginT(LM0{}, p, q, ps); };

    // pair mode (PAIR): the per-point sums below live as "one point per half-wave" - lanes 0..31 hold point A's value, lanes 32..63 point B's.  MODE 1 / 2: a full pass
    // (8 slots) of point A / B, its slot sums go to that half only; MODE 3: the shared pass of the second slot group, sums per half-wave
    float candE = -1.0f;          // energy of the point's residual into the newest frame without the outlier clamp (slot F - 1; candidate selection of the host)
    float gT0B = 0.0f;            // (pair mode) the first group's G entries of point B (gT[0]: point A's; gT[1]: the shared second group's)
    bool hasB = true;             // (pair mode) point B exists
    const int half = lane >> 5;
    auto back_xT = [&](auto lc, auto gc, auto mc, const unsigned p, const PtIn<NSG> &q, const Hit &T) {
        constexpr int g = decltype(gc)::value, MODE = decltype(mc)::value;
        auto ACC = [&](float &X, const float v) {
            if constexpr (MODE == 0) X += sum_slots(v, a16, a32);
            else if constexpr (MODE == 3) X += sum_half(v, a16);
            else { const float s_ = sum_slots(v, a16, a32); X += (half == MODE - 1) ? s_ : 0.0f; }
        };
#if LD_OPAQUE_K
        int k = k_; asm volatile("" : "+v"(k));
#endif
        // the uniform scalars of the slot record sit in the m of lanes 3 (energy), 5 (state), 6 (activity) of its 8-lane group
        int qState[NSG], qActive[NSG];
        float qEnergy[NSG];
        {
            const float h1 = qmlane(lc, J5{}, q, g), h2 = qmlane(lc, J6{}, q, g), l3 = qmlane(lc, J3{}, q, g);
            qState[g] = __builtin_bit_cast(int, h1); qActive[g] = __builtin_bit_cast(int, h2); qEnergy[g] = l3;
        }
        {
            const int t = (MODE == 3) ? 8 + (s & 3) : g * 8 + s;
            const unsigned slot = p * (unsigned) FS + (unsigned) t;
            const bool exists = (t < F) && (q.rflat[g] >= 0) && flagged;
            const bool isLin = MARG ? false : (exists && (q.rlin[g] != 0));
            // resetOOB (Residuals.h): MARG always; stepMode bit 1 = the optimize() preamble on every non-linearised residual (FullSystem.cc:744-748)
            const bool reset = MARG || ((stepMode & 2) && !isLin);
            const int st = exists ? (reset ? RES_IN : qState[g]) : RES_OOB;
            const DevPair &pr = sPair[t];

            int newState = st;
            float newEnergy = (exists && !reset) ? qEnergy[g] : 0.0f;
            float newEnergyWO = -1.0f;
            int activeNew = exists ? qActive[g] : 0;
            float jp = exists ? q.jp[g] : 0.0f;     // this lane's component k of JpJdF
            float c0 = 0, c1 = 0, c2 = 0;
            int toRemove = 0;
            double ret = 0.0;                                      // linearize() return value
            const bool doLin = exists && !isLin;

            // ================= active-set residual: PointFrameResidual::linearize ==================
            if (doLin && st == RES_OOB) { newState = RES_OOB; ret = (double) newEnergy; if (FIX) toRemove = 1; }
            // (wave-uniform structure below is predicated per 8-lane group)
            bool compute = doLin && st != RES_OOB;
            // ---- centre projection at the linearisation point (ResidualProjections.h:57-84) --------
            float KliP0 = (pu + 0 - cx) * fxi, KliP1 = (pv + 0 - cy) * fyi;
            float ptp0 = ((pr.R0[0] * KliP0 + pr.R0[1] * KliP1) + pr.R0[2] * 1.0f) + pr.t0[0] * idz;
            float ptp1 = ((pr.R0[3] * KliP0 + pr.R0[4] * KliP1) + pr.R0[5] * 1.0f) + pr.t0[1] * idz;
            float ptp2 = ((pr.R0[6] * KliP0 + pr.R0[7] * KliP1) + pr.R0[8] * 1.0f) + pr.t0[2] * idz;
            float drescale = 1.0f / ptp2;
            float new_idepth = idz * drescale;
            float uu = ptp0 * drescale, vv = ptp1 * drescale;
            float cKu = uu * fx + cx, cKv = vv * fy + cy;
            bool centerOK = (drescale > 0) && cKu > 1.1f && cKv > 1.1f && cKu < wM3G && cKv < hM3G;
            // ---- pattern pixel projection at the current state: done by the front half (ResidualProjections.h:24-33) ---------
            const float Ku = T.Ku, Kv = T.Kv;
            bool pixOK = Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G;
            // ---- bilinear Vec3f sample of the target image: interp() above ---------------------
            float hit0 = 0, hit1 = 0, hit2 = 0;
            {
                const bool smp = compute && centerOK && pixOK;
                hit0 = smp ? T.h0 : 0.0f; hit1 = smp ? T.h1 : 0.0f; hit2 = smp ? T.h2 : 0.0f;
            }
            bool laneBad = compute && (!centerOK || !pixOK || !isfinite(hit0));
            unsigned long long badMask = __ballot(laneBad);
            bool anyBad = ((badMask >> (s * 8)) & 0xFFull) != 0;
            if (compute && anyBad) { newState = RES_OOB; ret = (double) newEnergy; compute = false; }

            // ---- photometric terms (Residuals.cc:126-188) ---------------------------------------------
            float residual = hit0 - (float) (pr.aff[0] * color + pr.aff[1]);
            float drdA = (color - pr.b0);
            float w_ = sqrtf(S.outlierTHSumComponent / (S.outlierTHSumComponent + (hit1 * hit1 + hit2 * hit2)));
            w_ = 0.5f * (w_ + wgt);
            float hw = fabsf(residual) < S.huberTH ? 1 : S.huberTH / fabsf(residual);
            float eTerm = w_ * w_ * hw * residual * residual * (2 - hw);
            if (hw < 1) hw = sqrtf(hw);
            hw = hw * w_;
            float gx = hit1 * hw, gy = hit2 * hw;
            float resF = residual * hw;
            float jab0 = drdA * hw, jab1 = hw;
            if (!compute) { eTerm = 0; gx = 0; gy = 0; resF = 0; jab0 = 0; jab1 = 0; }

            float energyLeft = seq8(eTerm, k, lane);
            float wJI2_sum = seq8(hw * hw * (gx * gx + gy * gy), k, lane);
            float JI00 = sum8(gx * gx), JI11 = sum8(gy * gy), JI10 = sum8(gx * gy);
            float JabJI00 = sum8(drdA * hw * gx), JabJI01 = sum8(drdA * hw * gy), JabJI10 = sum8(hw * gx), JabJI11 = sum8(hw * gy);
            float Jab00 = sum8(drdA * drdA * hw * hw), Jab01 = sum8(drdA * hw * hw), Jab11 = sum8(hw * hw);
            if (S.affineOptModeA < 0) jab0 = 0;     // J->JabF zeroed AFTER the 2x2 sums (Residuals.cc:184-185)
            if (S.affineOptModeB < 0) jab1 = 0;

            // ---- geometric Jacobians at the linearisation point (Residuals.cc:67-104) ----------------
            float Jpdd0 = drescale * (pr.t0[0] - pr.t0[2] * uu) * 1.0f * fx;
            float Jpdd1 = drescale * (pr.t0[1] - pr.t0[2] * vv) * 1.0f * fy;
            float dCx2 = drescale * (pr.R0[6] * uu - pr.R0[0]);
            float dCx3 = fx * drescale * (pr.R0[7] * uu - pr.R0[1]) * fyi;
            float dCx0 = KliP0 * dCx2, dCx1 = KliP1 * dCx3;
            float dCy2 = fy * drescale * (pr.R0[6] * vv - pr.R0[3]) * fxi;
            float dCy3 = drescale * (pr.R0[7] * vv - pr.R0[4]);
            float dCy0 = KliP0 * dCy2, dCy1 = KliP1 * dCy3;
            float x[10], y[10];
            x[0] = (dCx0 + uu) * 50.0f; x[1] = dCx1 * 50.0f; x[2] = (dCx2 + 1) * 50.0f; x[3] = dCx3 * 50.0f;
            y[0] = dCy0 * 50.0f; y[1] = (dCy1 + vv) * 50.0f; y[2] = dCy2 * 50.0f; y[3] = (dCy3 + 1) * 50.0f;
            x[4] = new_idepth * fx; x[5] = 0; x[6] = -new_idepth * uu * fx; x[7] = -uu * vv * fx; x[8] = (1 + uu * uu) * fx; x[9] = -vv * fx;
            y[4] = 0; y[5] = new_idepth * fy; y[6] = -new_idepth * vv * fy; y[7] = -(1 + vv * vv) * fy; y[8] = uu * vv * fy; y[9] = uu * fy;

            float resAcc = resF;          // the residual column of the accumulators: resF (mode 0) or res_toZeroF (MARG, mode 2)
            if (MARG) {
                float dpx = 0, dpy = 0;
#pragma unroll
                for (int i = 0; i < 6; i++) { dpx += x[4 + i] * pr.dp[i]; dpy += y[4 + i] * pr.dp[i]; }
                const float dcx = ((x[0] * cD0 + x[1] * cD1) + x[2] * cD2) + x[3] * cD3;
                const float dcy = ((y[0] * cD0 + y[1] * cD1) + y[2] * cD2) + y[3] * cD3;
                const float Jp_delta_x = dpx + dcx + Jpdd0 * deltaF, Jp_delta_y = dpy + dcy + Jpdd1 * deltaF;
                float rtz = resF;
                rtz = rtz - gx * Jp_delta_x; rtz = rtz - gy * Jp_delta_y; rtz = rtz - jab0 * pr.dp[6]; rtz = rtz - jab1 * pr.dp[7];
                resAcc = compute ? rtz : 0.0f;
            }
            if (compute) {
                newEnergyWO = energyLeft;
                float th = pr.thMax;
                if (energyLeft > th || wJI2_sum < 2) { energyLeft = th; newState = RES_OUTLIER; }
                else newState = RES_IN;
                newEnergy = energyLeft;
                ret = (double) energyLeft;
                c0 = cKu; c1 = cKv; c2 = new_idepth;
                // stepMode bit 2 (re-chunking, ba_api.hip rechunk()): the applied state is linearised again only to re-form the per-chunk
                // partial sums - the decisions of the pass that produced it stand (its energy thresholds have moved on since)
                if (stepMode & 4) { newState = st; newEnergy = qEnergy[g]; ret = (double) newEnergy; }
            }

            // ================= applyRes(true) (Residuals.h:70-87) ======================================
            if (doLin && st != RES_OOB) {
                if (newState == RES_IN) {
                    activeNew = 1;
                    // takeData (Residuals.h:123-128)
                    float v0 = JI00 * Jpdd0 + JI10 * Jpdd1, v1 = JI10 * Jpdd0 + JI11 * Jpdd1;
                    const float jx = selk6(k, x[4], x[5], x[6], x[7], x[8], x[9]);
                    const float jy = selk6(k, y[4], y[5], y[6], y[7], y[8], y[9]);
                    float j6 = JabJI00 * Jpdd0 + JabJI01 * Jpdd1, j7 = JabJI10 * Jpdd0 + JabJI11 * Jpdd1;
                    jp = (k < 6) ? (jx * v0 + jy * v1) : (k == 6 ? j6 : j7);
                } else {
                    activeNew = 0;
                }
                if (FIX) {
                    if (activeNew) {
                        if (q.rnew[g]) {
                            // FullSystem.cc:1518-1534: relative baseline of new residuals
                            float inf0 = (pr.KRKi[0] * pu + pr.KRKi[1] * pv) + pr.KRKi[2] * 1.0f;
                            float inf1 = (pr.KRKi[3] * pu + pr.KRKi[4] * pv) + pr.KRKi[5] * 1.0f;
                            float inf2 = (pr.KRKi[6] * pu + pr.KRKi[7] * pv) + pr.KRKi[8] * 1.0f;
                            float r0 = inf0 + pr.Kt[0] * idp, r1 = inf1 + pr.Kt[1] * idp, r2 = inf2 + pr.Kt[2] * idp;
                            float ax = inf0 / inf2 - r0 / r2, ay = inf1 / inf2 - r1 / r2;
                            float relBS = (float) (0.01 * (double) sqrtf(ax * ax + ay * ay));
                            if (relBS > maxRelBS) maxRelBS = relBS;     // merged across slots after the loop
                        }
                    } else toRemove = 1;
                }
            }
            unsigned long long newGoodMask = 0;
            if (FIX) newGoodMask = __ballot(doLin && st != RES_OOB && activeNew && q.rnew[g] && k == 0);

            // ================= accumulate: active residual, mode 0 (AccumulatedTopHessian.cc) ==========
            const bool accHere = doLin && activeNew && compute;
            // residual column: Jab_r uses the (possibly zeroed) JabF, Jab2 / JabJIdx the un-zeroed sums
            const float z10 = (S.affineOptModeA < 0) ? 0.0f : 1.0f, z11 = (S.affineOptModeB < 0) ? 0.0f : 1.0f;
            const float JI_r0 = sum8(resAcc * gx), JI_r1 = sum8(resAcc * gy);
            const float Jab_r0 = z10 * sum8(drdA * hw * resAcc), Jab_r1 = z11 * sum8(hw * resAcc), rr = sum8(resAcc * resAcc);
            if (accHere) {
                const float xr1 = selk8(k, x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
                const float yr1 = selk8(k, y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7]);
                const float t1a = __builtin_fmaf(JI00, xr1, JI10 * yr1), t1b = __builtin_fmaf(JI10, xr1, JI11 * yr1);
#pragma unroll
                for (int c = 0; c < 10; c++) accR[g][c] = __builtin_fmaf(t1a, x[c], __builtin_fmaf(t1b, y[c], accR[g][c]));
                accR[g][10] = __builtin_fmaf(xr1, JabJI00, __builtin_fmaf(yr1, JabJI01, accR[g][10]));
                accR[g][11] = __builtin_fmaf(xr1, JabJI10, __builtin_fmaf(yr1, JabJI11, accR[g][11]));
                accR[g][12] = __builtin_fmaf(xr1, JI_r0, __builtin_fmaf(yr1, JI_r1, accR[g][12]));
                // second row: 8, 9 (geometric) on lanes 0, 1; 10, 11 (affine) and 12 (residual) on lanes 2, 3, 4
                const float xr2 = sel2(k == 0, x[8], x[9]), yr2 = sel2(k == 0, y[8], y[9]);
                const float t2a = __builtin_fmaf(JI00, xr2, JI10 * yr2), t2b = __builtin_fmaf(JI10, xr2, JI11 * yr2);
                const float g8 = __builtin_fmaf(t2a, x[8], t2b * y[8]), g9 = __builtin_fmaf(t2a, x[9], t2b * y[9]);
                const float g10 = __builtin_fmaf(xr2, JabJI00, yr2 * JabJI01), g11 = __builtin_fmaf(xr2, JabJI10, yr2 * JabJI11), g12 = __builtin_fmaf(xr2, JI_r0, yr2 * JI_r1);
                const bool geo = k < 2;
                accR[g][13] += geo ? g8 : 0.0f;
                accR[g][14] += geo ? g9 : 0.0f;
                accR[g][15] += geo ? g10 : (k == 2) ? Jab00 : 0.0f;
                accR[g][16] += geo ? g11 : (k == 2) ? Jab01 : (k == 3) ? Jab11 : 0.0f;
                accR[g][17] += geo ? g12 : (k == 2) ? Jab_r0 : (k == 3) ? Jab_r1 : (k == 4) ? rr : 0.0f;
            }
            // per-slot contributions to the point sums (same value in all 8 lanes of the slot)
            float Ji2_0 = JI00 * Jpdd0 + JI10 * Jpdd1, Ji2_1 = JI10 * Jpdd0 + JI11 * Jpdd1;
            float sbd = accHere ? (JI_r0 * Jpdd0 + JI_r1 * Jpdd1) : 0.0f;
            float sHdd = accHere ? (Ji2_0 * Jpdd0 + Ji2_1 * Jpdd1) : 0.0f;
            float sHc0 = accHere ? (x[0] * Ji2_0 + y[0] * Ji2_1) : 0.0f, sHc1 = accHere ? (x[1] * Ji2_0 + y[1] * Ji2_1) : 0.0f;
            float sHc2 = accHere ? (x[2] * Ji2_0 + y[2] * Ji2_1) : 0.0f, sHc3 = accHere ? (x[3] * Ji2_0 + y[3] * Ji2_1) : 0.0f;
            // sum over the 8 slots of this pass
            ACC(bdA, sbd); ACC(HddA, sHdd);
            ACC(HcdA0, sHc0); ACC(HcdA1, sHc1);
            ACC(HcdA2, sHc2); ACC(HcdA3, sHc3);
            if (accHere && k == 0) nresA++;

            // ================= linearised residual, mode 1 (AccumulatedTopHessian.cc:29-31,44-63) =======
            if (HAS_L) {
                const bool accL = isLin && activeNew;
                float lsbd = 0, lsHdd = 0, lH0 = 0, lH1 = 0, lH2 = 0, lH3 = 0;
                if (accL) {
                    const ldso_rawjac_t &J = B.Jlin[(unsigned) q.rlidx[g]];
                    const float *rtz = B.rtz + (unsigned) q.rlidx[g] * 8u;
                    float dpx = 0, dpy = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) { dpx += J.Jpdxi[0][i] * pr.dp[i]; dpy += J.Jpdxi[1][i] * pr.dp[i]; }
                    float dcx = ((J.Jpdc[0][0] * cD0 + J.Jpdc[0][1] * cD1) + J.Jpdc[0][2] * cD2) + J.Jpdc[0][3] * cD3;
                    float dcy = ((J.Jpdc[1][0] * cD0 + J.Jpdc[1][1] * cD1) + J.Jpdc[1][2] * cD2) + J.Jpdc[1][3] * cD3;
                    float Jp_delta_x = dpx + dcx + J.Jpdd[0] * deltaF;
                    float Jp_delta_y = dpy + dcy + J.Jpdd[1] * deltaF;
                    float lgx = J.JIdx[0][k], lgy = J.JIdx[1][k], la0 = J.JabF[0][k], la1 = J.JabF[1][k];
                    float ra = rtz[k];
                    ra = ra + lgx * Jp_delta_x; ra = ra + lgy * Jp_delta_y; ra = ra + la0 * pr.dp[6]; ra = ra + la1 * pr.dp[7];
                    float lJI_r0 = seq8(ra * lgx, k, lane), lJI_r1 = seq8(ra * lgy, k, lane);
                    float lJab_r0 = seq8(ra * la0, k, lane), lJab_r1 = seq8(ra * la1, k, lane), lrr = seq8(ra * ra, k, lane);
                    float a = J.JIdx2[0], b = J.JIdx2[1], c = J.JIdx2[3];
                    float lx[10], ly[10];
#pragma unroll
                    for (int i = 0; i < 4; i++) { lx[i] = J.Jpdc[0][i]; ly[i] = J.Jpdc[1][i]; }
#pragma unroll
                    for (int i = 0; i < 6; i++) { lx[4 + i] = J.Jpdxi[0][i]; ly[4 + i] = J.Jpdxi[1][i]; }
                    {
                        // AccumulatorApprox::update / updateTopRight / updateBotRight of this residual into the wave's own cell of the slot: lane k
                        // owns row k (columns k..12) and, for k < 5, row k + 8 - the distribution of the mode-0 register accumulators.  Plain
                        // read-modify-write of LDS words with exactly one writer lane each: deterministic, no atomics.  (Until round 3 lane 0
                        // performed all 91 updates while 7 lanes idled.)
                        float *dst = sTopL + (wave * FS + t) * LD_TOPN;
                        const float lxr = (k == 0) ? lx[0] : (k == 1) ? lx[1] : (k == 2) ? lx[2] : (k == 3) ? lx[3] : (k == 4) ? lx[4] : (k == 5) ? lx[5] : (k == 6) ? lx[6] : lx[7];
                        const float lyr = (k == 0) ? ly[0] : (k == 1) ? ly[1] : (k == 2) ? ly[2] : (k == 3) ? ly[3] : (k == 4) ? ly[4] : (k == 5) ? ly[5] : (k == 6) ? ly[6] : ly[7];
                        const int b1 = k * 13 - (k * (k - 1)) / 2 - k;                       // tri13(k, cc) = b1 + cc
#pragma unroll
                        for (int cc = 0; cc < 10; cc++)
                            if (cc >= k) dst[b1 + cc] += a * lx[cc] * lxr + c * ly[cc] * lyr + b * (lx[cc] * lyr + ly[cc] * lxr);
                        dst[b1 + 10] += lxr * J.JabJIdx[0] + lyr * J.JabJIdx[1];
                        dst[b1 + 11] += lxr * J.JabJIdx[2] + lyr * J.JabJIdx[3];
                        dst[b1 + 12] += lxr * lJI_r0 + lyr * lJI_r1;
                        if (k < 2) {             // rows 8, 9 (geometric)
                            const int r2 = k + 8, b2 = r2 * 13 - (r2 * (r2 - 1)) / 2 - r2;
                            const float lx2 = (k == 0) ? lx[8] : lx[9], ly2 = (k == 0) ? ly[8] : ly[9];
                            if (k == 0) dst[b2 + 8] += a * lx[8] * lx2 + c * ly[8] * ly2 + b * (lx[8] * ly2 + ly[8] * lx2);
                            dst[b2 + 9] += a * lx[9] * lx2 + c * ly[9] * ly2 + b * (lx[9] * ly2 + ly[9] * lx2);
                            dst[b2 + 10] += lx2 * J.JabJIdx[0] + ly2 * J.JabJIdx[1];
                            dst[b2 + 11] += lx2 * J.JabJIdx[2] + ly2 * J.JabJIdx[3];
                            dst[b2 + 12] += lx2 * lJI_r0 + ly2 * lJI_r1;
                        } else if (k == 2) {     // row 10
                            dst[tri13(10, 10)] += J.Jab2[0]; dst[tri13(10, 11)] += J.Jab2[1]; dst[tri13(10, 12)] += lJab_r0;
                        } else if (k == 3) {     // row 11
                            dst[tri13(11, 11)] += J.Jab2[3]; dst[tri13(11, 12)] += lJab_r1;
                        } else if (k == 4) {     // row 12
                            dst[tri13(12, 12)] += lrr;
                        }
                        if (k == 0) nresL++;
                    }
                    float lJi0 = a * J.Jpdd[0] + b * J.Jpdd[1], lJi1 = b * J.Jpdd[0] + c * J.Jpdd[1];
                    lsbd = lJI_r0 * J.Jpdd[0] + lJI_r1 * J.Jpdd[1];
                    lsHdd = lJi0 * J.Jpdd[0] + lJi1 * J.Jpdd[1];
                    lH0 = lx[0] * lJi0 + ly[0] * lJi1; lH1 = lx[1] * lJi0 + ly[1] * lJi1; lH2 = lx[2] * lJi0 + ly[2] * lJi1; lH3 = lx[3] * lJi0 + ly[3] * lJi1;
                }
                bdL += sum_slots(lsbd, a16, a32); HddL += sum_slots(lsHdd, a16, a32);
                HcdL0 += sum_slots(lH0, a16, a32); HcdL1 += sum_slots(lH1, a16, a32);
                HcdL2 += sum_slots(lH2, a16, a32); HcdL3 += sum_slots(lH3, a16, a32);
            }

            // ================= lifted Schur row: target block and this slot's share of the host block ==
            float tgt = 0.0f, hpart = 0.0f;
            {
                // component k of Ad^T JpJdF (target and host adjoint): lane k sums column k ^ r against the JpJdF component of lane k ^ r of its slot, r = 0..7 - the
                // permuted copies are 7 DPP moves (i ^ 7 = half mirror, i ^ 4..6 = half mirror of i ^ 3..1) and both products one packed multiply-add per r.
                // (Until round 6: all 8 components gathered into every lane - 8 DPP moves + 8 selects - and 16 multiply-adds whose operands the compiler packed with 12 moves.)
                float vx[8];
                vx[0] = jp; vx[1] = dpp_quad_xor1(jp); vx[2] = dpp_quad_xor2(jp); vx[3] = dpp_quad_xor3(jp); vx[7] = dpp_half_mirror(jp);
                vx[6] = dpp_half_mirror(vx[1]); vx[5] = dpp_half_mirror(vx[2]); vx[4] = dpp_half_mirror(vx[3]);
                if (exists && activeNew) {
                    const v4f_t *ad = (const v4f_t *) (sAd + (t * 64 + k * 8) * 2);
                    v2f_t acc = {0.0f, 0.0f};
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        const v4f_t c = ad[r >> 1];
                        const v2f_t c0_ = {c.x, c.y}, c1_ = {c.z, c.w}, v0_ = {vx[r], vx[r]}, v1_ = {vx[r + 1], vx[r + 1]};
                        acc = __builtin_elementwise_fma(c0_, v0_, acc);
                        acc = __builtin_elementwise_fma(c1_, v1_, acc);
                    }
                    tgt = acc.x; hpart = acc.y;
                }
            }
            ACC(hostPart, hpart);
            if constexpr (MODE == 2) gT0B = tgt; else gT[g] = tgt;
            {
                const unsigned long long am = __ballot(exists && activeNew && k == 0);
                if constexpr (MODE == 0) nActive += __popcll(am);
                else if constexpr (MODE == 3) nActive += __popcll(am & (half ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull));
                else nActive += (half == MODE - 1) ? __popcll(am) : 0;
            }
            if (FIX) numGood += __popcll(newGoodMask);
            if (FIX) { float m = maxRelBS; m = fmaxf(m, __shfl_xor(m, 8, 64)); m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64)); maxRelBS = m; }

            // ---- per-slot outputs: the slot's SlotRec of the next set, lane k stores its own pair (one dwordx2 store per lane) ------------
            {
                // stored by EVERY lane, also for the padding slots behind F (the table is dense [P][FS]; they keep the zeros of the upload): a store under a
                // branch cannot be counted by the in-order memory counter, and everything in flight in front of it would be waited for with vmcnt(0)
                const v16i_t o0 = ldg16<DESC, OFF_S0>(&nxt);
                v2f_t *o_slot = GP(v2f_t, o0, RS_SLOT);
                float *o_cand = GP(float, o0, RS_CAND);
                const float ewo_ = doLin ? newEnergyWO : -1.0f;
                const float cenK = compute ? (k == 0 ? c0 : k == 1 ? c1 : c2) : q.m[g];      // k < 3 only: lanes 0..2 hold the old centre in their m
                float mOut = (k < 3) ? cenK : (k == LD_SM_ENERGY) ? newEnergy : (k == LD_SM_EWO) ? ewo_
                           : __builtin_bit_cast(float, (k == LD_SM_STATE) ? newState : (k == LD_SM_ACTIVE) ? activeNew : toRemove);
                v2f_t e; e.x = (t < F) ? jp : 0.0f; e.y = (t < F) ? mOut : 0.0f;
                if (MODE != 3 || hasB || half == 0) AT(o_slot, slot * 8 + (unsigned) k) = e;
                if constexpr (PAIR) { if (k == 0 && t == F - 1 && (MODE != 3 || hasB || half == 0)) AT(o_cand, p) = ewo_; }
                else { const int lf = (F - 1) - g * 8; if (lf >= 0 && lf < 8) candE = RLF(ewo_, lf * 8); (void) o_cand; }          // stored by back_end
                if (k == 0 && t < F && doLin) energySum += ret;
            }
            if (!DESC && dumpJ != nullptr && compute) {          // (debug dump: the step-wise entry points only)
                auto &o = *(gptr_t<ldso_rawjac_t>) (unsigned long long) (dumpJ + q.rflat[g]);          // global, not flat: a pending flat access makes every later wait a vmcnt(0)
                o.resF[k] = resF; o.JIdx[0][k] = gx; o.JIdx[1][k] = gy; o.JabF[0][k] = jab0; o.JabF[1][k] = jab1;
                if (k == 0) {
                    for (int i = 0; i < 6; i++) { o.Jpdxi[0][i] = x[4 + i]; o.Jpdxi[1][i] = y[4 + i]; }
                    for (int i = 0; i < 4; i++) { o.Jpdc[0][i] = x[i]; o.Jpdc[1][i] = y[i]; }
                    o.Jpdd[0] = Jpdd0; o.Jpdd[1] = Jpdd1;
                    o.JIdx2[0] = JI00; o.JIdx2[1] = JI10; o.JIdx2[2] = JI10; o.JIdx2[3] = JI11;
                    o.JabJIdx[0] = JabJI00; o.JabJIdx[1] = JabJI01; o.JabJIdx[2] = JabJI10; o.JabJIdx[3] = JabJI11;
                    o.Jab2[0] = Jab00; o.Jab2[1] = Jab01; o.Jab2[2] = Jab01; o.Jab2[3] = Jab11;
                }
            }
        }   // slot group
    };   // back_x
    auto back_x = [&](auto gc, auto mc, const unsigned p, const PtIn<NSG> &q, const Hit &T) { back_xT(LM0{}, gc, mc, p, q, T); };
    auto back_g = [&](auto gc, const unsigned p, const PtIn<NSG> &q, const TapsG &T) { back_x(gc, M0{}, p, q, interp(T)); };

    auto back_end = [&](const unsigned p, const PtIn<NSG> &q) {
        (void) q;
#if LD_OPAQUE_K
        int lane = lane_, k = k_; asm volatile("" : "+v"(lane), "+v"(k));
#endif
        // ================= per-point Schur quantities (AccumulatedSCHessian.cc:9-31) =====================
        float HdiF = 0, bdSumF = 0, idH = 0;
        float Hc0 = HcdA0 + HcdL0, Hc1 = HcdA1 + HcdL1, Hc2 = HcdA2 + HcdL2, Hc3 = HcdA3 + HcdL3;
        if (nActive > 0) {
            float H = HddA + HddL + priorF;
            if (H < 1e-10) H = 1e-10;
            idH = H;
            HdiF = (float) (1.0 / (double) H);
            bdSumF = bdA + bdL;
            if (!MARG) bdSumF += priorF * deltaF;       // shiftPriorToZero = true in accumulateSCF_MT, false in marginalizePointsF
        }
        // ---- store G row: [8*FS frame entries | Hcd 4, bdSum, HdiF, 0, 0] --------------------------------
        {
            const v16i_t o0 = ldg16<DESC, OFF_S0>(&nxt);
            float *Gall = GP(float, o0, RS_G);
            const unsigned g0 = p * GSu;              // P * GS floats stay far below 2^32 bytes
#pragma unroll
            for (int g = 0; g < NSG; g++) {
                const int t = g * 8 + s;
                float val = (t == h) ? hostPart : gT[g];
                if (nActive == 0) val = 0.0f;
                AT(Gall, g0 + 8u * (unsigned) t + (unsigned) k) = val;
            }
            {
                // (lane-indexed choices as select trees over the bits of the lane index, computed by every lane, and only the store under the lane condition: written as
                // chains of lane == c they became a tree of exec-mask branches around the stores - which the in-order memory counter then cannot count, see the pipeline)
                const bool z4 = nActive == 0;
                const float e = selk8(lane & 7, z4 ? 0.0f : Hc0, z4 ? 0.0f : Hc1, z4 ? 0.0f : Hc2, z4 ? 0.0f : Hc3, bdSumF, HdiF, 0.0f, 0.0f);
                // (stored by EVERY lane - eight copies of the same eight dwords: a store under a lane condition sits behind an exec-mask branch, which the in-order
                // memory counter cannot count, and the wait for the next point's records in front of it becomes a wait for every store of this point)
                static_assert(LD_GEXTRA == 8, "G row extras: one dword per lane & 7");
                AT(Gall, g0 + 8u * (unsigned) FS + (unsigned) (lane & 7)) = e;
            }
            // the point's PtRec of the next set: lanes 0..3 store one dwordx4 each (ONE store instruction, 64 contiguous bytes); lane 4 the
            // accumulator scalars only the fetch functions read
            v4f_t *o_pt = GP(v4f_t, o0, RS_PT), *o_acc = GP(v4f_t, o0, RS_ACC);
            {
                const int l3 = lane & 3;
                v4f_t v;
                v.x = selk4(l3, HdiF, HcdA0, HcdL0, maxRelBS);
                v.y = selk4(l3, bdSumF, HcdA1, HcdL1, __builtin_bit_cast(float, numGood));
                v.z = selk4(l3, idH, HcdA2, HcdL2, 0.0f);
                v.w = selk4(l3, __builtin_bit_cast(float, nActive), HcdA3, HcdL3, 0.0f);
                AT(o_pt, p * 4 + (unsigned) l3) = v;          // (every lane, sixteen copies of the same 64 bytes: see above)
                v4f_t w; w.x = HddA; w.y = bdA; w.z = HddL; w.w = bdL;
                AT(o_acc, p) = w;
                if constexpr (!PAIR) { float *o_cand = GP(float, o0, RS_CAND); AT(o_cand, p) = candE; }
            }
            if (lane == 0) { nidSum += fabsf(idp); nidCnt++; }
        }
    };   // back_end

    if constexpr (PAIR) {
        static_assert(!PAIR || (NSG == 2 && DESC && !HAS_L && !FIX && !MARG), "pair mode: the GN-iteration kernels of windows of 9..12 key frames");
        using G0 = std::integral_constant<int, 0>; using G1 = std::integral_constant<int, 1>;
        using M1 = std::integral_constant<int, 1>; using M2 = std::integral_constant<int, 2>; using M3 = std::integral_constant<int, 3>;
        // point step of a pair -> inverse depths, one point per half-wave
        auto pstep_pair = [&](const unsigned pA, const unsigned dAB, const PairIn &q, float &idpH, float &idzH) {
#define HSEL(v, i) (half ? RLF(v, 32 + (i)) : RLF(v, (i)))
            idpH = HSEL(q.rgeo, GEO_IDP); idzH = HSEL(q.rgeo, GEO_IDZ);
            if (stepMode & 1) {
                auto active_of = [&](const float m) { float l2, h2; group_bcast_pair<2>(m, k, l2, h2); (void) l2; return __builtin_bit_cast(int, h2) != 0; };
                const int t1 = 8 + (s & 3);
                float sA = seq8(sXa[s * 8 + k] * q.jpA0, k, lane), sB = seq8(sXa[s * 8 + k] * q.jpB0, k, lane), s1 = seq8(sXa[t1 * 8 + k] * q.jp1, k, lane);
                sA = ((s < F) && q.rfA0 >= 0 && active_of(q.mA0)) ? sA : 0.0f;
                sB = ((s < F) && q.rfB0 >= 0 && active_of(q.mB0)) ? sB : 0.0f;
                s1 = ((t1 < F) && q.rf1 >= 0 && active_of(q.m1)) ? s1 : 0.0f;
                const float sumA = sum_slots(sA, a16, a32), sumB = sum_slots(sB, a16, a32), sum1 = sum_half(s1, a16);
                const float rHdi = HSEL(q.rrec, REC_HDI), rBd = HSEL(q.rrec, REC_BDSUM), rIdH = HSEL(q.rrec, REC_IDH);
                const int rNact = half ? RLI(q.rrec, 32 + REC_NACT) : RLI(q.rrec, REC_NACT);
                float dot = 0;
                dot += xc0 * (HSEL(q.rrec, REC_HCDA + 0) + HSEL(q.rrec, REC_HCDL + 0)); dot += xc1 * (HSEL(q.rrec, REC_HCDA + 1) + HSEL(q.rrec, REC_HCDL + 1));
                dot += xc2 * (HSEL(q.rrec, REC_HCDA + 2) + HSEL(q.rrec, REC_HCDL + 2)); dot += xc3 * (HSEL(q.rrec, REC_HCDA + 3) + HSEL(q.rrec, REC_HCDL + 3));
                float b = rBd;
                b -= dot;
                b -= (half ? sumB : sumA);          // (the order of the reference's sum: first group, then second)
                b -= sum1;
                float step = 0.0f;
                const bool mine = (half == 0) || (dAB != 0);          // lanes of a point B that does not exist store nothing
                if (rNact > 0) {
                    if (isfinite(b)) step = -b * rHdi;
                    else { step = HSEL(q.rgeo, GEO_STEP); if ((lane & 31) == 0 && mine) *(gptr_t<double>) (unsigned long long) (B.scalars + 4) = 1.0; }
                }
                const float ni = idpH + 1.0f * step;
                {
                    const v16i_t b0 = ldg16<DESC, OFF_B0>(&B);
                    v4f_t *w_geo = GP(v4f_t, b0, BP_GEO);
                    const int l5 = lane & 31;
                    if ((l5 == 1 || l5 == 2) && mine) {
                        v4f_t v;
                        v.x = (l5 == 1) ? ni : rHdi; v.y = (l5 == 1) ? ni : rBd; v.z = (l5 == 1) ? step : rIdH; v.w = (l5 == 1) ? idpH : 0.0f;
                        AT(w_geo, (pA + (half ? dAB : 0u)) * 4 + (unsigned) l5) = v;
                    }
                }
                idpH = ni; idzH = ni;
            }
        };
        PairIn ca = {}, cn = {};
        float idpH = 0, idzH = 0, idpHn = 0, idzHn = 0;
        TapsG ta = {}, tb = {};
        auto viewA0 = [&](const PairIn &q) { PtIn<NSG> v = {}; v.rflat[0] = q.rfA0; v.jp[0] = q.jpA0; v.m[0] = q.mA0; return v; };
        auto viewB0 = [&](const PairIn &q) { PtIn<NSG> v = {}; v.rflat[0] = q.rfB0; v.jp[0] = q.jpB0; v.m[0] = q.mB0; return v; };
        auto view1 = [&](const PairIn &q, const bool withB) { PtIn<NSG> v = {}; v.rflat[1] = (half && !withB) ? -1 : q.rf1; v.jp[1] = q.jp1; v.m[1] = q.m1; return v; };
        if (pi < np) {
            const unsigned dA0 = (pi + LD_WAVES < np) ? LD_WAVES : 0;
            load_pair<DESC>(ca, B, cur, FS, (unsigned) (p0 + pi), dA0, s, k, lane);          // (the classic first record load above is not used in pair mode)
            pstep_pair((unsigned) (p0 + pi), dA0, ca, idpH, idzH);
            front_x(G0{}, M0{}, viewA0(ca), RLF(ca.rgeo, GEO_U), RLF(ca.rgeo, GEO_V), RLF(idpH, 0), ta);
#pragma clang loop unroll(disable)
            do {
                const unsigned pA = (unsigned) (p0 + pi);
                hasB = pi + LD_WAVES < np;
                const unsigned dAB = hasB ? LD_WAVES : 0, pB = pA + dAB;
                const bool n1 = pi + 2 * LD_WAVES < np;
                const unsigned pA1 = n1 ? pA + 2 * LD_WAVES : pA, dAB1 = (pi + 3 * LD_WAVES < np) ? LD_WAVES : 0;
                // ---- the next pair's records, the first-group taps of point B ----
                load_pair<DESC>(cn, B, cur, FS, pA1, dAB1, s, k, lane);
                front_x(G0{}, M0{}, viewB0(ca), RLF(ca.rgeo, 32 + GEO_U), RLF(ca.rgeo, 32 + GEO_V), RLF(idpH, 32), tb);
                // ---- per-point sums, one point per half-wave ----
                HddA = 0; bdA = 0; HcdA0 = 0; HcdA1 = 0; HcdA2 = 0; HcdA3 = 0; hostPart = 0.0f; nActive = 0;
                flagged = true; recNActive = 0;
                // ---- first group of A (all 64 lanes) ----
                pu = RLF(ca.rgeo, GEO_U); pv = RLF(ca.rgeo, GEO_V); color = ca.colA; wgt = ca.wgtA; idp = RLF(idpH, 0); idz = RLF(idzH, 0); deltaF = idp - idz;
                back_x(G0{}, M1{}, pA, viewA0(ca), interp(ta));
                // ---- taps of the shared second group (per-lane point) ----
                {
                    const float puH = HSEL(ca.rgeo, GEO_U), pvH = HSEL(ca.rgeo, GEO_V);
                    front_x(G1{}, M3{}, view1(ca, hasB), puH, pvH, idpH, ta);
                }
                // ---- first group of B ----
                if (hasB) {
                    pu = RLF(ca.rgeo, 32 + GEO_U); pv = RLF(ca.rgeo, 32 + GEO_V); color = ca.colB; wgt = ca.wgtB; idp = RLF(idpH, 32); idz = RLF(idzH, 32); deltaF = idp - idz;
                    back_x(G0{}, M2{}, pB, viewB0(ca), interp(tb));
                }
                // ---- the next pair: point steps, first-group taps of its A ----
                if (n1) pstep_pair(pA1, dAB1, cn, idpHn, idzHn);
                front_x(G0{}, M0{}, viewA0(cn), RLF(cn.rgeo, GEO_U), RLF(cn.rgeo, GEO_V), RLF(idpHn, 0), tb);
                // ---- the shared second group ----
                pu = HSEL(ca.rgeo, GEO_U); pv = HSEL(ca.rgeo, GEO_V); color = half ? ca.colB : ca.colA; wgt = half ? ca.wgtB : ca.wgtA; idp = idpH; idz = idzH; deltaF = idp - idz;
                back_x(G1{}, M3{}, pA + (half ? dAB : 0u), view1(ca, hasB), interp(ta));
                // ---- the two points' Schur rows: the half-wave values made wave-uniform, then the classic tail ----
                {
                    const float H_ = HddA, b_ = bdA, c0_ = HcdA0, c1_ = HcdA1, c2_ = HcdA2, c3_ = HcdA3, hp_ = hostPart, g0A_ = gT[0], g1_ = gT[1];
                    const int na_ = nActive;
                    const float hpO = __shfl_xor(hp_, 32, 64), g1O = __shfl_xor(g1_, 32, 64);
#pragma unroll
                    for (int X = 0; X < 2; X++) {
                        if (X == 1 && !hasB) break;
                        const int L0 = 32 * X;
                        HddA = RLF(H_, L0); bdA = RLF(b_, L0); HcdA0 = RLF(c0_, L0); HcdA1 = RLF(c1_, L0); HcdA2 = RLF(c2_, L0); HcdA3 = RLF(c3_, L0);
                        nActive = RLI(na_, L0);
                        hostPart = (half == X) ? hp_ : hpO;
                        gT[0] = X ? gT0B : g0A_;
                        gT[1] = (s < 4) ? ((half == X) ? g1_ : g1O) : 0.0f;          // lanes s >= 4 of the second group: the targets 12..15, not in the window
                        priorF = RLF(ca.rgeo, L0 + GEO_PRIOR);
                        idp = RLF(idpH, L0); idz = RLF(idzH, L0); deltaF = idp - idz;
                        const int rN = RLI(ca.rrec, L0 + REC_NACT);
                        maxRelBS = ((stepMode & 1) && rN <= 0) ? 0.0f : RLF(ca.rrec, L0 + REC_MAXRELBS);
                        numGood = RLI(ca.rrec, L0 + REC_NUMGOOD);
                        back_end(X ? pB : pA, PtIn<NSG>{});
                    }
                }
                ca = cn; idpH = idpHn; idzH = idzHn; ta = tb;
                pi += 2 * LD_WAVES;
            } while (pi < np);
        }
#undef HSEL
    } else
    {
        // The software pipeline.  F <= 8 (one slot group per point): the taps of point i + 1 are in flight while point i is worked on.  F > 8: the unit is
        // the slot group - the taps of the point's second group, then of the next point's first group, are in flight behind the group that is worked on.
        // The records of point i + 2 are in flight in both cases.
        using G0 = std::integral_constant<int, 0>;
        using G1 = std::integral_constant<int, NSG - 1>;
        constexpr bool PIPE = DESC || (NSG == 1 && LD_PIPE_ARGS);
        // One slot group per point, the batched kernel (round 6, second half): the records of the point in work are parked in the wavefront's own LDS by the
        // iteration in front (two ds_write_b128) and read back where the back half starts (two ds_read_b128) instead of being rotated through a third register set
        // (qa = qb, qb = qc: 16 vector moves per point of an issue-bound loop; LDS instructions have their own issue port)
        // (the batched kernel only: in k_linearize_one a wavefront has one to three points - nothing to amortise the LDS round trips and the peeled copy of the loop
        // body against; measured at C4, 12 points per chunk: 16.36 us with the stash, 15.85 without, profiles/r06_linearize_peel_stash_ab.log)
        constexpr bool STASH = DESC && !ONE && NSG == 1 && !HAS_L && !FIX && LD_STASH;
        int par = 0;
        float *const wRec = sRec + wave * 1024;                      // [parity][half][lane][4 dwords]
        float *const myRec = wRec + lane * 4;
        auto stash = [&](const int pr_, const PtIn<NSG> &q) {
            v4f_t a_, b_;
            a_.x = q.rgeo; a_.y = q.rrec; a_.z = q.color; a_.w = q.wgt;
            b_.x = __builtin_bit_cast(float, q.rflat[0]); b_.y = __builtin_bit_cast(float, q.rlin[0]); b_.z = q.jp[0]; b_.w = q.m[0];
            *(v4f_t *) (myRec + pr_ * 512) = a_; *(v4f_t *) (myRec + pr_ * 512 + 256) = b_;
        };
        auto unstash = [&](const int pr_) {
            const v4f_t a_ = *(const v4f_t *) (myRec + pr_ * 512), b_ = *(const v4f_t *) (myRec + pr_ * 512 + 256);
            PtIn<NSG> q = {};
            q.ls = wRec + pr_ * 512;
            q.rgeo = a_.x; q.rrec = a_.y; q.color = a_.z; q.wgt = a_.w;
            // (scalars first: __builtin_bit_cast on an ELEMENT of an ext-vector returns element 0 whatever the index - the trap of round 4, met again in round 6 the other
            // way round: every residual's is-linearised flag read back as its index, every residual skipped, energies 0)
            const float bx_ = b_.x, by_ = b_.y;
            q.rflat[0] = __builtin_bit_cast(int, bx_); q.rlin[0] = __builtin_bit_cast(int, by_); q.jp[0] = b_.z; q.m[0] = b_.w;
            return q;
        };
        PtStep sa = {0, 0}, sb = {0, 0};
        TapsG ta = {}, tb = {};
#if LD_STAMP_ON
        // cycle accounting of the pipelined loop (wave 0 of chunk 0 -> energyLog[24 + i]): shader-clock cycles per phase, summed over the wave's points
        long long cy_[6] = {0, 0, 0, 0, 0, 0}, ct_ = 0;
#define LCYC(i) do { const long long n_ = (long long) __builtin_readcyclecounter(); if ((i) > 0) cy_[i] += n_ - ct_; ct_ = n_; cy_[5] += ((i) == 0); } while (0)
#else
#define LCYC(i) do { } while (0)
#endif
        // The loads of the pipeline are issued UNCONDITIONALLY (behind the last point of the wavefront they re-read that point, nobody uses the result): the
        // memory counter of a wavefront is in order, so a load can only be waited for precisely (vmcnt(N), N = operations issued behind it) when the compiler
        // can COUNT what is issued behind it - a load or store under a branch counts as zero, and the first version of this loop (loads under `if (next point
        // exists)`) waited with vmcnt(0) for everything in flight in front of every use: no overlap at all (r6 call 1: 219 -> 207 us for the batch).
        if (pi < np) {
            sa = pstep((unsigned) (p0 + pi), qa); front_g(G0{}, qa, sa, ta);
            if (ONE && PIPE && np <= LD_WAVES) {          // (k_linearize_one only: the argument-based kernels spill with two copies of the point's code, and the batched kernel
                                                          // must stay below 192 VGPRs - it shares its CUs with the other half-batch's k_reduce_batch_dense, 128 VGPRs at 4 wavefronts per SIMD)
                // one point per wavefront (a single window spread over the whole chip, C3): nothing to overlap it with - the point straight through, without
                // the pipeline's look-ahead loads (measured with them: 13.5 -> 14.7 us at C3); above 8 key frames both slot groups' taps are in flight together
                const unsigned p = (unsigned) (p0 + pi);
                if constexpr (NSG > 1) front_g(G1{}, qa, sa, tb);
                back_begin(p, qa, sa);
                back_g(G0{}, p, qa, ta);
                if constexpr (NSG > 1) back_g(G1{}, p, qa, tb);
                back_end(p, qa);
            } else {
            if constexpr (PIPE) load_point<NSG, HAS_L, FIX, DESC>(qb, B, cur, FS, (unsigned) (p0 + ((pi + LD_WAVES < np) ? pi + LD_WAVES : pi)), s, k, lane);
            if constexpr (STASH) stash(0, qa);
            auto one_point = [&]() {
                const unsigned p = (unsigned) (p0 + pi);
                const bool n1 = pi + LD_WAVES < np, n2 = pi + 2 * LD_WAVES < np;
                const unsigned p1 = n1 ? p + LD_WAVES : p, p2 = n2 ? p + 2 * LD_WAVES : p1;
                if constexpr (!PIPE) {
                    // the cold variants above 8 key frames (fix / linearised / marginalisation passes): no taps in flight behind the arithmetic - with two slot
                    // groups the pipelined form does not fit the register file (100+ spilled registers, measured)
                    (void) n1; (void) n2; (void) p1; (void) p2; (void) qb; (void) qc; (void) sb; (void) tb;
                    if (pi != waveU) { load_point<NSG, HAS_L, FIX, DESC>(qa, B, cur, FS, p, s, k, lane); sa = pstep(p, qa); front_g(G0{}, qa, sa, ta); }
                    back_begin(p, qa, sa);
                    back_g(G0{}, p, qa, ta);
                    front_g(G1{}, qa, sa, ta);
                    back_g(G1{}, p, qa, ta);
                    back_end(p, qa);
                } else if constexpr (NSG == 1) {
                    LCYC(0);
                    Hit ha = interp(ta);          // the taps of this point are consumed here: the next point's go into the same registers
                    // (pinned: the sample is plain arithmetic, which the compiler sinks to its first use - behind the next point's tap loads, with the old taps alive across
                    // them in a second register set and 14 moves per point)
                    asm volatile("" : "+v"(ha.h0), "+v"(ha.h1), "+v"(ha.h2), "+v"(ha.Ku), "+v"(ha.Kv));
                    using LMS = std::bool_constant<STASH && LD_STASH_LM>;
                    if constexpr (STASH) { stash(par ^ 1, qb); qb.ls = wRec + (par ^ 1) * 512; }
                    if (n1) sb = pstepT(LMS{}, p1, qb);          // (the fused point step stores: only for a real next point)
                    front_gT(LMS{}, G0{}, qb, sb, ta);
                    LCYC(1);          // point step + projection + tap issue of the next point
                    load_point<NSG, HAS_L, FIX, DESC>(qc, B, cur, FS, p2, s, k, lane);
                    LCYC(2);          // record loads issued
                    if constexpr (STASH) qa = unstash(par);
                    back_beginT(LMS{}, p, qa, sa);
                    back_xT(LMS{}, G0{}, M0{}, p, qa, ha);
                    back_end(p, qa);
                    LCYC(3);          // everything behind the taps of this point
#if LD_STAMP_ON
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    LCYC(4);          // what is still in flight at the end of the iteration (stamps build only: waits for the stores too)
                    par ^= 1;
                } else {
                    LCYC(0);
                    front_g(G1{}, qa, sa, tb);
                    back_begin(p, qa, sa);
                    back_g(G0{}, p, qa, ta);
                    LCYC(1);          // second group's taps issued, first group worked through
                    if (n1) sb = pstep(p1, qb);
                    front_g(G0{}, qb, sb, ta);
                    load_point<NSG, HAS_L, FIX, DESC>(qc, B, cur, FS, p2, s, k, lane);
                    LCYC(2);          // next point: step, first group's taps, records of the point behind it
                    back_g(G1{}, p, qa, tb);
                    back_end(p, qa);
                    LCYC(3);          // second group + the point's Schur row
#if LD_STAMP_ON
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    LCYC(4);
                }
                if constexpr (PIPE) {
                    if constexpr (STASH) { qb = qc; sa = sb; }          // (qc is loaded behind the last use of qb: the same registers, no copy)
                    else { qa = qb; qb = qc; sa = sb; }
                }
                pi += LD_WAVES;
            };
            if constexpr (STASH && LD_PEEL) {
                // The first point peeled off the loop: the wait for the next point's records (the top of the body) counts the stores issued behind their loads, and at
                // the loop header the compiler takes the minimum over the ways into it - entered straight from the prologue (no store yet) that is vmcnt(0) for every
                // iteration, i.e. a wait for the stores of the point just finished; entered from a copy of the body it is vmcnt(<stores per point>)
                one_point();
#pragma clang loop unroll(disable)
                while (pi < np) one_point();
            } else {
#pragma clang loop unroll(disable)
                do one_point(); while (pi < np);
            }
            }
        }
#if LD_STAMP_ON
        if (chunk == 0 && tid == 0) for (int u = 0; u < 6; u++) B.energyLog[24 + u] = (double) cy_[u];
#endif
    }   // points of this wave

    LSTAMP(6);
    // ================= block reduction of the top accumulators =============================================
    // every lane stores the upper-triangle part of its rows (each of the 91 cells of a slot has exactly one writer), then the
    // waves of the block are summed through LDS in fixed order
#pragma unroll
    for (int g = 0; g < NSG; g++) {
        float *cell = sRed + (size_t) (wave * FS + g * 8 + s) * LD_TOPN;
        const int b1 = k * 13 - (k * (k - 1)) / 2 - k;                       // tri13(k, c) = b1 + c
        const int r2 = k + 8, b2 = r2 * 13 - (r2 * (r2 - 1)) / 2 - r2;
#pragma unroll
        for (int c = 0; c < 13; c++) if (c >= k) cell[b1 + c] = accR[g][c];
#pragma unroll
        for (int c = 8; c < 13; c++) if (k < 5 && c >= r2) cell[b2 + c] = accR[g][13 + c - 8];
    }
    // energy / counters: wave reduce, then LDS (own cells: no second barrier)
    double *sE = (double *) (sXa + FS * 8);
    int *sC = (int *) (sE + LD_WAVES);
    float *sN = (float *) (sC + 4 * LD_WAVES);
    {
        double e = energySum;
        int na = nresA, nl = nresL;
        for (int o = 32; o > 0; o >>= 1) { double e2 = __shfl_xor(e, o, 64); int a2 = __shfl_xor(na, o, 64), l2 = __shfl_xor(nl, o, 64); e += e2; na += a2; nl += l2; }
        if (lane == 0) { sE[wave] = e; sC[wave * 4 + 0] = na; sC[wave * 4 + 1] = nl; sC[wave * 4 + 2] = nidCnt; sN[wave] = nidSum; }
    }
    __syncthreads();
    LSTAMP(7);
    const v16i_t e2 = ldg16<DESC, OFF_S0>(&nxt);
    float *o_topA = GP(float, e2, RS_TOPA), *o_topL = GP(float, e2, RS_TOPL);
    double *o_chunkE = GP(double, e2, RS_CHUNKE);
    for (int i = tid; i < FS * LD_TOPN; i += blockDim.x) {
        float a = 0;
#pragma unroll
        for (int wv = 0; wv < LD_WAVES; wv++) a += sRed[wv * FS * LD_TOPN + i];
        if constexpr (PAIR) {
            // pair mode: the lanes 32..63 of the shared pass accumulated the targets 8..11 of their point in the cells of the slots 12..15
            const int t_ = i / LD_TOPN;
            if (t_ >= 12) a = 0.0f;
            else if (t_ >= 8) {
#pragma unroll
                for (int wv = 0; wv < LD_WAVES; wv++) a += sRed[wv * FS * LD_TOPN + i + 4 * LD_TOPN];
            }
        }
        o_topA[(size_t) chunk * FS * LD_TOPN + i] = a;
        if (HAS_L) {
            float l = 0;
#pragma unroll
            for (int wv = 0; wv < LD_WAVES; wv++) l += sTopL[wv * FS * LD_TOPN + i];
            o_topL[(size_t) chunk * FS * LD_TOPN + i] = l;
        }
    }
    if (tid == 0) {
        double e = 0; int na = 0, nl = 0, nc = 0; float ns = 0;
        for (int wv = 0; wv < LD_WAVES; wv++) { e += sE[wv]; na += sC[wv * 4 + 0]; nl += sC[wv * 4 + 1]; nc += sC[wv * 4 + 2]; ns += sN[wv]; }
        o_chunkE[chunk] = e;
        const v4i_t e3 = ldg4<DESC, (int) offsetof(ResSet, chunkCnt)>(&nxt);
        int32_t *o_cnt = GP(int32_t, e3, 0); float *o_nid = GP(float, e3, 1);
        o_cnt[chunk * 2 + 0] = na; o_cnt[chunk * 2 + 1] = nl;
        o_nid[chunk * 2 + 0] = ns; o_nid[chunk * 2 + 1] = (float) nc;
    }
}

template <int NSG, bool HAS_L, bool FIX, bool MARG>
__global__ __launch_bounds__(64 * LD_WAVES) void k_linearize(BaPtrs B, BaDims D, ResSet cur, ResSet nxt, ldso_settings_t S, int stepMode, GnInit gi,
                                                             const int32_t *__restrict__ margFlags) {
    static_assert(sizeof(BaPtrs) + sizeof(BaDims) + 2 * sizeof(ResSet) + sizeof(ldso_settings_t) + sizeof(GnInit) + 256 >= 14 * 64 - 60, "k_linearize: ld_touch_kernarg<14> must stay inside the explicit arguments + the 256 bytes of hidden arguments behind them (the kernel uses dynamic LDS: the whole hidden block is part of its kernarg segment)");
    ld_touch_kernarg<14>();          // 952 bytes of arguments into the scalar cache with one wait (ba_dev.h)
    const int chunk = (int) blockIdx.x;
    linearize_body<NSG, HAS_L, FIX, MARG, false>(B, D, cur, nxt, S, stepMode, gi, margFlags, chunk, (int) gridDim.x, B.chunk_p0[chunk], B.chunk_n[chunk], B.chunk_host[chunk]);
}

// Batched windows (one 7-keyframe window is tiny for the chip): the chunks of nWin independent windows in one
// launch.  Workgroup -> window by the prefix of chunk counts (items[w].linBlock0); `cur` = which residual set is the applied one
// (all windows of a batch iterate in lockstep).
template <int NSG>
__global__ __launch_bounds__(64 * LD_WAVES) void k_linearize_batch(const BatchItem *__restrict__ items, const BatchBlock *__restrict__ blocks, const int32_t *__restrict__ wgStart, int cur,
                                                                   ldso_settings_t S, int stepMode, float calibPrior, int itCheck) {
    // wgStart != nullptr (round 6): this workgroup works through the blocks [wgStart[w], wgStart[w + 1]) - ldso_ba_batch_create cut the windows so that every
    // workgroup of the launch (one per CU) carries the same load; nullptr: one block per workgroup
    int b0 = (int) blockIdx.x, b1 = b0 + 1;
    if (wgStart != nullptr) { b0 = wgStart[blockIdx.x]; b1 = wgStart[blockIdx.x + 1]; }
#pragma clang loop unroll(disable)
    for (int b = b0; b < b1; b++) {
        const BatchBlock bb = blocks[b];                            // one scalar 16-byte load: window, first point, point count, host | chunk
        const BatchItem &it = items[bb.win];
        GnInit gi; gi.enable = 1; gi.hasPrior = it.hasPrior; gi.calibPrior = calibPrior; gi.itCheck = itCheck;
        linearize_body<NSG, false, false, false, true>(it.B, it.D, it.set[cur], it.set[cur ^ 1], S, stepMode, gi, nullptr, bb.host_chunk >> 8, it.D.nChunks, bb.p0, bb.np, bb.host_chunk & 0xFF);
        __syncthreads();                                            // the next block re-uses the operand / reduction LDS
    }
}

// The GN-iteration linearisation of ONE window with everything a workgroup needs before its first operand load in the KERNEL ARGUMENTS: the whole
// descriptor (BaPtrs / BaDims / the two residual sets) by value - it stays in the kernarg segment, the body fetches its pointer groups from there just
// in time exactly as from a BatchItem in device memory (DESC = true: `&B` is an address in the constant kernarg segment) - and the chunk of the
// workgroup computed from the by-value host tables of LinHead instead of a table entry in memory.  The prologue of this kernel is a chain of dependent
// memory levels and nothing else (window of 5 MB on a chip that moves that in 0.7 us): kernel arguments -> workgroup table entry -> descriptor ->
// operands -> LDS was four levels (k_linearize_batch<NSG, false>), this is two.
struct OneArgs { BaPtrs B; BaDims D; ResSet cur, nxt; ldso_settings_t S; int stepMode; GnInit gi; LinHead hd; };
template <int NSG, bool PAIR = false>
__global__ __launch_bounds__(64 * LD_WAVES) void k_linearize_one(OneArgs a) {
    // The descriptor is addressed IN the kernarg segment (the only parameter starts at its offset 0): taking the address of the by-value parameter
    // itself would make the compiler copy it to scratch memory (536 bytes per lane, and a scalar load from a private address is meaningless).
    const OneArgs &A = *(const OneArgs *) __builtin_amdgcn_kernarg_segment_ptr();
    static_assert(sizeof(OneArgs) >= 12 * 64 && sizeof(OneArgs) <= 16 * 64, "k_linearize_one: 16 lines = the explicit arguments (>= 768 bytes) + the 256 bytes of hidden arguments behind them");
    ld_touch_kernarg<16>();          // the whole argument block into the scalar cache with one wait (ba_dev.h): 10.6 -> 9.9 us at C3, 43.0 -> 42.3 us at C5
    const LinHead &hd = a.hd;
    const int chunk = (int) blockIdx.x;
    int h = 0;
#pragma unroll
    for (int i = 1; i < LD_MAXF; i++) h += (i < hd.F && chunk >= hd.cs[i]) ? 1 : 0;          // hosts without points have cs[i] == cs[i + 1]: they are skipped
    int c0 = 0, q0 = 0, q1 = 0;
#pragma unroll
    for (int i = 0; i < LD_MAXF; i++) { const bool m = (i == h); c0 = m ? hd.cs[i] : c0; q0 = m ? hd.hostP0[i] : q0; q1 = m ? hd.hostP0[i + 1] : q1; }
    const int p0 = q0 + (chunk - c0) * hd.CH, np = min(hd.CH, q1 - p0);
    linearize_body<NSG, false, false, false, true, true, PAIR>(A.B, A.D, A.cur, A.nxt, a.S, a.stepMode, a.gi, nullptr, chunk, (int) gridDim.x, p0, np, h);
}

// ---------------------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------------------
size_t ba_linearize_lds_bytes(int FS, bool hasL, bool stash = false) {
    size_t fl = (size_t) FS * (sizeof(DevPair) / 4) + 2 * (size_t) FS * 64 + (size_t) LD_WAVES * FS * LD_TOPN + (hasL ? (size_t) LD_WAVES * FS * LD_TOPN : 0) + (size_t) FS * 8 + LD_TAIL_FLOATS + 2 * (size_t) FS;
    return fl * sizeof(float) + 256 + ((stash && FS == 8) ? (size_t) LD_WAVES * 4096 : 0);          // + the record stash of the batched one-slot-group kernel
}

template <int NSG, bool HAS_L, bool FIX, bool MARG = false>
static hipError_t launch_one(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, int stepMode, const GnInit &gi, hipStream_t st,
                            const int32_t *margFlags = nullptr) {
    size_t lds = ba_linearize_lds_bytes(D.FS, HAS_L);
    auto kfn = k_linearize<NSG, HAS_L, FIX, MARG>;
    if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipLaunchKernelGGL(kfn, dim3(D.nChunks), dim3(64 * LD_WAVES), lds, st, B, D, cur, nxt, S, stepMode, gi, margFlags);
    return hipGetLastError();
}

hipError_t ba_launch_linearize(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S,
                               bool hasL, bool fix, int stepMode, const GnInit &gi, hipStream_t st) {
    if (D.nChunks == 0) return hipSuccess;
    if (D.nsg == 1) {
        if (hasL) return fix ? launch_one<1, true, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<1, true, false>(B, D, cur, nxt, S, stepMode, gi, st);
        return fix ? launch_one<1, false, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<1, false, false>(B, D, cur, nxt, S, stepMode, gi, st);
    } else {
        if (hasL) return fix ? launch_one<2, true, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<2, true, false>(B, D, cur, nxt, S, stepMode, gi, st);
        return fix ? launch_one<2, false, true>(B, D, cur, nxt, S, stepMode, gi, st) : launch_one<2, false, false>(B, D, cur, nxt, S, stepMode, gi, st);
    }
}

hipError_t ba_launch_linearize_one(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, int stepMode, const GnInit &gi, const LinHead &hd,
                                   hipStream_t st) {
    if (D.nChunks == 0) return hipSuccess;
    const size_t lds = ba_linearize_lds_bytes(D.FS, false);
    OneArgs a;
    a.B = B; a.D = D; a.cur = cur; a.nxt = nxt; a.S = S; a.stepMode = stepMode; a.gi = gi; a.hd = hd;
    if (D.nsg == 1) {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_linearize_one<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL(k_linearize_one<1>, dim3(D.nChunks), dim3(64 * LD_WAVES), lds, st, a);
    } else if (D.F <= 12 && LD_PAIR) {
        // 9..12 key frames: the second slot group uses 4 of its 8 slots - two points share its pass (pair mode, round 6)
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_linearize_one<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL((k_linearize_one<2, true>), dim3(D.nChunks), dim3(64 * LD_WAVES), lds, st, a);
    } else {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_linearize_one<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL(k_linearize_one<2>, dim3(D.nChunks), dim3(64 * LD_WAVES), lds, st, a);
    }
    return hipGetLastError();
}

// marginalizePointsF accumulate for the flagged points (see the MARG note at k_linearize); `nxt` is scratch
hipError_t ba_launch_linearize_marg(const BaPtrs &B, const BaDims &D, const ResSet &cur, const ResSet &nxt, const ldso_settings_t &S, const int32_t *margFlags,
                                    hipStream_t st) {
    if (D.nChunks == 0) return hipSuccess;
    const GnInit gi{0, 0, 0.0f};
    if (D.nsg == 1) return launch_one<1, false, false, true>(B, D, cur, nxt, S, 0, gi, st, margFlags);
    return launch_one<2, false, false, true>(B, D, cur, nxt, S, 0, gi, st, margFlags);
}

hipError_t ba_launch_linearize_batch(const BatchItem *d_items, const BatchBlock *d_blocks, int totalChunks, const int32_t *d_wgStart, int nWG, int FS, int cur, const ldso_settings_t &S, int stepMode,
                                     float calibPrior, hipStream_t st, int itCheck) {
    if (totalChunks == 0) return hipSuccess;
    const size_t lds = ba_linearize_lds_bytes(FS, false, true);
    const int grid = d_wgStart != nullptr ? nWG : totalChunks;
    if (grid <= 0) return hipSuccess;
    if (FS == 8) {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_linearize_batch<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL(k_linearize_batch<1>, dim3(grid), dim3(64 * LD_WAVES), lds, st, d_items, d_blocks, d_wgStart, cur, S, stepMode, calibPrior, itCheck);
    } else {
        if (lds > 48 * 1024) (void) hipFuncSetAttribute((const void *) k_linearize_batch<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        hipLaunchKernelGGL(k_linearize_batch<2>, dim3(grid), dim3(64 * LD_WAVES), lds, st, d_items, d_blocks, d_wgStart, cur, S, stepMode, calibPrior, itCheck);
    }
    return hipGetLastError();
}
