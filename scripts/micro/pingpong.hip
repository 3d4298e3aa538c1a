// Round trip of one 64-bit word between two workgroups (one lane each), gfx950: what a hand-over between cooperating workgroups costs, by the kind of access and by where
// the two workgroups sit.  Workgroups go to the XCDs round robin (block b -> XCD b % 8, printed from HW_REG_XCC_ID): block 0 <-> block 8 share an XCD (one L2), block 0 <-> block 1
// do not.
//   mode 0  device-scope relaxed atomic store / load (sc1: performed at the memory side) - what the tracker's hand-over words use
//   mode 1  plain vector store (writes through the CU's L1 into the XCD's L2), polled with a SCALAR load behind s_dcache_inv (the scalar cache misses into the L2)
//   mode 2  plain vector store, polled with a plain vector load behind buffer_inv sc1 (invalidates the L1)
//   mode 3  plain vector store, polled with a vector load carrying sc0 (workgroup scope: served by the L1 - expected never to see the other CU's store)
// A mode that cannot work for a placement (stale data) runs into the spin limit and is reported as "no hand-over".
// hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2000
#define SPIN_LIMIT 3000000
typedef unsigned long long u64;
__device__ __forceinline__ void st_plain(u64 *p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u64 ld_scalar_inv(const u64 *p) {
    u64 v;
    asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ u64 ld_inv_vec(const u64 *p) {
    u64 v;
    asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u64 ld_sc0(const u64 *p) {
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE> __device__ __forceinline__ void put(u64 *p, u64 v) {
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else st_plain(p, v);
}
template <int MODE> __device__ __forceinline__ u64 get(const u64 *p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) return ld_scalar_inv(p);
    if (MODE == 2) return ld_inv_vec(p);
    return ld_sc0(p);
}
template <int MODE> __global__ void k(u64 *words /* [0]: leader -> partner, [16]: partner -> leader (own 128-byte lines) */, int partner, int *xcc, long long *res) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[b] = (int) (id & 15); }
    if (b != 0 && b != partner) return;
    if (threadIdx.x != 0) return;
    u64 *a = words, *r = words + 16;
    int fail = 0;
    if (b == 0) {
        const long long t0 = wall_clock64();
        for (int i = 1; i <= N && !fail; i++) {
            put<MODE>(a, (u64) i);
            int spins = 0;
            while (get<MODE>(r) != (u64) i) if (++spins > SPIN_LIMIT) { fail = 1; break; }
        }
        res[0] = wall_clock64() - t0; res[1] = fail;
        if (fail) put<0>(a, ~0ull);                          // release the partner
    } else {
        for (int i = 1; i <= N && !fail; i++) {
            int spins = 0;
            for (;;) { const u64 v = get<MODE>(a); if (v == (u64) i) break; if (v == ~0ull || ++spins > 2 * SPIN_LIMIT) { fail = 1; break; } if (MODE != 0 && (spins & 1023) == 0 && get<0>(a) == ~0ull) { fail = 1; break; } }
            if (!fail) put<MODE>(r, (u64) i);
        }
    }
}
template <int MODE> static void run(const char *name, u64 *words, int *xcc, long long *res) {
    for (int partner : {8, 1}) {
        hipMemset(words, 0, 64 * 8); hipMemset(res, 0, 16);
        hipLaunchKernelGGL(k<MODE>, dim3(16), dim3(64), 0, 0, words, partner, xcc, res);
        hipError_t e = hipDeviceSynchronize();
        long long h[2]; int x[16];
        hipMemcpy(h, res, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 64, hipMemcpyDeviceToHost);
        printf("%-58s block 0 (XCD %d) <-> block %d (XCD %d): ", name, x[0], partner, x[partner]);
        if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
        else if (h[1]) printf("no hand-over (spin limit)\n");
        else printf("%7.1f ns per round trip\n", h[0] * 10.0 / N);
    }
}
int main() {
    u64 *words; int *xcc; long long *res;
    hipMalloc(&words, 64 * 8); hipMalloc(&xcc, 64); hipMalloc(&res, 16);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("device-scope atomic store / load", words, xcc, res);
        run<1>("plain store / scalar load behind s_dcache_inv", words, xcc, res);
        run<2>("plain store / vector load behind buffer_inv sc1", words, xcc, res);
        run<3>("plain store / vector load sc0", words, xcc, res);
    }
    return 0;
}
