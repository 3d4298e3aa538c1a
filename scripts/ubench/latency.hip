// Dependent-issue latency of the operations on the serial critical paths (LDL^T pivots, LM solves), one wavefront, gfx950.
// hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP> __global__ void k(double *out, long long *t, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    float xf = (float) x;
    long long t0 = wall_clock64(), c0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        if (OP == 0) x = __builtin_fma(x, y, 1e-9);                       // v_fma_f64
        if (OP == 1) x = __builtin_amdgcn_rcp(x) + 0.0;                    // v_rcp_f64 (+ add)
        if (OP == 2) x = __builtin_amdgcn_rcp(x);                          // v_rcp_f64
        if (OP == 3) xf = __builtin_fmaf(xf, 1.0000001f, 1e-9f);           // v_fma_f32
        if (OP == 4) xf = __builtin_amdgcn_rcpf(xf);                       // v_rcp_f32
        if (OP == 5) x = (double) (float) x;                               // cvt f64->f32->f64
        if (OP == 6) { unsigned long long u = __builtin_bit_cast(unsigned long long, x);
                       unsigned lo = __builtin_amdgcn_readlane((int) u, 3), hi = __builtin_amdgcn_readlane((int) (u >> 32), 3);
                       x = __builtin_bit_cast(double, ((unsigned long long) hi << 32) | lo) + 1e-9; }   // readlane x2 + add
        if (OP == 7) { int v = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) << 2, __builtin_bit_cast(int, xf)); xf = __builtin_bit_cast(float, v) + 1e-9f; }  // ds_bpermute + add
        if (OP == 8) { int v = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xf), 0xB1, 0xF, 0xF, true); xf = __builtin_bit_cast(float, v) + 1e-9f; }        // dpp mov + add
        if (OP == 9) x = 1.0 / x;                                          // IEEE division
        if (OP == 10) x = sqrt(x + 2.0);
        if (OP == 11) x = sin(x);
    }
    long long t1 = wall_clock64(), c1 = clock64();
    out[threadIdx.x] = x + xf;
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = c1 - c0; }
}
int main() {
    double *o; long long *t, h[2];
    hipMalloc(&o, 64 * 8); hipMalloc(&t, 16);
    const char *names[] = {"v_fma_f64", "v_rcp_f64+add", "v_rcp_f64", "v_fma_f32", "v_rcp_f32", "cvt f64->f32->f64", "2x readlane+add_f64", "ds_bpermute+add_f32", "dpp mov+add_f32", "fp64 division", "fp64 sqrt+add", "fp64 sin"};
#define RUN(OP) for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, o, t, 1.3); hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); \
        if (rep) printf("%-22s %7.2f ns/op   (s_memtime ticks/op %.2f)\n", names[OP], h[0] * 10.0 / N, (double) h[1] / N); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    return 0;
}
