// fp64 VALU issue rates on gfx950, one wave per SIMD (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#define FENCE(x) asm volatile("" :: "v"(x)); __builtin_amdgcn_sched_barrier(0)
__global__ __launch_bounds__(256) void k(double *o, long long *t, const double *in) {
    const int tid = threadIdx.x;
    double a[8], b[8], c[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = in[tid + i * 256]; b[i] = in[tid + 2048 + i * 256]; c[i] = in[tid + 4096 + i * 256]; }
    FENCE(a[7]); FENCE(c[7]); FENCE(b[7]);
    long long c0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 32; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_fma(b[i], c[(i + r) & 7], a[i]);      // 256 independent-ish fma, 3 VGPR operands
#pragma unroll
    for (int i = 0; i < 8; i++) { FENCE(a[i]); }
    long long c1 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 32; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = b[i] * c[(i + r) & 7];                          // 256 mul
#pragma unroll
    for (int i = 0; i < 8; i++) { FENCE(b[i]); }
    long long c2 = clock64(); __builtin_amdgcn_sched_barrier(0);
    float fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { fa[i] = (float) a[i]; fb[i] = (float) b[i]; }
#pragma unroll
    for (int i = 0; i < 8; i++) { FENCE(fa[i]); FENCE(fb[i]); }
    long long c3 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 32; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = __builtin_fmaf(fb[i], fb[(i + r) & 7], fa[i]);  // 256 f32 fma
#pragma unroll
    for (int i = 0; i < 8; i++) { FENCE(fa[i]); }
    long long c4 = clock64(); __builtin_amdgcn_sched_barrier(0);
    // 64-bit selects
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = (tid > r + i) ? a[i] : c[i];
#pragma unroll
    for (int i = 0; i < 8; i++) { FENCE(c[i]); }
    long long c5 = clock64(); __builtin_amdgcn_sched_barrier(0);
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + b[i] + c[i] + fa[i];
    o[tid] = s;
    if (tid == 0) { t[0] = c1 - c0; t[1] = c2 - c1; t[2] = c4 - c3; t[3] = c5 - c4; }
}
int main() {
    double *o, *in; long long *t; hipMalloc(&o, 256 * 8); hipMalloc(&t, 64); hipMalloc(&in, 8192 * 8);
    hipMemset(in, 0, 8192 * 8);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, t, in);
    long long h[8]; hipMemcpy(h, t, 32, hipMemcpyDeviceToHost);
    printf("cycles per instruction (throughput, 1 wave/SIMD): fma_f64 %.2f | mul_f64 %.2f | fma_f32 %.2f | select64 (cmp+2 cndmask) %.2f\n",
           h[0] / 256.0, h[1] / 256.0, h[2] / 256.0, h[3] / 128.0);
    return 0;
}
