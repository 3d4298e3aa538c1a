// dependent-issue latencies on gfx950, one wave per SIMD (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double *o, long long *t, double seed) {
    __shared__ double sh[1024];
    const int tid = threadIdx.x;
    double x = seed + tid * 1e-9, y = 1.0000001;
    asm volatile("" :: "v"(x)); __builtin_amdgcn_sched_barrier(0);
    long long c0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 256; i++) x = __builtin_fma(x, y, 1e-9);
    asm volatile("" :: "v"(x)); __builtin_amdgcn_sched_barrier(0);
    long long c1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    double r = x;
#pragma unroll
    for (int i = 0; i < 64; i++) r = __builtin_amdgcn_rcp(r) + 1.0;
    asm volatile("" :: "v"(r)); __builtin_amdgcn_sched_barrier(0);
    long long c2 = clock64(); __builtin_amdgcn_sched_barrier(0);
    float f = (float) r;
#pragma unroll
    for (int i = 0; i < 256; i++) f = __builtin_fmaf(f, 1.0000001f, 1e-9f);
    asm volatile("" :: "v"(f)); __builtin_amdgcn_sched_barrier(0);
    long long c3 = clock64(); __builtin_amdgcn_sched_barrier(0);
    // LDS round trip chain
    sh[tid] = f; sh[tid + 256] = r;
    __syncthreads();
    int idx = tid;
    __builtin_amdgcn_sched_barrier(0);
    long long c4 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; i++) { idx = (int) sh[idx] & 255; }
    asm volatile("" :: "v"(idx)); __builtin_amdgcn_sched_barrier(0);
    long long c5 = clock64(); __builtin_amdgcn_sched_barrier(0);
    double m = x;
#pragma unroll
    for (int i = 0; i < 128; i++) m = m * 1.0000001;
    asm volatile("" :: "v"(m)); __builtin_amdgcn_sched_barrier(0);
    long long c6 = clock64(); __builtin_amdgcn_sched_barrier(0);
    // barrier cost
#pragma unroll
    for (int i = 0; i < 16; i++) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    long long c7 = clock64(); __builtin_amdgcn_sched_barrier(0);
    double z = m;
#pragma unroll
    for (int i = 0; i < 64; i++) z = (z > 0.5) ? z * 0.999 : 0.0;      // mul + cmp + cndmask
    asm volatile("" :: "v"(z)); __builtin_amdgcn_sched_barrier(0);
    long long c8 = clock64(); __builtin_amdgcn_sched_barrier(0);
    o[tid] = x + r + f + idx + m + z;
    if (tid == 0) { t[0] = c1 - c0; t[1] = c2 - c1; t[2] = c3 - c2; t[3] = c5 - c4; t[4] = c6 - c5; t[5] = c7 - c6; t[6] = c8 - c7; }
}
int main() {
    double *o; long long *t; hipMalloc(&o, 256 * 8); hipMalloc(&t, 64);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, t, 0.5);
    long long h[8]; hipMemcpy(h, t, 56, hipMemcpyDeviceToHost);
    printf("per op cycles: fma_f64 %.1f | rcp_f64+add %.1f | fma_f32 %.1f | lds dependent read %.1f | mul_f64 %.1f | barrier %.1f | mul+cmp+cndmask f64 %.1f\n",
           h[0] / 256.0, h[1] / 64.0, h[2] / 256.0, h[3] / 32.0, h[4] / 128.0, h[5] / 16.0, h[6] / 64.0);
    return 0;
}
