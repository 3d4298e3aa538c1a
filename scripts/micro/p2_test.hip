// phase-2 pattern of the LDL^T (10 tiles x C dependent fma with negated operand) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define FENCE(x) asm volatile("" :: "v"(x)); __builtin_amdgcn_sched_barrier(0)
template <int C> __device__ void run(double *v, const double (*fi)[C], const double (*gj)[C]) {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) {
            double w = v[a * (a + 1) / 2 + b];
#pragma unroll
            for (int q = 0; q < C; q++) w = __builtin_fma(-fi[a][q], gj[b][q], w);
            v[a * (a + 1) / 2 + b] = w;
        }
}
__global__ __launch_bounds__(256) void k(double *o, long long *t, const double *in) {
    const int tid = threadIdx.x;
    double v[10], fi[4][8], gj[4][8];
    for (int i = 0; i < 10; i++) v[i] = in[tid + i * 256];
    for (int a = 0; a < 4; a++) for (int q = 0; q < 8; q++) { fi[a][q] = in[tid + (a * 8 + q) * 256]; gj[a][q] = in[tid + 8192 + (a * 8 + q) * 256]; }
    for (int i = 0; i < 10; i++) { FENCE(v[i]); }
    for (int a = 0; a < 4; a++) for (int q = 0; q < 8; q++) { FENCE(fi[a][q]); FENCE(gj[a][q]); }
    long long c0 = clock64(); __builtin_amdgcn_sched_barrier(0);
    run<8>(v, fi, gj);
    for (int i = 0; i < 10; i++) { FENCE(v[i]); }
    long long c1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    run<8>(v, fi, gj);
    for (int i = 0; i < 10; i++) { FENCE(v[i]); }
    long long c2 = clock64(); __builtin_amdgcn_sched_barrier(0);
    double s = 0; for (int i = 0; i < 10; i++) s += v[i];
    o[tid] = s;
    if (tid == 0) { t[0] = c1 - c0; t[1] = c2 - c1; }
}
int main() {
    double *o, *in; long long *t; hipMalloc(&o, 256 * 8); hipMalloc(&t, 64); hipMalloc(&in, 16384 * 8);
    hipMemset(in, 0, 16384 * 8);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, t, in);
        long long h[8]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("rep %d: 80 fma: first pass %lld cycles (%.2f/instr), second pass %lld (%.2f/instr)\n", rep, h[0], h[0] / 80.0, h[1], h[1] / 80.0);
    }
    return 0;
}
