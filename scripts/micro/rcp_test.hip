// accuracy of v_rcp_f64 / v_rsq_f64 + Newton steps (run on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *d, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = d[i];
    double r0 = __builtin_amdgcn_rcp(x);
    double r1 = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
    double r2 = __builtin_fma(__builtin_fma(-x, r1, 1.0), r1, r1);
    double q0 = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * q0), q0, 1.0); double q1 = __builtin_fma(0.5 * q0, e, q0);
    e = __builtin_fma(-(x * q1), q1, 1.0); double q2 = __builtin_fma(0.5 * q1, e, q1);
    o[i * 6 + 0] = r0; o[i * 6 + 1] = r1; o[i * 6 + 2] = r2; o[i * 6 + 3] = q0; o[i * 6 + 4] = q1; o[i * 6 + 5] = q2;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n), o(n * 6);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = exp((u - 0.5) * 60.0); }
    double *d, *od; hipMalloc(&d, n * 8); hipMalloc(&od, n * 48);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, od, n);
    hipMemcpy(o.data(), od, n * 48, hipMemcpyDeviceToHost);
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        long double x = h[i], rc = 1.0L / x, rs = 1.0L / sqrtl(x);
        for (int j = 0; j < 3; j++) { double e = (double) fabsl((o[i * 6 + j] - rc) / rc); if (e > m[j]) m[j] = e; }
        for (int j = 3; j < 6; j++) { double e = (double) fabsl((o[i * 6 + j] - rs) / rs); if (e > m[j]) m[j] = e; }
    }
    printf("rcp: raw %.3e  1NR %.3e  2NR %.3e | rsq: raw %.3e 1NR %.3e 2NR %.3e (eps 1.1e-16)\n", m[0], m[1], m[2], m[3], m[4], m[5]);
    return 0;
}
