// known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for 4-byte-per-lane accesses
// (the linearize kernel's access width).  rd4: reads N floats with dword loads; wr4: writes N floats with dword stores;
// gather12: reads 12-byte AoS pixels at random positions (4 taps x 3 floats like the bilinear gather).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void rd4(const float *in, float *out, size_t n) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; float s = 0;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) s += in[i];
    if (s == 12345.678f) out[0] = s;
}
__global__ void wr4(float *out, size_t n) {
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = (float) i;
}
__global__ void gather12(const float *img, const int *pos, float *out, int n, int w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    const float *bp = img + 3 * pos[i], *bq = bp + 3 * w; float s = 0;
    for (int c = 0; c < 6; c++) s += bp[c] + bq[c];
    if (s == 12345.678f) out[0] = s;
}
int main() {
    const size_t N = 64ull << 20;      // 256 MB of floats
    float *a, *b; hipMalloc(&a, N * 4); hipMalloc(&b, N * 4); hipMemset(a, 0, N * 4);
    hipLaunchKernelGGL(rd4, dim3(4096), dim3(256), 0, 0, a, b, N);
    hipLaunchKernelGGL(wr4, dim3(4096), dim3(256), 0, 0, b, N);
    const int w = 640, h = 480, n = 1 << 20; std::vector<int> p(n);
    unsigned s = 12345; for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; p[i] = (s >> 8) % (w * (h - 1) - 1); }
    int *dp; hipMalloc(&dp, n * 4); hipMemcpy(dp, p.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gather12, dim3(n / 256), dim3(256), 0, 0, a, dp, b, n, w);
    hipDeviceSynchronize();
    printf("rd4: %zu bytes read; wr4: %zu bytes written; gather12: %d taps x 48 B = %zu algorithmic bytes over a %d-byte image\n", N * 4, N * 4, n, (size_t) n * 48, w * h * 12);
    return 0;
}
