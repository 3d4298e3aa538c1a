// images.hip — FrameHessian::makeImages on the device (reference src/internal/FrameHessian.cc:44-113): from the level-0
// irradiance image build the pyramid of 12-byte AoS pixels (I, dx, dy) that the bundle adjustment (level 0) and the coarse
// tracker (all levels) sample.  Level l >= 1 is the 2x2 MEAN of level l-1 (:72-78); gradients are central differences over
// the flat index range [w, w(h-1)) including the row wrap at x = 0 / w-1 (:81-89); rows 0 and h-1 keep zero gradients (the
// reference leaves them uninitialised and never samples them).  Pure streaming: 4 B in / 12 B out per pixel at level 0.
// absSquaredGrad (pixel selector only) is not produced.
#include <hip/hip_runtime.h>

// channel 0 of a level (copy of the input or 2x2 mean of the level below), gradients zeroed
__global__ __launch_bounds__(256) void k_img_intensity(const float *__restrict__ color, const float *__restrict__ below, float *__restrict__ dI, int wl, int hl, int wlm1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wl * hl) return;
    float v;
    if (below == nullptr) v = color[i];
    else {
        const int x = i % wl, y = i / wl;
        const float *p = below + 3 * (size_t) (2 * x + 2 * y * wlm1);
        v = 0.25f * (((p[0] + p[3]) + p[3 * wlm1]) + p[3 * wlm1 + 3]);
    }
    dI[3 * (size_t) i] = v; dI[3 * (size_t) i + 1] = 0.0f; dI[3 * (size_t) i + 2] = 0.0f;
}

__global__ __launch_bounds__(256) void k_img_gradient(float *__restrict__ dI, int wl, int hl) {
    const int idx = wl + blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= wl * (hl - 1)) return;
    float dx = 0.5f * (dI[3 * (size_t) (idx + 1)] - dI[3 * (size_t) (idx - 1)]);
    float dy = 0.5f * (dI[3 * (size_t) (idx + wl)] - dI[3 * (size_t) (idx - wl)]);
    if (!(fabsf(dx) <= 255.0f)) dx = 0;        // NaN or |.| > 255 (FrameHessian.cc:86-87)
    if (!(fabsf(dy) <= 255.0f)) dy = 0;
    dI[3 * (size_t) idx + 1] = dx;
    dI[3 * (size_t) idx + 2] = dy;
}

// d_color: w*h floats on the device; d_levels[l]: (w>>l)*(h>>l)*3 floats on the device
extern "C" hipError_t img_launch_make_images(const float *d_color, int w, int h, int levels, float *const *d_levels, hipStream_t st) {
    for (int l = 0; l < levels; l++) {
        const int wl = w >> l, hl = h >> l, n = wl * hl;
        hipLaunchKernelGGL(k_img_intensity, dim3((n + 255) / 256), dim3(256), 0, st, l == 0 ? d_color : nullptr, l == 0 ? nullptr : d_levels[l - 1], d_levels[l], wl, hl, w >> (l > 0 ? l - 1 : 0));
        const int m = wl * (hl - 2);
        if (m > 0) hipLaunchKernelGGL(k_img_gradient, dim3((m + 255) / 256), dim3(256), 0, st, d_levels[l], wl, hl);
    }
    return hipGetLastError();
}
