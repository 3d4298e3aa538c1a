// images.hip — FrameHessian::makeImages on the device (reference src/internal/FrameHessian.cc:44-113): from the level-0
// irradiance image build the pyramid of 12-byte AoS pixels (I, dx, dy) that the bundle adjustment (level 0) and the coarse
// tracker (all levels) sample.  Level l >= 1 is the 2x2 MEAN of level l-1 (:72-78); gradients are central differences over
// the flat index range [w, w(h-1)) including the row wrap at x = 0 / w-1 (:81-89); rows 0 and h-1 keep zero gradients (the
// reference leaves them uninitialised and never samples them).  Pure streaming: 4 B in / 12 B out per pixel at level 0.
// absSquaredGrad (pixel selector only) is not produced.
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/ldso_hip.h"
#include "pyramid.h"

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

// ---------------------------------------------------------------------------------------------------------
// ldso_pyramid_t: FrameHessian::dIp of one frame, resident in HBM and shared by pointer - what the reference does on the host
// (CoarseTracker::setCoarseTrackingRef / trackNewestCoarse read fh->dIp[lvl], ImmaturePoint::traceOn and
// PointFrameResidual::linearize read fh->dI = dIp[0]).  One 4-byte-per-pixel upload per frame, every consumer zero-copy.
// ---------------------------------------------------------------------------------------------------------
void ldso_set_error(const std::string &s);
#define PCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ldso_set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return LDSO_E_HIP; } } while (0)
#define PREQ(c, msg) do { if (!(c)) { ldso_set_error(msg); return LDSO_E_INVALID; } } while (0)

extern "C" {

int ldso_pyr_create(int device, int w, int h, int levels, ldso_pyramid_t **out) {
    PREQ(out && w > 16 && h > 16 && levels >= 1 && levels <= LDSO_PYR_LEVELS && (w >> (levels - 1)) > 2 && (h >> (levels - 1)) > 2, "ldso_pyr_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { ldso_set_error("no HIP device visible"); return LDSO_E_NODEVICE; }
    PREQ(device >= 0 && device < ndev, "ldso_pyr_create: device index out of range");
    PCHK(hipSetDevice(device));
    ldso_pyramid *P = new ldso_pyramid();
    P->device = device; P->w = w; P->h = h; P->levels = levels;
    size_t total = 0;
    for (int l = 0; l < levels; l++) total += (size_t) (w >> l) * (h >> l) * 3;
    // ONE allocation for all levels (each level 16-byte aligned) + the raw image
    size_t off = 0, offs[LDSO_PYR_LEVELS];
    for (int l = 0; l < levels; l++) { offs[l] = off; off += (((size_t) (w >> l) * (h >> l) * 3) + 3) & ~(size_t) 3; }
    if (hipMalloc((void **) &P->base, (off + (size_t) w * h) * sizeof(float)) != hipSuccess) { delete P; ldso_set_error("ldso_pyr_create: out of device memory"); return LDSO_E_HIP; }
    for (int l = 0; l < levels; l++) P->lv[l] = P->base + offs[l];
    P->d_color = P->base + off;
    if (hipEventCreateWithFlags(&P->ready, hipEventDisableTiming) != hipSuccess) { hipFree(P->base); delete P; ldso_set_error("ldso_pyr_create: hipEventCreate failed"); return LDSO_E_HIP; }
    *out = P;
    return LDSO_OK;
}

int ldso_pyr_destroy(ldso_pyramid_t *P) {
    if (!P) return LDSO_OK;
    hipSetDevice(P->device);
    hipDeviceSynchronize();          // consumers may still be reading the levels
    if (P->ready) hipEventDestroy(P->ready);
    hipFree(P->base);
    delete P;
    return LDSO_OK;
}

// FrameHessian::makeImages (FrameHessian.cc:44-113) from a device-resident irradiance image: enqueued on `hip_stream`, consumers wait
// on the pyramid's event (no host synchronisation).
int ldso_pyr_make_images_device(ldso_pyramid_t *P, const void *irradiance_dev, void *hip_stream) {
    PREQ(P && irradiance_dev, "ldso_pyr_make_images_device: bad arguments");
    PCHK(hipSetDevice(P->device));
    hipStream_t st = (hipStream_t) hip_stream;
    PCHK(img_launch_make_images((const float *) irradiance_dev, P->w, P->h, P->levels, P->lv, st));
    PCHK(hipEventRecord(P->ready, st));
    P->built = true;
    return LDSO_OK;
}

// the same from a host image (w*h floats): 4 bytes per pixel cross PCIe, once per frame
int ldso_pyr_make_images(ldso_pyramid_t *P, const float *irradiance, void *hip_stream) {
    PREQ(P && irradiance, "ldso_pyr_make_images: bad arguments");
    PCHK(hipSetDevice(P->device));
    hipStream_t st = (hipStream_t) hip_stream;
    PCHK(hipMemcpyAsync(P->d_color, irradiance, (size_t) P->w * P->h * sizeof(float), hipMemcpyHostToDevice, st));
    return ldso_pyr_make_images_device(P, P->d_color, hip_stream);
}

int ldso_pyr_level(ldso_pyramid_t *P, int lvl, const void **dev_ptr, int *wl, int *hl) {
    PREQ(P && lvl >= 0 && lvl < P->levels, "ldso_pyr_level: bad arguments");
    if (dev_ptr) *dev_ptr = P->lv[lvl];
    if (wl) *wl = P->w >> lvl;
    if (hl) *hl = P->h >> lvl;
    return LDSO_OK;
}

int ldso_pyr_get_level(ldso_pyramid_t *P, int lvl, float *out) {
    PREQ(P && out && lvl >= 0 && lvl < P->levels && P->built, "ldso_pyr_get_level: bad arguments (or no image yet)");
    PCHK(hipSetDevice(P->device));
    PCHK(hipEventSynchronize(P->ready));
    PCHK(hipMemcpy(out, P->lv[lvl], (size_t) (P->w >> lvl) * (P->h >> lvl) * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return LDSO_OK;
}

}  // extern "C"
