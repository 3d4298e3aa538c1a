// trace.hip — immature-point tracing on the device (gfx950): FullSystem::traceNewCoarse (reference
// src/frontend/FullSystem.cc:1012-1050) = ImmaturePoint::traceOn (src/internal/ImmaturePoint.cc:47-310) for every immature point
// of the window against a new frame, ONE launch.
//
// Mapping: one wavefront per immature point.  The discrete epipolar search is the parallel part: lane l evaluates the 8-pattern
// Huber energy at search steps l and l+64 (numSteps <= 99) - the position of step i is reached with i sequential float additions
// exactly like the reference's `ptx += dx` loop, the 8 pattern terms are summed in pattern order, so every energy is bit-identical
// to the CPU path; the best step (first minimum) and the second-best outside the +-radius window are wave reductions.  The
// short Gauss-Newton refinement (<= 3 iterations of 8 samples) and the scalar bookkeeping run uniformly in all lanes.
// Memory: per point 128 B record in / out + 4-byte taps of the level-0 image along the epipolar line (L2 resident).
#include <hip/hip_runtime.h>
#include <vector>
#include <string>
#include <cstring>
#include <cmath>
#include "../../include/ldso_hip.h"

void ldso_set_error(const std::string &s);
#include "pyramid.h"

struct TraceArgs {
    ldso_immature_t *pts;
    int n;
    const float *img;       // level-0 image of the new frame, 12-byte AoS (I, dx, dy)
    int w, h;
    const float *KRKi, *Kt, *aff;      // per host: 9, 3, 2 floats
    int nHosts;
    ldso_trace_settings_t s;
    int *counts;            // [6] per resulting status
};

// getInterpolatedElement31 (GlobalFuncs.h:146-159)
static __device__ __forceinline__ float interp31(const float *img, float x, float y, int w) {
    const int ix = (int) x, iy = (int) y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float *bp = img + 3 * (ix + iy * w);
    return dxdy * bp[3 + 3 * w] + (dy - dxdy) * bp[3 * w] + (dx - dxdy) * bp[3] + (1 - dx - dy + dxdy) * bp[0];
}

static __device__ __forceinline__ void trace_fail(ldso_immature_t *p, int lane, int status, int *counts) {
    if (lane == 0) { p->lastTraceUV[0] = -1; p->lastTraceUV[1] = -1; p->lastTracePixelInterval = 0; p->lastTraceStatus = status; atomicAdd(&counts[status], 1); }
}

__global__ __launch_bounds__(256) void k_trace_on(TraceArgs A) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane((int) ((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (i >= A.n) return;
    ldso_immature_t *P = A.pts + i;
    const int hst = P->host;
    if (hst < 0 || hst >= A.nHosts) return;
    const ldso_trace_settings_t &S = A.s;
    const int w = A.w, h = A.h;
    const int prevStatus = P->lastTraceStatus;
    if (prevStatus == LDSO_IPS_OOB) { if (lane == 0) atomicAdd(&A.counts[LDSO_IPS_OOB], 1); return; }     // :53
    const float *KRKi = A.KRKi + 9 * hst, *Kt = A.Kt + 3 * hst, *aff = A.aff + 2 * hst;
    const float pu = P->u, pv = P->v, idmin = P->idepth_min, idmax = P->idepth_max;
    const float maxPixSearch = (w + h) * S.maxPixSearch;
    // ---- project idepth_min / idepth_max (:59-124) ----
    const float pr0 = (KRKi[0] * pu + KRKi[1] * pv) + KRKi[2] * 1.0f, pr1 = (KRKi[3] * pu + KRKi[4] * pv) + KRKi[5] * 1.0f,
                pr2 = (KRKi[6] * pu + KRKi[7] * pv) + KRKi[8] * 1.0f;
    const float m0 = pr0 + Kt[0] * idmin, m1 = pr1 + Kt[1] * idmin, m2 = pr2 + Kt[2] * idmin;
    const float uMin = m0 / m2, vMin = m1 / m2;
    if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) { trace_fail(P, lane, LDSO_IPS_OOB, A.counts); return; }
    float dist, uMax, vMax;
    const bool finiteMax = isfinite(idmax);
    if (finiteMax) {
        const float x0 = pr0 + Kt[0] * idmax, x1 = pr1 + Kt[1] * idmax, x2 = pr2 + Kt[2] * idmax;
        uMax = x0 / x2; vMax = x1 / x2;
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) { trace_fail(P, lane, LDSO_IPS_OOB, A.counts); return; }
        dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
        dist = sqrtf(dist);
        if (dist < S.trace_slackInterval) {                                                          // :91-96
            if (lane == 0) { P->lastTraceUV[0] = (uMax + uMin) * 0.5f; P->lastTraceUV[1] = (vMax + vMin) * 0.5f; P->lastTracePixelInterval = dist;
                             P->lastTraceStatus = LDSO_IPS_SKIPPED; atomicAdd(&A.counts[LDSO_IPS_SKIPPED], 1); }
            return;
        }
    } else {
        dist = maxPixSearch;
        const float q0 = pr0 + Kt[0] * 0.01f, q1 = pr1 + Kt[1] * 0.01f, q2 = pr2 + Kt[2] * 0.01f;
        uMax = q0 / q2; vMax = q1 / q2;
        const float ddx = uMax - uMin, ddy = vMax - vMin;
        const float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
        uMax = uMin + dist * ddx * d; vMax = vMin + dist * ddy * d;
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) { trace_fail(P, lane, LDSO_IPS_OOB, A.counts); return; }
    }
    if (!(idmin < 0 || (m2 > 0.75f && m2 < 1.5f))) { trace_fail(P, lane, LDSO_IPS_OOB, A.counts); return; }      // :127-131
    // ---- error bound in pixels (:134-147) ----
    float dx = S.trace_stepsize * (uMax - uMin), dy = S.trace_stepsize * (vMax - vMin);
    const float g0 = P->gradH[0], g1 = P->gradH[1], g2 = P->gradH[2], g3 = P->gradH[3];
    const float a = (dx * g0 + dy * g2) * dx + (dx * g1 + dy * g3) * dy;
    const float b = (dy * g0 + -dx * g2) * dy + (dy * g1 + -dx * g3) * -dx;
    float errorInPixel = 0.2f + 0.2f * (a + b) / a;
    if (errorInPixel * S.trace_minImprovementFactor > dist && finiteMax) {
        if (lane == 0) { P->lastTraceUV[0] = (uMax + uMin) * 0.5f; P->lastTraceUV[1] = (vMax + vMin) * 0.5f; P->lastTracePixelInterval = dist;
                         P->lastTraceStatus = LDSO_IPS_BADCONDITION; atomicAdd(&A.counts[LDSO_IPS_BADCONDITION], 1); }
        return;
    }
    if (errorInPixel > 10) errorInPixel = 10;
    // ---- discrete search (:150-205), lanes = steps ----
    dx /= dist; dy /= dist;
    if (dist > maxPixSearch) { uMax = uMin + maxPixSearch * dx; vMax = vMin + maxPixSearch * dy; dist = maxPixSearch; }
    int numSteps = (int) (1.9999f + dist / S.trace_stepsize);
    const float randShift = uMin * 1000 - floorf(uMin * 1000);
    const float ptx0 = uMin - randShift * dx, pty0 = vMin - randShift * dy;
    float rx[8], ry[8], col[8];
    {
        const int ox[8] = {0, -1, 1, -2, 0, 2, -1, 0}, oy[8] = {-2, -1, -1, 0, 0, 0, 1, 2};           // staticPattern[8], Setting.cc:221
#pragma unroll
        for (int k = 0; k < 8; k++) { rx[k] = KRKi[0] * ox[k] + KRKi[1] * oy[k]; ry[k] = KRKi[3] * ox[k] + KRKi[4] * oy[k]; col[k] = P->color[k]; }
    }
    if (!isfinite(dx) || !isfinite(dy)) { trace_fail(P, lane, LDSO_IPS_OOB, A.counts); return; }
    if (numSteps >= 100) numSteps = 99;
    const float aff0 = aff[0], aff1 = aff[1];
    float e[2], px[2], py[2];
    {
        // position of step `lane`, then 64 more additions for step lane + 64 (sequential float accumulation as in the reference loop)
        float x = ptx0, y = pty0;
        for (int j = 0; j < 63; j++) if (j < lane) { x += dx; y += dy; }
        px[0] = x; py[0] = y;
        for (int j = 0; j < 64; j++) { x += dx; y += dy; }
        px[1] = x; py[1] = y;
    }
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int si = lane + 64 * pass;
        float energy = 0;
        if (si < numSteps) {
            float hit[8];
#pragma unroll
            for (int k = 0; k < 8; k++) hit[k] = interp31(A.img, px[pass] + rx[k], py[pass] + ry[k], w);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!isfinite(hit[k])) { energy += 1e5f; continue; }
                const float r = hit[k] - (float) (aff0 * col[k] + aff1);
                const float hw = fabsf(r) < S.huberTH ? 1 : S.huberTH / fabsf(r);
                energy += hw * r * r * (2 - hw);
            }
        }
        e[pass] = energy;
    }
    // best step: first minimum below 1e10 (strict < in index order)
    float bE; int bI;
    {
        const bool v0 = lane < numSteps && e[0] < 1e10f, v1 = lane + 64 < numSteps && e[1] < 1e10f;
        bE = v0 ? e[0] : 3e38f; bI = v0 ? lane : 1 << 20;
        if (v1 && e[1] < bE) { bE = e[1]; bI = lane + 64; }
        for (int o = 32; o > 0; o >>= 1) {
            const float e2 = __shfl_xor(bE, o, 64); const int i2 = __shfl_xor(bI, o, 64);
            if (e2 < bE || (e2 == bE && i2 < bI)) { bE = e2; bI = i2; }
        }
    }
    float bestU = 0, bestV = 0, bestEnergy = 1e10f;
    int bestIdx = -1;
    if (bI < (1 << 20)) {
        bestIdx = bI; bestEnergy = bE;
        const float sx = (bI >= 64) ? px[1] : px[0], sy = (bI >= 64) ? py[1] : py[0];
        bestU = __shfl(sx, bI & 63, 64); bestV = __shfl(sy, bI & 63, 64);
    }
    float secondBest = 1e10f;                                                                        // :208-215
    {
        float sb = 1e10f;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int si = lane + 64 * pass;
            if (si < numSteps && (si < bestIdx - S.minTraceTestRadius || si > bestIdx + S.minTraceTestRadius) && e[pass] < sb) sb = e[pass];
        }
        for (int o = 32; o > 0; o >>= 1) sb = fminf(sb, __shfl_xor(sb, o, 64));
        secondBest = sb;
    }
    float quality = P->quality;
    const float newQuality = secondBest / bestEnergy;
    if (newQuality < quality || numSteps > 10) quality = newQuality;
    // ---- Gauss-Newton refinement along the line (:218-268), uniform in all lanes ----
    float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
    if (S.trace_GNIterations > 0) bestEnergy = 1e5f;
    float wgt[8];
#pragma unroll
    for (int k = 0; k < 8; k++) wgt[k] = P->weights[k];
    for (int it = 0; it < S.trace_GNIterations; it++) {
        float H = 1, bb = 0, energy = 0;
        float h0[8], h1[8], h2[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float x = bestU + rx[k], y = bestV + ry[k];
            const int ix = (int) x, iy = (int) y;
            const float fx = x - ix, fy = y - iy, fxy = fx * fy;
            const float *bp = A.img + 3 * (ix + iy * w);
            const float w11 = fxy, w01 = fy - fxy, w10 = fx - fxy, w00 = 1 - fx - fy + fxy;
            h0[k] = w11 * bp[3 + 3 * w] + w01 * bp[3 * w] + w10 * bp[3] + w00 * bp[0];
            h1[k] = w11 * bp[4 + 3 * w] + w01 * bp[3 * w + 1] + w10 * bp[4] + w00 * bp[1];
            h2[k] = w11 * bp[5 + 3 * w] + w01 * bp[3 * w + 2] + w10 * bp[5] + w00 * bp[2];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (!isfinite(h0[k])) { energy += 1e5f; continue; }
            const float r = h0[k] - (aff0 * col[k] + aff1);
            const float dResdDist = dx * h1[k] + dy * h2[k];
            const float hw = fabsf(r) < S.huberTH ? 1 : S.huberTH / fabsf(r);
            H += hw * dResdDist * dResdDist;
            bb += hw * r * dResdDist;
            energy += wgt[k] * wgt[k] * hw * r * r * (2 - hw);
        }
        if (energy > bestEnergy) {
            stepBack *= 0.5f;
            bestU = uBak + stepBack * dx; bestV = vBak + stepBack * dy;
        } else {
            float step = -gnstepsize * bb / H;
            if (step < -0.5f) step = -0.5f; else if (step > 0.5f) step = 0.5f;
            if (!isfinite(step)) step = 0;
            uBak = bestU; vBak = bestV; stepBack = step;
            bestU += step * dx; bestV += step * dy;
            bestEnergy = energy;
        }
        if (fabsf(stepBack) < S.trace_GNThreshold) break;
    }
    if (lane == 0) P->quality = quality;
    // ---- energy-based outlier (:271-278) ----
    if (!(bestEnergy < P->energyTH * S.trace_extraSlackOnTH)) {
        trace_fail(P, lane, prevStatus == LDSO_IPS_OUTLIER ? LDSO_IPS_OOB : LDSO_IPS_OUTLIER, A.counts);
        return;
    }
    // ---- new interval (:281-300) ----
    float nmin, nmax;
    if (dx * dx > dy * dy) {
        nmin = (pr2 * (bestU - errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
        nmax = (pr2 * (bestU + errorInPixel * dx) - pr0) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
    } else {
        nmin = (pr2 * (bestV - errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
        nmax = (pr2 * (bestV + errorInPixel * dy) - pr1) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
    }
    if (nmin > nmax) { const float t = nmin; nmin = nmax; nmax = t; }
    if (lane == 0) { P->idepth_min = nmin; P->idepth_max = nmax; }
    if (!isfinite(nmin) || !isfinite(nmax) || (nmax < 0)) { trace_fail(P, lane, LDSO_IPS_OUTLIER, A.counts); return; }
    if (lane == 0) {
        P->lastTracePixelInterval = 2 * errorInPixel;
        P->lastTraceUV[0] = bestU; P->lastTraceUV[1] = bestV;
        P->lastTraceStatus = LDSO_IPS_GOOD;
        atomicAdd(&A.counts[LDSO_IPS_GOOD], 1);
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct ldso_tracer {
    int device = 0, w = 0, h = 0, maxPoints = 0, n = 0;
    hipStream_t stream = nullptr;
    ldso_trace_settings_t settings;
    ldso_immature_t *d_pts = nullptr;
    float *d_img = nullptr, *d_color = nullptr, *d_pose = nullptr;      // pose: [LDSO_MAX_FRAMES][14]
    const float *img = nullptr;       // the frame traced on: d_img, or level 0 of a shared ldso_pyramid_t
    int *d_counts = nullptr;
    bool haveFrame = false;
};

#define TCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ldso_set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return LDSO_E_HIP; } } while (0)
#define TREQ(c, msg) do { if (!(c)) { ldso_set_error(msg); return LDSO_E_INVALID; } } while (0)

extern "C" {

int ldso_trace_settings_default(ldso_trace_settings_t *s) {
    if (!s) return LDSO_E_INVALID;
    memset(s, 0, sizeof(*s));
    s->maxPixSearch = 0.027f; s->trace_stepsize = 1.0f; s->trace_GNThreshold = 0.1f; s->trace_extraSlackOnTH = 1.2f;
    s->trace_slackInterval = 1.5f; s->trace_minImprovementFactor = 2; s->huberTH = 9; s->trace_GNIterations = 3; s->minTraceTestRadius = 2;
    return LDSO_OK;
}

int ldso_trace_create(int device, int w, int h, int max_points, ldso_tracer_t **out) {
    TREQ(out && w > 16 && h > 16 && max_points > 0, "ldso_trace_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { ldso_set_error("no HIP device visible"); return LDSO_E_NODEVICE; }
    TREQ(device >= 0 && device < ndev, "ldso_trace_create: device index out of range");
    TCHK(hipSetDevice(device));
    ldso_tracer *T = new ldso_tracer();
    T->device = device; T->w = w; T->h = h; T->maxPoints = max_points;
    ldso_trace_settings_default(&T->settings);
    TCHK(hipStreamCreateWithFlags(&T->stream, hipStreamNonBlocking));
    TCHK(hipMalloc(&T->d_pts, (size_t) max_points * sizeof(ldso_immature_t)));
    TCHK(hipMalloc(&T->d_img, (size_t) w * h * 12));
    TCHK(hipMalloc(&T->d_color, (size_t) w * h * 4));
    TCHK(hipMalloc(&T->d_pose, (size_t) LDSO_MAX_FRAMES * 14 * 4));
    TCHK(hipMalloc(&T->d_counts, 8 * 4));
    *out = T;
    return LDSO_OK;
}

int ldso_trace_destroy(ldso_tracer_t *T) {
    if (!T) return LDSO_OK;
    hipSetDevice(T->device);
    hipDeviceSynchronize();
    hipFree(T->d_pts); hipFree(T->d_img); hipFree(T->d_color); hipFree(T->d_pose); hipFree(T->d_counts);
    if (T->stream) hipStreamDestroy(T->stream);
    delete T;
    return LDSO_OK;
}

int ldso_trace_set_settings(ldso_tracer_t *T, const ldso_trace_settings_t *s) {
    TREQ(T && s, "null argument");
    TREQ(s->trace_GNIterations >= 0 && s->trace_stepsize > 0 && s->minTraceTestRadius >= 0, "ldso_trace_set_settings: bad values");
    T->settings = *s;
    return LDSO_OK;
}

int ldso_trace_set_points(ldso_tracer_t *T, int n, const ldso_immature_t *pts) {
    TREQ(T && n >= 0 && n <= T->maxPoints && (n == 0 || pts), "ldso_trace_set_points: bad arguments");
    TCHK(hipSetDevice(T->device));
    if (n) TCHK(hipMemcpyAsync(T->d_pts, pts, (size_t) n * sizeof(ldso_immature_t), hipMemcpyHostToDevice, T->stream));
    TCHK(hipStreamSynchronize(T->stream));
    T->n = n;
    return LDSO_OK;
}

int ldso_trace_get_points(ldso_tracer_t *T, ldso_immature_t *out) {
    TREQ(T && (T->n == 0 || out), "ldso_trace_get_points: bad arguments");
    TCHK(hipSetDevice(T->device));
    if (T->n) TCHK(hipMemcpyAsync(out, T->d_pts, (size_t) T->n * sizeof(ldso_immature_t), hipMemcpyDeviceToHost, T->stream));
    TCHK(hipStreamSynchronize(T->stream));
    return LDSO_OK;
}

int ldso_trace_set_frame(ldso_tracer_t *T, const float *dI) {
    TREQ(T && dI, "ldso_trace_set_frame: bad arguments");
    TCHK(hipSetDevice(T->device));
    TCHK(hipMemcpyAsync(T->d_img, dI, (size_t) T->w * T->h * 12, hipMemcpyHostToDevice, T->stream));
    TCHK(hipStreamSynchronize(T->stream));
    T->haveFrame = true; T->img = T->d_img;
    return LDSO_OK;
}

int ldso_trace_set_frame_raw(ldso_tracer_t *T, const float *irradiance) {
    TREQ(T && irradiance, "ldso_trace_set_frame_raw: bad arguments");
    TCHK(hipSetDevice(T->device));
    TCHK(hipMemcpyAsync(T->d_color, irradiance, (size_t) T->w * T->h * 4, hipMemcpyHostToDevice, T->stream));
    float *lv[1] = {T->d_img};
    TCHK(img_launch_make_images(T->d_color, T->w, T->h, 1, lv, T->stream));
    TCHK(hipStreamSynchronize(T->stream));
    T->haveFrame = true; T->img = T->d_img;
    return LDSO_OK;
}

// the new frame as a resident ldso_pyramid_t (zero-copy: ImmaturePoint::traceOn samples frame->dI = level 0)
int ldso_trace_set_frame_pyramid(ldso_tracer_t *T, ldso_pyramid_t *pyr) {
    TREQ(T && pyr, "ldso_trace_set_frame_pyramid: bad arguments");
    TREQ(pyr->built && pyr->device == T->device && pyr->w == T->w && pyr->h == T->h, "ldso_trace_set_frame_pyramid: pyramid does not match the tracer (device, size) or holds no image");
    TCHK(hipSetDevice(T->device));
    TCHK(hipStreamWaitEvent(T->stream, pyr->ready, 0));
    T->haveFrame = true; T->img = pyr->lv[0];
    return LDSO_OK;
}

int ldso_trace_on(ldso_tracer_t *T, int n_hosts, const float *KRKi, const float *Kt, const float *aff, int *counts_out) {
    TREQ(T && n_hosts > 0 && n_hosts <= LDSO_MAX_FRAMES && KRKi && Kt && aff, "ldso_trace_on: bad arguments");
    TREQ(T->haveFrame, "ldso_trace_on: set the new frame first");
    TCHK(hipSetDevice(T->device));
    std::vector<float> pose((size_t) n_hosts * 14);
    memcpy(pose.data(), KRKi, (size_t) n_hosts * 9 * 4);
    memcpy(pose.data() + n_hosts * 9, Kt, (size_t) n_hosts * 3 * 4);
    memcpy(pose.data() + n_hosts * 12, aff, (size_t) n_hosts * 2 * 4);
    TCHK(hipMemcpyAsync(T->d_pose, pose.data(), pose.size() * 4, hipMemcpyHostToDevice, T->stream));
    TCHK(hipMemsetAsync(T->d_counts, 0, 8 * 4, T->stream));
    if (T->n > 0) {
        TraceArgs A;
        A.pts = T->d_pts; A.n = T->n; A.img = T->img; A.w = T->w; A.h = T->h;
        A.KRKi = T->d_pose; A.Kt = T->d_pose + n_hosts * 9; A.aff = T->d_pose + n_hosts * 12; A.nHosts = n_hosts;
        A.s = T->settings; A.counts = T->d_counts;
        const int waves = T->n, blocks = (waves + 3) / 4;
        hipLaunchKernelGGL(k_trace_on, dim3(blocks), dim3(256), 0, T->stream, A);
        TCHK(hipGetLastError());
    }
    int c[8] = {0};
    TCHK(hipMemcpyAsync(c, T->d_counts, 8 * 4, hipMemcpyDeviceToHost, T->stream));
    TCHK(hipStreamSynchronize(T->stream));
    if (counts_out) for (int i = 0; i < 6; i++) counts_out[i] = c[i];
    return LDSO_OK;
}

}  // extern "C"
