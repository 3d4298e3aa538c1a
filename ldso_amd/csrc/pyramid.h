// ldso_pyramid_t (images.hip): one frame's FrameHessian::dIp in HBM, shared zero-copy by the tracker, the tracer and the bundle adjustment
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ldso_window.h"

struct ldso_pyramid {
    int device = 0, w = 0, h = 0, levels = 0;
    float *base = nullptr;                 // one allocation: the levels, then the raw image
    float *lv[LDSO_PYR_LEVELS] = {};       // level l: (w>>l)*(h>>l) pixels of (I, dx, dy)
    float *d_color = nullptr;              // staging of the raw irradiance (host uploads)
    hipEvent_t ready = nullptr;            // recorded after the last build: consumers wait on it with hipStreamWaitEvent
    bool built = false;
};

extern "C" hipError_t img_launch_make_images(const float *d_color, int w, int h, int levels, float *const *d_levels, hipStream_t st);
