// adapter_capi.cc — C entry points for tests/test_adapter_gpu.py: run the compiled drop-in adapter (ldso_gpu_adapter.cc) on reference object
// graphs that oracle/ref_driver.cc builds (libldso_ref.so: the reference's own translation units).  Test plumbing, not part of the adapter.
#include <cstdio>
#include <exception>
#include <memory>
#include <vector>
#include <cmath>
#include "ldso_gpu_adapter.h"
#include "internal/PointHessian.h"
#include "internal/Residuals.h"

using namespace ldso;
using namespace ldso::internal;

static thread_local char g_err[512];
#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception &e) { std::snprintf(g_err, sizeof(g_err), "%s", e.what()); return -1; }

extern "C" {

const char *adp_last_error() { return g_err; }
void *adp_create(int device, int maxFrames, int maxPoints) {
    try { return new GpuBackend(device, maxFrames, maxPoints); } catch (const std::exception &e) { std::snprintf(g_err, sizeof(g_err), "%s", e.what()); return nullptr; }
}
void adp_destroy(void *b) { delete (GpuBackend *) b; }
// fs = ref_fs_handle(window) of libldso_ref.so
int adp_optimize(void *b, void *fs, int iterations, float *rmse, int *executed, int *lost) {
    GUARD(*rmse = ((GpuBackend *) b)->optimize(*(FullSystem *) fs, iterations); *executed = ((GpuBackend *) b)->lastIterations; *lost = ((FullSystem *) fs)->isLost ? 1 : 0)
}
// fs = ref_tr_prepare(tracker ...), tracker = ref_tr_coarse_tracker, fhs = ref_tr_frame_hessians, newfh = ref_tr_new_frame_hessian, calib = ref_tr_calib_hessian
int adp_track_new_coarse(void *b, void *fs_, void *tracker, void *fhs, void *newfh, void *calib, double *result4) {
    GUARD(
        GpuBackend &B = *(GpuBackend *) b; FullSystem &fs = *(FullSystem *) fs_; CoarseTracker &tr = *(CoarseTracker *) tracker;
        B.makeK(tr, *(std::shared_ptr<CalibHessian> *) calib);
        B.setCoarseTrackingRef(tr, *(std::vector<std::shared_ptr<FrameHessian>> *) fhs);
        Vec4 r = B.trackNewCoarse(fs, *(std::shared_ptr<FrameHessian> *) newfh);
        for (int i = 0; i < 4; i++) result4[i] = r[i])
}
// one CoarseTracker::trackNewestCoarse through the adapter (T row-major [R|t] in / out)
int adp_track_newest_coarse(void *b, void *tracker, void *fhs, void *newfh, void *calib, double *T, float *ab, int coarsestLvl, const double *minRes, double *lastResiduals, int *ok) {
    GUARD(
        GpuBackend &B = *(GpuBackend *) b; CoarseTracker &tr = *(CoarseTracker *) tracker;
        B.makeK(tr, *(std::shared_ptr<CalibHessian> *) calib);
        B.setCoarseTrackingRef(tr, *(std::vector<std::shared_ptr<FrameHessian>> *) fhs);
        Mat33 R; Vec3 t;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = T[i * 4 + j]; t[i] = T[i * 4 + 3]; }
        SE3 pose(R, t); AffLight aff(ab[0], ab[1]); Vec5 mr;
        for (int i = 0; i < 5; i++) mr[i] = minRes[i];
        *ok = B.trackNewestCoarse(tr, *(std::shared_ptr<FrameHessian> *) newfh, pose, aff, coarsestLvl, mr) ? 1 : 0;
        Eigen::Matrix<double, 3, 4> M = pose.matrix3x4();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) T[i * 4 + j] = M(i, j);
        ab[0] = aff.a; ab[1] = aff.b;
        for (int i = 0; i < 5; i++) lastResiduals[i] = tr.lastResiduals[i])
}

// toOptimize = ref_fs_build_immature(window, ...) of libldso_ref.so.  out: per candidate the verdict, the new point's inverse depth, and per
// window frame 0 where the new point got a PointFrameResidual (state IN) / -1 elsewhere; last[k][0..1] = state of lastResiduals[0..1] (-1: no residual)
int adp_activate_points(void *b, void *fs_, void *toOptimize, int n, int F, int *ok, float *idepth, int *res_target /*n*F*/, int *last /*n*2*/) {
    GUARD(
        auto &cand = *(std::vector<std::shared_ptr<ImmaturePoint>> *) toOptimize;
        std::vector<std::shared_ptr<PointHessian>> made;
        ((GpuBackend *) b)->activatePoints(*(FullSystem *) fs_, cand, made);
        for (int i = 0; i < n; i++) {
            ok[i] = made[i] ? 1 : 0; idepth[i] = made[i] ? made[i]->idepth : NAN;
            for (int t = 0; t < F; t++) res_target[i * F + t] = -1;
            last[2 * i] = last[2 * i + 1] = -1;
            if (!made[i]) continue;
            for (auto &r : made[i]->residuals) res_target[i * F + r->target.lock()->idx] = (int) r->state_state;
            for (int j = 0; j < 2; j++) last[2 * i + j] = made[i]->lastResiduals[j].first ? (int) made[i]->lastResiduals[j].second : -1;
        })
}

// fh = ref_fs_new_frame(window, ...): GpuBackend::traceNewCoarse in place of FullSystem::traceNewCoarse; counts = its six trace_* counters
int adp_trace_new_coarse(void *b, void *fs, void *fh, int *counts) {
    GUARD(((GpuBackend *) b)->traceNewCoarse(*(FullSystem *) fs, *(std::shared_ptr<FrameHessian> *) fh); for (int i = 0; i < 6; i++) counts[i] = ((GpuBackend *) b)->lastTraceCounts[i])
}

}  // extern "C"
