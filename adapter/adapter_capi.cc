// adapter_capi.cc — C entry points for tests/test_adapter_gpu.py: run the compiled drop-in adapter (ldso_gpu_adapter.cc) on reference object
// graphs that oracle/ref_driver.cc builds (libldso_ref.so: the reference's own translation units).  Test plumbing, not part of the adapter.
#include <cstdio>
#include <exception>
#include <memory>
#include <vector>
#include <cmath>
// test plumbing reaches into FullSystem's private members exactly as the adapter's translation unit does (see ldso_gpu_adapter.h)
#include <deque>
#include <list>
#include <map>
#include <mutex>
#include <queue>
#include <set>
#include <string>
#include <thread>
#include <Eigen/Core>
#include <glog/logging.h>
#define private public
#define protected public
#include "frontend/CoarseTracker.h"
#include "frontend/FullSystem.h"
#undef private
#undef protected
#include "ldso_gpu_adapter.h"
#include "internal/PointHessian.h"
#include "internal/Residuals.h"
#include "internal/ImmaturePoint.h"
#include "internal/GlobalCalib.h"
#include "internal/OptimizationBackend/EnergyFunctional.h"
#include <chrono>

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

int adp_set_device_pyramids(void *b, int on) { ((GpuBackend *) b)->useDevicePyramids = on != 0; return 0; }
int adp_pyramids_built(void *b) { return ((GpuBackend *) b)->pyramidsBuilt; }
int adp_set_write_back_jacobians(void *b, int on) { ((GpuBackend *) b)->writeBackJacobians = on != 0; return 0; }
// the window resident across optimize() calls (ldso_ba_update_window) or flattened and uploaded every time; how many uploads of each kind so far
int adp_set_resident_window(void *b, int on) { ((GpuBackend *) b)->residentWindow = on != 0; return 0; }
int adp_upload_counts(void *b, int *delta, int *fresh) { *delta = ((GpuBackend *) b)->uploadsDelta; *fresh = ((GpuBackend *) b)->uploadsFresh; return 0; }
// wall-clock split of the last GpuBackend::optimize (seconds): flatten + upload, device (ldso_ba_optimize incl. its read-back of the energies), fetch, write-back into the objects
int adp_last_optimize_times(void *b, double *out4) { for (int i = 0; i < 4; i++) out4[i] = ((GpuBackend *) b)->lastOptimizeSeconds[i]; return 0; }
int adp_last_upload_times(void *b, double *out6) { for (int i = 0; i < 6; i++) out6[i] = ((GpuBackend *) b)->lastUploadSeconds[i]; return 0; }

// ---- one key frame in the order of FullSystem::makeKeyFrame (FullSystem.cc:410-640) on a reference object graph -----------------------------
// b == nullptr: the reference's own members everywhere.  b != nullptr: GpuBackend::traceNewCoarse / activatePoints / optimize in place of
// FullSystem::traceNewCoarse (:429), the optimizeImmaturePoint loop of activatePointsMT (:1157-1166) and FullSystem::optimize (:478); everything
// else - ef->insertFrame, the new residuals of old points, removeOutliers, flagPointsForRemoval, dropPointsF, marginalizePointsF,
// marginalizeFrame - is the reference's host code on both graphs, as in a first integration of the drop-in.
// What is POLICY upstream of the hot path is decided by the caller, identically for both graphs: which frame is flagged for marginalisation
// (flagFramesForMarginalization -> margIdx, -1: none), the key-frame number (globalMap->NumFrames() -> kfId), and the candidate selection of
// activatePointsMT, which is restated below WITHOUT the distance map (every immature point that passes the canActivate rule and projects into
// the newest frame is a candidate).  The coarse-tracker swap (:515-522) and makeNewTraces (:539: pixel selection) are the caller's business.
// newfh = shared_ptr<FrameHessian>* of ref_fs_new_frame.  stats: [candidates, activated, residuals added for old points, points after, lost]
static bool deviceMarginalisation = false;
int adp_set_device_marginalisation(int on) { deviceMarginalisation = on != 0; return 0; }
int adp_make_keyframe(void *b, void *fs_, void *newfh, int margIdx, int kfId, int iterations, float *rmse, int *stats) {
    GUARD(
        GpuBackend *B = (GpuBackend *) b; FullSystem &fs = *(FullSystem *) fs_;
        shared_ptr<FrameHessian> fh = *(shared_ptr<FrameHessian> *) newfh;
        for (int i = 0; i < 5; i++) stats[i] = 0;
        // :429 trace new keyframe
        if (B) B->traceNewCoarse(fs, fh); else fs.traceNewCoarse(fh);
        // :434 flag frames to be marginalised (policy: the caller's choice)
        if (margIdx >= 0) fs.frames[margIdx]->frameHessian->flaggedForMarginalization = true;
        // :437-442 add the new frame
        fh->idx = fs.frames.size();
        fs.frames.push_back(fh->frame);
        fh->frame->kfId = fh->frameID = kfId;
        fs.ef->insertFrame(fh, fs.Hcalib->mpCH);
        fs.setPrecalcValues();
        // :447-470 new residuals for old points
        for (auto fht : fs.frames) {
            shared_ptr<FrameHessian> &fh1 = fht->frameHessian;
            if (fh1 == fh) continue;
            for (auto feat : fht->features) {
                if (feat->status == Feature::FeatureStatus::VALID && feat->point->status == Point::PointStatus::ACTIVE) {
                    shared_ptr<PointHessian> ph = feat->point->mpPH;
                    shared_ptr<PointFrameResidual> r(new PointFrameResidual(ph, fh1, fh));
                    r->setState(ResState::IN);
                    ph->residuals.push_back(r);
                    fs.ef->insertResidual(r);
                    ph->lastResiduals[1] = ph->lastResiduals[0];
                    ph->lastResiduals[0] = std::pair<shared_ptr<PointFrameResidual>, ResState>(r, ResState::IN);
                    stats[2]++;
                }
            }
        }
        // :473 activatePointsMT: candidate rule of :1090-1150 (no distance map), the optimizeImmaturePoint loop, the object hand-over of :1168-1190
        {
            auto newestFr = fs.frames.back();
            fs.coarseDistanceMap->makeK(fs.Hcalib->mpCH);
            std::vector<shared_ptr<ImmaturePoint>> toOptimize;
            for (auto fr : fs.frames) {
                shared_ptr<FrameHessian> host = fr->frameHessian;
                if (host == newestFr->frameHessian) continue;
                SE3 fhToNew = newestFr->frameHessian->PRE_worldToCam * host->PRE_camToWorld;
                Mat33f KRKi = (fs.coarseDistanceMap->K[1] * fhToNew.rotationMatrix().cast<float>() * fs.coarseDistanceMap->Ki[0]);
                Vec3f Kt = (fs.coarseDistanceMap->K[1] * fhToNew.translation().cast<float>());
                for (size_t i = 0; i < host->frame->features.size(); i++) {
                    shared_ptr<Feature> &feat = host->frame->features[i];
                    if (!(feat->status == Feature::FeatureStatus::IMMATURE && feat->ip)) continue;
                    shared_ptr<ImmaturePoint> &ph = feat->ip;
                    ph->idxInImmaturePoints = i;
                    if (!std::isfinite(ph->idepth_max) || ph->lastTraceStatus == IPS_OUTLIER) { feat->status = Feature::FeatureStatus::OUTLIER; feat->ReleaseImmature(); continue; }
                    bool canActivate = (ph->lastTraceStatus == IPS_GOOD || ph->lastTraceStatus == IPS_SKIPPED || ph->lastTraceStatus == IPS_BADCONDITION || ph->lastTraceStatus == IPS_OOB)
                                       && ph->lastTracePixelInterval < 8 && ph->quality > setting_minTraceQuality && (ph->idepth_max + ph->idepth_min) > 0;
                    if (!canActivate) {
                        if (ph->feature->host.lock()->frameHessian->flaggedForMarginalization || ph->lastTraceStatus == IPS_OOB) { feat->status = Feature::FeatureStatus::OUTLIER; feat->ReleaseImmature(); }
                        continue;
                    }
                    Vec3f ptp = KRKi * Vec3f(feat->uv[0], feat->uv[1], 1) + Kt * (0.5f * (ph->idepth_max + ph->idepth_min));
                    int u = ptp[0] / ptp[2] + 0.5f;
                    int v = ptp[1] / ptp[2] + 0.5f;
                    if ((u > 0 && v > 0 && u < wG[1] && v < hG[1])) toOptimize.push_back(ph);
                    else { feat->status = Feature::FeatureStatus::OUTLIER; feat->ReleaseImmature(); }
                }
            }
            stats[0] = (int) toOptimize.size();
            std::vector<shared_ptr<PointHessian>> optimized(toOptimize.size());
            if (B) { if (!toOptimize.empty()) B->activatePoints(fs, toOptimize, optimized); }
            else fs.activatePointsMT_Reductor(&optimized, &toOptimize, 0, (int) toOptimize.size(), 0, 0);
            for (size_t k = 0; k < toOptimize.size(); k++) {
                shared_ptr<PointHessian> newpoint = optimized[k];
                shared_ptr<ImmaturePoint> ph = toOptimize[k];
                if (newpoint != nullptr) {
                    ph->feature->status = Feature::FeatureStatus::VALID;
                    ph->feature->point->mpPH = newpoint;
                    ph->feature->ReleaseImmature();
                    newpoint->takeData();
                    for (auto r : newpoint->residuals) fs.ef->insertResidual(r);
                    stats[1]++;
                } else if (newpoint == nullptr || ph->lastTraceStatus == IPS_OOB) {
                    ph->feature->status = Feature::FeatureStatus::OUTLIER;
                    ph->feature->ReleaseImmature();
                }
            }
        }
        fs.ef->makeIDX();
        // :477-478 optimize
        fh->frameEnergyTH = fs.frames.back()->frameHessian->frameEnergyTH;
        *rmse = B ? B->optimize(fs, iterations) : fs.optimize(iterations);
        if (fs.isLost) { stats[4] = 1; return 0; }
        // :511 remove outliers; :526-536 flag / drop / marginalise points
        fs.removeOutliers();
        if (B && deviceMarginalisation) {
            // the device keeps the window of optimize(): the policy on the host, the re-linearise / fix / accumulate of the marginalised points on the device
            B->flagPointsForRemoval(fs);
            fs.ef->dropPointsF();
            fs.getNullspaces(fs.ef->lastNullspaces_pose, fs.ef->lastNullspaces_scale, fs.ef->lastNullspaces_affA, fs.ef->lastNullspaces_affB);
            B->marginalizePoints(fs);
        } else {
            fs.flagPointsForRemoval();
            fs.ef->dropPointsF();
            fs.getNullspaces(fs.ef->lastNullspaces_pose, fs.ef->lastNullspaces_scale, fs.ef->lastNullspaces_affA, fs.ef->lastNullspaces_affB);
            fs.ef->marginalizePointsF();
        }
        // :594-603 marginalise the flagged frames (their pyramids live in the driver's image store: ~FrameHessian must not delete[] them)
        for (unsigned int i = 0; i < fs.frames.size(); i++)
            if (fs.frames[i]->frameHessian->flaggedForMarginalization) {
                shared_ptr<Frame> fr = fs.frames[i];
                for (int l = 0; l < PYR_LEVELS; l++) { fr->frameHessian->dIp[l] = nullptr; fr->frameHessian->absSquaredGrad[l] = nullptr; }
                if (B && deviceMarginalisation) B->marginalizeFrame(fs, fr); else fs.marginalizeFrame(fr);
                i = 0;
            }
        for (auto &fr : fs.frames) for (auto &feat : fr->features) if (feat->status == Feature::FeatureStatus::VALID && feat->point && feat->point->status == Point::PointStatus::ACTIVE) stats[3]++;
    )
}

// What a trajectory / map comparison needs of the graph after a key frame: per frame [Frame::id, camToWorld 3x4 row-major (PRE_camToWorld), a, b, active points
// hosted, residuals hosted, immature points hosted, frameEnergyTH] (18 doubles), then HM (n x n) and bM (n) with n = 8 F + 4; returns F
int adp_graph_summary(void *fs_, int capFrames, double *frames18, double *HM, double *bM, double *calib4) {
    FullSystem &fs = *(FullSystem *) fs_;
    const int F = (int) fs.frames.size();
    for (int f = 0; f < F && f < capFrames; f++) {
        FrameHessian &fh = *fs.frames[f]->frameHessian;
        double *o = frames18 + 18 * f;
        o[0] = (double) fs.frames[f]->id;
        Eigen::Matrix<double, 3, 4> M = fh.PRE_camToWorld.matrix3x4();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) o[1 + i * 4 + j] = M(i, j);
        o[13] = fh.aff_g2l().a; o[14] = fh.aff_g2l().b;
        int np = 0, nr = 0, ni = 0;
        for (auto &feat : fs.frames[f]->features) {
            if (feat->status == Feature::FeatureStatus::VALID && feat->point && feat->point->status == Point::PointStatus::ACTIVE) { np++; nr += (int) feat->point->mpPH->residuals.size(); }
            else if (feat->status == Feature::FeatureStatus::IMMATURE && feat->ip) ni++;
        }
        o[15] = np; o[16] = nr; o[17] = ni;
    }
    const int n = (int) fs.ef->HM.rows();
    if (HM) for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HM[(size_t) i * n + j] = fs.ef->HM(i, j); bM[i] = fs.ef->bM[i]; }
    if (calib4) for (int i = 0; i < 4; i++) calib4[i] = fs.Hcalib->mpCH->value[i];
    return F;
}
// inverse depths of the active points in traversal order (frames, then their features) - up to cap; returns the count.  host = Frame::id of the host
// key frame, uv = the point's pixel: (host, u, v) identifies a point across two object graphs whose point sets differ by a few threshold cases
int adp_graph_idepths(void *fs_, int cap, float *idepth, int *host, float *uv) {
    FullSystem &fs = *(FullSystem *) fs_;
    int n = 0;
    for (size_t f = 0; f < fs.frames.size(); f++)
        for (auto &feat : fs.frames[f]->features)
            if (feat->status == Feature::FeatureStatus::VALID && feat->point && feat->point->status == Point::PointStatus::ACTIVE) {
                if (n < cap) { idepth[n] = feat->point->mpPH->idepth; host[n] = (int) fs.frames[f]->id; uv[2 * n] = feat->uv[0]; uv[2 * n + 1] = feat->uv[1]; }
                n++;
            }
    return n;
}

}  // extern "C"
