// ldso_gpu_adapter.cc — see ldso_gpu_adapter.h.  Compiles against the reference's headers (LDSO include/ + its Eigen / Sophus) and links
// libldso_hip.so; every function is the replacement body of the reference member function it cites.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#ifndef LDSO_ADAPTER_OPEN_PRIVATE
#define LDSO_ADAPTER_OPEN_PRIVATE 1      // see the header: stands in for `friend class ldso::GpuBackend;` in FullSystem.h / CoarseTracker.h
#endif
#if LDSO_ADAPTER_OPEN_PRIVATE
#include <Eigen/Core>
#include <glog/logging.h>
#define private public
#define protected public
#include "frontend/CoarseTracker.h"
#include "frontend/FullSystem.h"
#undef private
#undef protected
#endif
#include "ldso_gpu_adapter.h"

#include "internal/CalibHessian.h"
#include "internal/FrameHessian.h"
#include "internal/GlobalCalib.h"
#include "internal/GlobalFuncs.h"
#include "internal/OptimizationBackend/EnergyFunctional.h"
#include "internal/PointHessian.h"
#include "internal/Residuals.h"

using namespace ldso::internal;

namespace ldso {

namespace {

void throwOn(int rc, const char *what) {
    if (rc != LDSO_OK && rc != LDSO_E_NONFINITE) throw std::runtime_error(std::string(what) + ": " + ldso_last_error());
}
void toRowMajor34(const SE3 &T, double *m) {
    Eigen::Matrix<double, 3, 4> M = T.matrix3x4();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) m[i * 4 + j] = M(i, j);
}
SE3 fromRowMajor34(const double *m) {
    Mat33 R; Vec3 t;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = m[i * 4 + j]; t[i] = m[i * 4 + 3]; }
    return SE3(R, t);
}
ldso_calib_t flatCalib(const CalibHessian &c) {
    ldso_calib_t o;
    for (int i = 0; i < 4; i++) { o.value[i] = c.value[i]; o.value_zero[i] = c.value_zero[i]; }
    return o;
}
ldso_settings_t flatSettings() {      // the setting_* globals the hot path reads (Settings.h), as the library's ldso_settings_t
    ldso_settings_t s;
    ldso_settings_default(&s);
    s.huberTH = setting_huberTH; s.outlierTHSumComponent = setting_outlierTHSumComponent;
    s.affineOptModeA = setting_affineOptModeA; s.affineOptModeB = setting_affineOptModeB;
    s.frameEnergyTHN = setting_frameEnergyTHN; s.frameEnergyTHFacMedian = setting_frameEnergyTHFacMedian;
    s.frameEnergyTHConstWeight = setting_frameEnergyTHConstWeight; s.overallEnergyTHWeight = setting_overallEnergyTHWeight;
    s.initialCalibHessian = setting_initialCalibHessian; s.margWeightFac = setting_margWeightFac;
    s.idepthFixPriorMargFac = setting_idepthFixPriorMargFac; s.thOptIterations = setting_thOptIterations;
    s.coarseCutoffTH = setting_coarseCutoffTH; s.minOptIterations = setting_minOptIterations;
    s.solverMode = setting_solverMode; s.forceAcceptStep = setting_forceAceptStep ? 1 : 0; s.solverModeDelta = setting_solverModeDelta;
    return s;
}
void toRaw(const RawResidualJacobian &J, ldso_rawjac_t &j) {
    for (int i = 0; i < 8; i++) { j.resF[i] = J.resF[i]; j.JIdx[0][i] = J.JIdx[0][i]; j.JIdx[1][i] = J.JIdx[1][i]; j.JabF[0][i] = J.JabF[0][i]; j.JabF[1][i] = J.JabF[1][i]; }
    for (int i = 0; i < 6; i++) { j.Jpdxi[0][i] = J.Jpdxi[0][i]; j.Jpdxi[1][i] = J.Jpdxi[1][i]; }
    for (int i = 0; i < 4; i++) { j.Jpdc[0][i] = J.Jpdc[0][i]; j.Jpdc[1][i] = J.Jpdc[1][i]; }
    j.Jpdd[0] = J.Jpdd[0]; j.Jpdd[1] = J.Jpdd[1];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { j.JIdx2[a * 2 + b] = J.JIdx2(a, b); j.JabJIdx[a * 2 + b] = J.JabJIdx(a, b); j.Jab2[a * 2 + b] = J.Jab2(a, b); }
}
void fromRaw(const ldso_rawjac_t &j, RawResidualJacobian &J) {
    for (int i = 0; i < 8; i++) { J.resF[i] = j.resF[i]; J.JIdx[0][i] = j.JIdx[0][i]; J.JIdx[1][i] = j.JIdx[1][i]; J.JabF[0][i] = j.JabF[0][i]; J.JabF[1][i] = j.JabF[1][i]; }
    for (int i = 0; i < 6; i++) { J.Jpdxi[0][i] = j.Jpdxi[0][i]; J.Jpdxi[1][i] = j.Jpdxi[1][i]; }
    for (int i = 0; i < 4; i++) { J.Jpdc[0][i] = j.Jpdc[0][i]; J.Jpdc[1][i] = j.Jpdc[1][i]; }
    J.Jpdd[0] = j.Jpdd[0]; J.Jpdd[1] = j.Jpdd[1];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { J.JIdx2(a, b) = j.JIdx2[a * 2 + b]; J.JabJIdx(a, b) = j.JabJIdx[a * 2 + b]; J.Jab2(a, b) = j.Jab2[a * 2 + b]; }
}

}  // namespace

GpuBackend::GpuBackend(int device, int maxFrames, int maxPoints) : device_(device), maxFrames_(maxFrames), maxPoints_(maxPoints) {
    throwOn(ldso_ba_create(device, wG[0], hG[0], maxFrames, maxPoints, &ba_), "ldso_ba_create");
    slotOwner_.assign(maxFrames, -1);
}

GpuBackend::~GpuBackend() {
    for (auto &kv : trackers_) ldso_tr_destroy(kv.second);
    if (tracer_) ldso_trace_destroy(tracer_);
    if (ba_) ldso_ba_destroy(ba_);
    slotPyr_.clear(); tracerPyr_.reset(); trackerPyr_.clear(); pyr_.clear();          // the pyramids, after their consumers
}

// ------------------------------------------------------------------------------------------------------------------------------------
// FrameHessian::dIp on the device (FrameHessian.cc:44-113): built once per frame from channel 0 of dIp[0] (= the irradiance makeImages started from;
// the device build is bit-identical to the host arrays, tests/test_pyramid_gpu.py), keyed by Frame::id
// ------------------------------------------------------------------------------------------------------------------------------------
GpuBackend::PyrHolder::~PyrHolder() { if (p) ldso_pyr_destroy(p); }

GpuBackend::PyrRef GpuBackend::pyramidOf(const shared_ptr<FrameHessian> &fh) {
    const unsigned long id = fh->frame->id;
    std::lock_guard<std::mutex> lk(pyrMutex_);
    auto it = pyr_.find(id);
    if (it != pyr_.end()) return it->second;
    PyrRef e = std::make_shared<PyrHolder>();
    const size_t n = (size_t) wG[0] * hG[0];
    e->irradiance.resize(n);
    const Vec3f *src = fh->dIp[0];
    for (size_t i = 0; i < n; i++) e->irradiance[i] = src[i][0];
    int rc = ldso_pyr_create(device_, wG[0], hG[0], pyrLevelsUsed, &e->p);
    if (rc == LDSO_OK) rc = ldso_pyr_make_images(e->p, e->irradiance.data(), nullptr);
    throwOn(rc, "ldso_pyr_create / ldso_pyr_make_images");          // e (and its half-built pyramid) goes with the exception
    pyr_[id] = e;
    pyramidsBuilt++;
    return e;
}

// Frames that left the window AND that no consumer holds any more (use_count 1 = the map alone).  A consumer can only gain a reference through pyramidOf,
// i.e. under pyrMutex_: a count of 1 seen here cannot grow behind our back; a consumer letting go concurrently only delays the release by one call.
void GpuBackend::releasePyramids(FullSystem &fs) {
    std::set<unsigned long> keep;
    for (auto &fr : fs.frames) keep.insert(fr->id);
    std::vector<PyrRef> dying;                                       // destroyed outside the lock
    {
        std::lock_guard<std::mutex> lk(pyrMutex_);
        for (auto it = pyr_.begin(); it != pyr_.end();) {
            if (!keep.count(it->first) && it->second.use_count() == 1) { dying.push_back(std::move(it->second)); it = pyr_.erase(it); } else ++it;
        }
    }
}

const char *GpuBackend::lastError() const { return ldso_last_error(); }

// ------------------------------------------------------------------------------------------------------------------------------------
// Images: FrameHessian::dIp[0] of a key frame goes to the device once, when the frame first appears in the window; its slot is
// recycled when the frame has left (FullSystem::marginalizeFrame).
// ------------------------------------------------------------------------------------------------------------------------------------
void GpuBackend::syncImageSlots(FullSystem &fs, std::vector<int32_t> &slots) {
    std::set<unsigned long> live;
    for (auto &fr : fs.frames) live.insert(fr->id);
    for (size_t s = 0; s < slotOwner_.size(); s++)
        if (slotOwner_[s] >= 0 && !live.count((unsigned long) slotOwner_[s])) { slotOf_.erase((unsigned long) slotOwner_[s]); slotOwner_[s] = -1; if (s < slotPyr_.size()) slotPyr_[s].reset(); }
    slots.clear();
    for (auto &fr : fs.frames) {
        FrameHessian *fh = fr->frameHessian.get();
        auto it = slotOf_.find(fr->id);
        if (it == slotOf_.end()) {
            size_t s = 0;
            while (s < slotOwner_.size() && slotOwner_[s] >= 0) s++;
            if (s == slotOwner_.size()) throw std::runtime_error("GpuBackend: more key frames than maxFrames");
            if (useDevicePyramids && fr->frameHessian->frame) {      // zero-copy: level 0 of the frame's pyramid, held for as long as the slot names it
                PyrRef pr = pyramidOf(fr->frameHessian);
                throwOn(ldso_ba_set_image_pyramid(ba_, (int) s, pr->p), "ldso_ba_set_image_pyramid");
                if (slotPyr_.size() < slotOwner_.size()) slotPyr_.resize(slotOwner_.size());
                slotPyr_[s] = pr;
            } else throwOn(ldso_ba_set_image(ba_, (int) s, (const float *) fh->dIp[0]), "ldso_ba_set_image");      // Vec3f AoS (I, dx, dy): a straight copy
            slotOwner_[s] = (long) fr->id; it = slotOf_.emplace(fr->id, (int) s).first;
        }
        slots.push_back(it->second);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Flatten the window in the reference's own orders: frames = FullSystem::frames (= EnergyFunctional::frames, idx = position), points in
// the order of EnergyFunctional::makeIDX (EnergyFunctional.cc:380-401: frames, then the host frame's features), residuals in
// PointHessian::residuals order; a residual's slot is (hostIDX, targetIDX).  allPoints / flat give the write-back its objects.
// ------------------------------------------------------------------------------------------------------------------------------------
// one point of the window as ldso_ba_set_window / ldso_ba_update_window take it
static void flatPoint(const PointHessian &ph, int host, ldso_point_t &p) {
    memset(&p, 0, sizeof(p));
    p.u = ph.u; p.v = ph.v; p.idepth = ph.idepth; p.idepth_zero = ph.idepth_zero; p.priorF = ph.priorF;
    memcpy(p.color, ph.color, sizeof(p.color)); memcpy(p.weights, ph.weights, sizeof(p.weights));
    p.host = host;
}
static void flatResidual(const PointFrameResidual &r, int point, int host, int target, ldso_residual_t &q) {
    q.point = point; q.host = host; q.target = target; q.state_state = (int32_t) r.state_state;
    q.is_linearized = r.isLinearized ? 1 : 0; q.is_active = r.isActive() ? 1 : 0; q.is_new = r.isNew ? 1 : 0; q.state_energy = (float) r.state_energy;
}

// The whole window from the objects (the first optimize() of a handle, windows with linearised residuals, after invalidateWindow(), or when the delta path below
// gave up).  Residuals of a point go over target-ascending: the flat order ldso_ba_update_window keeps, so that both paths describe the same window.
void GpuBackend::uploadFresh(FullSystem &fs, const std::vector<int32_t> &slots, std::vector<shared_ptr<PointHessian>> &allPoints, bool trustIndices) {
    const int F = (int) fs.frames.size();
    allPoints.clear(); flat_.clear(); rows_.clear();
    resBegin_.assign(1, 0);
    std::vector<ldso_point_t> P; std::vector<ldso_residual_t> R; std::vector<ldso_rawjac_t> LJ; std::vector<float> RTZ, mrb; std::vector<int32_t> ngr;
    P.reserve(4096); R.reserve(32768); flat_.reserve(32768); rows_.reserve(4096);
    // linearised residuals (none in LDSO's own flow: flagPointsForRemoval clears isLinearized, FullSystem.cc:1243) carry their Jacobian and
    // res_toZeroF across; the 296-byte records are only built from the first one on (the residuals in front of it get empty ones then)
    bool anyLin = false;
    for (int f = 0; f < F; f++)
        for (shared_ptr<Feature> &feat : fs.frames[f]->features) {
            if (!(feat->status == Feature::FeatureStatus::VALID && feat->point && feat->point->status == Point::PointStatus::ACTIVE)) continue;
            shared_ptr<PointHessian> ph = feat->point->mpPH;
            ldso_point_t p;
            flatPoint(*ph, f, p);
            p.res_begin = (int32_t) R.size(); p.res_count = (int32_t) ph->residuals.size();
            // PRECONDITION of trustIndices: ef->makeIDX() ran after the last insertResidual (makeKeyFrame does, FullSystem.cc:474) - insertResidual does not
            // clear EFIndicesValid, so a caller that adds residuals and skips makeIDX would hand over stale / zero target indices.  Fresh residuals are appended:
            // the NEWEST residual of the point is checked against the frame it names (one weak_ptr::lock per point, not per residual); a mismatch derives the
            // indices of this point the slow way
            bool trust = trustIndices && EFIndicesValid;
            if (trust && !ph->residuals.empty()) {
                const PointFrameResidual &rl = *ph->residuals.back();
                trust = rl.targetIDX >= 0 && rl.targetIDX < F && fs.frames[rl.targetIDX]->frameHessian.get() == rl.target.lock().get();
            }
            Row row; row.ph = ph.get(); row.feat = feat.get(); row.mask = 0; row.host = f;
            for (int c = 0; c < kMaxCols; c++) row.res[c] = nullptr;
            for (shared_ptr<PointFrameResidual> &r : ph->residuals) {
                // makeIDX (EnergyFunctional.cc:380-401): LDSO's own flow reaches optimize() with valid indices (two weak_ptr::lock() per residual are two
                // atomic read-modify-write pairs, 24 000 of them a quarter of the flatten at C3)
                r->hostIDX = f;
                if (!trust) r->targetIDX = r->target.lock()->idx;
                const int t = r->targetIDX;
                if (t < 0 || t >= F || t == f || ((row.mask >> t) & 1u)) throw std::runtime_error("GpuBackend: a residual names a frame outside the window, its host, or a target twice");
                row.mask |= 1u << t; row.res[t] = r.get();
            }
            for (int t = 0; t < F; t++) {
                if (!((row.mask >> t) & 1u)) continue;
                PointFrameResidual &r = *row.res[t];
                ldso_residual_t q;
                flatResidual(r, (int32_t) P.size(), f, t, q);
                if (r.isLinearized && !anyLin) { anyLin = true; LJ.resize(R.size()); RTZ.resize(R.size() * 8, 0.0f); for (auto &j : LJ) memset(&j, 0, sizeof(j)); }
                R.push_back(q); flat_.push_back(&r);
                if (anyLin) {
                    ldso_rawjac_t j;
                    memset(&j, 0, sizeof(j));
                    if (r.isLinearized) toRaw(*r.J, j);
                    LJ.push_back(j);
                    for (int k = 0; k < 8; k++) RTZ.push_back(r.isLinearized ? r.res_toZeroF[k] : 0.0f);
                }
            }
            P.push_back(p); allPoints.push_back(ph); rows_.push_back(row);
            mrb.push_back(ph->maxRelBaseline); ngr.push_back(ph->numGoodResiduals);
            resBegin_.push_back((int32_t) R.size());
        }
    if (P.empty()) return;
    lastUploadSeconds[1] += lapSince();
    throwOn(ldso_ba_set_window(ba_, F, slots.data(), (int) P.size(), P.data(), (int) R.size(), R.data(), anyLin ? LJ.data() : nullptr, anyLin ? RTZ.data() : nullptr), "ldso_ba_set_window");
    lastUploadSeconds[2] += lapSince();
    throwOn(ldso_ba_set_point_stats(ba_, mrb.data(), ngr.data()), "ldso_ba_set_point_stats");
    lastUploadSeconds[3] += lapSince();
    residentValid_ = !anyLin;                                        // ldso_ba_update_window does not carry linearised residuals: the next call uploads in full again
    if (anyLin) uploadsFreshBecauseLinearized++;                     // (counted: in LDSO's own flow isLinearized is never set - FullSystem.cc:1241-1250 fixes and marginalises in one go)
}

// The window as a delta against the one the last optimize() left on the device (ldso_ba_update_window).  Surviving points are recognised by walking the new
// window (frames, then the host frame's features: makeIDX order) against the rows of the resident one - both in the same stable order, so one pass with two
// cursors; a row is skipped when its point is gone (host frame left, point no longer ACTIVE).  Returns false when the resident window cannot express the new
// one (linearised residuals, an inconsistency): the caller uploads everything.
bool GpuBackend::uploadDelta(FullSystem &fs, const std::vector<int32_t> &slots, std::vector<shared_ptr<PointHessian>> &allPoints) {
    const int F = (int) fs.frames.size(), oF = (int) rowFrames_.size();
    if (F > kMaxCols || oF > kMaxCols || rows_.empty()) return false;
    int32_t frameFrom[kMaxCols], oldToNew[kMaxCols];
    for (int c = 0; c < kMaxCols; c++) { frameFrom[c] = -1; oldToNew[c] = -1; }
    int lastOld = -1, nInserted = 0;
    for (int f = 0; f < F; f++) {
        for (int o = 0; o < oF; o++) if (rowFrames_[o].get() == fs.frames[f].get()) frameFrom[f] = o;          // identity, not Frame::id: another object graph may reuse the ids
        if (frameFrom[f] >= 0) { if (frameFrom[f] <= lastOld) return false; lastOld = frameFrom[f]; oldToNew[frameFrom[f]] = f; } else nInserted++;
    }
    if (nInserted == F) return false;          // nothing of the resident window is left
    const size_t oP = rows_.size();
    auto alive = [&](const Row &r) {          // the frame first: features of a frame that left the window may be gone with it
        return oldToNew[r.host] >= 0 && r.feat->status == Feature::FeatureStatus::VALID && r.feat->point && r.feat->point->status == Point::PointStatus::ACTIVE
               && r.feat->point->mpPH.get() == r.ph;
    };
    std::vector<Row> rows; rows.reserve(oP + 512);
    std::vector<int32_t> pointFrom; pointFrom.reserve(oP + 512);
    std::vector<uint32_t> mask; mask.reserve(oP + 512);
    std::vector<shared_ptr<PointHessian>> pts; pts.reserve(oP + 512);
    std::vector<ldso_point_t> fresh; std::vector<ldso_residual_t> freshRes; std::vector<float> fmrb; std::vector<int32_t> fngr;
    size_t j = 0;
    auto columnOf = [&](const PointFrameResidual &r) { const shared_ptr<FrameHessian> t = r.target.lock(); return t ? t->idx : -1; };      // FrameHessian::idx = window position (set above)
    for (int f = 0; f < F; f++)
        for (shared_ptr<Feature> &feat : fs.frames[f]->features) {
            if (!(feat->status == Feature::FeatureStatus::VALID && feat->point && feat->point->status == Point::PointStatus::ACTIVE)) continue;
            PointHessian *ph = feat->point->mpPH.get();
            while (j < oP && rows_[j].ph != ph && !alive(rows_[j])) j++;
            Row n; n.ph = ph; n.feat = feat.get(); n.mask = 0; n.host = f;
            for (int c = 0; c < kMaxCols; c++) n.res[c] = nullptr;
            const int have = (int) ph->residuals.size();
            auto reread = [&]() -> bool {          // the point's residual list from the objects
                n.mask = 0;
                for (int c = 0; c < kMaxCols; c++) n.res[c] = nullptr;
                for (shared_ptr<PointFrameResidual> &r : ph->residuals) {
                    const int t = columnOf(*r);
                    if (r->isLinearized || t < 0 || t >= F || t == f || ((n.mask >> t) & 1u)) return false;
                    n.mask |= 1u << t; n.res[t] = r.get();
                }
                return true;
            };
            if (j < oP && rows_[j].ph == ph) {
                // ---- a point of the resident window: its residuals by column, minus the frames that left, plus what was appended since ----
                const Row &o = rows_[j];
                if (oldToNew[o.host] != f) return false;
                for (int c = 0; c < oF; c++)
                    if (((o.mask >> c) & 1u) && oldToNew[c] >= 0) { n.mask |= 1u << oldToNew[c]; n.res[oldToNew[c]] = o.res[c]; }
                int cnt = __builtin_popcount(n.mask);
                // (a point that got NO residual for a frame inserted since - LDSO gives every active point one, FullSystem.cc:447-470 - may as well have lost one
                // and gained one: equal counts prove nothing then, its list is re-read)
                bool ok = have >= cnt && have - cnt <= nInserted && !(nInserted > 0 && have == cnt);
                for (int k = cnt; ok && k < have; k++) {          // insertResidual appends (FullSystem.cc:447-470): the new ones are the last
                    PointFrameResidual &r = *ph->residuals[k];
                    const int t = columnOf(r);
                    ok = !r.isLinearized && t >= 0 && t < F && t != f && frameFrom[t] < 0 && !((n.mask >> t) & 1u);
                    if (ok) { n.mask |= 1u << t; n.res[t] = &r; }
                }
                // every residual of the point's CURRENT list must be one of the pointers cached above, and every cached pointer must still be in the list: a host
                // that dropped a residual and created another one for the same target between two optimize() calls (equal counts, nothing inserted) would
                // otherwise leave a dangling PointFrameResidual* in flat_ (round-5 advisor).  targetIDX is the hint (makeIDX keeps it current in LDSO's own
                // flow: one compare per residual), the scan over the row is the fallback.
                if (ok) {
                    uint32_t seen = 0;
                    for (int k = 0; ok && k < have; k++) {
                        const PointFrameResidual *r = ph->residuals[k].get();
                        int c = r->targetIDX;
                        if (!(c >= 0 && c < F && n.res[c] == r)) { c = -1; for (int t = 0; t < F; t++) if (n.res[t] == r) { c = t; break; } }
                        if (c < 0 || ((seen >> c) & 1u)) ok = false; else seen |= 1u << c;
                    }
                    ok = ok && seen == n.mask;
                    if (!ok) residentPointerMismatches++;
                }
                if (!ok && !reread()) return false;
                pointFrom.push_back((int32_t) j);
                pts.push_back(lastPoints_[j]);
                j++;
            } else {
                // ---- a fresh point (activated since the last optimize()): its records cross PCIe ----
                if (!reread()) return false;
                const int k = (int) fresh.size();
                ldso_point_t p;
                flatPoint(*ph, f, p);
                p.res_begin = (int32_t) freshRes.size(); p.res_count = have;
                fresh.push_back(p); fmrb.push_back(ph->maxRelBaseline); fngr.push_back(ph->numGoodResiduals);
                for (int t = 0; t < F; t++) if ((n.mask >> t) & 1u) { ldso_residual_t q; flatResidual(*n.res[t], k, f, t, q); freshRes.push_back(q); }
                pointFrom.push_back(-1 - k);
                pts.push_back(feat->point->mpPH);
            }
            rows.push_back(n); mask.push_back(n.mask);
        }
    for (; j < oP; j++) if (alive(rows_[j])) return false;          // a point of the resident window that the walk did not meet
    if (rows.empty()) return false;
    lastUploadSeconds[1] += lapSince();
    const int rc = ldso_ba_update_window(ba_, F, slots.data(), frameFrom, (int) rows.size(), pointFrom.data(), mask.data(), (int) fresh.size(), fresh.data(), (int) freshRes.size(),
                                         freshRes.data(), fmrb.data(), fngr.data());
    lastUploadSeconds[2] += lapSince();
    if (rc != LDSO_OK) { LOG(WARNING) << "GpuBackend: ldso_ba_update_window refused the delta (" << ldso_last_error() << "), uploading the whole window"; return false; }
    rows_.swap(rows); allPoints.swap(pts);
    flat_.clear(); resBegin_.assign(1, 0);
    for (const Row &r : rows_) {
        for (int t = 0; t < F; t++) if ((r.mask >> t) & 1u) flat_.push_back(r.res[t]);
        resBegin_.push_back((int32_t) flat_.size());
    }
    return true;
}

int GpuBackend::uploadWindow(FullSystem &fs, std::vector<shared_ptr<PointHessian>> &allPoints, bool trustIndices) {
    lapSince();
    for (double &t : lastUploadSeconds) t = 0;
    const ldso_settings_t st = flatSettings();
    throwOn(ldso_ba_set_settings(ba_, &st), "ldso_ba_set_settings");
    std::vector<int32_t> slots;
    syncImageSlots(fs, slots);
    lastUploadSeconds[0] += lapSince();
    const int F = (int) fs.frames.size();
    std::vector<ldso_frame_t> Fv((size_t) F);
    for (int f = 0; f < F; f++) {
        FrameHessian &fh = *fs.frames[f]->frameHessian;
        fh.idx = f;
        ldso_frame_t &o = Fv[f];
        memset(&o, 0, sizeof(o));
        toRowMajor34(fh.get_worldToCam_evalPT(), o.worldToCam_evalPT);
        for (int i = 0; i < 10; i++) { o.state[i] = fh.get_state()[i]; o.state_zero[i] = fh.get_state_zero()[i]; }
        for (int i = 0; i < 8; i++) o.prior[i] = fh.prior[i];                                      // FrameHessian::takeData (FrameHessian.cc:115)
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) o.nullspaces_pose[r * 6 + c] = fh.nullspaces_pose(r, c);
        for (int r = 0; r < 6; r++) o.nullspaces_scale[r] = fh.nullspaces_scale[r];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 2; c++) o.nullspaces_affine[r * 2 + c] = fh.nullspaces_affine(r, c);
        o.ab_exposure = fh.ab_exposure; o.frameEnergyTH = fh.frameEnergyTH; o.frameID = (int32_t) fh.frame->id;   // getPrior keys on frame->id == 0
    }
    lastUploadWasDelta = residentWindow && residentValid_ && uploadDelta(fs, slots, allPoints);
    if (!lastUploadWasDelta) { residentValid_ = false; uploadFresh(fs, slots, allPoints, trustIndices); }
    (lastUploadWasDelta ? uploadsDelta : uploadsFresh)++;
    if (allPoints.empty()) { residentValid_ = false; return 0; }
    rowFrames_.assign(fs.frames.begin(), fs.frames.end());
    lastPoints_ = allPoints;                                         // keeps the rows' PointHessian objects alive while the window is resident
    lapSince();
    const ldso_calib_t c = flatCalib(*fs.Hcalib->mpCH);
    throwOn(ldso_ba_set_frames(ba_, Fv.data(), &c), "ldso_ba_set_frames");                      // setAdjointsF + setPrecalcValues on the device
    lastUploadSeconds[4] += lapSince();
    const int n = CPARS + 8 * F;
    if ((int) fs.ef->HM.rows() == n) {
        std::vector<double> HM((size_t) n * n), bM((size_t) n);
        for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HM[(size_t) i * n + j] = fs.ef->HM(i, j); bM[i] = fs.ef->bM[i]; }
        throwOn(ldso_ba_set_prior(ba_, HM.data(), bM.data()), "ldso_ba_set_prior");
    }
    lastUploadSeconds[5] += lapSince();
    return (int) allPoints.size();
}

// ------------------------------------------------------------------------------------------------------------------------------------
// float FullSystem::optimize(int mnumOptIts)                                                                       FullSystem.cc:725-864
// ------------------------------------------------------------------------------------------------------------------------------------
float GpuBackend::optimize(FullSystem &fs, int mnumOptIts) {
    if (fs.frames.size() < 2) return 0;
    if (fs.frames.size() < 3) mnumOptIts = 20;
    if (fs.frames.size() < 4) mnumOptIts = 15;

    std::vector<shared_ptr<PointHessian>> allPoints;
    if (useDevicePyramids) releasePyramids(fs);
    const auto tWall0 = std::chrono::steady_clock::now();
    auto lap = [&](int i, std::chrono::steady_clock::time_point &t) { const auto n_ = std::chrono::steady_clock::now(); lastOptimizeSeconds[i] = std::chrono::duration<double>(n_ - t).count(); t = n_; };
    auto tLap = tWall0;
    if (uploadWindow(fs, allPoints, /*trustIndices*/ true) == 0) return 0;
    lap(0, tLap);
    const int F = (int) fs.frames.size(), P = (int) allPoints.size(), R = (int) flat_.size();
    const std::vector<PointFrameResidual *> &flat = flat_;

    // activeResiduals (:735-755): only FullSystem::optimize itself reads the list (see fillActiveResiduals in the header)
    fs.activeResiduals.clear();
    if (fillActiveResiduals)
        for (int k = 0; k < P; k++) for (shared_ptr<PointFrameResidual> &r : allPoints[k]->residuals) if (!r->isLinearized) fs.activeResiduals.push_back(r);

    // the whole loop of :757-843 and the tail :845-851 (re-anchor the newest frame, adjoints, precalc, linearizeAll(true)) on the device
    float rmse = 0;
    const int rc = ldso_ba_optimize(ba_, mnumOptIts, /*force_all_iterations*/ 0, &rmse, &lastIterations);
    throwOn(rc, "ldso_ba_optimize");
    if (rc == LDSO_E_NONFINITE) { LOG(WARNING) << "KF Tracking failed: LOST!"; fs.isLost = true; }         // :853-857
    lap(1, tLap);

    // ---- write back ----------------------------------------------------------------------------------------------------------------
    // the landing buffers are members: 1 MB of fresh vectors per call is 1 MB of page faults and zero fill per call; everything in them is overwritten by the fetch
    if (wbRes_.size() < (size_t) R) { wbRes_.resize((size_t) R); wbState_.resize((size_t) R); wbActive_.resize((size_t) R); wbRemove_.resize((size_t) R); }
    if (wbPoints_.size() < (size_t) P) wbPoints_.resize((size_t) P);
    if (wbFrames_.size() < (size_t) F) { wbFrames_.resize((size_t) F); wbStep_.resize((size_t) F * 10); }
    std::vector<ldso_res_out_t> &ro = wbRes_; std::vector<int32_t> &st = wbState_, &act = wbActive_, &rem = wbRemove_;
    std::vector<ldso_point_out_t> &po = wbPoints_; std::vector<ldso_frame_t> &fo = wbFrames_; std::vector<double> &fstep = wbStep_;
    double cv[4], cs[4];
    throwOn(ldso_ba_get_results(ba_, ro.data(), st.data(), act.data(), rem.data(), po.data(), fo.data(), fstep.data(), cv, cs), "ldso_ba_get_results");      // one synchronisation
    std::vector<ldso_rawjac_t> Jv;
    if (writeBackJacobians) {
        std::vector<int32_t> ids((size_t) R);
        for (int i = 0; i < R; i++) ids[i] = i;
        Jv.resize((size_t) R);
        throwOn(ldso_ba_get_jacobians(ba_, ids.data(), R, Jv.data()), "ldso_ba_get_jacobians");
    }

    lap(2, tLap);
    // calibration and frames: the states through the reference's own setters (they rebuild state_scaled, PRE_worldToCam / PRE_camToWorld);
    // the newest frame was re-anchored (:845-848: setEvalPT(PRE_worldToCam, [0..0, a, b, 0, 0]))
    VecC v, vs;
    for (int i = 0; i < 4; i++) { v[i] = cv[i]; vs[i] = cs[i]; }
    fs.Hcalib->mpCH->setValue(v); fs.Hcalib->mpCH->step = vs;
    for (int f = 0; f < F; f++) {
        FrameHessian &fh = *fs.frames[f]->frameHessian;
        Vec10 s;
        for (int i = 0; i < 10; i++) { s[i] = fo[f].state[i]; fh.step[i] = fstep[(size_t) f * 10 + i]; }
        if (f == F - 1) fh.setEvalPT(fromRowMajor34(fo[f].worldToCam_evalPT), s); else fh.setState(s);
        fh.frameEnergyTH = fo[f].frameEnergyTH;                                                  // setNewFrameEnergyTH (:1762-1793)
    }
    // points (doStepFromBackup :1595-1606, AccumulatedSCHessian.cc:9-51, the fixing pass :1521-1536)
    for (int k = 0; k < P; k++) {
        if (k + 8 < P) { const char *nx = (const char *) allPoints[k + 8].get(); __builtin_prefetch(nx, 1); __builtin_prefetch(nx + 64, 1); __builtin_prefetch(nx + 192, 1); __builtin_prefetch(nx + 256, 1); }      // 304-byte objects
        PointHessian &ph = *allPoints[k];
        ph.setIdepth(po[k].idepth); ph.setIdepthZero(po[k].idepth);
        ph.step = po[k].step; ph.HdiF = po[k].HdiF; ph.bdSumF = po[k].bdSumF; ph.idepth_hessian = po[k].idepth_hessian;
        ph.Hdd_accAF = po[k].Hdd_accAF; ph.bd_accAF = po[k].bd_accAF; ph.Hdd_accLF = po[k].Hdd_accLF; ph.bd_accLF = po[k].bd_accLF;
        for (int i = 0; i < 4; i++) { ph.Hcd_accAF[i] = po[k].Hcd_accAF[i]; ph.Hcd_accLF[i] = po[k].Hcd_accLF[i]; }
        ph.maxRelBaseline = po[k].maxRelBaseline; ph.numGoodResiduals = po[k].numGoodResiduals;
    }
    // residuals: applyRes(true) of the fixing pass (Residuals.h:70-87), lastResiduals bookkeeping and removal (:1472-1489)
    for (int i = 0; i < R; i++) {
        // 12 000 separately allocated objects: the walk is a chain of cache misses unless the next ones are already on their way
        if (i + 12 < R) { const char *nx = (const char *) flat[i + 12]; __builtin_prefetch(nx, 1); __builtin_prefetch(nx + 176, 1); __builtin_prefetch(nx + 240, 1); __builtin_prefetch(nx + 287, 1); }      // state | centre | JpJdF (288-byte objects)
        PointFrameResidual &r = *flat[i];
        if (r.isLinearized) continue;                                                            // not in activeResiduals: untouched by optimize()
        r.state_NewEnergy = ro[i].state_NewEnergy; r.state_NewEnergyWithOutlier = ro[i].state_NewEnergyWithOutlier; r.state_NewState = (ResState) ro[i].state_NewState;
        r.state_state = (ResState) st[i]; r.state_energy = ro[i].state_NewEnergy; r.isActiveAndIsGoodNEW = act[i] != 0;
        for (int k = 0; k < 3; k++) r.centerProjectedTo[k] = ro[i].centerProjectedTo[k];
        for (int k = 0; k < 8; k++) r.JpJdF[k] = ro[i].JpJdF[k];
        if (writeBackJacobians && r.state_NewState != ResState::OOB) fromRaw(Jv[i], *r.J);
    }
    // (the residuals of point k are flat[resBegin[k] .. resBegin[k + 1]): no weak_ptr::lock() per residual to find the point)
    for (int k = 0; k < P; k++) {
        PointHessian &ph = *allPoints[k];
        for (int i = resBegin_[k]; i < resBegin_[k + 1]; i++) {
            PointFrameResidual *r = flat[i];
            if (r->isLinearized) continue;
            if (ph.lastResiduals[0].first.get() == r) ph.lastResiduals[0].second = r->state_state;
            else if (ph.lastResiduals[1].first.get() == r) ph.lastResiduals[1].second = r->state_state;
        }
    }
    for (int k = 0; k < P; k++) {
        PointHessian &ph = *allPoints[k];
        for (int i = resBegin_[k]; i < resBegin_[k + 1]; i++) {
            if (!rem[i]) continue;
            shared_ptr<PointFrameResidual> r;
            for (shared_ptr<PointFrameResidual> &q : ph.residuals) if (q.get() == flat[i]) { r = q; break; }
            if (!r) continue;
            if (ph.lastResiduals[0].first == r) ph.lastResiduals[0].first = 0;
            else if (ph.lastResiduals[1].first == r) ph.lastResiduals[1].first = 0;
            fs.ef->dropResidual(r);                                                               // EnergyFunctional.cc:44-56
            // ... and out of the resident window's host view: the next delta says dropResidual through the cleared bit
            Row &row = rows_[k];
            for (int t = 0; t < kMaxCols; t++) if (row.res[t] == flat[i]) { row.res[t] = nullptr; row.mask &= ~(1u << t); }
        }
    }
    int resInA = 0, resInL = 0;
    throwOn(ldso_ba_get_counts(ba_, &resInA, &resInL), "ldso_ba_get_counts");
    fs.ef->resInA = resInA; fs.ef->resInL = resInL;

    // host mirrors of what the device computed on its side (:849-851): adjoints, pair precalc, deltas - the reference's own functions
    EFDeltaValid = false; EFAdjointsValid = false;
    fs.ef->setAdjointsF(fs.Hcalib->mpCH);
    fs.setPrecalcValues();

    // :859-869 unchanged: hand the estimated poses to the frames
    {
        unique_lock<mutex> crlock(fs.shellPoseMutex);
        for (auto fr : fs.frames) {
            fr->setPose(fr->frameHessian->PRE_camToWorld.inverse());
            if (fr->kfId >= fs.globalMap->getLatestOptimizedKfId()) fr->setPoseOpti(Sim3(fr->getPose().matrix()));
            fr->aff_g2l = fr->frameHessian->aff_g2l();
        }
    }
    lap(3, tLap);
    return rmse;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// void FullSystem::flagPointsForRemoval()                                                                       FullSystem.cc:1208-1270
// The policy, line for line; the inner loop of :1241-1250 (re-linearise and fix the residuals of a point that is marginalised) is what
// ldso_ba_marginalize_points runs on the device for the points flagged here.
// ------------------------------------------------------------------------------------------------------------------------------------
void GpuBackend::flagPointsForRemoval(FullSystem &fs) {
    std::vector<shared_ptr<FrameHessian>> fhsToMargPoints;
    for (int i = 0; i < (int) fs.frames.size(); i++)
        if (fs.frames[i]->frameHessian->flaggedForMarginalization) fhsToMargPoints.push_back(fs.frames[i]->frameHessian);
    for (auto &fr : fs.frames) {
        shared_ptr<FrameHessian> host = fr->frameHessian;
        for (auto &feat : fr->features) {
            if (!(feat->status == Feature::FeatureStatus::VALID && feat->point->status == Point::PointStatus::ACTIVE)) continue;
            shared_ptr<PointHessian> ph = feat->point->mpPH;
            if (ph->idepth_scaled < 0 || ph->residuals.size() == 0) {
                ph->point->status = Point::PointStatus::OUTLIER;
                feat->status = Feature::FeatureStatus::OUTLIER;
            } else if (ph->isOOB(fhsToMargPoints) || host->flaggedForMarginalization) {
                if (ph->isInlierNew()) {
                    // (:1241-1250 happens on the device, see marginalizePoints)
                    if (ph->idepth_hessian > setting_minIdepthH_marg) ph->point->status = Point::PointStatus::MARGINALIZED;
                    else ph->point->status = Point::PointStatus::OUT;
                } else ph->point->status = Point::PointStatus::OUT;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// void EnergyFunctional::marginalizePointsF()                                                          EnergyFunctional.cc:165-222
// ------------------------------------------------------------------------------------------------------------------------------------
void GpuBackend::marginalizePoints(FullSystem &fs) {
    EnergyFunctional &ef = *fs.ef;
    const int P = (int) lastPoints_.size();
    if (P == 0) throw std::runtime_error("GpuBackend::marginalizePoints: no window resident (call optimize first)");
    std::vector<int32_t> flags((size_t) P, 0);
    int nMarg = 0;
    for (int k = 0; k < P; k++)
        if (lastPoints_[k]->point && lastPoints_[k]->point->status == Point::PointStatus::MARGINALIZED) { flags[k] = 1; nMarg++; }
    const int n = CPARS + 8 * (int) fs.frames.size();
    if ((int) ef.HM.rows() != n) throw std::runtime_error("GpuBackend::marginalizePoints: the prior does not have the window's dimension");
    if (nMarg > 0) {
        // (the settings the device pass reads - margWeightFac, idepthFixPriorMargFac - went up with optimize(); so did ef.HM / ef.bM)
        std::vector<double> HM((size_t) n * n), bM((size_t) n);
        throwOn(ldso_ba_marginalize_points(ba_, flags.data(), HM.data(), bM.data()), "ldso_ba_marginalize_points");
        for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) ef.HM(i, j) = HM[(size_t) i * n + j]; ef.bM[i] = bM[i]; }
    }
    // the reference's bookkeeping around the accumulation (:169-184, :193, :199, :219-220)
    ef.allPointsToMarg.clear();
    for (int k = 0; k < P; k++) {
        if (!flags[k]) continue;
        shared_ptr<PointHessian> p = lastPoints_[k];
        p->priorF *= setting_idepthFixPriorMargFac;
        for (auto r : p->residuals)
            if (r->isActive()) { ef.connectivityMap[(((uint64_t) r->host.lock()->frameID) << 32) + ((uint64_t) r->target.lock()->frameID)][1]++; ef.resInM++; }
        ef.allPointsToMarg.push_back(p);
    }
    for (auto p : ef.allPointsToMarg) ef.removePoint(p);
    EFIndicesValid = false;
    ef.makeIDX();
}

// ------------------------------------------------------------------------------------------------------------------------------------
// void FullSystem::marginalizeFrame(shared_ptr<Frame> &frame)                                                      FullSystem.cc:602-645
// ------------------------------------------------------------------------------------------------------------------------------------
void GpuBackend::marginalizeFrame(FullSystem &fs, shared_ptr<Frame> &frame) {
    EnergyFunctional &ef = *fs.ef;
    shared_ptr<FrameHessian> fh = frame->frameHessian;
    const int F = (int) fs.frames.size(), n = CPARS + 8 * F, nd = n - 8;
    int col = -1;
    for (int f = 0; f < F; f++) if (fs.frames[f].get() == frame.get()) col = f;
    const bool resident = residentValid_ && (int) rowFrames_.size() == F && col >= 0 && rowFrames_[col].get() == frame.get();
    if (!resident || (int) ef.HM.rows() != n) { margFrameHostFallback++; fs.marginalizeFrame(frame); return; }          // no window of these frames on the device: the reference's member (counted)
    margFrameDevice++;
    // ---- EnergyFunctional::marginalizeFrame: the arithmetic (:80-136) on the device, on ef's own prior ----
    {
        std::vector<double> HM((size_t) n * n), bM((size_t) n), oH((size_t) nd * nd), ob((size_t) nd);
        for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HM[(size_t) i * n + j] = ef.HM(i, j); bM[i] = ef.bM[i]; }
        throwOn(ldso_ba_set_prior(ba_, HM.data(), bM.data()), "ldso_ba_set_prior");
        throwOn(ldso_ba_marginalize_frame(ba_, col, oH.data(), ob.data()), "ldso_ba_marginalize_frame");
        ef.HM.resize(nd, nd); ef.bM.resize(nd);
        for (int i = 0; i < nd; i++) { for (int j = 0; j < nd; j++) ef.HM(i, j) = oH[(size_t) i * nd + j]; ef.bM[i] = ob[i]; }
    }
    // ... its bookkeeping (:138-150)
    for (unsigned int i = fh->idx; i + 1 < ef.frames.size(); i++) { ef.frames[i] = ef.frames[i + 1]; ef.frames[i]->idx = i; }
    ef.frames.pop_back();
    ef.nFrames--;
    EFIndicesValid = false; EFAdjointsValid = false; EFDeltaValid = false;
    ef.makeIDX();
    // ---- drop all observations of existing points in that frame (:607-632): column `col` of the resident rows ----
    for (Row &row : rows_) {
        PointFrameResidual *rp = row.res[col];
        if (!rp || row.host == col) continue;
        if (!(row.feat->status == Feature::FeatureStatus::VALID && row.feat->point && row.feat->point->status == Point::PointStatus::ACTIVE)) continue;
        PointHessian &ph = *row.ph;
        shared_ptr<PointFrameResidual> r;
        for (shared_ptr<PointFrameResidual> &q : ph.residuals) if (q.get() == rp) { r = q; break; }
        if (!r) continue;
        if (ph.lastResiduals[0].first == r) ph.lastResiduals[0].first = nullptr;
        else if (ph.lastResiduals[1].first == r) ph.lastResiduals[1].first = nullptr;
        ef.dropResidual(r);
        row.res[col] = nullptr; row.mask &= ~(1u << col);
    }
    // ---- :634-644 ----
    frame->ReleaseAll();
    ldso::internal::deleteOutOrder<shared_ptr<Frame>>(fs.frames, frame);
    for (unsigned int i = 0; i < fs.frames.size(); i++) fs.frames[i]->frameHessian->idx = i;
    fs.setPrecalcValues();
    ef.setAdjointsF(fs.Hcalib->mpCH);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Point activation: the optimizeImmaturePoint calls of FullSystem::activatePointsMT_Reductor (:1196-1206) as one device call, then the
// object construction of optimizeImmaturePoint's tail (:977-1008) on the host.
// ------------------------------------------------------------------------------------------------------------------------------------
void GpuBackend::activatePoints(FullSystem &fs, std::vector<shared_ptr<ImmaturePoint>> &toOptimize, std::vector<shared_ptr<PointHessian>> &optimized) {
    optimized.assign(toOptimize.size(), nullptr);
    if (toOptimize.empty()) return;
    std::vector<shared_ptr<PointHessian>> allPoints;
    if (uploadWindow(fs, allPoints) == 0) throw std::runtime_error("GpuBackend::activatePoints: the window holds no active point yet (activate on the host)");
    const int F = (int) fs.frames.size();
    std::vector<ldso_immature_t> in(toOptimize.size());
    for (size_t k = 0; k < toOptimize.size(); k++) {
        ImmaturePoint &ip = *toOptimize[k];
        ldso_immature_t &q = in[k];
        memset(&q, 0, sizeof(q));
        q.u = ip.feature->uv[0]; q.v = ip.feature->uv[1];
        memcpy(q.color, ip.color, sizeof(q.color)); memcpy(q.weights, ip.weights, sizeof(q.weights));
        q.gradH[0] = ip.gradH(0, 0); q.gradH[1] = ip.gradH(0, 1); q.gradH[2] = ip.gradH(1, 0); q.gradH[3] = ip.gradH(1, 1);
        q.energyTH = ip.energyTH; q.idepth_min = ip.idepth_min; q.idepth_max = ip.idepth_max; q.quality = ip.quality;
        q.lastTraceStatus = (int32_t) ip.lastTraceStatus; q.lastTraceUV[0] = ip.lastTraceUV[0]; q.lastTraceUV[1] = ip.lastTraceUV[1];
        q.lastTracePixelInterval = ip.lastTracePixelInterval;
        q.host = ip.feature->host.lock()->frameHessian->idx;
    }
    std::vector<ldso_activation_t> out(toOptimize.size());
    throwOn(ldso_ba_activate_points(ba_, (int) in.size(), in.data(), /*minObs*/ 1, setting_minIdepthH_act, setting_GNItsOnPointActivation, out.data()), "ldso_ba_activate_points");
    for (size_t k = 0; k < toOptimize.size(); k++) {
        if (!out[k].ok) continue;                                                                // return 0 / nullptr (:924-926, :945-947, :968-975)
        shared_ptr<ImmaturePoint> point = toOptimize[k];
        point->feature->CreateFromImmature();                                                    // :977
        shared_ptr<PointHessian> p = point->feature->point->mpPH;
        p->lastResiduals[0].first = nullptr; p->lastResiduals[0].second = ResState::OOB;
        p->lastResiduals[1].first = nullptr; p->lastResiduals[1].second = ResState::OOB;
        p->setIdepthZero(out[k].idepth); p->setIdepth(out[k].idepth);
        shared_ptr<FrameHessian> host = point->feature->host.lock()->frameHessian;
        for (int t = 0; t < F; t++) {
            if (out[k].res_state[t] != 0) continue;                                              // ResState::IN only (:989)
            shared_ptr<FrameHessian> target = fs.frames[t]->frameHessian;
            shared_ptr<PointFrameResidual> r(new PointFrameResidual(p, host, target));
            r->state_NewEnergy = r->state_energy = 0; r->state_NewState = ResState::OUTLIER; r->setState(ResState::IN);
            p->residuals.push_back(r);
            if (target == fs.frames.back()->frameHessian) { p->lastResiduals[0].first = r; p->lastResiduals[0].second = ResState::IN; }
            else if (target == (fs.frames.size() < 2 ? nullptr : fs.frames[fs.frames.size() - 2]->frameHessian)) { p->lastResiduals[1].first = r; p->lastResiduals[1].second = ResState::IN; }
        }
        optimized[k] = p;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// void FullSystem::traceNewCoarse(shared_ptr<FrameHessian> fh)                                                   FullSystem.cc:1012-1050
// Every immature point of the window's key frames is traced into the new frame by one ldso_trace_on call (one wavefront per point);
// the per-host KRKi / Kt / affine transfer are formed here exactly as the reference forms them (its own Eigen / Sophus expressions).
// ------------------------------------------------------------------------------------------------------------------------------------
void GpuBackend::traceNewCoarse(FullSystem &fs, shared_ptr<FrameHessian> fh) {
    unique_lock<mutex> lock(fs.mapMutex);
    for (int i = 0; i < 6; i++) lastTraceCounts[i] = 0;
    const int F = (int) fs.frames.size();
    std::vector<ldso_immature_t> rec;
    std::vector<ImmaturePoint *> who;
    for (int f = 0; f < F; f++)
        for (auto &feat : fs.frames[f]->features) {
            if (!(feat->status == Feature::FeatureStatus::IMMATURE && feat->ip)) continue;
            ImmaturePoint &ip = *feat->ip;
            ldso_immature_t q;
            memset(&q, 0, sizeof(q));
            q.u = feat->uv[0]; q.v = feat->uv[1];
            memcpy(q.color, ip.color, sizeof(q.color)); memcpy(q.weights, ip.weights, sizeof(q.weights));
            q.gradH[0] = ip.gradH(0, 0); q.gradH[1] = ip.gradH(0, 1); q.gradH[2] = ip.gradH(1, 0); q.gradH[3] = ip.gradH(1, 1);
            q.energyTH = ip.energyTH; q.idepth_min = ip.idepth_min; q.idepth_max = ip.idepth_max; q.quality = ip.quality;
            q.lastTraceStatus = (int32_t) ip.lastTraceStatus; q.lastTraceUV[0] = ip.lastTraceUV[0]; q.lastTraceUV[1] = ip.lastTraceUV[1];
            q.lastTracePixelInterval = ip.lastTracePixelInterval; q.host = f;
            rec.push_back(q); who.push_back(&ip);
        }
    if (rec.empty()) return;
    if (!tracer_ || (int) rec.size() > tracerCap_) {
        if (tracer_) ldso_trace_destroy(tracer_);
        tracer_ = nullptr;
        tracerCap_ = std::max((int) rec.size() * 2, 16384);
        throwOn(ldso_trace_create(device_, wG[0], hG[0], tracerCap_, &tracer_), "ldso_trace_create");
    }
    ldso_trace_settings_t ts;
    ldso_trace_settings_default(&ts);
    ts.maxPixSearch = setting_maxPixSearch; ts.trace_stepsize = setting_trace_stepsize; ts.trace_GNThreshold = setting_trace_GNThreshold;
    ts.trace_extraSlackOnTH = setting_trace_extraSlackOnTH; ts.trace_slackInterval = setting_trace_slackInterval;
    ts.trace_minImprovementFactor = setting_trace_minImprovementFactor; ts.huberTH = setting_huberTH;
    ts.trace_GNIterations = setting_trace_GNIterations; ts.minTraceTestRadius = setting_minTraceTestRadius;
    throwOn(ldso_trace_set_settings(tracer_, &ts), "ldso_trace_set_settings");
    // :1018-1032, per host frame
    Mat33f K = Mat33f::Identity();
    K(0, 0) = fs.Hcalib->mpCH->fxl(); K(1, 1) = fs.Hcalib->mpCH->fyl(); K(0, 2) = fs.Hcalib->mpCH->cxl(); K(1, 2) = fs.Hcalib->mpCH->cyl();
    std::vector<float> KRKi((size_t) F * 9), Kt((size_t) F * 3), aff((size_t) F * 2);
    for (int f = 0; f < F; f++) {
        shared_ptr<FrameHessian> host = fs.frames[f]->frameHessian;
        SE3 hostToNew = fh->PRE_worldToCam * host->PRE_camToWorld;
        Mat33f M = K * hostToNew.rotationMatrix().cast<float>() * K.inverse();
        Vec3f t = K * hostToNew.translation().cast<float>();
        Vec2f a = AffLight::fromToVecExposure(host->ab_exposure, fh->ab_exposure, host->aff_g2l(), fh->aff_g2l()).cast<float>();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) KRKi[(size_t) f * 9 + r * 3 + c] = M(r, c); Kt[(size_t) f * 3 + r] = t[r]; }
        aff[(size_t) f * 2] = a[0]; aff[(size_t) f * 2 + 1] = a[1];
    }
    throwOn(ldso_trace_set_points(tracer_, (int) rec.size(), rec.data()), "ldso_trace_set_points");
    if (useDevicePyramids && fh->frame) { tracerPyr_ = pyramidOf(fh); throwOn(ldso_trace_set_frame_pyramid(tracer_, tracerPyr_->p), "ldso_trace_set_frame_pyramid"); }
    else throwOn(ldso_trace_set_frame(tracer_, (const float *) fh->dIp[0]), "ldso_trace_set_frame");
    throwOn(ldso_trace_on(tracer_, F, KRKi.data(), Kt.data(), aff.data(), lastTraceCounts), "ldso_trace_on");
    throwOn(ldso_trace_get_points(tracer_, rec.data()), "ldso_trace_get_points");
    for (size_t i = 0; i < rec.size(); i++) {          // what ImmaturePoint::traceOn leaves in the object (ImmaturePoint.cc:47-310)
        ImmaturePoint &ip = *who[i]; const ldso_immature_t &q = rec[i];
        ip.idepth_min = q.idepth_min; ip.idepth_max = q.idepth_max; ip.quality = q.quality;
        ip.lastTraceStatus = (ImmaturePointStatus) q.lastTraceStatus;
        ip.lastTraceUV = Vec2f(q.lastTraceUV[0], q.lastTraceUV[1]); ip.lastTracePixelInterval = q.lastTracePixelInterval;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// CoarseTracker
// ------------------------------------------------------------------------------------------------------------------------------------
ldso_tracker_t *GpuBackend::trackerOf(CoarseTracker &tr) {
    std::lock_guard<std::mutex> lk(handlesMutex_);
    auto it = trackers_.find(&tr);
    if (it != trackers_.end()) return it->second;
    ldso_tracker_t *t = nullptr;
    throwOn(ldso_tr_create(device_, wG[0], hG[0], pyrLevelsUsed, &t), "ldso_tr_create");
    trackers_[&tr] = t;
    return t;
}

// true when handle t already holds this frame's pyramid as its new frame; otherwise records that it is about to
bool GpuBackend::newFrameResident(ldso_tracker_t *t, const shared_ptr<FrameHessian> &fh) {
    const unsigned long id = fh->frame ? fh->frame->id : ~0ul;
    std::lock_guard<std::mutex> lk(handlesMutex_);
    auto it = trackerNewFrameId_.find(t);
    if (it != trackerNewFrameId_.end() && it->second == id && id != ~0ul) return true;
    trackerNewFrameId_[t] = id;
    return false;
}

// void CoarseTracker::makeK(shared_ptr<CalibHessian>)                                                            CoarseTracker.cc:219-246
void GpuBackend::makeK(CoarseTracker &tr, shared_ptr<CalibHessian> HCalib) {
    tr.makeK(HCalib);                                  // the host copies (w[], h[], K[] ...) other LDSO code reads; 20 scalar operations
    ldso_tracker_t *t = trackerOf(tr);
    const ldso_settings_t st = flatSettings();
    throwOn(ldso_tr_set_settings(t, &st), "ldso_tr_set_settings");
    const ldso_calib_t c = flatCalib(*HCalib);
    throwOn(ldso_tr_make_k(t, &c), "ldso_tr_make_k");
}

// void CoarseTracker::setCoarseTrackingRef(std::vector<shared_ptr<FrameHessian>> &)                               CoarseTracker.cc:248-256
void GpuBackend::setCoarseTrackingRef(CoarseTracker &tr, std::vector<shared_ptr<FrameHessian>> &frameHessians) {
    tr.lastRef = frameHessians.back();
    // makeCoarseDepthL0's inputs (:264-283): every active point whose newest residual is IN, as (Ku, Kv, new_idepth, HdiF); the scatter, the
    // pyramid of the depth map, the dilation and the point-cloud build (:285-438) run on the device
    std::vector<float> pts;
    for (shared_ptr<FrameHessian> &fh : frameHessians)
        for (shared_ptr<Feature> &feat : fh->frame->features) {
            if (!(feat->status == Feature::FeatureStatus::VALID && feat->point->status == Point::PointStatus::ACTIVE)) continue;
            shared_ptr<PointHessian> ph = feat->point->mpPH;
            if (ph->lastResiduals[0].first != 0 && ph->lastResiduals[0].second == ResState::IN) {
                shared_ptr<PointFrameResidual> r = ph->lastResiduals[0].first;
                pts.push_back(r->centerProjectedTo[0]); pts.push_back(r->centerProjectedTo[1]); pts.push_back(r->centerProjectedTo[2]); pts.push_back(ph->HdiF);
            }
        }
    tr.refFrameID = tr.lastRef->frame->id;
    tr.lastRef_aff_g2l = tr.lastRef->aff_g2l();
    tr.firstCoarseRMSE = -1;
    if (useDevicePyramids && tr.lastRef->frame) {
        ldso_tracker_t *t = trackerOf(tr);
        PyrRef pr = pyramidOf(tr.lastRef);
        { std::lock_guard<std::mutex> lk(handlesMutex_); trackerPyr_[t].ref = pr; }
        throwOn(ldso_tr_set_ref_pyramid(t, pr->p, tr.lastRef_aff_g2l.a, tr.lastRef_aff_g2l.b, tr.lastRef->ab_exposure, pts.data(), (int) (pts.size() / 4)),
                "ldso_tr_set_ref_pyramid");
        return;
    }
    const float *pyr[PYR_LEVELS];
    for (int l = 0; l < pyrLevelsUsed; l++) pyr[l] = (const float *) tr.lastRef->dIp[l];
    throwOn(ldso_tr_set_ref(trackerOf(tr), pyr, tr.lastRef_aff_g2l.a, tr.lastRef_aff_g2l.b, tr.lastRef->ab_exposure, pts.data(), (int) (pts.size() / 4)), "ldso_tr_set_ref");
}

// bool CoarseTracker::trackNewestCoarse(newFrameHessian, lastToNew_out, aff_g2l_out, coarsestLvl, minResForAbort)  CoarseTracker.cc:61-217
bool GpuBackend::trackNewestCoarse(CoarseTracker &tr, shared_ptr<FrameHessian> newFrameHessian, SE3 &lastToNew_out, AffLight &aff_g2l_out, int coarsestLvl,
                                   Vec5 minResForAbort) {
    ldso_tracker_t *t = trackerOf(tr);
    tr.newFrame = newFrameHessian;
    if (!newFrameResident(t, newFrameHessian)) {              // the pyramid goes over once per frame, not once per hypothesis
        if (useDevicePyramids && newFrameHessian->frame) {
            PyrRef pr = pyramidOf(newFrameHessian);
            { std::lock_guard<std::mutex> lk(handlesMutex_); trackerPyr_[t].newFrame = pr; }
            throwOn(ldso_tr_set_new_frame_pyramid(t, pr->p, newFrameHessian->ab_exposure), "ldso_tr_set_new_frame_pyramid");
        }
        else {
            const float *pyr[PYR_LEVELS];
            for (int l = 0; l < pyrLevelsUsed; l++) pyr[l] = (const float *) newFrameHessian->dIp[l];
            throwOn(ldso_tr_set_new_frame(t, pyr, newFrameHessian->ab_exposure), "ldso_tr_set_new_frame");
        }
    }
    double T[12], mr[5], lr[5], fl[3];
    float ab[2] = {(float) aff_g2l_out.a, (float) aff_g2l_out.b};
    toRowMajor34(lastToNew_out, T);
    for (int i = 0; i < 5; i++) mr[i] = minResForAbort[i];
    int ok = 0, its = 0;
    throwOn(ldso_tr_track(t, T, ab, coarsestLvl, mr, lr, fl, &ok, &its), "ldso_tr_track");
    for (int i = 0; i < 5; i++) tr.lastResiduals[i] = lr[i];
    for (int i = 0; i < 3; i++) tr.lastFlowIndicators[i] = fl[i];
    // `return false` from the level loop (:188-189, residual above 1.5 x minResForAbort) leaves the outputs untouched; the affine sanity
    // checks (:202-211) fail AFTER "set!" (:198-199)
    bool aborted = false;
    for (int l = coarsestLvl; l >= 0 && !aborted; l--) aborted = lr[l] > 1.5 * mr[l];
    if (!aborted) { lastToNew_out = fromRowMajor34(T); aff_g2l_out = AffLight(ab[0], ab[1]); }
    return ok != 0;
}

// Vec4 FullSystem::trackNewCoarse(shared_ptr<FrameHessian> fh)                                                   FullSystem.cc:179-386
Vec4 GpuBackend::trackNewCoarse(FullSystem &fs, shared_ptr<FrameHessian> fh) {
    CoarseTracker &tr = *fs.coarseTracker;
    ldso_tracker_t *t = trackerOf(tr);
    shared_ptr<FrameHessian> lastF = tr.lastRef;
    tr.newFrame = fh;
    if (!newFrameResident(t, fh)) {
        if (useDevicePyramids && fh->frame) {
            PyrRef pr = pyramidOf(fh);
            { std::lock_guard<std::mutex> lk(handlesMutex_); trackerPyr_[t].newFrame = pr; }
            throwOn(ldso_tr_set_new_frame_pyramid(t, pr->p, fh->ab_exposure), "ldso_tr_set_new_frame_pyramid");
        }
        else {
            const float *pyr[PYR_LEVELS];
            for (int l = 0; l < pyrLevelsUsed; l++) pyr[l] = (const float *) fh->dIp[l];
            throwOn(ldso_tr_set_new_frame(t, pyr, fh->ab_exposure), "ldso_tr_set_new_frame");
        }
    }
    double sprelast[12], slast[12], lastFw2c[12];
    float aff_last[2] = {0, 0};
    int posesValid = 0;
    if (fs.allFrameHistory.size() != 2) {                    // :192-195: with two frames in the history the reference ends up with an empty try list (its TODO)
        shared_ptr<Frame> s1 = fs.allFrameHistory[fs.allFrameHistory.size() - 2], s2 = fs.allFrameHistory[fs.allFrameHistory.size() - 3];
        unique_lock<mutex> crlock(fs.shellPoseMutex);
        toRowMajor34(s2->getPose(), sprelast); toRowMajor34(s1->getPose(), slast); toRowMajor34(lastF->frame->getPose(), lastFw2c);
        aff_last[0] = (float) s1->aff_g2l.a; aff_last[1] = (float) s1->aff_g2l.b;
        posesValid = (s1->poseValid && s2->poseValid && lastF->frame->poseValid) ? 1 : 0;
    } else {
        const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        memcpy(sprelast, I, sizeof(I)); memcpy(slast, I, sizeof(I)); toRowMajor34(lastF->frame->getPose(), lastFw2c);
    }
    double rmse[5], res4[4], w2c[12];
    float aff_out[2];
    int tries = 0, good = 0;
    for (int i = 0; i < 5; i++) rmse[i] = fs.lastCoarseRMSE[i];
    throwOn(ldso_tr_track_new_coarse(t, sprelast, slast, lastFw2c, posesValid, aff_last, rmse, setting_reTrackThreshold, res4, w2c, aff_out, &tries, &good), "ldso_tr_track_new_coarse");
    if (!good) LOG(WARNING) << "BIG ERROR! tracking failed entirely. Take predicted pose and hope we may somehow recover." << endl;
    for (int i = 0; i < 5; i++) fs.lastCoarseRMSE[i] = rmse[i];
    fh->frame->setPose(fromRowMajor34(w2c));                                                     // :376-377
    fh->frame->aff_g2l = AffLight(aff_out[0], aff_out[1]);
    if (tr.firstCoarseRMSE < 0) tr.firstCoarseRMSE = rmse[0];
    return Vec4(res4[0], res4[1], res4[2], res4[3]);
}

}  // namespace ldso
