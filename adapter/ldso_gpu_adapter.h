// ldso_gpu_adapter.h — the host-side drop-in: LDSO's own types in, libldso_hip.so (include/ldso_hip.h, C-ABI) underneath.
//
// This is the code INTEGRATION.md describes, as a real translation unit against the reference's headers
// (include/frontend/FullSystem.h, include/frontend/CoarseTracker.h, include/internal/*.h): every function below is the new BODY of the
// reference member function it names.  Nothing of LDSO's object graph changes: Frame / Feature / Point / FrameHessian / PointHessian /
// PointFrameResidual / CalibHessian / EnergyFunctional stay the owners of all state; the library owns device memory only.
//
//   GpuBackend::optimize(fs, n)                     float FullSystem::optimize(int mnumOptIts)                      FullSystem.cc:725-864
//   GpuBackend::makeK / setCoarseTrackingRef        CoarseTracker::makeK / setCoarseTrackingRef                     CoarseTracker.cc:219-256
//   GpuBackend::trackNewestCoarse                   bool CoarseTracker::trackNewestCoarse(...)                      CoarseTracker.cc:61-217
//   GpuBackend::trackNewCoarse(fs, fh)              Vec4 FullSystem::trackNewCoarse(shared_ptr<FrameHessian>)       FullSystem.cc:179-386
//   GpuBackend::activatePoints(fs, ...)             the optimizeImmaturePoint loop of activatePointsMT              FullSystem.cc:892-1010,1196-1206
//   GpuBackend::traceNewCoarse(fs, fh)              void FullSystem::traceNewCoarse(shared_ptr<FrameHessian>)       FullSystem.cc:1012-1050
//   GpuBackend::flagPointsForRemoval(fs)            the policy of void FullSystem::flagPointsForRemoval()           FullSystem.cc:1208-1270
//   GpuBackend::marginalizePoints(fs)               void EnergyFunctional::marginalizePointsF() + FullSystem.cc:1241-1250   EnergyFunctional.cc:165-222
//
// The functions reach into private members of FullSystem / CoarseTracker (frames, ef, activeResiduals, allFrameHistory, lastCoarseRMSE,
// shellPoseMutex ...): a maintainer makes them member functions or adds `friend class ldso::GpuBackend;` to the two classes.  The reference tree
// is read-only in this repository, so ldso_gpu_adapter.cc opens the access specifiers for its own inclusion of the two headers instead
// (LDSO_ADAPTER_OPEN_PRIVATE) - the class layout is unchanged, the object files link against the reference's own.
#pragma once
#include <map>
#include <memory>
#include <chrono>
#include <mutex>
#include <vector>

#include "frontend/CoarseTracker.h"
#include "frontend/FullSystem.h"
#include "internal/ImmaturePoint.h"
#include "ldso_hip.h"

namespace ldso {

class GpuBackend {
public:
    // One backend per FullSystem (the BA handle belongs to the mapping thread, the tracker handles to whoever holds the respective
    // CoarseTracker - the reference's mutex discipline, FullSystem.h:272-306).  Image size / pyramid depth come from the globals of
    // internal/GlobalCalib.h (wG[0], hG[0], pyrLevelsUsed), like the reference's own constructors.
    GpuBackend(int device, int maxFrames, int maxPoints);
    ~GpuBackend();
    GpuBackend(const GpuBackend &) = delete;
    GpuBackend &operator=(const GpuBackend &) = delete;

    // ---- windowed bundle adjustment -------------------------------------------------------------------------------------------------
    float optimize(FullSystem &fs, int mnumOptIts);
    // also store RawResidualJacobian (296 B per residual: 3.5 MB at C3) back into r->J.  Off by default: nothing of makeKeyFrame reads the J that
    // optimize() leaves behind - flagPointsForRemoval re-linearises the residuals it marginalises (FullSystem.cc:1241-1250: r->linearize writes
    // r->J itself) and the next optimize() re-linearises everything.  On for host code that still accumulates from r->J.
    bool writeBackJacobians = false;
    // The window stays RESIDENT on the device between two optimize() calls: the second call uploads a delta (ldso_ba_update_window) - which frames and points
    // stayed, one bit per residual, the records of the freshly activated points - instead of flattening and uploading 2000 points and 12 000 residuals again.
    // What the delta path relies on is what LDSO does: between two optimize() calls the host does not change a SURVIVING point's u / v / colours / inverse depth
    // or the state of its residuals (it only removes points and residuals, adds one residual per point for the new key frame, and activates new points;
    // FullSystem::makeKeyFrame :429-640).  Residual counts are checked per point and a point whose residual list changed in another way is re-read in full;
    // invalidateWindow() (or residentWindow = false) forces the next optimize() to upload everything.
    bool residentWindow = true;
    void invalidateWindow() { residentValid_ = false; }
    bool lastUploadWasDelta = false;
    int uploadsDelta = 0, uploadsFresh = 0;          // how the windows of this backend's lifetime went over
    int uploadsFreshBecauseLinearized = 0;           // ... of the full uploads: windows that held linearised residuals (ldso_ba_update_window does not carry them)
    int residentPointerMismatches = 0;               // resident points whose cached residual pointers no longer matched PointHessian::residuals (their list was re-read)
    // marginalizeFrame(): how often the device path ran, and how often the call fell back to the reference's HOST member (FullSystem::marginalizeFrame ->
    // EnergyFunctional.cc:72-151) because no window of these frames was resident - a maintainer sees here when the drop-in did not take the device path
    int margFrameDevice = 0, margFrameHostFallback = 0;
    // FullSystem::activeResiduals is only read inside FullSystem::optimize itself (FullSystem.cc:735-1770: linearizeAll / applyRes / setNewFrameEnergyTH - all of
    // which run on the device here); filling it costs 12 000 shared_ptr copies per call.  On for host code that wants the list anyway.
    bool fillActiveResiduals = false;
    int lastIterations = 0;              // GN iterations the device executed in the last optimize()
    double lastUploadSeconds[6] = {0, 0, 0, 0, 0, 0};  // of that: settings + image slots | flatten (host walk) | ldso_ba_set_window | set_point_stats | set_frames | set_prior
    double lastOptimizeSeconds[4] = {0, 0, 0, 0};      // wall clock of the last optimize(): flatten + upload | device | fetch | write-back into the objects

    // void FullSystem::marginalizeFrame(shared_ptr<Frame> &frame) (FullSystem.cc:602-645): the Schur complement of EnergyFunctional::marginalizeFrame
    // (EnergyFunctional.cc:72-151) through ldso_ba_marginalize_frame on the window optimize() left resident (ef's H_M / b_M go up first: 29 KB), the reference's
    // bookkeeping on the host - with the residuals that target the frame found through the resident window's host view (one pointer per point) instead of a
    // weak_ptr::lock() per residual of the window.
    void marginalizeFrame(FullSystem &fs, shared_ptr<Frame> &frame);
    // ---- point marginalisation on the device (what follows optimize() in makeKeyFrame, FullSystem.cc:526-536) ----------------------------
    // flagPointsForRemoval: the POLICY of FullSystem::flagPointsForRemoval (FullSystem.cc:1208-1270: which points go OUT / OUTLIER / MARGINALIZED) without its inner
    // loop (:1241-1250: resetOOB + linearize + applyRes + fixLinearizationF of every residual of a point that gets marginalised) - the device pass of
    // marginalizePoints does exactly that to exactly those points.  marginalizePoints: EnergyFunctional::marginalizePointsF (EnergyFunctional.cc:165-222) through
    // ldso_ba_marginalize_points on the window GpuBackend::optimize left resident (same frames, same point order), H_M / b_M written back into ef, the
    // reference's bookkeeping (priorF, connectivity counts, removePoint, makeIDX) on the host.  Call order as the reference: flagPointsForRemoval,
    // ef->dropPointsF(), marginalizePoints.
    void flagPointsForRemoval(FullSystem &fs);
    void marginalizePoints(FullSystem &fs);

    // ---- coarse tracker -------------------------------------------------------------------------------------------------------------
    void makeK(CoarseTracker &tr, shared_ptr<CalibHessian> HCalib);
    void setCoarseTrackingRef(CoarseTracker &tr, std::vector<shared_ptr<FrameHessian>> &frameHessians);
    bool trackNewestCoarse(CoarseTracker &tr, shared_ptr<FrameHessian> newFrameHessian, SE3 &lastToNew_out, AffLight &aff_g2l_out, int coarsestLvl,
                           Vec5 minResForAbort);
    Vec4 trackNewCoarse(FullSystem &fs, shared_ptr<FrameHessian> fh);

    // ---- point activation (the optimizeImmaturePoint calls of activatePointsMT_Reductor) ---------------------------------------------
    // optimized[k] = the new PointHessian of toOptimize[k] or nullptr, exactly what FullSystem::optimizeImmaturePoint returns
    void activatePoints(FullSystem &fs, std::vector<shared_ptr<internal::ImmaturePoint>> &toOptimize, std::vector<shared_ptr<PointHessian>> &optimized);

    // ---- immature-point tracing: void FullSystem::traceNewCoarse(shared_ptr<FrameHessian> fh)                   FullSystem.cc:1012-1050
    void traceNewCoarse(FullSystem &fs, shared_ptr<FrameHessian> fh);
    int lastTraceCounts[6] = {0, 0, 0, 0, 0, 0};        // good, oob, outlier, skipped, bad condition, uninitialised (the function's trace_* counters)

    // ---- FrameHessian::dIp on the device: one ldso_pyramid_t per frame (FrameHessian.cc:44-113 on the device from the frame's irradiance, 4 bytes per pixel
    // over PCIe once per frame), shared zero-copy by the BA image slot, the tracer and both roles of the coarse trackers - as the reference shares fh->dIp by
    // pointer.  Off: every consumer uploads the host arrays it needs (12 bytes per pixel and consumer).  Pyramids of frames that left the window and are neither
    // a tracker's reference nor among the most recent frames are released at the next optimize().
    bool useDevicePyramids = true;
    int pyramidsBuilt = 0;               // statistics: pyramids built so far (one per frame seen)

    const char *lastError() const;

private:
    ldso_ba_t *ba_ = nullptr;
    ldso_tracer_t *tracer_ = nullptr;
    int tracerCap_ = 0;
    std::map<CoarseTracker *, ldso_tracker_t *> trackers_;          // the reference double-buffers two CoarseTrackers (FullSystem.h:296-297)
    std::map<unsigned long, int> slotOf_;                            // key frame (Frame::id: addresses get reused) -> image slot of the BA handle
    std::vector<long> slotOwner_;                                    // slot -> Frame::id, -1 = free
    std::vector<shared_ptr<PointHessian>> lastPoints_;               // the points of the window optimize() left on the device, in its order
    // One device pyramid per frame (Frame::id), REFERENCE-COUNTED: the map below holds one reference, every consumer that names the pyramid on the device
    // holds another (the BA image slot, the tracer, a tracker's reference frame / new frame).  releasePyramids() only drops the MAP's reference of frames that
    // left the window and that nobody else holds; the device memory goes when the last holder lets go - no guessing which tracker may still read it, no
    // reading of another thread's state (the keep-set was built from an unsynchronised read of the trackers' maps and of lastRef).
    struct PyrHolder {
        ldso_pyramid_t *p = nullptr;
        std::vector<float> irradiance;                               // the host copy of channel 0 the build read from
        ~PyrHolder();
    };
    typedef std::shared_ptr<PyrHolder> PyrRef;
    std::map<unsigned long, PyrRef> pyr_;                            // Frame::id -> pyramid
    std::mutex pyrMutex_;                                            // tracking thread (new frame / reference) and mapping thread (window, tracer) share the map
    std::vector<PyrRef> slotPyr_;                                    // BA image slot -> the pyramid it aliases (mapping thread only)
    PyrRef tracerPyr_;                                               // the frame the tracer currently reads (mapping thread only)
    struct TrackerPyr { PyrRef ref, newFrame; };
    std::map<ldso_tracker_t *, TrackerPyr> trackerPyr_;              // guarded by handlesMutex_
    PyrRef pyramidOf(const shared_ptr<FrameHessian> &fh);
    void releasePyramids(FullSystem &fs);
    std::vector<int32_t> resBegin_;                                  // last upload: the residuals of point k are flat_[resBegin_[k] .. resBegin_[k + 1])
    // the resident window as the host sees it: one row per point in device order, its residual objects by window column (= target frame index)
    static constexpr int kMaxCols = 16;
    static_assert(kMaxCols >= LDSO_MAX_FRAMES, "a row of the resident window holds one residual pointer per window column");
    struct Row { PointHessian *ph; Feature *feat; uint32_t mask; int host; PointFrameResidual *res[kMaxCols]; };
    std::vector<Row> rows_;
    // optimize()'s write-back: where ldso_ba_get_results lands (kept across calls)
    std::vector<ldso_res_out_t> wbRes_; std::vector<int32_t> wbState_, wbActive_, wbRemove_;
    std::vector<ldso_point_out_t> wbPoints_; std::vector<ldso_frame_t> wbFrames_; std::vector<double> wbStep_;
    std::vector<shared_ptr<Frame>> rowFrames_;                       // the frame of every column of the resident window (held: the rows point into their features)
    std::vector<PointFrameResidual *> flat_;                         // the residual objects in the device's flat order (point-major, target-ascending)
    bool residentValid_ = false;
    std::chrono::steady_clock::time_point lapT_;
    double lapSince() { const auto n = std::chrono::steady_clock::now(); const double d = std::chrono::duration<double>(n - lapT_).count(); lapT_ = n; return d; }
    bool uploadDelta(FullSystem &fs, const std::vector<int32_t> &slots, std::vector<shared_ptr<PointHessian>> &allPoints);
    void uploadFresh(FullSystem &fs, const std::vector<int32_t> &slots, std::vector<shared_ptr<PointHessian>> &allPoints, bool trustIndices);
    int device_, maxFrames_, maxPoints_;
    // whose pyramid a tracker handle currently holds as "new frame": keyed by Frame::id, not by address (LDSO releases the FrameHessian of a
    // non-key frame after tracking, the allocator may hand the same address to the next frame)
    std::map<ldso_tracker_t *, unsigned long> trackerNewFrameId_;
    // trackerOf / newFrameResident run on the tracking thread (coarseTracker) AND on the mapping thread (coarseTracker_forNewKF under
    // coarseTrackerSwapMutex, FullSystem.cc makeKeyFrame): the two maps above are shared between them
    std::mutex handlesMutex_;
    bool newFrameResident(ldso_tracker_t *t, const shared_ptr<FrameHessian> &fh);
    ldso_tracker_t *trackerOf(CoarseTracker &tr);
    int uploadWindow(FullSystem &fs, std::vector<shared_ptr<PointHessian>> &allPoints, bool trustIndices = false);
    void syncImageSlots(FullSystem &fs, std::vector<int32_t> &slots);
};

}  // namespace ldso
