// ORACLE — TEST INFRASTRUCTURE ONLY.  Link-time stand-ins for the parts of the reference's front end that FullSystem.cc refers to but the
// reference-compiled pin (oracle/_ref/libldso_ref.so, see oracle/Makefile) never executes: the pin runs the reference's odometry slice
// (FullSystem::optimize and its helpers, optimizeImmaturePoint, trackNewCoarse) with loop closing disabled (setting_enableLoopClosing =
// false), DSO's own pixel selector (setting_pointSelection = 0) and no viewer (FullSystem::viewer == nullptr).  The translation units behind
// these symbols need OpenCV / Pangolin / g2o / DBoW3 (src/frontend/{LoopClosing,FeatureDetector,DSOViewer}.cc, src/Map.cc) and are not part
// of the hot path (out of scope).  Every stand-in that would change arithmetic if it were reached aborts loudly.
#include <cstdio>
#include <cstdlib>
#include "frontend/FullSystem.h"

namespace {
[[noreturn]] void unreachable(const char *what) {
    std::fprintf(stderr, "oracle/ref_stubs.cc: %s was reached - the reference pin does not cover that part of LDSO\n", what);
    std::abort();
}
}

namespace ldso {

FeatureDetector::FeatureDetector() {}
FeatureDetector::~FeatureDetector() {}
int FeatureDetector::DetectCorners(int, shared_ptr<Frame> &) { unreachable("FeatureDetector::DetectCorners (setting_pointSelection == 1)"); }

LoopClosing::LoopClosing(FullSystem *) { unreachable("LoopClosing (setting_enableLoopClosing)"); }
void LoopClosing::InsertKeyFrame(shared_ptr<Frame> &) { unreachable("LoopClosing::InsertKeyFrame"); }

// The global map only collects key frames for the pose graph / map output: nothing on the odometry path reads it back
// (getLatestOptimizedKfId() stays 0, which is what FullSystem::optimize sees without loop closing).
void Map::AddKeyFrame(shared_ptr<Frame>) {}
bool Map::OptimizeALLKFs() { unreachable("Map::OptimizeALLKFs (pose graph)"); }
void Map::lastOptimizeAllKFs() { unreachable("Map::lastOptimizeAllKFs (pose graph)"); }
void Map::UpdateAllWorldPoints() {}          // called by ~FullSystem -> blockUntilMappingIsFinished; map output only

void PangolinDSOViewer::publishKeyframes(std::vector<shared_ptr<Frame>> &, bool, shared_ptr<CalibHessian>) { unreachable("PangolinDSOViewer"); }
void PangolinDSOViewer::publishCamPose(shared_ptr<Frame>, shared_ptr<CalibHessian>) { unreachable("PangolinDSOViewer"); }

}  // namespace ldso
