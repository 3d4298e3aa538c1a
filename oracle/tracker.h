// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h header).  Pinned BIT FOR BIT to the reference-compiled CoarseTracker.cc by tests/test_ref_pin.py (see linalg.h).
// tracker.h — CPU restatement of CoarseTracker (src/frontend/CoarseTracker.cc:30-632,
// include/frontend/CoarseTracker.h:17-127): makeK, makeCoarseDepthL0, calcRes, calcGSSSE,
// trackNewestCoarse.  The object graph inputs of makeCoarseDepthL0 (points with lastResiduals[0] IN)
// arrive flattened as (Ku, Kv, new_idepth, HdiF) tuples in the reference's iteration order.
#pragma once
#include "backend.h"

namespace orc {

struct CoarseTracker {
    const Globals *g;
    int pyrLevelsUsed;
    Mat33f K[LDSO_PYR_LEVELS], Ki[LDSO_PYR_LEVELS];
    float fx[LDSO_PYR_LEVELS], fy[LDSO_PYR_LEVELS], fxi[LDSO_PYR_LEVELS], fyi[LDSO_PYR_LEVELS];
    float cx[LDSO_PYR_LEVELS], cy[LDSO_PYR_LEVELS], cxi[LDSO_PYR_LEVELS], cyi[LDSO_PYR_LEVELS];
    int w[LDSO_PYR_LEVELS], h[LDSO_PYR_LEVELS];

    // reference frame
    const float *lastRef_dIp[LDSO_PYR_LEVELS];
    float lastRef_ab_exposure = 1;
    AffLight lastRef_aff_g2l;
    // new frame
    const float *newFrame_dIp[LDSO_PYR_LEVELS];
    float newFrame_ab_exposure = 1;

    Vec5 lastResiduals;
    Vec3 lastFlowIndicators;

    std::vector<float> idepth[LDSO_PYR_LEVELS], weightSums[LDSO_PYR_LEVELS], weightSums_bak[LDSO_PYR_LEVELS];
    std::vector<float> pc_u[LDSO_PYR_LEVELS], pc_v[LDSO_PYR_LEVELS], pc_idepth[LDSO_PYR_LEVELS], pc_color[LDSO_PYR_LEVELS];
    int pc_n[LDSO_PYR_LEVELS];
    std::vector<float> buf_warped_idepth, buf_warped_u, buf_warped_v, buf_warped_dx, buf_warped_dy, buf_warped_residual, buf_warped_weight, buf_warped_refColor;
    int buf_warped_n = 0;
    Accumulator9 acc;

    CoarseTracker(int ww, int hh, const Globals *g_) : g(g_), pyrLevelsUsed(g_->pyrLevelsUsed) {
        for (int lvl = 0; lvl < pyrLevelsUsed; lvl++) {
            int wl = ww >> lvl, hl = hh >> lvl;
            idepth[lvl].assign(wl * hl + 8, 0); weightSums[lvl].assign(wl * hl + 8, 0); weightSums_bak[lvl].assign(wl * hl + 8, 0);
            pc_u[lvl].assign(wl * hl + 8, 0); pc_v[lvl].assign(wl * hl + 8, 0); pc_idepth[lvl].assign(wl * hl + 8, 0); pc_color[lvl].assign(wl * hl + 8, 0);
            pc_n[lvl] = 0;
        }
        size_t n = (size_t) ww * hh + 8;
        buf_warped_idepth.assign(n, 0); buf_warped_u.assign(n, 0); buf_warped_v.assign(n, 0); buf_warped_dx.assign(n, 0); buf_warped_dy.assign(n, 0);
        buf_warped_residual.assign(n, 0); buf_warped_weight.assign(n, 0); buf_warped_refColor.assign(n, 0);
        w[0] = h[0] = 0;
    }

    void makeK(CalibHessian *HCalib);                                   // CoarseTracker.cc:219-246
    // CoarseTracker.cc:258-438; pts = n x (Ku, Kv, new_idepth, HdiF)
    void makeCoarseDepthL0(const float *pts, int n);
    Vec6 calcRes(int lvl, const SE3 &refToNew, AffLight aff_g2l, float cutoffTH);              // :440-572
    void calcGSSSE(int lvl, Mat88 &H_out, Vec8 &b_out, const SE3 &refToNew, AffLight aff_g2l);  // :574-632
    bool trackNewestCoarse(SE3 &lastToNew_out, AffLight &aff_g2l_out, int coarsestLvl, Vec5 minResForAbort, int *iterations_out);   // :61-217
};

}  // namespace orc
