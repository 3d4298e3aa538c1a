// ORACLE — TEST INFRASTRUCTURE ONLY.  Driver of the REFERENCE-COMPILED library oracle/_ref/libldso_ref.so.
//
// The hot-path translation units of /root/reference are compiled where they lie, unmodified (oracle/Makefile, target _ref):
//   src/internal/Residuals.cc, src/internal/OptimizationBackend/{AccumulatedTopHessian,AccumulatedSCHessian,EnergyFunctional}.cc,
//   include/internal/OptimizationBackend/MatrixAccumulators.h, src/internal/{FrameFramePrecalc,FrameHessian,GlobalCalib,PointHessian,
//   ImmaturePoint}.cc, src/frontend/{CoarseTracker,CoarseInitializer,PixelSelector2}.cc, src/{Setting,Camera}.cc
// against the header shim oracle/ref_shim (Eigen / Sophus / glog / DBoW3 / OpenCV stand-ins, see ref_shim/Eigen/Core for what
// that replaces).  This file builds the reference's own object graph (Frame / FrameHessian / Feature / Point / PointHessian /
// PointFrameResidual / CalibHessian / EnergyFunctional) from the flattened window of include/ldso_window.h and exposes the
// reference's functions one by one through the same C entry points as oracle/capi.cc (prefix ref_ instead of orc_), so that
// tests/test_ref_pin.py can run the oracle's restatement and the reference side by side on identical inputs.
//
// Since round 3 src/frontend/FullSystem.cc and src/{Frame,Feature,Point}.cc are compiled into the library as well (unmodified; the
// front-end pieces they refer to but never execute here - viewer, loop closing, corner detector, global map - are link-time stand-ins in
// ref_stubs.cc).  The ref_fs_* entry points below attach a real FullSystem object to the window and call the reference's own
// optimize / linearizeAll / setNewFrameEnergyTH / backupState / doStepFromBackup / loadSateBackup / solveSystem / calcLEnergy / calcMEnergy /
// optimizeImmaturePoint / trackNewCoarse (private members: this file, and only this file, is compiled with `#define private public`).
// The stage calls of the older entry points (ref_linearize_all, ref_solve_system, ref_flag_points, ref_marginalize_frame) keep their
// restated FullSystem loops, each citing its source: setPrecalcValues (:1423-1431), getNullspaces (:1711-1760), the residual loop of
// linearizeAll_Reductor (:1442-1470, without setNewFrameEnergyTH), the re-linearise / fix loop of flagPointsForRemoval (:1241-1250), the
// residual drop of marginalizeFrame (:602-640) - tests/test_ref_pin.py checks them against the real members.
#include <vector>
#include <memory>
#include <map>
#include <set>
#include <mutex>
#include <thread>
#include <functional>
#include <condition_variable>
#include <cstring>
#include <cstdio>
#include <iostream>
#include <fstream>
#include <sstream>
#include <string>
#include <chrono>
#include <deque>
#include <list>
#include <queue>
#include <type_traits>
#include <unistd.h>
#include <Eigen/Core>
#include <glog/logging.h>
#define private public
#define protected public
#include "Frame.h"
#include "Feature.h"
#include "Point.h"
#include "Camera.h"
#include "Settings.h"
#include "internal/FrameHessian.h"
#include "internal/PointHessian.h"
#include "internal/CalibHessian.h"
#include "internal/Residuals.h"
#include "internal/GlobalCalib.h"
#include "internal/GlobalFuncs.h"
#include "internal/ImmaturePoint.h"
#include "internal/OptimizationBackend/EnergyFunctional.h"
#include "frontend/CoarseTracker.h"
#include "frontend/CoarseInitializer.h"
#include "frontend/FullSystem.h"
#undef private
#undef protected
#include "../include/ldso_window.h"

using namespace ldso;
using namespace ldso::internal;

namespace {

struct RefWindow {
    shared_ptr<Camera> cam;
    shared_ptr<CalibHessian> Hcalib;
    shared_ptr<EnergyFunctional> ef;
    IndexThreadReduce<Vec10> red;
    std::vector<shared_ptr<Frame>> frames;
    std::vector<shared_ptr<PointHessian>> pointByFlat;
    std::vector<shared_ptr<PointFrameResidual>> resByFlat;
    std::vector<shared_ptr<PointFrameResidual>> activeResiduals;
    std::vector<std::vector<float>> imageStore;
    int levels = 0;
    FullSystem *fs = nullptr;         // ref_fs_attach: the reference's own FullSystem driving this window
    std::string fsLog;                // what the reference streamed into LOG(...) during the last ref_fs_* call (glog stand-in)
};

static void apply_settings(const ldso_settings_t *s) {
    setting_huberTH = s->huberTH; setting_outlierTHSumComponent = s->outlierTHSumComponent;
    setting_affineOptModeA = s->affineOptModeA; setting_affineOptModeB = s->affineOptModeB;
    setting_frameEnergyTHN = s->frameEnergyTHN; setting_frameEnergyTHFacMedian = s->frameEnergyTHFacMedian;
    setting_frameEnergyTHConstWeight = s->frameEnergyTHConstWeight; setting_overallEnergyTHWeight = s->overallEnergyTHWeight;
    setting_initialCalibHessian = s->initialCalibHessian; setting_margWeightFac = s->margWeightFac;
    setting_idepthFixPriorMargFac = s->idepthFixPriorMargFac; setting_thOptIterations = s->thOptIterations;
    setting_coarseCutoffTH = s->coarseCutoffTH; setting_minOptIterations = s->minOptIterations;
    setting_solverMode = s->solverMode; setting_forceAceptStep = s->forceAcceptStep != 0; setting_solverModeDelta = s->solverModeDelta;
    multiThreading = false;
}

static SE3 se3_from34(const double *m) {
    Mat33 R; Vec3 t;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = m[i * 4 + j]; t[i] = m[i * 4 + 3]; }
    return SE3(R, t);
}

// FullSystem::setPrecalcValues (FullSystem.cc:1423-1431)
static void set_precalc(RefWindow *W) {
    for (auto &fr : W->frames) {
        fr->frameHessian->targetPrecalc.resize(W->frames.size());
        for (size_t i = 0; i < W->frames.size(); i++)
            fr->frameHessian->targetPrecalc[i].Set(fr->frameHessian, W->frames[i]->frameHessian, W->Hcalib);
    }
    W->ef->setDeltaF(W->Hcalib);
}

// FullSystem::getNullspaces (FullSystem.cc:1711-1760)
static std::vector<VecX> get_nullspaces(RefWindow *W, std::vector<VecX> &np, std::vector<VecX> &ns, std::vector<VecX> &na, std::vector<VecX> &nb) {
    np.clear(); ns.clear(); na.clear(); nb.clear();
    int n = CPARS + W->frames.size() * 8;
    std::vector<VecX> pre;
    for (int i = 0; i < 6; i++) {
        VecX v(n); v.setZero();
        for (auto fr : W->frames) {
            auto fh = fr->frameHessian;
            v.segment<6>(CPARS + fh->idx * 8) = fh->nullspaces_pose.col(i);
            v.segment<3>(CPARS + fh->idx * 8) *= SCALE_XI_TRANS_INVERSE;
            v.segment<3>(CPARS + fh->idx * 8 + 3) *= SCALE_XI_ROT_INVERSE;
        }
        pre.push_back(v); np.push_back(v);
    }
    for (int i = 0; i < 2; i++) {
        VecX v(n); v.setZero();
        for (auto fr : W->frames) {
            auto fh = fr->frameHessian;
            v.segment<2>(CPARS + fh->idx * 8 + 6) = fh->nullspaces_affine.col(i).head<2>();
            v[CPARS + fh->idx * 8 + 6] *= SCALE_A_INVERSE;
            v[CPARS + fh->idx * 8 + 7] *= SCALE_B_INVERSE;
        }
        pre.push_back(v);
        if (i == 0) na.push_back(v);
        if (i == 1) nb.push_back(v);
    }
    VecX v(n); v.setZero();
    for (auto fr : W->frames) {
        auto fh = fr->frameHessian;
        v.segment<6>(CPARS + fh->idx * 8) = fh->nullspaces_scale;
        v.segment<3>(CPARS + fh->idx * 8) *= SCALE_XI_TRANS_INVERSE;
        v.segment<3>(CPARS + fh->idx * 8 + 3) *= SCALE_XI_ROT_INVERSE;
    }
    pre.push_back(v); ns.push_back(v);
    return pre;
}

static void put_jac(const RawResidualJacobian &J, ldso_rawjac_t &j) {
    for (int k = 0; k < 8; k++) { j.resF[k] = J.resF[k]; j.JIdx[0][k] = J.JIdx[0][k]; j.JIdx[1][k] = J.JIdx[1][k]; j.JabF[0][k] = J.JabF[0][k]; j.JabF[1][k] = J.JabF[1][k]; }
    for (int k = 0; k < 6; k++) { j.Jpdxi[0][k] = J.Jpdxi[0][k]; j.Jpdxi[1][k] = J.Jpdxi[1][k]; }
    for (int k = 0; k < 4; k++) { j.Jpdc[0][k] = J.Jpdc[0][k]; j.Jpdc[1][k] = J.Jpdc[1][k]; }
    j.Jpdd[0] = J.Jpdd[0]; j.Jpdd[1] = J.Jpdd[1];
    j.JIdx2[0] = J.JIdx2(0, 0); j.JIdx2[1] = J.JIdx2(0, 1); j.JIdx2[2] = J.JIdx2(1, 0); j.JIdx2[3] = J.JIdx2(1, 1);
    j.JabJIdx[0] = J.JabJIdx(0, 0); j.JabJIdx[1] = J.JabJIdx(0, 1); j.JabJIdx[2] = J.JabJIdx(1, 0); j.JabJIdx[3] = J.JabJIdx(1, 1);
    j.Jab2[0] = J.Jab2(0, 0); j.Jab2[1] = J.Jab2(0, 1); j.Jab2[2] = J.Jab2(1, 0); j.Jab2[3] = J.Jab2(1, 1);
}

}  // namespace

extern "C" {

void *ref_create(int w, int h, int levels, const ldso_settings_t *settings, const ldso_calib_t *calib,
                 int F, const ldso_frame_t *frames, const float *const *images,
                 int P, const ldso_point_t *points, int R, const ldso_residual_t *residuals,
                 const ldso_rawjac_t *linJ, const float *lin_res_toZeroF,
                 const double *HM, const double *bM, int /*multithreading*/) {
    RefWindow *W = new RefWindow();
    apply_settings(settings);
    W->levels = levels;
    // globals of GlobalCalib.cc: K from the scaled calibration value (CalibHessian.h:86-100: value_scaled = SCALE_F / SCALE_C * value)
    Eigen::Matrix3f K = Eigen::Matrix3f::Zero();
    K(0, 0) = (float) (SCALE_F * calib->value[0]); K(1, 1) = (float) (SCALE_F * calib->value[1]);
    K(0, 2) = (float) (SCALE_C * calib->value[2]); K(1, 2) = (float) (SCALE_C * calib->value[3]); K(2, 2) = 1;
    setGlobalCalib(w, h, K);
    pyrLevelsUsed = levels;

    W->cam.reset(new Camera(K(0, 0), K(1, 1), K(0, 2), K(1, 2)));
    W->cam->CreateCH(W->cam);
    W->Hcalib = W->cam->mpCH;
    VecC v0, vz;
    for (int i = 0; i < 4; i++) { v0[i] = calib->value[i]; vz[i] = calib->value_zero[i]; }
    W->Hcalib->value_zero = vz;
    W->Hcalib->setValue(v0);

    W->ef.reset(new EnergyFunctional());
    W->ef->red = &W->red;
    W->imageStore.resize((size_t) F * levels);
    for (int f = 0; f < F; f++) {
        const ldso_frame_t &in = frames[f];
        shared_ptr<Frame> fr(new Frame());
        fr->id = in.frameID;                          // FrameHessian::getPrior keys on frame->id == 0
        fr->CreateFH(fr);
        shared_ptr<FrameHessian> fh = fr->frameHessian;
        fh->frameID = in.frameID; fh->ab_exposure = in.ab_exposure; fh->frameEnergyTH = in.frameEnergyTH;
        for (int l = 0; l < PYR_LEVELS; l++) { fh->dIp[l] = nullptr; fh->absSquaredGrad[l] = nullptr; }
        for (int l = 0; l < levels; l++) {
            size_t n = (size_t) (w >> l) * (h >> l) * 3;
            const float *src = images[f * levels + l];
            if (src) { W->imageStore[f * levels + l].assign(src, src + n); fh->dIp[l] = (Vec3f *) W->imageStore[f * levels + l].data(); }
        }
        fh->dI = fh->dIp[0];
        fh->worldToCam_evalPT = se3_from34(in.worldToCam_evalPT);
        Vec10 st, sz;
        for (int i = 0; i < 10; i++) { st[i] = in.state[i]; sz[i] = in.state_zero[i]; }
        fh->state_zero = sz;
        fh->setState(st);
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) fh->nullspaces_pose(r, c) = in.nullspaces_pose[r * 6 + c];
        for (int r = 0; r < 6; r++) fh->nullspaces_scale[r] = in.nullspaces_scale[r];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 2; c++) fh->nullspaces_affine(r, c) = in.nullspaces_affine[r * 2 + c];
        W->frames.push_back(fr);
    }
    for (auto &fr : W->frames) W->ef->insertFrame(fr->frameHessian, W->Hcalib);
    // EnergyFunctional::insertFrame sets idx = frames.size() (1-based) and relies on makeIDX for the window index
    W->pointByFlat.resize(P);
    W->resByFlat.resize(R);
    for (int k = 0; k < P; k++) {
        const ldso_point_t &in = points[k];
        shared_ptr<Frame> host = W->frames[in.host];
        shared_ptr<Feature> feat(new Feature(in.u, in.v, host));
        feat->status = Feature::FeatureStatus::VALID;
        shared_ptr<Point> pt(new Point());
        pt->status = Point::PointStatus::ACTIVE;
        pt->mHostFeature = feat;
        feat->point = pt;
        shared_ptr<PointHessian> p(new PointHessian());
        pt->mpPH = p; p->point = pt;
        host->features.push_back(feat);
        p->u = in.u; p->v = in.v;
        p->setIdepth(in.idepth);
        p->setIdepthZero(in.idepth_zero);
        memcpy(p->color, in.color, sizeof(p->color));
        memcpy(p->weights, in.weights, sizeof(p->weights));
        p->hasDepthPrior = in.priorF != 0;
        p->takeData();
        p->priorF = in.priorF;
        W->pointByFlat[k] = p;
        W->ef->nPoints++;
        for (int j = 0; j < in.res_count; j++) {
            int ri = in.res_begin + j;
            const ldso_residual_t &rin = residuals[ri];
            shared_ptr<PointFrameResidual> r(new PointFrameResidual(p, W->frames[rin.host]->frameHessian, W->frames[rin.target]->frameHessian));
            r->state_state = (ResState) rin.state_state;
            r->state_energy = rin.state_energy;
            r->isLinearized = rin.is_linearized != 0;
            r->isActiveAndIsGoodNEW = rin.is_active != 0;
            r->isNew = rin.is_new != 0;
            if (r->isLinearized && linJ) {
                const ldso_rawjac_t &j74 = linJ[ri];
                RawResidualJacobian &J = *r->J;
                for (int i = 0; i < 8; i++) { J.resF[i] = j74.resF[i]; J.JIdx[0][i] = j74.JIdx[0][i]; J.JIdx[1][i] = j74.JIdx[1][i]; J.JabF[0][i] = j74.JabF[0][i]; J.JabF[1][i] = j74.JabF[1][i]; }
                for (int i = 0; i < 6; i++) { J.Jpdxi[0][i] = j74.Jpdxi[0][i]; J.Jpdxi[1][i] = j74.Jpdxi[1][i]; }
                for (int i = 0; i < 4; i++) { J.Jpdc[0][i] = j74.Jpdc[0][i]; J.Jpdc[1][i] = j74.Jpdc[1][i]; }
                J.Jpdd[0] = j74.Jpdd[0]; J.Jpdd[1] = j74.Jpdd[1];
                J.JIdx2(0, 0) = j74.JIdx2[0]; J.JIdx2(0, 1) = j74.JIdx2[1]; J.JIdx2(1, 0) = j74.JIdx2[2]; J.JIdx2(1, 1) = j74.JIdx2[3];
                J.JabJIdx(0, 0) = j74.JabJIdx[0]; J.JabJIdx(0, 1) = j74.JabJIdx[1]; J.JabJIdx(1, 0) = j74.JabJIdx[2]; J.JabJIdx(1, 1) = j74.JabJIdx[3];
                J.Jab2(0, 0) = j74.Jab2[0]; J.Jab2(0, 1) = j74.Jab2[1]; J.Jab2(1, 0) = j74.Jab2[2]; J.Jab2(1, 1) = j74.Jab2[3];
                for (int i = 0; i < 8; i++) r->res_toZeroF[i] = lin_res_toZeroF[ri * 8 + i];
            }
            p->residuals.push_back(r);
            if (r->isLinearized && linJ) W->ef->insertResidual(r); else W->ef->nResiduals++;      // insertResidual = takeData + counters (EF.cc:26-30)
            W->resByFlat[ri] = r;
        }
        // a freshly activated point starts with lastResiduals = {nullptr, OOB} (FullSystem::optimizeImmaturePoint, FullSystem.cc:981-984) ...
        p->lastResiduals[0] = {nullptr, ResState::OOB}; p->lastResiduals[1] = {nullptr, ResState::OOB};
        for (auto &r : p->residuals) {      // ... and FullSystem.cc:446-469 / :1000-1006 point them at the residuals into the two newest frames
            if (r->target.lock() == W->frames.back()->frameHessian) p->lastResiduals[0] = {r, ResState::IN};
            else if (F >= 2 && r->target.lock() == W->frames[F - 2]->frameHessian) p->lastResiduals[1] = {r, ResState::IN};
        }
    }
    int n = 8 * F + 4;
    if (HM) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) W->ef->HM(i, j) = HM[i * n + j];
    if (bM) for (int i = 0; i < n; i++) W->ef->bM[i] = bM[i];
    W->ef->makeIDX();
    set_precalc(W);
    return W;
}

void ref_destroy(void *h) {
    RefWindow *W = (RefWindow *) h;
    if (W->fs) {
        // the FullSystem shares the window's objects; its tracker / initializer hold no frames here
        for (auto &fr : W->fs->frames) if (fr->frameHessian) for (int l = 0; l < PYR_LEVELS; l++) { fr->frameHessian->dIp[l] = nullptr; fr->frameHessian->absSquaredGrad[l] = nullptr; }
        W->ef->red = &W->red;
        delete W->fs; W->fs = nullptr;
    }
    // ~FrameHessian delete[]s its pyramid levels (FrameHessian.h:24-29); here they point into imageStore
    for (auto &fh : W->ef->frames) for (int l = 0; l < PYR_LEVELS; l++) { fh->dIp[l] = nullptr; fh->absSquaredGrad[l] = nullptr; }
    for (auto &fr : W->frames) if (fr->frameHessian) for (int l = 0; l < PYR_LEVELS; l++) { fr->frameHessian->dIp[l] = nullptr; fr->frameHessian->absSquaredGrad[l] = nullptr; }
    delete W;
}

// the activeResiduals list of FullSystem::optimize (FullSystem.cc:735-755); reset_oob = the resetOOB of that loop
void ref_collect_active(void *h, int reset_oob) {
    RefWindow *W = (RefWindow *) h;
    W->activeResiduals.clear();
    for (auto &fr : W->frames)
        for (auto &feat : fr->features)
            if (feat->status == Feature::FeatureStatus::VALID && feat->point->status == Point::PointStatus::ACTIVE)
                for (auto &r : feat->point->mpPH->residuals)
                    if (!r->isLinearized) { W->activeResiduals.push_back(r); if (reset_oob) r->resetOOB(); }
}

// the residual loop of FullSystem::linearizeAll_Reductor (FullSystem.cc:1442-1470): PointFrameResidual::linearize on every
// active residual; returns the energy sum.  (setNewFrameEnergyTH is FullSystem code and is not part of this call.)
double ref_linearize_all(void *h) {
    RefWindow *W = (RefWindow *) h;
    double E = 0;
    for (auto &r : W->activeResiduals) E += r->linearize(W->Hcalib);
    return E;
}

void ref_apply_res(void *h) { RefWindow *W = (RefWindow *) h; for (auto &r : W->activeResiduals) r->applyRes(true); }

void ref_set_frame_energy_th(void *h, int f, float th) { ((RefWindow *) h)->frames[f]->frameHessian->frameEnergyTH = th; }

void ref_set_precalc(void *h) { set_precalc((RefWindow *) h); }

// FullSystem::solveSystem (FullSystem.cc:1694-1704): nullspaces, then EnergyFunctional::solveSystemF
void ref_solve_system(void *h, int iteration, double lambda) {
    RefWindow *W = (RefWindow *) h;
    W->ef->lastNullspaces_forLogging = get_nullspaces(W, W->ef->lastNullspaces_pose, W->ef->lastNullspaces_scale, W->ef->lastNullspaces_affA, W->ef->lastNullspaces_affB);
    W->ef->solveSystemF(iteration, lambda, W->Hcalib);
}

void ref_calc_lm_energies(void *h, double *EM, double *EL) {
    EnergyFunctional *ef = ((RefWindow *) h)->ef.get();
    *EM = ef->calcMEnergyF();
    *EL = ef->calcLEnergyF_MT();
}

int ref_num_frames(void *h) { return ((RefWindow *) h)->ef->nFrames; }
void ref_counts(void *h, int *a, int *l, int *m) { auto ef = ((RefWindow *) h)->ef; *a = ef->resInA; *l = ef->resInL; *m = ef->resInM; }

void ref_get_residuals(void *h, ldso_res_out_t *out, ldso_rawjac_t *J, int32_t *state_state, int32_t *is_active, float *res_toZeroF, int32_t *is_linearized, int32_t *alive) {
    RefWindow *W = (RefWindow *) h;
    std::set<PointFrameResidual *> live;
    for (auto &p : W->pointByFlat) for (auto &r : p->residuals) live.insert(r.get());
    for (size_t i = 0; i < W->resByFlat.size(); i++) {
        PointFrameResidual *r = W->resByFlat[i].get();
        if (out) {
            out[i].state_NewEnergy = (float) r->state_NewEnergy;
            out[i].state_NewEnergyWithOutlier = (float) r->state_NewEnergyWithOutlier;
            out[i].state_NewState = r->state_NewState;
            for (int k = 0; k < 3; k++) out[i].centerProjectedTo[k] = r->centerProjectedTo[k];
            for (int k = 0; k < 8; k++) out[i].JpJdF[k] = r->JpJdF[k];
        }
        if (J) put_jac(*r->J, J[i]);
        if (state_state) state_state[i] = r->state_state;
        if (is_active) is_active[i] = r->isActiveAndIsGoodNEW ? 1 : 0;
        if (res_toZeroF) for (int k = 0; k < 8; k++) res_toZeroF[i * 8 + k] = r->res_toZeroF[k];
        if (is_linearized) is_linearized[i] = r->isLinearized ? 1 : 0;
        if (alive) alive[i] = live.count(r) ? 1 : 0;
    }
}

void ref_get_points(void *h, ldso_point_out_t *out, int32_t *status) {
    RefWindow *W = (RefWindow *) h;
    for (size_t i = 0; i < W->pointByFlat.size(); i++) {
        PointHessian *p = W->pointByFlat[i].get();
        if (out) {
            ldso_point_out_t &o = out[i];
            o.step = p->step; o.HdiF = p->HdiF; o.bdSumF = p->bdSumF; o.idepth_hessian = p->idepth_hessian;
            o.Hdd_accAF = p->Hdd_accAF; o.bd_accAF = p->bd_accAF; o.Hdd_accLF = p->Hdd_accLF; o.bd_accLF = p->bd_accLF;
            for (int k = 0; k < 4; k++) { o.Hcd_accAF[k] = p->Hcd_accAF[k]; o.Hcd_accLF[k] = p->Hcd_accLF[k]; }
            o.idepth = p->idepth; o.maxRelBaseline = p->maxRelBaseline; o.numGoodResiduals = p->numGoodResiduals;
        }
        if (status) status[i] = !p->point ? -1 /* released with its host frame (Frame::ReleaseAll) */ : p->alreadyRemoved ? 100 + (int) p->point->status : (int) p->point->status;
    }
}

void ref_get_frames(void *h, ldso_frame_t *out, double *step, double *calib_value, double *calib_step, double *pre_worldToCam) {
    RefWindow *W = (RefWindow *) h;
    for (size_t f = 0; f < W->ef->frames.size(); f++) {
        FrameHessian *fh = W->ef->frames[f].get();
        if (out) {
            ldso_frame_t &o = out[f];
            Eigen::Matrix<double, 3, 4> M = fh->worldToCam_evalPT.matrix3x4();
            for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) o.worldToCam_evalPT[i * 4 + j] = M(i, j);
            for (int i = 0; i < 10; i++) { o.state[i] = fh->state[i]; o.state_zero[i] = fh->state_zero[i]; }
            for (int i = 0; i < 8; i++) o.prior[i] = fh->prior[i];
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) o.nullspaces_pose[r * 6 + c] = fh->nullspaces_pose(r, c);
            for (int r = 0; r < 6; r++) o.nullspaces_scale[r] = fh->nullspaces_scale[r];
            for (int r = 0; r < 4; r++) for (int c = 0; c < 2; c++) o.nullspaces_affine[r * 2 + c] = fh->nullspaces_affine(r, c);
            o.ab_exposure = fh->ab_exposure; o.frameEnergyTH = fh->frameEnergyTH; o.frameID = fh->frameID; o.pad_ = 0;
        }
        if (step) for (int i = 0; i < 10; i++) step[f * 10 + i] = fh->step[i];
        if (pre_worldToCam) {
            Eigen::Matrix<double, 3, 4> M = fh->PRE_worldToCam.matrix3x4();
            for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) pre_worldToCam[f * 12 + i * 4 + j] = M(i, j);
        }
    }
    if (calib_value) for (int i = 0; i < 4; i++) calib_value[i] = W->Hcalib->value[i];
    if (calib_step) for (int i = 0; i < 4; i++) calib_step[i] = W->Hcalib->step[i];
}

// F*F pair precalc: 9 KRKi, 3 Kt, 9 R0, 3 t0, 2 aff, 1 b0 (27 floats, matrices row-major) at [h*F + t]
void ref_get_precalc(void *h, float *out) {
    RefWindow *W = (RefWindow *) h;
    int F = W->ef->frames.size();
    for (int a = 0; a < F; a++)
        for (int t = 0; t < F; t++) {
            const FrameFramePrecalc &p = W->ef->frames[a]->targetPrecalc[t];
            float *o = out + (a * F + t) * 27;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { o[i * 3 + j] = p.PRE_KRKiTll(i, j); o[12 + i * 3 + j] = p.PRE_RTll_0(i, j); }
            for (int i = 0; i < 3; i++) { o[9 + i] = p.PRE_KtTll[i]; o[21 + i] = p.PRE_tTll_0[i]; }
            o[24] = p.PRE_aff_mode[0]; o[25] = p.PRE_aff_mode[1]; o[26] = p.PRE_b0_mode;
        }
}

// adjoints: adHost / adTarget F*F*64 doubles (row-major 8x8) at [h + t*F]; adHTdeltaF F*F*8 floats
void ref_get_adjoints(void *h, double *adHost, double *adTarget, float *adHTdeltaF) {
    EnergyFunctional *ef = ((RefWindow *) h)->ef.get();
    int n = ef->nFrames * ef->nFrames;
    for (int i = 0; i < n; i++) {
        for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) {
            if (adHost) adHost[i * 64 + r * 8 + c] = ef->adHost[i](r, c);
            if (adTarget) adTarget[i * 64 + r * 8 + c] = ef->adTarget[i](r, c);
        }
        if (adHTdeltaF) for (int c = 0; c < 8; c++) adHTdeltaF[i * 8 + c] = ef->adHTdeltaF[i][c];
    }
}

// raw fp32 accumulators after the last solve (single-thread slot 0), row-major:
// topA/topL: F*F*169 at [h + t*F]; accD F^3*64; accE F^2*32; accEB F^2*8; accHcc 16; accbc 4
void ref_get_accumulators(void *h, float *topA, float *topL, float *accD, float *accE, float *accEB, float *accHcc, float *accbc) {
    EnergyFunctional *ef = ((RefWindow *) h)->ef.get();
    int F = ef->nFrames;
    for (int i = 0; i < F * F; i++) {
        if (topA) { ef->accSSE_top_A->acc[0][i].finish(); for (int r = 0; r < 13; r++) for (int c = 0; c < 13; c++) topA[i * 169 + r * 13 + c] = ef->accSSE_top_A->acc[0][i].H(r, c); }
        if (topL) { ef->accSSE_top_L->acc[0][i].finish(); for (int r = 0; r < 13; r++) for (int c = 0; c < 13; c++) topL[i * 169 + r * 13 + c] = ef->accSSE_top_L->acc[0][i].H(r, c); }
        if (accE) for (int r = 0; r < 8; r++) for (int c = 0; c < 4; c++) accE[i * 32 + r * 4 + c] = ef->accSSE_bot->accE[0][i].A1m(r, c);
        if (accEB) for (int r = 0; r < 8; r++) accEB[i * 8 + r] = ef->accSSE_bot->accEB[0][i].A1m(r, 0);
    }
    if (accD) for (int i = 0; i < F * F * F; i++) for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) accD[i * 64 + r * 8 + c] = ef->accSSE_bot->accD[0][i].A1m(r, c);
    if (accHcc) for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) accHcc[r * 4 + c] = ef->accSSE_bot->accHcc[0].A1m(r, c);
    if (accbc) for (int r = 0; r < 4; r++) accbc[r] = ef->accSSE_bot->accbc[0].A1m(r, 0);
}

// what EnergyFunctional keeps of the last solve: lastHS = H_A + H_L + H_M - H_sc, lastbS, lastX (EF.cc:296-345)
void ref_get_system(void *h, double *lastHS, double *lastbS, double *x) {
    EnergyFunctional *ef = ((RefWindow *) h)->ef.get();
    int n = ef->lastHS.rows();
    if (lastHS) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) lastHS[i * n + j] = ef->lastHS(i, j);
    if (lastbS) for (int i = 0; i < n; i++) lastbS[i] = ef->lastbS[i];
    if (x) for (int i = 0; i < (int) ef->lastX.size(); i++) x[i] = ef->lastX[i];
}

void ref_get_prior(void *h, double *HM, double *bM) {
    EnergyFunctional *ef = ((RefWindow *) h)->ef.get();
    int n = ef->HM.rows();
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) HM[i * n + j] = ef->HM(i, j); bM[i] = ef->bM[i]; }
}

// ---- marginalisation ---------------------------------------------------------------------------------------------------------
// status per flat point as decided by FullSystem::flagPointsForRemoval (host policy, taken from the caller): 0 keep, 3 marginalise
// (PS_MARGINALIZED), 1 / 2 drop.  For the points to marginalise the loop of FullSystem.cc:1241-1250 runs on the reference objects:
// resetOOB, linearize, applyRes(true), fixLinearizationF for the active ones.
void ref_flag_points(void *h, const int32_t *status) {
    RefWindow *W = (RefWindow *) h;
    for (size_t i = 0; i < W->pointByFlat.size(); i++) {
        shared_ptr<PointHessian> ph = W->pointByFlat[i];
        int st = status[i] % 100;
        if (st == 3) {
            for (auto &r : ph->residuals) {
                r->resetOOB();
                r->linearize(W->Hcalib);
                r->isLinearized = false;
                r->applyRes(true);
                if (r->isActive()) r->fixLinearizationF(W->ef);
            }
            ph->point->status = Point::PointStatus::MARGINALIZED;
        } else if (st == 1 || st == 2) {
            ph->point->status = (st == 1) ? Point::PointStatus::OUTLIER : Point::PointStatus::OUT;
        }
    }
}
void ref_drop_points(void *h) { ((RefWindow *) h)->ef->dropPointsF(); }
void ref_marginalize_points(void *h) {
    RefWindow *W = (RefWindow *) h;
    W->ef->lastNullspaces_forLogging = get_nullspaces(W, W->ef->lastNullspaces_pose, W->ef->lastNullspaces_scale, W->ef->lastNullspaces_affA, W->ef->lastNullspaces_affB);
    W->ef->marginalizePointsF();
    for (auto &p : W->ef->allPointsToMarg) p->alreadyRemoved = true;
}
// FullSystem::marginalizeFrame (FullSystem.cc:602-640): EnergyFunctional::marginalizeFrame, then drop the residuals that target it
void ref_marginalize_frame(void *h, int idx) {
    RefWindow *W = (RefWindow *) h;
    shared_ptr<FrameHessian> fh = W->ef->frames[idx];
    W->ef->marginalizeFrame(fh);
    for (auto &fr : W->frames) {
        if (fr->frameHessian == fh) continue;
        for (auto &feat : fr->features) {
            if (feat->status != Feature::FeatureStatus::VALID || feat->point->status != Point::PointStatus::ACTIVE) continue;
            shared_ptr<PointHessian> ph = feat->point->mpPH;
            size_t n = ph->residuals.size();
            for (size_t i = 0; i < n; i++) {
                shared_ptr<PointFrameResidual> r = ph->residuals[i];
                if (r->target.lock() == fh) {
                    if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = nullptr;
                    else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = nullptr;
                    W->ef->dropResidual(r);
                    i--; n--;
                }
            }
        }
    }
    for (size_t i = 0; i < W->frames.size(); i++) if (W->frames[i]->frameHessian == fh) { W->frames.erase(W->frames.begin() + i); break; }
    W->ef->setAdjointsF(W->Hcalib);
    set_precalc(W);
}

// ---- the reference's own FullSystem on this window (FullSystem.cc compiled unmodified) -------------------------------------------
// ref_fs_attach builds a FullSystem (its constructor starts the idle mapping thread and allocates trackers / selector for wG[0] x hG[0])
// and hands it the window: frames, the EnergyFunctional (ef->red = &threadReduce as FullSystem.cc:41 does) and the camera.
static FullSystem *fs_of(RefWindow *W) {
    if (!W->fs) {
        setting_enableLoopClosing = false;
        W->fs = new FullSystem(nullptr);
        W->fs->linearizeOperation = true;
        W->fs->ef = W->ef;
        W->ef->red = &W->fs->threadReduce;
        W->fs->Hcalib = W->cam;
    }
    W->fs->frames = W->frames;
    return W->fs;
}
struct FsCall {          // capture LOG(...) for the duration of a call, write the frame list back afterwards
    RefWindow *W; FullSystem *fs;
    explicit FsCall(void *h) : W((RefWindow *) h), fs(fs_of(W)) { W->fsLog.clear(); ref_shim::log_capture_slot() = &W->fsLog; }
    ~FsCall() { ref_shim::log_capture_slot() = nullptr; W->frames = fs->frames; W->activeResiduals = fs->activeResiduals; }
};

void ref_fs_attach(void *h, int multithreading) { FsCall c(h); multiThreading = multithreading != 0; }
// the FullSystem object itself, for the compiled drop-in adapter (adapter/adapter_capi.cc) that runs on this window instead of optimize()
void *ref_fs_handle(void *h) { FsCall c(h); return c.fs; }
void ref_fs_sync_back(void *h) { RefWindow *W = (RefWindow *) h; if (W->fs) { W->frames = W->fs->frames; W->activeResiduals = W->fs->activeResiduals; } }
int ref_fs_log(void *h, char *out, int cap) {
    RefWindow *W = (RefWindow *) h;
    int n = (int) W->fsLog.size();
    if (out && cap > 0) { int m = n < cap - 1 ? n : cap - 1; memcpy(out, W->fsLog.data(), m); out[m] = 0; }
    return n;
}
// float FullSystem::optimize(int mnumOptIts) (FullSystem.cc:725-864); the per-iteration energies are in the log (printOptRes)
float ref_fs_optimize(void *h, int iterations) { FsCall c(h); return c.fs->optimize(iterations); }
// wall time of one FullSystem::optimize(iterations) call (the cpu_baseline leg of bench.py: the reference's own code, shared_ptr graph,
// IndexThreadReduce and all - with the header shim's Eigen, see ref_shim/Eigen/Core)
double ref_fs_time_optimize(void *h, int iterations) {
    RefWindow *W = (RefWindow *) h;
    FullSystem *fs = fs_of(W);
    auto t0 = std::chrono::steady_clock::now();
    fs->optimize(iterations);
    auto t1 = std::chrono::steady_clock::now();
    W->frames = fs->frames; W->activeResiduals = fs->activeResiduals;
    return std::chrono::duration<double>(t1 - t0).count();
}
int ref_fs_is_lost(void *h) { return ((RefWindow *) h)->fs && ((RefWindow *) h)->fs->isLost ? 1 : 0; }
// the activeResiduals list of optimize (:735-755) - restated (it is inline in optimize); everything below is the reference's member
void ref_fs_collect_active(void *h, int reset_oob) { ref_collect_active(h, reset_oob); FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; }
// Vec3 FullSystem::linearizeAll(bool fixLinearization) (:1442-1492): linearize + setNewFrameEnergyTH (+ state bookkeeping / residual removal)
void ref_fs_linearize_all(void *h, int fix, double out[3]) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; Vec3 e = c.fs->linearizeAll(fix != 0); for (int i = 0; i < 3; i++) out[i] = e[i]; }
void ref_fs_apply_res(void *h) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; Vec10 st; c.fs->applyRes_Reductor(true, 0, (int) c.fs->activeResiduals.size(), &st, 0); }
void ref_fs_set_new_frame_energy_th(void *h) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; c.fs->setNewFrameEnergyTH(); }
void ref_fs_backup_state(void *h, int backup_last_step) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; c.fs->backupState(backup_last_step != 0); }
int ref_fs_do_step(void *h, float stepfac) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; return c.fs->doStepFromBackup(stepfac, stepfac, stepfac, stepfac, stepfac) ? 1 : 0; }
void ref_fs_load_state_backup(void *h) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; c.fs->loadSateBackup(); }
void ref_fs_solve_system(void *h, int iteration, double lambda) { FsCall c(h); c.fs->activeResiduals = c.W->activeResiduals; c.fs->solveSystem(iteration, lambda); }
void ref_fs_calc_energies(void *h, double *EL, double *EM) { FsCall c(h); *EL = c.fs->calcLEnergy(); *EM = c.fs->calcMEnergy(); }
void ref_fs_set_precalc(void *h) { FsCall c(h); c.fs->setPrecalcValues(); }
// FullSystem::flagPointsForRemoval (:1208-1290) and FullSystem::marginalizeFrame (:602-640): the members themselves
void ref_fs_flag_points_for_removal(void *h) { FsCall c(h); c.fs->flagPointsForRemoval(); }
void ref_fs_flag_frame(void *h, int idx) { ((RefWindow *) h)->ef->frames[idx]->flaggedForMarginalization = true; }
void ref_fs_marginalize_frame(void *h, int idx) {
    FsCall c(h);
    shared_ptr<Frame> fr = c.fs->frames[idx];
    // marginalizeFrame ends with frame->ReleaseAll(): ~FrameHessian would delete[] the pyramid, which lives in imageStore here
    for (int l = 0; l < PYR_LEVELS; l++) { fr->frameHessian->dIp[l] = nullptr; fr->frameHessian->absSquaredGrad[l] = nullptr; }
    c.fs->marginalizeFrame(fr);
}

// current-state pair transforms as ImmaturePoint::linearizeResidual reads them (FrameFramePrecalc: PRE_RTll 9, PRE_tTll 3, PRE_aff_mode 2)
// at [host*F + target]: the oracle's / the device's activation takes the same 14 floats
void ref_get_pair_rt(void *h, float *out) {
    RefWindow *W = (RefWindow *) h;
    int F = W->ef->frames.size();
    for (int a = 0; a < F; a++)
        for (int t = 0; t < F; t++) {
            const FrameFramePrecalc &p = W->ef->frames[a]->targetPrecalc[t];
            float *o = out + (a * F + t) * 14;
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o[i * 3 + j] = p.PRE_RTll(i, j); o[9 + i] = p.PRE_tTll[i]; }
            o[12] = p.PRE_aff_mode[0]; o[13] = p.PRE_aff_mode[1];
        }
}

// an ImmaturePoint of the reference on the window's host frame from a flat record: the constructor's own sampling of the host image is
// replaced by the record's values (colour, weights, gradH, energyTH, depth interval), so that every side of a comparison starts identically
static shared_ptr<ImmaturePoint> make_immature(RefWindow *W, FullSystem *fs, const ldso_immature_t &q) {
    const int w = wG[0], hh = hG[0];
    shared_ptr<Frame> host = fs->frames[q.host];
    shared_ptr<Feature> feat(new Feature(q.u, q.v, host));
    float uc = std::min(std::max(q.u, 8.0f), (float) w - 9), vc = std::min(std::max(q.v, 8.0f), (float) hh - 9);
    feat->uv = Vec2f(uc, vc);                                  // keep the constructor's own sampling inside the image; its results are replaced by the record's
    shared_ptr<ImmaturePoint> ip(new ImmaturePoint(host, feat, 1, W->Hcalib));
    feat->uv = Vec2f(q.u, q.v);
    feat->ip = ip;
    memcpy(ip->color, q.color, sizeof(q.color)); memcpy(ip->weights, q.weights, sizeof(q.weights));
    ip->gradH(0, 0) = q.gradH[0]; ip->gradH(0, 1) = q.gradH[1]; ip->gradH(1, 0) = q.gradH[2]; ip->gradH(1, 1) = q.gradH[3];
    ip->energyTH = q.energyTH; ip->idepth_min = q.idepth_min; ip->idepth_max = q.idepth_max; ip->quality = q.quality;
    ip->lastTraceStatus = (ImmaturePointStatus) q.lastTraceStatus;
    return ip;
}

// shared_ptr<PointHessian> FullSystem::optimizeImmaturePoint(point, minObs, residuals) (FullSystem.cc:892-1010) - the member itself - for n
// immature-point records against the key frames of the window.  Observable results: the verdict (non-null return), the new point's
// inverse depth, and the caller-owned temporary residuals' final states (the energy / Hdd / bd locals of the function are not).
void ref_fs_activate_points(void *h, int n, const ldso_immature_t *pts, int min_obs, float min_idepth_hessian, int gn_iterations, ldso_activation_t *out) {
    FsCall c(h);
    setting_minIdepthH_act = min_idepth_hessian; setting_GNItsOnPointActivation = gn_iterations;
    const int F = (int) c.fs->frames.size();
    for (int i = 0; i < n; i++) {
        const ldso_immature_t &q = pts[i];
        ldso_activation_t &o = out[i];
        memset(&o, 0, sizeof(o));
        for (int f = 0; f < LDSO_MAX_FRAMES; f++) o.res_state[f] = -1;
        o.energy = o.Hdd = o.bd = NAN; o.iterations = -1;
        shared_ptr<ImmaturePoint> ip = make_immature(c.W, c.fs, q);
        shared_ptr<Feature> feat = ip->feature;
        std::vector<shared_ptr<ImmaturePointTemporaryResidual>> tr;
        for (int f = 0; f + 1 < F; f++) tr.push_back(shared_ptr<ImmaturePointTemporaryResidual>(new ImmaturePointTemporaryResidual()));
        shared_ptr<PointHessian> ph = c.fs->optimizeImmaturePoint(ip, min_obs, tr);
        o.ok = ph ? 1 : 0;
        o.idepth = ph ? ph->idepth : NAN;
        int good = 0;
        for (auto &t : tr) {
            shared_ptr<FrameHessian> tf = t->target.lock();
            if (!tf) continue;
            o.res_state[tf->idx] = (int) t->state_state;
            if (t->state_state == ResState::IN) good++;
        }
        o.numGoodRes = good;
        feat->ReleaseAll();
    }
}

// ---- FullSystem::traceNewCoarse (FullSystem.cc:1012-1050), the member itself ---------------------------------------------------------------
// ref_fs_add_immature hangs immature points (Feature::IMMATURE + ImmaturePoint) on the window's key frames, ref_fs_new_frame builds the frame to
// trace into (level-0 image, pose through setEvalPT_scaled as makeKeyFrame / makeNonKeyFrame do, :596 / :430), ref_fs_trace_new_coarse calls
// the member, ref_fs_get_immature reads the records back in traversal order (frames, then their features).
void ref_fs_add_immature(void *h, int n, const ldso_immature_t *pts) {
    FsCall c(h);
    for (int i = 0; i < n; i++) {
        shared_ptr<ImmaturePoint> ip = make_immature(c.W, c.fs, pts[i]);
        ip->lastTraceUV = Vec2f(pts[i].lastTraceUV[0], pts[i].lastTraceUV[1]); ip->lastTracePixelInterval = pts[i].lastTracePixelInterval;
        ip->feature->status = Feature::FeatureStatus::IMMATURE;
        c.fs->frames[pts[i].host]->features.push_back(ip->feature);
    }
}
static std::vector<shared_ptr<Frame>> g_newFrames;          // frames handed out by ref_fs_new_frame (kept alive for the adapter's borrowed pointers)
static std::vector<std::vector<float>> g_newFrameImages;
void *ref_fs_new_frame(void *h, const float *dI_level0, const double *w2c, float aff_a, float aff_b, float exposure) {
    FsCall c(h);
    const int w = wG[0], hh = hG[0];
    shared_ptr<Frame> fr(new Frame());
    fr->CreateFH(fr);
    auto fh = fr->frameHessian;
    for (int l = 0; l < PYR_LEVELS; l++) { fh->dIp[l] = nullptr; fh->absSquaredGrad[l] = nullptr; }
    g_newFrameImages.emplace_back(dI_level0, dI_level0 + (size_t) w * hh * 3);
    fh->dIp[0] = (Vec3f *) g_newFrameImages.back().data(); fh->dI = fh->dIp[0];
    fh->ab_exposure = exposure;
    fr->setPose(se3_from34(w2c)); fr->aff_g2l = AffLight(aff_a, aff_b);
    fh->setEvalPT_scaled(fr->getPose(), fr->aff_g2l);
    g_newFrames.push_back(fr);
    return &g_newFrames.back()->frameHessian;               // shared_ptr<FrameHessian> *
}
// Frame::id comes from a process-wide counter (Frame.cc:13-20): two object graphs built side by side get different ids, and FrameHessian::getPrior
// / the adapter's image slots key on it - a comparison run gives the frames of both graphs the same ids
void ref_fs_set_frame_id(void *fh_shared_ptr, long id) {
    shared_ptr<FrameHessian> &fh = *(shared_ptr<FrameHessian> *) fh_shared_ptr;
    fh->frame->id = (unsigned long) id;
}
void ref_fs_release_new_frames() {
    for (auto &fr : g_newFrames) if (fr->frameHessian) for (int l = 0; l < PYR_LEVELS; l++) { fr->frameHessian->dIp[l] = nullptr; fr->frameHessian->absSquaredGrad[l] = nullptr; }
    g_newFrames.clear(); g_newFrameImages.clear();
}
void ref_fs_trace_new_coarse(void *h, void *fh_shared_ptr) { FsCall c(h); c.fs->traceNewCoarse(*(shared_ptr<FrameHessian> *) fh_shared_ptr); }
int ref_fs_get_immature(void *h, ldso_immature_t *out, int cap) {
    FsCall c(h);
    int n = 0;
    for (size_t f = 0; f < c.fs->frames.size(); f++)
        for (auto &feat : c.fs->frames[f]->features) {
            if (!(feat->status == Feature::FeatureStatus::IMMATURE && feat->ip)) continue;
            if (n < cap) {
                ldso_immature_t &q = out[n]; ImmaturePoint &ip = *feat->ip;
                memset(&q, 0, sizeof(q));
                q.u = feat->uv[0]; q.v = feat->uv[1];
                memcpy(q.color, ip.color, sizeof(q.color)); memcpy(q.weights, ip.weights, sizeof(q.weights));
                q.gradH[0] = ip.gradH(0, 0); q.gradH[1] = ip.gradH(0, 1); q.gradH[2] = ip.gradH(1, 0); q.gradH[3] = ip.gradH(1, 1);
                q.energyTH = ip.energyTH; q.idepth_min = ip.idepth_min; q.idepth_max = ip.idepth_max; q.quality = ip.quality;
                q.lastTraceStatus = (int32_t) ip.lastTraceStatus; q.lastTraceUV[0] = ip.lastTraceUV[0]; q.lastTraceUV[1] = ip.lastTraceUV[1];
                q.lastTracePixelInterval = ip.lastTracePixelInterval; q.host = (int32_t) f;
            }
            n++;
        }
    return n;
}

// for the compiled adapter's activation test (adapter/adapter_capi.cc): the same ImmaturePoint objects as ref_fs_activate_points builds,
// handed over as a heap-allocated std::vector<shared_ptr<ImmaturePoint>> (ref_fs_free_immature releases them and what they created)
void *ref_fs_build_immature(void *h, int n, const ldso_immature_t *pts, int min_obs_unused, float min_idepth_hessian, int gn_iterations) {
    FsCall c(h);
    (void) min_obs_unused;
    setting_minIdepthH_act = min_idepth_hessian; setting_GNItsOnPointActivation = gn_iterations;
    auto *v = new std::vector<shared_ptr<ImmaturePoint>>();
    for (int i = 0; i < n; i++) v->push_back(make_immature(c.W, c.fs, pts[i]));
    return v;
}
void ref_fs_free_immature(void *vec) {
    auto *v = (std::vector<shared_ptr<ImmaturePoint>> *) vec;
    // make_immature's features are owned by their immature point alone (they are in no frame's list): Feature::ReleaseImmature (Feature.cc:26-31) drops that
    // owner FIRST (ip->feature = nullptr) and then writes its own member - into a Feature that no longer exists unless somebody else holds it.  Hold it here.
    // (An AddressSanitizer build of this library found it: heap-use-after-free, an occasional "corrupted double-linked list" in tests/test_adapter_gpu.py.)
    for (auto &ip : *v) if (ip) { shared_ptr<Feature> f = ip->feature; if (f) f->ReleaseAll(); }
    delete v;
}

// ---- FrameHessian::makeImages (FrameHessian.cc:44-113) -----------------------------------------------------------------------
// out[l]: (w>>l)*(h>>l)*3 floats
void ref_make_images(int w, int h, int levels, const float *color, float *const *out) {
    Eigen::Matrix3f K = Eigen::Matrix3f::Identity();
    K(0, 0) = K(1, 1) = 1; K(0, 2) = w / 2.0f; K(1, 2) = h / 2.0f;
    setGlobalCalib(w, h, K);
    pyrLevelsUsed = levels;
    setting_enableLoopClosing = false;
    shared_ptr<Frame> fr(new Frame());
    fr->CreateFH(fr);
    shared_ptr<FrameHessian> fh = fr->frameHessian;
    std::vector<float> c(color, color + (size_t) w * h);
    fh->makeImages(c.data(), nullptr);
    for (int l = 0; l < levels; l++) memcpy(out[l], fh->dIp[l], (size_t) (w >> l) * (h >> l) * 12);
}


// ---- CoarseTracker (src/frontend/CoarseTracker.cc, compiled unmodified) --------------------------------------------------------
namespace {
struct RefTracker {
    shared_ptr<Camera> cam;
    shared_ptr<CoarseTracker> tr;
    shared_ptr<Frame> refFrame, newFrame;
    std::vector<shared_ptr<FrameHessian>> fhs;
    std::vector<shared_ptr<PointFrameResidual>> keep;
    std::vector<std::vector<float>> refImgs, newImgs;
    int levels = 0, w = 0, h = 0;
    FullSystem *fs = nullptr;          // ref_tr_track_new_coarse: the reference's FullSystem around this tracker
};
static void detach_images(shared_ptr<Frame> &fr) { if (fr && fr->frameHessian) for (int l = 0; l < PYR_LEVELS; l++) { fr->frameHessian->dIp[l] = nullptr; fr->frameHessian->absSquaredGrad[l] = nullptr; } }
static shared_ptr<Frame> make_frame(std::vector<std::vector<float>> &store, const float *const *dIp, int w, int h, int levels, float exposure) {
    shared_ptr<Frame> fr(new Frame());
    fr->CreateFH(fr);
    auto fh = fr->frameHessian;
    for (int l = 0; l < PYR_LEVELS; l++) { fh->dIp[l] = nullptr; fh->absSquaredGrad[l] = nullptr; }
    store.resize(levels);
    for (int l = 0; l < levels; l++) { size_t n = (size_t) (w >> l) * (h >> l) * 3; store[l].assign(dIp[l], dIp[l] + n); fh->dIp[l] = (Vec3f *) store[l].data(); }
    fh->dI = fh->dIp[0];
    fh->ab_exposure = exposure;
    return fr;
}
}  // namespace

void *ref_tr_create(int w, int h, int levels, const ldso_settings_t *settings, const ldso_calib_t *calib) {
    RefTracker *T = new RefTracker();
    apply_settings(settings);
    T->levels = levels; T->w = w; T->h = h;
    Eigen::Matrix3f K = Eigen::Matrix3f::Zero();
    K(0, 0) = (float) (SCALE_F * calib->value[0]); K(1, 1) = (float) (SCALE_F * calib->value[1]);
    K(0, 2) = (float) (SCALE_C * calib->value[2]); K(1, 2) = (float) (SCALE_C * calib->value[3]); K(2, 2) = 1;
    setGlobalCalib(w, h, K);
    pyrLevelsUsed = levels;
    for (int l = 0; l < levels; l++) { wG[l] = w >> l; hG[l] = h >> l; }
    T->cam.reset(new Camera(K(0, 0), K(1, 1), K(0, 2), K(1, 2)));
    T->cam->CreateCH(T->cam);
    VecC v0, vz;
    for (int i = 0; i < 4; i++) { v0[i] = calib->value[i]; vz[i] = calib->value_zero[i]; }
    T->cam->mpCH->value_zero = vz;
    T->cam->mpCH->setValue(v0);
    T->tr.reset(new CoarseTracker(w, h));
    T->tr->makeK(T->cam->mpCH);
    return T;
}
void ref_tr_destroy(void *h) { RefTracker *T = (RefTracker *) h; detach_images(T->refFrame); detach_images(T->newFrame); if (T->fs) { T->fs->allFrameHistory.clear(); delete T->fs; } delete T; }

// Vec4 FullSystem::trackNewCoarse(shared_ptr<FrameHessian> fh) (FullSystem.cc:179-386) - the member itself: the motion-hypothesis list from
// the poses of the last two frames and the reference key frame, the try loop with the growing achievedRes thresholds, the pose / affine
// hand-over to the new frame.  The FullSystem gets this driver's tracker (reference + new frame set) and a three-frame history.
// the FullSystem around this tracker with a three-frame history (sprelast, slast, the new frame): what trackNewCoarse reads
static FullSystem *tr_prepare_history(RefTracker *T, const double *sprelast, const double *slast, const double *lastF, int posesValid, const float *aff_last,
                                      const double *lastCoarseRMSE, double reTrackThreshold) {
    if (!T->fs) { setting_enableLoopClosing = false; T->fs = new FullSystem(nullptr); }
    FullSystem *fs = T->fs;
    fs->coarseTracker = T->tr;
    setting_reTrackThreshold = reTrackThreshold;
    shared_ptr<Frame> f0(new Frame()), f1(new Frame());
    f0->setPose(se3_from34(sprelast)); f1->setPose(se3_from34(slast)); T->refFrame->setPose(se3_from34(lastF));
    f0->poseValid = f1->poseValid = T->refFrame->poseValid = posesValid != 0;
    f1->aff_g2l = AffLight(aff_last[0], aff_last[1]);
    fs->allFrameHistory.clear();
    fs->allFrameHistory.push_back(f0); fs->allFrameHistory.push_back(f1); fs->allFrameHistory.push_back(T->newFrame);
    for (int i = 0; i < 5; i++) fs->lastCoarseRMSE[i] = lastCoarseRMSE[i];
    return fs;
}
// for the compiled drop-in adapter: the prepared FullSystem, the tracker's frame-hessian list and the new frame (borrowed pointers)
void *ref_tr_prepare(void *h, const double *sprelast, const double *slast, const double *lastF, int posesValid, const float *aff_last,
                     const double *lastCoarseRMSE, double reTrackThreshold) {
    return tr_prepare_history((RefTracker *) h, sprelast, slast, lastF, posesValid, aff_last, lastCoarseRMSE, reTrackThreshold);
}
void *ref_tr_coarse_tracker(void *h) { return ((RefTracker *) h)->tr.get(); }
void *ref_tr_frame_hessians(void *h) { return &((RefTracker *) h)->fhs; }
void *ref_tr_new_frame_hessian(void *h) { return &((RefTracker *) h)->newFrame->frameHessian; }
void *ref_tr_calib_hessian(void *h) { return &((RefTracker *) h)->cam->mpCH; }
void ref_tr_read_result(void *h, double *lastCoarseRMSE, double *new_w2c, float *aff_out) {
    RefTracker *T = (RefTracker *) h;
    for (int i = 0; i < 5; i++) lastCoarseRMSE[i] = T->fs->lastCoarseRMSE[i];
    Eigen::Matrix<double, 3, 4> M = T->newFrame->getPose().matrix3x4();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) new_w2c[i * 4 + j] = M(i, j);
    aff_out[0] = T->newFrame->aff_g2l.a; aff_out[1] = T->newFrame->aff_g2l.b;
}

int ref_tr_track_new_coarse(void *h, const double *sprelast, const double *slast, const double *lastF, int posesValid, const float *aff_last,
                            double *lastCoarseRMSE, double reTrackThreshold, double *result4, double *new_w2c, float *aff_out) {
    RefTracker *T = (RefTracker *) h;
    FullSystem *fs = tr_prepare_history(T, sprelast, slast, lastF, posesValid, aff_last, lastCoarseRMSE, reTrackThreshold);
    Vec4 r = fs->trackNewCoarse(T->newFrame->frameHessian);
    for (int i = 0; i < 4; i++) result4[i] = r[i];
    for (int i = 0; i < 5; i++) lastCoarseRMSE[i] = fs->lastCoarseRMSE[i];
    Eigen::Matrix<double, 3, 4> M = T->newFrame->getPose().matrix3x4();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) new_w2c[i * 4 + j] = M(i, j);
    aff_out[0] = T->newFrame->aff_g2l.a; aff_out[1] = T->newFrame->aff_g2l.b;
    return 0;
}

// setCoarseTrackingRef on a reference frame whose active points are given as (Ku, Kv, new_idepth, HdiF) x n: each becomes a
// Feature / Point / PointHessian with lastResiduals[0] = an active IN residual towards the reference frame (what makeCoarseDepthL0 reads)
void ref_tr_set_ref(void *h, const float *const *ref_dIp, float ref_a, float ref_b, float ref_exposure, const float *pts, int n) {
    RefTracker *T = (RefTracker *) h;
    detach_images(T->refFrame);
    T->refFrame = make_frame(T->refImgs, ref_dIp, T->w, T->h, T->levels, ref_exposure);
    auto fh = T->refFrame->frameHessian;
    Vec10 ss = Vec10::Zero(); ss[6] = ref_a; ss[7] = ref_b;
    fh->state_scaled = ss;                               // lastRef->aff_g2l() reads state_scaled[6..7] (FrameHessian.h:64-66)
    T->keep.clear();
    for (int i = 0; i < n; i++) {
        shared_ptr<Feature> feat(new Feature(pts[i * 4], pts[i * 4 + 1], T->refFrame));
        feat->status = Feature::FeatureStatus::VALID;
        shared_ptr<Point> pt(new Point());
        pt->status = Point::PointStatus::ACTIVE;
        pt->mHostFeature = feat; feat->point = pt;
        shared_ptr<PointHessian> ph(new PointHessian());
        pt->mpPH = ph; ph->point = pt;
        ph->HdiF = pts[i * 4 + 3];
        shared_ptr<PointFrameResidual> r(new PointFrameResidual(ph, fh, fh));
        r->centerProjectedTo = Vec3f(pts[i * 4], pts[i * 4 + 1], pts[i * 4 + 2]);
        r->isActiveAndIsGoodNEW = true;
        r->state_state = ResState::IN;
        ph->lastResiduals[0] = {r, ResState::IN};
        T->keep.push_back(r);
        T->refFrame->features.push_back(feat);
    }
    T->fhs.clear(); T->fhs.push_back(fh);
    T->tr->setCoarseTrackingRef(T->fhs);
}
void ref_tr_set_new_frame(void *h, const float *const *new_dIp, float exposure) {
    RefTracker *T = (RefTracker *) h;
    detach_images(T->newFrame);
    T->newFrame = make_frame(T->newImgs, new_dIp, T->w, T->h, T->levels, exposure);
    T->tr->newFrame = T->newFrame->frameHessian;
}
int ref_tr_pc_n(void *h, int lvl) { return ((RefTracker *) h)->tr->pc_n[lvl]; }
void ref_tr_get_pc(void *h, int lvl, float *u, float *v, float *idepth, float *color) {
    CoarseTracker *tr = ((RefTracker *) h)->tr.get();
    int n = tr->pc_n[lvl];
    memcpy(u, tr->pc_u[lvl], n * 4); memcpy(v, tr->pc_v[lvl], n * 4); memcpy(idepth, tr->pc_idepth[lvl], n * 4); memcpy(color, tr->pc_color[lvl], n * 4);
}
void ref_tr_get_K(void *h, float *fx, float *fy, float *cx, float *cy) {
    RefTracker *T = (RefTracker *) h;
    for (int l = 0; l < T->levels; l++) { fx[l] = T->tr->fx[l]; fy[l] = T->tr->fy[l]; cx[l] = T->tr->cx[l]; cy[l] = T->tr->cy[l]; }
}
int ref_tr_calc_res(void *h, int lvl, const double *T_ref2new, float a, float b, float cutoffTH, double *rs_out) {
    CoarseTracker *tr = ((RefTracker *) h)->tr.get();
    Vec6 rs = tr->calcRes(lvl, se3_from34(T_ref2new), AffLight(a, b), cutoffTH);
    for (int i = 0; i < 6; i++) rs_out[i] = rs[i];
    return tr->buf_warped_n;
}
void ref_tr_get_warped(void *h, float *idepth, float *u, float *v, float *dx, float *dy, float *residual, float *weight, float *refColor) {
    CoarseTracker *tr = ((RefTracker *) h)->tr.get();
    int n = tr->buf_warped_n;
    memcpy(idepth, tr->buf_warped_idepth, n * 4); memcpy(u, tr->buf_warped_u, n * 4); memcpy(v, tr->buf_warped_v, n * 4);
    memcpy(dx, tr->buf_warped_dx, n * 4); memcpy(dy, tr->buf_warped_dy, n * 4); memcpy(residual, tr->buf_warped_residual, n * 4);
    memcpy(weight, tr->buf_warped_weight, n * 4); memcpy(refColor, tr->buf_warped_refColor, n * 4);
}
void ref_tr_calc_gs(void *h, int lvl, const double *T_ref2new, float a, float b, double *H_out, double *b_out) {
    CoarseTracker *tr = ((RefTracker *) h)->tr.get();
    Mat88 H; Vec8 bb;
    tr->calcGSSSE(lvl, H, bb, se3_from34(T_ref2new), AffLight(a, b));
    for (int i = 0; i < 8; i++) { for (int j = 0; j < 8; j++) H_out[i * 8 + j] = H(i, j); b_out[i] = bb[i]; }
}
int ref_tr_track(void *h, double *T_inout, float *ab_inout, int coarsestLvl, const double *minResForAbort, double *lastResiduals, double *flow) {
    RefTracker *T = (RefTracker *) h;
    SE3 pose = se3_from34(T_inout);
    AffLight aff(ab_inout[0], ab_inout[1]);
    Vec5 mr; for (int i = 0; i < 5; i++) mr[i] = minResForAbort[i];
    bool ok = T->tr->trackNewestCoarse(T->newFrame->frameHessian, pose, aff, coarsestLvl, mr);
    Eigen::Matrix<double, 3, 4> M = pose.matrix3x4();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) T_inout[i * 4 + j] = M(i, j);
    ab_inout[0] = aff.a; ab_inout[1] = aff.b;
    for (int i = 0; i < 5; i++) lastResiduals[i] = T->tr->lastResiduals[i];
    for (int i = 0; i < 3; i++) flow[i] = T->tr->lastFlowIndicators[i];
    return ok ? 1 : 0;
}


// ---- ImmaturePoint::traceOn (src/internal/ImmaturePoint.cc:47-310, compiled unmodified) under the loop of
//      FullSystem::traceNewCoarse (FullSystem.cc:1012-1050): every immature point against the new frame; counts[6] per status -----
void ref_trace_on(int n, ldso_immature_t *pts, const float *dI, int w, int h, int n_hosts, const float *KRKi, const float *Kt, const float *aff,
                  const ldso_trace_settings_t *s, int *counts) {
    setting_maxPixSearch = s->maxPixSearch; setting_trace_stepsize = s->trace_stepsize; setting_trace_GNThreshold = s->trace_GNThreshold;
    setting_trace_extraSlackOnTH = s->trace_extraSlackOnTH; setting_trace_slackInterval = s->trace_slackInterval;
    setting_trace_minImprovementFactor = s->trace_minImprovementFactor; setting_huberTH = s->huberTH;
    setting_trace_GNIterations = s->trace_GNIterations; setting_minTraceTestRadius = s->minTraceTestRadius;
    Eigen::Matrix3f K = Eigen::Matrix3f::Identity(); K(0, 2) = w / 2.0f; K(1, 2) = h / 2.0f;
    setGlobalCalib(w, h, K);
    shared_ptr<Camera> cam(new Camera(1, 1, w / 2.0, h / 2.0));
    cam->CreateCH(cam);
    // the new frame (target of the search) and a blank host frame for the ImmaturePoint constructor, whose results are then
    // replaced by the record's (the constructor samples the HOST image, which the records already carry)
    std::vector<std::vector<float>> store, blankStore;
    const float *lv[1] = {dI};
    shared_ptr<Frame> fr = make_frame(store, lv, w, h, 1, 1.0f);
    std::vector<float> blank((size_t) w * h * 3, 0.0f);
    const float *bl[1] = {blank.data()};
    shared_ptr<Frame> host = make_frame(blankStore, bl, w, h, 1, 1.0f);
    if (counts) for (int i = 0; i < 6; i++) counts[i] = 0;
    for (int i = 0; i < n; i++) {
        ldso_immature_t &q = pts[i];
        if (q.host < 0 || q.host >= n_hosts) continue;
        shared_ptr<Feature> feat(new Feature(q.u, q.v, host));
        float uc = std::min(std::max(q.u, 8.0f), (float) w - 9), vc = std::min(std::max(q.v, 8.0f), (float) h - 9);
        feat->uv = Vec2f(uc, vc);                                  // keep the constructor's sampling inside the blank image
        shared_ptr<ImmaturePoint> ip(new ImmaturePoint(host, feat, 1, cam->mpCH));
        feat->uv = Vec2f(q.u, q.v);
        memcpy(ip->color, q.color, sizeof(q.color)); memcpy(ip->weights, q.weights, sizeof(q.weights));
        ip->gradH(0, 0) = q.gradH[0]; ip->gradH(0, 1) = q.gradH[1]; ip->gradH(1, 0) = q.gradH[2]; ip->gradH(1, 1) = q.gradH[3];
        ip->energyTH = q.energyTH; ip->idepth_min = q.idepth_min; ip->idepth_max = q.idepth_max; ip->quality = q.quality;
        ip->lastTraceStatus = (ImmaturePointStatus) q.lastTraceStatus;
        ip->lastTraceUV = Vec2f(q.lastTraceUV[0], q.lastTraceUV[1]); ip->lastTracePixelInterval = q.lastTracePixelInterval;
        Mat33f M; Vec3f t; Vec2f a;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M(r, c) = KRKi[9 * q.host + r * 3 + c]; t[r] = Kt[3 * q.host + r]; }
        a[0] = aff[2 * q.host]; a[1] = aff[2 * q.host + 1];
        ImmaturePointStatus st = ip->traceOn(fr->frameHessian, M, t, a, cam->mpCH);
        q.idepth_min = ip->idepth_min; q.idepth_max = ip->idepth_max; q.quality = ip->quality; q.lastTraceStatus = (int) ip->lastTraceStatus;
        q.lastTraceUV[0] = ip->lastTraceUV[0]; q.lastTraceUV[1] = ip->lastTraceUV[1]; q.lastTracePixelInterval = ip->lastTracePixelInterval;
        if (counts && (int) st >= 0 && (int) st < 6) counts[(int) st]++;
    }
    detach_images(fr); detach_images(host);
}


// ---- CoarseInitializer::trackFrame (src/frontend/CoarseInitializer.cc, compiled unmodified).  setFirst's pixel selection and
//      k-d tree are upstream of the hot path: the points arrive as records, and the state setFirst leaves behind (:608-616) is set here.
namespace {
struct RefInit {
    shared_ptr<Camera> cam;
    shared_ptr<CoarseInitializer> ci;
    shared_ptr<Frame> first, cur;
    std::vector<std::vector<float>> firstImgs, newImgs;
    int w, h, levels;
};
static void pnt_from(const ldso_init_point_t &q, Pnt &p) {
    p.u = q.u; p.v = q.v; p.idepth = q.idepth; p.isGood = q.isGood != 0; p.energy = Vec2f(q.energy[0], q.energy[1]); p.isGood_new = q.isGood_new != 0;
    p.idepth_new = q.idepth_new; p.energy_new = Vec2f(q.energy_new[0], q.energy_new[1]); p.iR = q.iR; p.iRSumNum = q.iRSumNum;
    p.lastHessian = q.lastHessian; p.lastHessian_new = q.lastHessian_new; p.maxstep = q.maxstep; p.parent = q.parent; p.parentDist = q.parentDist;
    for (int i = 0; i < 10; i++) { p.neighbours[i] = q.neighbours[i]; p.neighboursDist[i] = q.neighboursDist[i]; }
    p.my_type = q.my_type; p.outlierTH = q.outlierTH;
}
static void pnt_to(const Pnt &p, ldso_init_point_t &q) {
    q.u = p.u; q.v = p.v; q.idepth = p.idepth; q.isGood = p.isGood; q.energy[0] = p.energy[0]; q.energy[1] = p.energy[1]; q.isGood_new = p.isGood_new;
    q.idepth_new = p.idepth_new; q.energy_new[0] = p.energy_new[0]; q.energy_new[1] = p.energy_new[1]; q.iR = p.iR; q.iRSumNum = p.iRSumNum;
    q.lastHessian = p.lastHessian; q.lastHessian_new = p.lastHessian_new; q.maxstep = p.maxstep; q.parent = p.parent; q.parentDist = p.parentDist;
    for (int i = 0; i < 10; i++) { q.neighbours[i] = p.neighbours[i]; q.neighboursDist[i] = p.neighboursDist[i]; }
    q.my_type = p.my_type; q.outlierTH = p.outlierTH; q.pad_ = 0;
}
}  // namespace

void *ref_init_create(int w, int h, int levels) {
    RefInit *I = new RefInit();
    I->w = w; I->h = h; I->levels = levels;
    Eigen::Matrix3f K = Eigen::Matrix3f::Identity(); K(0, 2) = w / 2.0f; K(1, 2) = h / 2.0f;
    setGlobalCalib(w, h, K);
    pyrLevelsUsed = levels;
    for (int l = 0; l < levels; l++) { wG[l] = w >> l; hG[l] = h >> l; }
    I->ci.reset(new CoarseInitializer(w, h));
    return I;
}
void ref_init_destroy(void *p) { RefInit *I = (RefInit *) p; detach_images(I->first); detach_images(I->cur); delete I; }
void ref_init_set_first(void *p, const float *calib, const float *const *dIp, float exposure, const ldso_init_point_t *const *points, const int *n_points,
                        float huberTH, int fixAffine) {
    RefInit *I = (RefInit *) p;
    pyrLevelsUsed = I->levels;
    for (int l = 0; l < I->levels; l++) { wG[l] = I->w >> l; hG[l] = I->h >> l; }
    setting_huberTH = huberTH;
    I->cam.reset(new Camera(calib[0], calib[1], calib[2], calib[3]));
    I->cam->CreateCH(I->cam);
    CoarseInitializer *c = I->ci.get();
    c->makeK(I->cam->mpCH);
    detach_images(I->first);
    I->first = make_frame(I->firstImgs, dIp, I->w, I->h, I->levels, exposure);
    c->firstFrame = I->first->frameHessian;
    c->fixAffine = fixAffine != 0;
    for (int l = 0; l < I->levels; l++) {
        if (c->points[l] != 0) delete[] c->points[l];
        c->points[l] = new Pnt[n_points[l] > 0 ? n_points[l] : 1];
        for (int i = 0; i < n_points[l]; i++) pnt_from(points[l][i], c->points[l][i]);
        c->numPoints[l] = n_points[l];
    }
    // the state CoarseInitializer::setFirst leaves behind (CoarseInitializer.cc:608-616)
    c->thisToNext = SE3();
    c->snapped = false;
    c->frameID = c->snappedAt = 0;
    for (int i = 0; i < I->levels; i++) c->dGrads[i].setZero();
}
int ref_init_track_frame(void *p, const float *const *dIp, float exposure, ldso_init_state_t *s) {
    RefInit *I = (RefInit *) p;
    pyrLevelsUsed = I->levels;
    detach_images(I->cur);
    I->cur = make_frame(I->newImgs, dIp, I->w, I->h, I->levels, exposure);
    CoarseInitializer *c = I->ci.get();
    int r = c->trackFrame(I->cur->frameHessian) ? 1 : 0;
    if (s) {
        Eigen::Matrix<double, 3, 4> M = c->thisToNext.matrix3x4();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) s->thisToNext[i * 4 + j] = M(i, j);
        s->aff_a = c->thisToNext_aff.a; s->aff_b = c->thisToNext_aff.b;
        s->snapped = c->snapped; s->snappedAt = c->snappedAt; s->frameID = c->frameID; s->ready = r; s->evals = 0; s->pad_ = 0;
    }
    return r;
}
void ref_init_get_points(void *p, int lvl, ldso_init_point_t *out) {
    CoarseInitializer *c = ((RefInit *) p)->ci.get();
    for (int i = 0; i < c->numPoints[lvl]; i++) pnt_to(c->points[lvl][i], out[i]);
}

}  // extern "C"
