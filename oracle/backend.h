// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h header).  Pinned to reference-compiled code by tests/test_ref_pin.py (see linalg.h).
//
// backend.h — CPU restatement of the windowed photometric bundle-adjustment hot path of LDSO:
//   src/internal/Residuals.cc:13-242                      PointFrameResidual::linearize / fixLinearizationF
//   include/internal/Residuals.h:70-128                   applyRes / takeData / resetOOB
//   include/internal/ResidualProjections.h:24-84          projectPoint (both overloads)
//   include/internal/GlobalFuncs.h:89-103                 getInterpolatedElement33
//   src/internal/FrameFramePrecalc.cc:6-35                FrameFramePrecalc::Set
//   include/internal/FrameHessian.h:43-157, FrameHessian.cc:115-119   state scaling, priors, takeData
//   include/internal/CalibHessian.h:71-100                setValue / setValueScaled
//   include/AffLight.h:27-35                              fromToVecExposure
//   src/internal/OptimizationBackend/AccumulatedTopHessian.cc + .h:39-117
//   src/internal/OptimizationBackend/AccumulatedSCHessian.cc + .h:38-111
//   src/internal/OptimizationBackend/EnergyFunctional.cc (all)
//   src/frontend/FullSystem.cc:725-864 (optimize), :1208-1270 (flagPointsForRemoval), :1423-1793
//   include/internal/IndexThreadReduce.h:56-169           6-worker chunked reduction (timing mode)
// Float/double split is kept exactly where the reference has it.
#pragma once
#include <memory>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include "linalg.h"
#include "lie.h"
#include "accumulators.h"
#include "../include/ldso_window.h"

namespace orc {

// ---- Settings.h:8-43,163 ---------------------------------------------------------------------
static const int NUM_THREADS = 6;
static const int CPARS = 4;
static const int patternNum = 8;
static const float SCALE_IDEPTH = 1.0f, SCALE_XI_ROT = 1.0f, SCALE_XI_TRANS = 0.5f, SCALE_F = 50.0f, SCALE_C = 50.0f,
        SCALE_A = 10.0f, SCALE_B = 1000.0f;
static const float SCALE_IDEPTH_INVERSE = 1.0f / SCALE_IDEPTH, SCALE_XI_ROT_INVERSE = 1.0f / SCALE_XI_ROT,
        SCALE_XI_TRANS_INVERSE = 1.0f / SCALE_XI_TRANS, SCALE_F_INVERSE = 1.0f / SCALE_F, SCALE_C_INVERSE = 1.0f / SCALE_C,
        SCALE_A_INVERSE = 1.0f / SCALE_A, SCALE_B_INVERSE = 1.0f / SCALE_B;
static const int staticPattern8[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};   // Setting.cc:221
#define patternP staticPattern8

enum ResState { IN = 0, OOB, OUTLIER };
enum PointStatus { PS_ACTIVE = 0, PS_OUTLIER, PS_OUT, PS_MARGINALIZED };

struct Globals {   // GlobalCalib.cc + the setting_* the path reads
    int wG[LDSO_PYR_LEVELS], hG[LDSO_PYR_LEVELS];
    float wM3G, hM3G;
    int pyrLevelsUsed;
    ldso_settings_t s;
    float setting_minIdepthH_marg = 50;          // Setting.cc:26
    int setting_minGoodActiveResForMarg = 3;     // Setting.cc:55
    int setting_minGoodResForMarg = 4;           // Setting.cc:56
};

// ===================================================================================================
// GlobalFuncs.h:89-103  getInterpolatedElement33 (Vec3f AoS image)
// ===================================================================================================
inline Vec3f getInterpolatedElement33(const float *mat, const float x, const float y, const int width) {
    int ix = (int) x;
    int iy = (int) y;
    float dx = x - ix;
    float dy = y - iy;
    float dxdy = dx * dy;
    const float *bp = mat + 3 * (ix + iy * width);
    Vec3f r;
    for (int c = 0; c < 3; c++)
        r[c] = dxdy * bp[3 * (1 + width) + c] + (dy - dxdy) * bp[3 * width + c] + (dx - dxdy) * bp[3 + c] + (1 - dx - dy + dxdy) * bp[c];
    return r;
}


// ---- IndexThreadReduce.h:26-169 ---------------------------------------------------------------
struct IndexThreadReduce {
    typedef std::function<void(int, int, Vec10 *, int)> Fn;
    Vec10 stats;
    IndexThreadReduce() {
        for (int i = 0; i < NUM_THREADS; i++) { isDone[i] = false; gotOne[i] = true; workerThreads[i] = std::thread(&IndexThreadReduce::workerLoop, this, i); }
    }
    ~IndexThreadReduce() {
        running = false;
        { std::unique_lock<std::mutex> lock(exMutex); todo_signal.notify_all(); }
        for (int i = 0; i < NUM_THREADS; i++) workerThreads[i].join();
    }
    void reduce(Fn fn, int first, int end, int stepSize_ = 0) {
        stats.setZero();
        if (stepSize_ == 0) stepSize_ = ((end - first) + NUM_THREADS - 1) / NUM_THREADS;
        std::unique_lock<std::mutex> lock(exMutex);
        callPerIndex = fn; nextIndex = first; maxIndex = end; stepSize = stepSize_;
        for (int i = 0; i < NUM_THREADS; i++) { isDone[i] = false; gotOne[i] = false; }
        todo_signal.notify_all();
        while (true) {
            done_signal.wait(lock);
            bool allDone = true;
            for (int i = 0; i < NUM_THREADS; i++) allDone = allDone && isDone[i];
            if (allDone) break;
        }
        nextIndex = 0; maxIndex = 0; callPerIndex = nullptr;
    }
private:
    std::thread workerThreads[NUM_THREADS];
    bool isDone[NUM_THREADS], gotOne[NUM_THREADS];
    std::mutex exMutex;
    std::condition_variable todo_signal, done_signal;
    int nextIndex = 0, maxIndex = 0, stepSize = 1;
    bool running = true;
    Fn callPerIndex;
    void workerLoop(int idx) {
        std::unique_lock<std::mutex> lock(exMutex);
        while (running) {
            int todo = 0; bool gotSomething = false;
            if (nextIndex < maxIndex) { todo = nextIndex; nextIndex += stepSize; gotSomething = true; }
            if (gotSomething) {
                lock.unlock();
                Vec10 s; callPerIndex(todo, std::min(todo + stepSize, maxIndex), &s, idx);
                gotOne[idx] = true;
                lock.lock();
                stats += s;
            } else {
                if (!gotOne[idx]) {
                    lock.unlock();
                    Vec10 s; callPerIndex(0, 0, &s, idx);
                    gotOne[idx] = true;
                    lock.lock();
                    stats += s;
                }
                isDone[idx] = true;
                done_signal.notify_all();
                todo_signal.wait(lock);
            }
        }
    }
};

// ---- AffLight.h:27-35 ---------------------------------------------------------------------------
struct AffLight {
    float a = 0, b = 0;
    AffLight() {}
    AffLight(float a_, float b_) : a(a_), b(b_) {}
    static Vec2 fromToVecExposure(float exposureF, float exposureT, AffLight g2F, AffLight g2T) {
        if (exposureF == 0 || exposureT == 0) { exposureT = exposureF = 1; }
        float a = std::exp(g2T.a - g2F.a) * exposureT / exposureF;   // float overload of exp, as in the reference
        float b = g2T.b - a * g2F.b;
        Vec2 r; r[0] = a; r[1] = b;
        return r;
    }
};

// ---- CalibHessian.h:16-140 ----------------------------------------------------------------------
struct CalibHessian {
    VecC value_zero, value_scaled, value, step, step_backup, value_backup, value_minus_value_zero;
    VecCf value_scaledf, value_scaledi;
    float &fxl() { return value_scaledf[0]; }
    float &fyl() { return value_scaledf[1]; }
    float &cxl() { return value_scaledf[2]; }
    float &cyl() { return value_scaledf[3]; }
    float &fxli() { return value_scaledi[0]; }
    float &fyli() { return value_scaledi[1]; }
    float &cxli() { return value_scaledi[2]; }
    float &cyli() { return value_scaledi[3]; }
    void setValue(const VecC &v) {
        value = v;
        value_scaled[0] = SCALE_F * value[0]; value_scaled[1] = SCALE_F * value[1];
        value_scaled[2] = SCALE_C * value[2]; value_scaled[3] = SCALE_C * value[3];
        value_scaledf = value_scaled.cast<float>();
        value_scaledi[0] = 1.0f / value_scaledf[0];
        value_scaledi[1] = 1.0f / value_scaledf[1];
        value_scaledi[2] = -value_scaledf[2] / value_scaledf[0];
        value_scaledi[3] = -value_scaledf[3] / value_scaledf[1];
        value_minus_value_zero = value - value_zero;
    }
};

struct FrameHessian;
struct PointHessian;
struct PointFrameResidual;
struct EnergyFunctional;

// ---- FrameFramePrecalc.h:22-45 ----------------------------------------------------------------
struct FrameFramePrecalc {
    Mat33f PRE_RTll, PRE_KRKiTll, PRE_RKiTll, PRE_RTll_0;
    Vec2f PRE_aff_mode;
    float PRE_b0_mode = 0;
    Vec3f PRE_tTll, PRE_KtTll, PRE_tTll_0;
    float distanceLL = 0;
    void Set(FrameHessian *host, FrameHessian *target, CalibHessian *HCalib);
};

// ---- FrameHessian.h:27-214 -----------------------------------------------------------------------
struct FrameHessian {
    int frameID = 0;
    const float *dIp[LDSO_PYR_LEVELS] = {0};   // Vec3f AoS per level
    const float *dI = nullptr;
    float frameEnergyTH = 8 * 8 * patternNum;
    float ab_exposure = 0;
    bool flaggedForMarginalization = false;
    Mat66 nullspaces_pose;
    Mat42 nullspaces_affine;
    Vec6 nullspaces_scale;
    SE3 worldToCam_evalPT;
    Vec10 state, step, step_backup, state_backup, state_zero, state_scaled;
    SE3 PRE_worldToCam, PRE_camToWorld;
    std::vector<FrameFramePrecalc> targetPrecalc;
    Vec8 prior, delta_prior, delta;
    Vec10 priorFull;
    int idx = 0;
    std::vector<PointHessian *> features;   // hosted points (all statuses), frame->features order

    const Vec10 get_state_minus_stateZero() const { return state - state_zero; }
    Vec6 w2c_leftEps() const { Vec6 r; for (int i = 0; i < 6; i++) r[i] = state_scaled[i]; return r; }
    AffLight aff_g2l() const { return AffLight((float) state_scaled[6], (float) state_scaled[7]); }
    AffLight aff_g2l_0() const { return AffLight((float) (state_zero[6] * SCALE_A), (float) (state_zero[7] * SCALE_B)); }
    void setState(const Vec10 &st) {   // FrameHessian.h:78-90
        state = st;
        for (int i = 0; i < 3; i++) state_scaled[i] = SCALE_XI_TRANS * state[i];
        for (int i = 3; i < 6; i++) state_scaled[i] = SCALE_XI_ROT * state[i];
        state_scaled[6] = SCALE_A * state[6]; state_scaled[7] = SCALE_B * state[7];
        state_scaled[8] = SCALE_A * state[8]; state_scaled[9] = SCALE_B * state[9];
        PRE_worldToCam = SE3::exp(w2c_leftEps()) * worldToCam_evalPT;
        PRE_camToWorld = PRE_worldToCam.inverse();
    }
    void setStateZero(const Vec10 &sz);   // FrameHessian.cc:12-42 (nullspaces)
    void setEvalPT(const SE3 &w2c, const Vec10 &st) { worldToCam_evalPT = w2c; setState(st); setStateZero(st); }
    void takeData() {   // FrameHessian.cc:115-119
        for (int i = 0; i < 8; i++) { prior[i] = priorFull[i]; delta[i] = state[i] - state_zero[i]; delta_prior[i] = state[i]; }
    }
};

// ---- RawResidualJacobian.h:13-39 -----------------------------------------------------------------
struct RawResidualJacobian {
    VecNRf resF;
    Vec6f Jpdxi[2];
    VecCf Jpdc[2];
    Vec2f Jpdd;
    VecNRf JIdx[2];
    VecNRf JabF[2];
    Mat22f JIdx2, JabJIdx, Jab2;
};

// ---- PointHessian.h:19-132 -------------------------------------------------------------------------
struct PointHessian {
    int status = PS_ACTIVE;          // Point::PointStatus of the owning map point
    int flatIndex = -1;              // index in the caller's point array
    float u = 0, v = 0;
    bool hasDepthPrior = false;
    float idepth_scaled = 0, idepth_zero_scaled = 0, idepth_zero = 0, idepth = 0, step = 0, step_backup = 0, idepth_backup = 0;
    float nullspaces_scale = 0, idepth_hessian = 0, maxRelBaseline = 0;
    int numGoodResiduals = 0;
    std::vector<PointFrameResidual *> residuals;
    std::pair<PointFrameResidual *, ResState> lastResiduals[2];
    float color[8], weights[8];
    float priorF = 0, deltaF = 0;
    float bdSumF = 0, HdiF = 0, Hdd_accLF = 0, bd_accLF = 0, Hdd_accAF = 0, bd_accAF = 0;
    VecCf Hcd_accLF, Hcd_accAF;
    bool alreadyRemoved = false;
    FrameHessian *hostFrame = nullptr;

    PointHessian() { lastResiduals[0] = {nullptr, OOB}; lastResiduals[1] = {nullptr, OOB}; }
    void setIdepth(float id) { idepth = id; idepth_scaled = SCALE_IDEPTH * id; }
    void setIdepthZero(float id) { idepth_zero = id; idepth_zero_scaled = SCALE_IDEPTH * id; nullspaces_scale = -(id * 1.001 - id / 1.001) * 500; }
    bool isOOB(const std::vector<FrameHessian *> &toMarg, const Globals &g) const;   // PointHessian.h:53-73
    bool isInlierNew(const Globals &g) const {
        return (int) residuals.size() >= g.setting_minGoodActiveResForMarg && numGoodResiduals >= g.setting_minGoodResForMarg;
    }
};

// ---- Residuals.h:40-130 ---------------------------------------------------------------------------
struct PointFrameResidual {
    int flatIndex = -1;
    ResState state_state = OUTLIER;
    double state_energy = 0;
    ResState state_NewState = OUTLIER;
    double state_NewEnergy = 0;
    double state_NewEnergyWithOutlier = 0;
    PointHessian *point = nullptr;
    FrameHessian *host = nullptr, *target = nullptr;
    RawResidualJacobian J;
    bool isNew = true;
    Vec2f projectedTo[8];
    Vec3f centerProjectedTo;
    int hostIDX = 0, targetIDX = 0;
    VecNRf res_toZeroF;
    Vec8f JpJdF;
    bool isLinearized = false;
    bool isActiveAndIsGoodNEW = false;

    bool isActive() const { return isActiveAndIsGoodNEW; }
    void setState(ResState s) { state_state = s; }
    void resetOOB() { state_NewEnergy = state_energy = 0; state_NewState = OUTLIER; setState(IN); }
    double linearize(CalibHessian *HCalib, const Globals &g);
    void applyRes(bool copyJacobians) {
        if (copyJacobians) {
            if (state_state == OOB) return;
            if (state_NewState == IN) { isActiveAndIsGoodNEW = true; takeData(); }
            else isActiveAndIsGoodNEW = false;
        }
        state_state = state_NewState;
        state_energy = state_NewEnergy;
    }
    void takeData() {
        Vec2f JI_JI_Jd = J.JIdx2 * J.Jpdd;
        for (int i = 0; i < 6; i++) JpJdF[i] = J.Jpdxi[0][i] * JI_JI_Jd[0] + J.Jpdxi[1][i] * JI_JI_Jd[1];
        Vec2f t = J.JabJIdx * J.Jpdd;
        JpJdF[6] = t[0]; JpJdF[7] = t[1];
    }
    void fixLinearizationF(EnergyFunctional *ef);
};

// ---- AccumulatedTopHessian.h / .cc -------------------------------------------------------------------
struct AccumulatedTopHessianSSE {
    int nframes[NUM_THREADS];
    AccumulatorApprox *acc[NUM_THREADS];
    int nres[NUM_THREADS];
    AccumulatedTopHessianSSE() { for (int t = 0; t < NUM_THREADS; t++) { nres[t] = 0; acc[t] = 0; nframes[t] = 0; } }
    ~AccumulatedTopHessianSSE() { for (int t = 0; t < NUM_THREADS; t++) if (acc[t]) delete[] acc[t]; }
    void setZero(int nFrames, int min = 0, int max = 1, Vec10 *stats = 0, int tid = 0) {
        if (nFrames != nframes[tid]) { if (acc[tid]) delete[] acc[tid]; acc[tid] = new AccumulatorApprox[nFrames * nFrames]; }
        for (int i = 0; i < nFrames * nFrames; i++) acc[tid][i].initialize();
        nframes[tid] = nFrames; nres[tid] = 0;
    }
    template <int mode> void addPoint(PointHessian *p, EnergyFunctional const *const ef, int tid = 0);
    template <int mode> void addPointsInternal(std::vector<PointHessian *> *points, EnergyFunctional const *const ef, int min = 0, int max = 1, Vec10 *stats = 0, int tid = 0) {
        for (int i = min; i < max; i++) addPoint<mode>((*points)[i], ef, tid);
    }
    void stitchDouble(MatXX &H, VecX &b, EnergyFunctional const *const EF, bool usePrior, bool useDelta, int tid = 0);
    void stitchDoubleMT(IndexThreadReduce *red, MatXX &H, VecX &b, EnergyFunctional const *const EF, bool usePrior, bool MT);
    void stitchDoubleInternal(MatXX *H, VecX *b, EnergyFunctional const *const EF, bool usePrior, int min, int max, Vec10 *stats, int tid);
};

// ---- AccumulatedSCHessian.h / .cc --------------------------------------------------------------------
struct AccumulatedSCHessianSSE {
    AccumulatorXX<8, CPARS> *accE[NUM_THREADS];
    AccumulatorX<8> *accEB[NUM_THREADS];
    AccumulatorXX<8, 8> *accD[NUM_THREADS];
    AccumulatorXX<CPARS, CPARS> accHcc[NUM_THREADS];
    AccumulatorX<CPARS> accbc[NUM_THREADS];
    int nframes[NUM_THREADS];
    AccumulatedSCHessianSSE() { for (int i = 0; i < NUM_THREADS; i++) { accE[i] = 0; accEB[i] = 0; accD[i] = 0; nframes[i] = 0; } }
    ~AccumulatedSCHessianSSE() { for (int i = 0; i < NUM_THREADS; i++) { if (accE[i]) delete[] accE[i]; if (accEB[i]) delete[] accEB[i]; if (accD[i]) delete[] accD[i]; } }
    void setZero(int n, int min = 0, int max = 1, Vec10 *stats = 0, int tid = 0) {
        if (n != nframes[tid]) {
            if (accE[tid]) delete[] accE[tid];
            if (accEB[tid]) delete[] accEB[tid];
            if (accD[tid]) delete[] accD[tid];
            accE[tid] = new AccumulatorXX<8, CPARS>[n * n];
            accEB[tid] = new AccumulatorX<8>[n * n];
            accD[tid] = new AccumulatorXX<8, 8>[n * n * n];
        }
        accbc[tid].initialize(); accHcc[tid].initialize();
        for (int i = 0; i < n * n; i++) { accE[tid][i].initialize(); accEB[tid][i].initialize(); for (int j = 0; j < n; j++) accD[tid][i * n + j].initialize(); }
        nframes[tid] = n;
    }
    void addPoint(PointHessian *p, bool shiftPriorToZero, int tid = 0);
    void addPointsInternal(std::vector<PointHessian *> *points, bool shiftPriorToZero, int min = 0, int max = 1, Vec10 *stats = 0, int tid = 0) {
        for (int i = min; i < max; i++) addPoint((*points)[i], shiftPriorToZero, tid);
    }
    void stitchDouble(MatXX &H_sc, VecX &b_sc, const EnergyFunctional *const EF, int tid = 0);
    void stitchDoubleMT(IndexThreadReduce *red, MatXX &H, VecX &b, EnergyFunctional const *const EF, bool MT);
    void stitchDoubleInternal(MatXX *H, VecX *b, EnergyFunctional const *const EF, int min, int max, Vec10 *stats, int tid);
};

// ---- EnergyFunctional.h:54-234 ---------------------------------------------------------------------
struct EnergyFunctional {
    const Globals *g = nullptr;
    bool multiThreading = false;
    IndexThreadReduce *red = nullptr;
    std::vector<FrameHessian *> frames;
    int nPoints = 0, nFrames = 0, nResiduals = 0;
    MatXX HM;
    VecX bM;
    int resInA = 0, resInL = 0, resInM = 0;
    MatXX lastHS;
    VecX lastbS, lastX;
    std::vector<VecX> lastNullspaces_forLogging, lastNullspaces_pose, lastNullspaces_scale, lastNullspaces_affA, lastNullspaces_affB;
    std::vector<Mat88> adHost, adTarget;
    std::vector<Mat88f> adHostF, adTargetF;
    std::vector<Mat18f> adHTdeltaF;
    VecC cPrior;
    VecCf cDeltaF, cPriorF;
    AccumulatedTopHessianSSE *accSSE_top_L, *accSSE_top_A;
    AccumulatedSCHessianSSE *accSSE_bot;
    std::vector<PointHessian *> allPoints, allPointsToMarg;
    float currentLambda = 0;
    bool EFAdjointsValid = false, EFIndicesValid = false, EFDeltaValid = false;
    // kept for inspection by the tests (not in the reference): the three stitched systems of the last solve
    MatXX last_HA, last_HL, last_Hsc, last_HFinal;
    VecX last_bA, last_bL, last_bsc, last_bFinal;

    EnergyFunctional() : accSSE_top_L(new AccumulatedTopHessianSSE), accSSE_top_A(new AccumulatedTopHessianSSE), accSSE_bot(new AccumulatedSCHessianSSE) {}
    ~EnergyFunctional() { delete accSSE_top_L; delete accSSE_top_A; delete accSSE_bot; }

    void insertResidual(PointFrameResidual *r) { r->takeData(); nResiduals++; }
    void insertFrame(FrameHessian *fh, CalibHessian *Hcalib);
    void dropResidual(PointFrameResidual *r);
    void marginalizeFrame(FrameHessian *fh);
    void removePoint(PointHessian *ph);
    void marginalizePointsF();
    void dropPointsF();
    void solveSystemF(int iteration, double lambda, CalibHessian *HCalib);
    double calcMEnergyF();
    double calcLEnergyF_MT();
    void makeIDX();
    void setDeltaF(CalibHessian *HCalib);
    void setAdjointsF(CalibHessian *Hcalib);
    VecX getStitchedDeltaF() const {
        VecX d(CPARS + nFrames * 8);
        for (int i = 0; i < CPARS; i++) d[i] = (double) cDeltaF[i];
        for (int h = 0; h < nFrames; h++) for (int i = 0; i < 8; i++) d[CPARS + 8 * h + i] = frames[h]->delta[i];
        return d;
    }
    void resubstituteF_MT(const VecX &x, CalibHessian *HCalib, bool MT);
    void resubstituteFPt(const VecCf &xc, Mat18f *xAd, int min, int max, Vec10 *stats, int tid);
    void accumulateAF_MT(MatXX &H, VecX &b, bool MT);
    void accumulateLF_MT(MatXX &H, VecX &b, bool MT);
    void accumulateSCF_MT(MatXX &H, VecX &b, bool MT);
    void calcLEnergyPt(int min, int max, Vec10 *stats, int tid);
    void orthogonalize(VecX *b, MatXX *H);
};

// ---- the optimisation slice of FullSystem (FullSystem.cc:725-864, 1208-1270, 1423-1793) ---------------
struct FullSystem {
    Globals g;
    bool multiThreading = false;
    IndexThreadReduce *threadReduce = nullptr;
    std::vector<FrameHessian *> frames;
    CalibHessian Hcalib;
    EnergyFunctional *ef = nullptr;
    std::vector<PointFrameResidual *> activeResiduals;
    std::vector<float> allResVec;
    bool isLost = false;
    // ownership
    std::vector<std::unique_ptr<FrameHessian>> ownFrames;
    std::vector<std::unique_ptr<PointHessian>> ownPoints;
    std::vector<std::unique_ptr<PointFrameResidual>> ownResiduals;
    // test instrumentation: energy after every linearizeAll inside optimize()
    std::vector<double> energyLog;
    bool forceAllIterations = false;

    FullSystem() { ef = new EnergyFunctional(); }
    ~FullSystem() { delete ef; if (threadReduce) delete threadReduce; }
    void enableMT() { multiThreading = true; threadReduce = new IndexThreadReduce(); ef->multiThreading = true; ef->red = threadReduce; }

    void collectActiveResiduals();
    float optimize(int mnumOptIts);
    void setPrecalcValues();
    void solveSystem(int iteration, double lambda);
    Vec3 linearizeAll(bool fixLinearization);
    void linearizeAll_Reductor(bool fixLinearization, std::vector<PointFrameResidual *> *toRemove, int min, int max, Vec10 *stats, int tid);
    bool doStepFromBackup(float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD);
    void backupState(bool backupLastStep);
    void loadSateBackup();
    double calcLEnergy() { if (g.s.forceAcceptStep) return 0; return ef->calcLEnergyF_MT(); }
    double calcMEnergy() { if (g.s.forceAcceptStep) return 0; return ef->calcMEnergyF(); }
    void applyRes_Reductor(bool copyJacobians, int min, int max, Vec10 *stats, int tid) { for (int k = min; k < max; k++) activeResiduals[k]->applyRes(true); }
    std::vector<VecX> getNullspaces(std::vector<VecX> &nullspaces_pose, std::vector<VecX> &nullspaces_scale, std::vector<VecX> &nullspaces_affA, std::vector<VecX> &nullspaces_affB);
    void setNewFrameEnergyTH();
    void flagPointsForRemoval();
};

}  // namespace orc
