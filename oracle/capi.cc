// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h / backend.h headers).  Pinned to reference-compiled code by tests/test_ref_pin.py.
// capi.cc — C entry points so that tests/ (ctypes) can drive the CPU restatement with the same
// flattened window images (include/ldso_window.h) that the HIP library consumes.
#include "backend.h"
#include "tracker.h"
#include <chrono>

using namespace orc;

namespace {

struct OrcWindow {
    FullSystem fs;
    std::vector<PointHessian *> pointByFlat;          // flat point index -> object
    std::vector<PointFrameResidual *> resByFlat;      // flat residual index -> object
    std::vector<std::vector<float>> imageStore;       // owned copies, [frame*levels + lvl]
    int levels = 0;
};

static void fill_globals(Globals &g, int w, int h, int levels, const ldso_settings_t *s) {
    g.s = *s;
    g.pyrLevelsUsed = levels;
    for (int l = 0; l < LDSO_PYR_LEVELS; l++) { g.wG[l] = w >> l; g.hG[l] = h >> l; }
    g.wM3G = w - 3;
    g.hM3G = h - 3;
}

}  // namespace

extern "C" {

// images: F*levels pointers, index [f*levels + lvl], each Vec3f AoS of (w>>lvl)*(h>>lvl) pixels.
void *orc_create(int w, int h, int levels, const ldso_settings_t *settings, const ldso_calib_t *calib,
                 int F, const ldso_frame_t *frames, const float *const *images,
                 int P, const ldso_point_t *points, int R, const ldso_residual_t *residuals,
                 const ldso_rawjac_t *linJ, const float *lin_res_toZeroF,
                 const double *HM, const double *bM, int multithreading) {
    OrcWindow *W = new OrcWindow();
    FullSystem &fs = W->fs;
    fill_globals(fs.g, w, h, levels, settings);
    fs.ef->g = &fs.g;
    W->levels = levels;
    if (multithreading) fs.enableMT();

    VecC v0; for (int i = 0; i < 4; i++) { fs.Hcalib.value_zero[i] = calib->value_zero[i]; v0[i] = calib->value[i]; }
    fs.Hcalib.setValue(v0);

    W->imageStore.resize((size_t) F * levels);
    for (int f = 0; f < F; f++) {
        fs.ownFrames.emplace_back(new FrameHessian());
        FrameHessian *fh = fs.ownFrames.back().get();
        const ldso_frame_t &in = frames[f];
        fh->frameID = in.frameID;
        fh->ab_exposure = in.ab_exposure;
        fh->frameEnergyTH = in.frameEnergyTH;
        for (int l = 0; l < levels; l++) {
            size_t n = (size_t) (w >> l) * (h >> l) * 3;
            const float *src = images[f * levels + l];
            if (src) { W->imageStore[f * levels + l].assign(src, src + n); fh->dIp[l] = W->imageStore[f * levels + l].data(); }
        }
        fh->dI = fh->dIp[0];
        fh->worldToCam_evalPT = SE3::fromMatrix34(in.worldToCam_evalPT);
        Vec10 st, sz;
        for (int i = 0; i < 10; i++) { st[i] = in.state[i]; sz[i] = in.state_zero[i]; }
        fh->state_zero = sz;
        fh->setState(st);
        for (int i = 0; i < 8; i++) fh->priorFull[i] = in.prior[i];
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) fh->nullspaces_pose(r, c) = in.nullspaces_pose[r * 6 + c];
        for (int r = 0; r < 6; r++) fh->nullspaces_scale[r] = in.nullspaces_scale[r];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 2; c++) fh->nullspaces_affine(r, c) = in.nullspaces_affine[r * 2 + c];
        fs.frames.push_back(fh);
    }
    for (FrameHessian *fh : fs.frames) fs.ef->insertFrame(fh, &fs.Hcalib);

    W->pointByFlat.resize(P);
    W->resByFlat.resize(R);
    for (int k = 0; k < P; k++) {
        fs.ownPoints.emplace_back(new PointHessian());
        PointHessian *p = fs.ownPoints.back().get();
        const ldso_point_t &in = points[k];
        p->flatIndex = k;
        p->u = in.u; p->v = in.v;
        p->setIdepth(in.idepth);
        p->setIdepthZero(in.idepth_zero);
        memcpy(p->color, in.color, sizeof(p->color));
        memcpy(p->weights, in.weights, sizeof(p->weights));
        p->priorF = in.priorF;
        p->hasDepthPrior = in.priorF != 0;
        p->deltaF = p->idepth - p->idepth_zero;
        p->hostFrame = fs.frames[in.host];
        p->hostFrame->features.push_back(p);
        W->pointByFlat[k] = p;
        fs.ef->nPoints++;
        for (int j = 0; j < in.res_count; j++) {
            int ri = in.res_begin + j;
            const ldso_residual_t &rin = residuals[ri];
            fs.ownResiduals.emplace_back(new PointFrameResidual());
            PointFrameResidual *r = fs.ownResiduals.back().get();
            r->flatIndex = ri;
            r->point = p;
            r->host = fs.frames[rin.host];
            r->target = fs.frames[rin.target];
            r->resetOOB();
            r->state_state = (ResState) rin.state_state;
            r->state_energy = rin.state_energy;
            r->isLinearized = rin.is_linearized != 0;
            r->isActiveAndIsGoodNEW = rin.is_active != 0;
            r->isNew = rin.is_new != 0;
            if (r->isLinearized && linJ) {
                const ldso_rawjac_t &j74 = linJ[ri];
                for (int i = 0; i < 8; i++) { r->J.resF[i] = j74.resF[i]; r->J.JIdx[0][i] = j74.JIdx[0][i]; r->J.JIdx[1][i] = j74.JIdx[1][i]; r->J.JabF[0][i] = j74.JabF[0][i]; r->J.JabF[1][i] = j74.JabF[1][i]; }
                for (int i = 0; i < 6; i++) { r->J.Jpdxi[0][i] = j74.Jpdxi[0][i]; r->J.Jpdxi[1][i] = j74.Jpdxi[1][i]; }
                for (int i = 0; i < 4; i++) { r->J.Jpdc[0][i] = j74.Jpdc[0][i]; r->J.Jpdc[1][i] = j74.Jpdc[1][i]; r->J.JIdx2[i] = j74.JIdx2[i]; r->J.JabJIdx[i] = j74.JabJIdx[i]; r->J.Jab2[i] = j74.Jab2[i]; }
                r->J.Jpdd[0] = j74.Jpdd[0]; r->J.Jpdd[1] = j74.Jpdd[1];
                for (int i = 0; i < 8; i++) r->res_toZeroF[i] = lin_res_toZeroF[ri * 8 + i];
                r->takeData();
            }
            p->residuals.push_back(r);
            W->resByFlat[ri] = r;
            fs.ef->nResiduals++;
        }
        // lastResiduals: residuals to the newest / second-newest frame (FullSystem.cc:446-469 sets these)
        for (auto r : p->residuals) {
            if (r->target == fs.frames.back()) p->lastResiduals[0] = {r, IN};
            else if (F >= 2 && r->target == fs.frames[F - 2]) p->lastResiduals[1] = {r, IN};
        }
    }
    int n = 8 * F + 4;
    if (HM) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) fs.ef->HM(i, j) = HM[i * n + j];
    if (bM) for (int i = 0; i < n; i++) fs.ef->bM[i] = bM[i];
    fs.ef->makeIDX();
    fs.setPrecalcValues();
    return W;
}

void orc_destroy(void *h) { delete (OrcWindow *) h; }

void orc_set_force_all_iterations(void *h, int v) { ((OrcWindow *) h)->fs.forceAllIterations = v != 0; }

void orc_collect_active(void *h) { ((OrcWindow *) h)->fs.collectActiveResiduals(); }
// the activeResiduals list of collectActiveResiduals WITHOUT its resetOOB: continue from a transplanted mid-optimisation state
void orc_collect_active_keep_states(void *h) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    fs.activeResiduals.clear();
    for (FrameHessian *fr : fs.frames)
        for (PointHessian *ph : fr->features)
            if (ph->status == PS_ACTIVE && !ph->alreadyRemoved)
                for (auto &r : ph->residuals)
                    if (!r->isLinearized) fs.activeResiduals.push_back(r);
}

double orc_linearize_all(void *h, int fix) { return ((OrcWindow *) h)->fs.linearizeAll(fix != 0)[0]; }

void orc_apply_res(void *h) { FullSystem &fs = ((OrcWindow *) h)->fs; fs.applyRes_Reductor(true, 0, fs.activeResiduals.size(), 0, 0); }

void orc_set_precalc(void *h) { ((OrcWindow *) h)->fs.setPrecalcValues(); }

void orc_backup_state(void *h) { ((OrcWindow *) h)->fs.backupState(false); }

int orc_do_step(void *h) { return ((OrcWindow *) h)->fs.doStepFromBackup(1, 1, 1, 1, 1) ? 1 : 0; }

void orc_solve_system(void *h, int iteration, double lambda) { ((OrcWindow *) h)->fs.solveSystem(iteration, lambda); }

float orc_optimize(void *h, int niters) { return ((OrcWindow *) h)->fs.optimize(niters); }
int orc_is_lost(void *h) { return ((OrcWindow *) h)->fs.isLost ? 1 : 0; }      // FullSystem::isLost after optimize (FullSystem.cc:853-857)

int orc_energy_log(void *h, double *out, int cap) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    int n = std::min<int>(cap, fs.energyLog.size());
    for (int i = 0; i < n; i++) out[i] = fs.energyLog[i];
    return fs.energyLog.size();
}

// EnergyFunctional::calcMEnergyF / calcLEnergyF_MT (EnergyFunctional.cc:353-378) regardless of setting_forceAceptStep
void orc_calc_lm_energies(void *h, double *EM, double *EL) {
    EnergyFunctional *ef = ((OrcWindow *) h)->fs.ef;
    *EM = ef->calcMEnergyF();
    *EL = ef->calcLEnergyF_MT();
}

int orc_num_frames(void *h) { return ((OrcWindow *) h)->fs.ef->nFrames; }
int orc_num_active_residuals(void *h) { return ((OrcWindow *) h)->fs.activeResiduals.size(); }
int orc_num_all_points(void *h) { return ((OrcWindow *) h)->fs.ef->allPoints.size(); }
void orc_counts(void *h, int *resInA, int *resInL, int *resInM) { auto ef = ((OrcWindow *) h)->fs.ef; *resInA = ef->resInA; *resInL = ef->resInL; *resInM = ef->resInM; }

// per-residual outputs in FLAT order. J may be NULL.
void orc_get_residuals(void *h, ldso_res_out_t *out, ldso_rawjac_t *J, int32_t *state_state, int32_t *is_active, float *res_toZeroF, int32_t *is_linearized, int32_t *alive) {
    OrcWindow *W = (OrcWindow *) h;
    std::vector<char> live(W->resByFlat.size(), 0);
    for (auto &up : W->fs.ownPoints) for (auto r : up->residuals) live[r->flatIndex] = 1;
    for (size_t i = 0; i < W->resByFlat.size(); i++) {
        PointFrameResidual *r = W->resByFlat[i];
        if (out) {
            out[i].state_NewEnergy = (float) r->state_NewEnergy;
            out[i].state_NewEnergyWithOutlier = (float) r->state_NewEnergyWithOutlier;
            out[i].state_NewState = r->state_NewState;
            for (int k = 0; k < 3; k++) out[i].centerProjectedTo[k] = r->centerProjectedTo[k];
            for (int k = 0; k < 8; k++) out[i].JpJdF[k] = r->JpJdF[k];
        }
        if (J) {
            ldso_rawjac_t &j = J[i];
            for (int k = 0; k < 8; k++) { j.resF[k] = r->J.resF[k]; j.JIdx[0][k] = r->J.JIdx[0][k]; j.JIdx[1][k] = r->J.JIdx[1][k]; j.JabF[0][k] = r->J.JabF[0][k]; j.JabF[1][k] = r->J.JabF[1][k]; }
            for (int k = 0; k < 6; k++) { j.Jpdxi[0][k] = r->J.Jpdxi[0][k]; j.Jpdxi[1][k] = r->J.Jpdxi[1][k]; }
            for (int k = 0; k < 4; k++) { j.Jpdc[0][k] = r->J.Jpdc[0][k]; j.Jpdc[1][k] = r->J.Jpdc[1][k]; j.JIdx2[k] = r->J.JIdx2[k]; j.JabJIdx[k] = r->J.JabJIdx[k]; j.Jab2[k] = r->J.Jab2[k]; }
            j.Jpdd[0] = r->J.Jpdd[0]; j.Jpdd[1] = r->J.Jpdd[1];
        }
        if (state_state) state_state[i] = r->state_state;
        if (is_active) is_active[i] = r->isActiveAndIsGoodNEW ? 1 : 0;
        if (res_toZeroF) for (int k = 0; k < 8; k++) res_toZeroF[i * 8 + k] = r->res_toZeroF[k];
        if (is_linearized) is_linearized[i] = r->isLinearized ? 1 : 0;
        if (alive) alive[i] = live[i];
    }
}

void orc_get_points(void *h, ldso_point_out_t *out, int32_t *status) {
    OrcWindow *W = (OrcWindow *) h;
    for (size_t i = 0; i < W->pointByFlat.size(); i++) {
        PointHessian *p = W->pointByFlat[i];
        if (out) {
            ldso_point_out_t &o = out[i];
            o.step = p->step; o.HdiF = p->HdiF; o.bdSumF = p->bdSumF; o.idepth_hessian = p->idepth_hessian;
            o.Hdd_accAF = p->Hdd_accAF; o.bd_accAF = p->bd_accAF; o.Hdd_accLF = p->Hdd_accLF; o.bd_accLF = p->bd_accLF;
            for (int k = 0; k < 4; k++) { o.Hcd_accAF[k] = p->Hcd_accAF[k]; o.Hcd_accLF[k] = p->Hcd_accLF[k]; }
            o.idepth = p->idepth; o.maxRelBaseline = p->maxRelBaseline; o.numGoodResiduals = p->numGoodResiduals;
        }
        if (status) status[i] = p->alreadyRemoved ? 100 + p->status : p->status;
    }
}

// frames' current state/step/frameEnergyTH/evalPT in window order; calib value/step
void orc_get_frames(void *h, ldso_frame_t *out, double *step /*F*10*/, double *calib_value /*4*/, double *calib_step /*4*/, double *pre_worldToCam /*F*12*/) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    for (size_t f = 0; f < fs.frames.size(); f++) {
        FrameHessian *fh = fs.frames[f];
        if (out) {
            ldso_frame_t &o = out[f];
            Mat33 R = fh->worldToCam_evalPT.rotationMatrix();
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o.worldToCam_evalPT[i * 4 + j] = R(i, j); o.worldToCam_evalPT[i * 4 + 3] = fh->worldToCam_evalPT.t[i]; }
            for (int i = 0; i < 10; i++) { o.state[i] = fh->state[i]; o.state_zero[i] = fh->state_zero[i]; }
            for (int i = 0; i < 8; i++) o.prior[i] = fh->prior[i];
            for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) o.nullspaces_pose[r * 6 + c] = fh->nullspaces_pose(r, c);
            for (int r = 0; r < 6; r++) o.nullspaces_scale[r] = fh->nullspaces_scale[r];
            for (int r = 0; r < 4; r++) for (int c = 0; c < 2; c++) o.nullspaces_affine[r * 2 + c] = fh->nullspaces_affine(r, c);
            o.ab_exposure = fh->ab_exposure; o.frameEnergyTH = fh->frameEnergyTH; o.frameID = fh->frameID;
        }
        if (step) for (int i = 0; i < 10; i++) step[f * 10 + i] = fh->step[i];
        if (pre_worldToCam) {
            Mat33 R = fh->PRE_worldToCam.rotationMatrix();
            for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) pre_worldToCam[f * 12 + i * 4 + j] = R(i, j); pre_worldToCam[f * 12 + i * 4 + 3] = fh->PRE_worldToCam.t[i]; }
        }
    }
    if (calib_value) for (int i = 0; i < 4; i++) calib_value[i] = fs.Hcalib.value[i];
    if (calib_step) for (int i = 0; i < 4; i++) calib_step[i] = fs.Hcalib.step[i];
}

// F*F pair precalc: 9 KRKi, 3 Kt, 9 R0, 3 t0, 2 aff, 1 b0  (27 floats) at [h*F + t]
void orc_get_precalc(void *h, float *out) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    int F = fs.frames.size();
    for (int a = 0; a < F; a++)
        for (int t = 0; t < F; t++) {
            const FrameFramePrecalc &p = fs.frames[a]->targetPrecalc[t];
            float *o = out + (a * F + t) * 27;
            for (int i = 0; i < 9; i++) o[i] = p.PRE_KRKiTll.d[i];
            for (int i = 0; i < 3; i++) o[9 + i] = p.PRE_KtTll[i];
            for (int i = 0; i < 9; i++) o[12 + i] = p.PRE_RTll_0.d[i];
            for (int i = 0; i < 3; i++) o[21 + i] = p.PRE_tTll_0[i];
            o[24] = p.PRE_aff_mode[0]; o[25] = p.PRE_aff_mode[1]; o[26] = p.PRE_b0_mode;
        }
}

// adjoints: adHost/adTarget F*F*64 doubles at [h + t*F]; adHTdeltaF F*F*8 floats
void orc_get_adjoints(void *h, double *adHost, double *adTarget, float *adHTdeltaF) {
    EnergyFunctional *ef = ((OrcWindow *) h)->fs.ef;
    int n = ef->nFrames * ef->nFrames;
    for (int i = 0; i < n; i++) {
        if (adHost) memcpy(adHost + i * 64, ef->adHost[i].d, 64 * sizeof(double));
        if (adTarget) memcpy(adTarget + i * 64, ef->adTarget[i].d, 64 * sizeof(double));
        if (adHTdeltaF) memcpy(adHTdeltaF + i * 8, ef->adHTdeltaF[i].d, 8 * sizeof(float));
    }
}

// raw fp32 accumulators after the last solve (single-thread slot 0):
// topA/topL: F*F*169 (13x13 row-major) at [h + t*F]; accD F^3*64; accE F^2*32; accEB F^2*8; accHcc 16; accbc 4
void orc_get_accumulators(void *h, float *topA, float *topL, float *accD, float *accE, float *accEB, float *accHcc, float *accbc) {
    EnergyFunctional *ef = ((OrcWindow *) h)->fs.ef;
    int F = ef->nFrames;
    for (int i = 0; i < F * F; i++) {
        if (topA) { ef->accSSE_top_A->acc[0][i].finish(); memcpy(topA + i * 169, ef->accSSE_top_A->acc[0][i].H.d, 169 * 4); }
        if (topL) { ef->accSSE_top_L->acc[0][i].finish(); memcpy(topL + i * 169, ef->accSSE_top_L->acc[0][i].H.d, 169 * 4); }
        if (accE) memcpy(accE + i * 32, ef->accSSE_bot->accE[0][i].A1m.d, 32 * 4);
        if (accEB) memcpy(accEB + i * 8, ef->accSSE_bot->accEB[0][i].A1m.d, 8 * 4);
    }
    if (accD) for (int i = 0; i < F * F * F; i++) memcpy(accD + i * 64, ef->accSSE_bot->accD[0][i].A1m.d, 64 * 4);
    if (accHcc) memcpy(accHcc, ef->accSSE_bot->accHcc[0].A1m.d, 16 * 4);
    if (accbc) memcpy(accbc, ef->accSSE_bot->accbc[0].A1m.d, 4 * 4);
}

static void copyM(const MatXX &m, double *o) { if (o && m.d.size()) memcpy(o, m.d.data(), m.d.size() * sizeof(double)); }
static void copyV(const VecX &v, double *o) { if (o && v.d.size()) memcpy(o, v.d.data(), v.d.size() * sizeof(double)); }

void orc_get_system(void *h, double *HA, double *bA, double *HL, double *bL, double *Hsc, double *bsc, double *HFinal, double *bFinal, double *x, double *lastHS, double *lastbS) {
    EnergyFunctional *ef = ((OrcWindow *) h)->fs.ef;
    copyM(ef->last_HA, HA); copyV(ef->last_bA, bA); copyM(ef->last_HL, HL); copyV(ef->last_bL, bL);
    copyM(ef->last_Hsc, Hsc); copyV(ef->last_bsc, bsc); copyM(ef->last_HFinal, HFinal); copyV(ef->last_bFinal, bFinal);
    copyV(ef->lastX, x); copyM(ef->lastHS, lastHS); copyV(ef->lastbS, lastbS);
}

void orc_get_prior(void *h, double *HM, double *bM) { EnergyFunctional *ef = ((OrcWindow *) h)->fs.ef; copyM(ef->HM, HM); copyV(ef->bM, bM); }

// fixLinearizationF on selected residuals (flat ids), used to build the "mixed" windows that exercise mode 1
void orc_fix_linearization(void *h, const int32_t *ids, int n) {
    OrcWindow *W = (OrcWindow *) h;
    for (int i = 0; i < n; i++) { PointFrameResidual *r = W->resByFlat[ids[i]]; if (r->isActive()) r->fixLinearizationF(W->fs.ef); }
}

// --- marginalisation (config C5) -------------------------------------------------------------------
void orc_flag_frame(void *h, int frame_idx) { ((OrcWindow *) h)->fs.frames[frame_idx]->flaggedForMarginalization = true; }
void orc_flag_points_for_removal(void *h) { ((OrcWindow *) h)->fs.flagPointsForRemoval(); }
void orc_drop_points(void *h) { ((OrcWindow *) h)->fs.ef->dropPointsF(); }
void orc_marginalize_points(void *h) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    fs.ef->lastNullspaces_forLogging = fs.getNullspaces(fs.ef->lastNullspaces_pose, fs.ef->lastNullspaces_scale, fs.ef->lastNullspaces_affA, fs.ef->lastNullspaces_affB);
    fs.ef->marginalizePointsF();
}
// FullSystem::marginalizeFrame (FullSystem.cc:602-640): drop residuals that target the frame, then EF::marginalizeFrame
void orc_marginalize_frame(void *h, int frame_idx) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    FrameHessian *fh = fs.frames[frame_idx];
    fs.ef->marginalizeFrame(fh);
    for (FrameHessian *fr : fs.frames) {
        if (fr == fh) continue;
        for (PointHessian *ph : fr->features) {
            if (ph->alreadyRemoved || ph->status != PS_ACTIVE) continue;
            size_t n = ph->residuals.size();
            for (size_t i = 0; i < n; i++) {
                PointFrameResidual *r = ph->residuals[i];
                if (r->target == fh) {
                    if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = 0;
                    else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = 0;
                    fs.ef->dropResidual(r);
                    i--; n--;
                }
            }
        }
    }
    for (size_t i = 0; i < fs.frames.size(); i++) if (fs.frames[i] == fh) { fs.frames.erase(fs.frames.begin() + i); break; }
    for (size_t i = 0; i < fs.frames.size(); i++) fs.frames[i]->idx = i;
    fs.setPrecalcValues();
    fs.ef->setAdjointsF(&fs.Hcalib);
    fs.ef->setDeltaF(&fs.Hcalib);
}

// Export the CURRENT window back to flat arrays (after marginalisation etc.) in allPoints order.
// Returns counts via pointers. Arrays must have capacity for the original P / R.
void orc_export_window(void *h, int *F_out, int *P_out, int *R_out, ldso_point_t *points, ldso_residual_t *residuals,
                       ldso_rawjac_t *linJ, float *lin_rtz, int32_t *orig_point_index, int32_t *orig_res_index) {
    OrcWindow *W = (OrcWindow *) h;
    FullSystem &fs = W->fs;
    fs.ef->makeIDX();
    int P = 0, R = 0;
    for (PointHessian *p : fs.ef->allPoints) {
        ldso_point_t &o = points[P];
        o.u = p->u; o.v = p->v; o.idepth = p->idepth; o.idepth_zero = p->idepth_zero;
        memcpy(o.color, p->color, sizeof(o.color)); memcpy(o.weights, p->weights, sizeof(o.weights));
        o.priorF = p->priorF; o.host = p->hostFrame->idx; o.res_begin = R; o.res_count = p->residuals.size();
        if (orig_point_index) orig_point_index[P] = p->flatIndex;
        for (PointFrameResidual *r : p->residuals) {
            ldso_residual_t &ro = residuals[R];
            ro.point = P; ro.host = r->host->idx; ro.target = r->target->idx; ro.state_state = r->state_state;
            ro.is_linearized = r->isLinearized; ro.is_active = r->isActiveAndIsGoodNEW; ro.is_new = r->isNew; ro.state_energy = (float) r->state_energy;
            if (orig_res_index) orig_res_index[R] = r->flatIndex;
            if (linJ) {
                ldso_rawjac_t &j = linJ[R];
                for (int k = 0; k < 8; k++) { j.resF[k] = r->J.resF[k]; j.JIdx[0][k] = r->J.JIdx[0][k]; j.JIdx[1][k] = r->J.JIdx[1][k]; j.JabF[0][k] = r->J.JabF[0][k]; j.JabF[1][k] = r->J.JabF[1][k]; }
                for (int k = 0; k < 6; k++) { j.Jpdxi[0][k] = r->J.Jpdxi[0][k]; j.Jpdxi[1][k] = r->J.Jpdxi[1][k]; }
                for (int k = 0; k < 4; k++) { j.Jpdc[0][k] = r->J.Jpdc[0][k]; j.Jpdc[1][k] = r->J.Jpdc[1][k]; j.JIdx2[k] = r->J.JIdx2[k]; j.JabJIdx[k] = r->J.JabJIdx[k]; j.Jab2[k] = r->J.Jab2[k]; }
                j.Jpdd[0] = r->J.Jpdd[0]; j.Jpdd[1] = r->J.Jpdd[1];
            }
            if (lin_rtz) for (int k = 0; k < 8; k++) lin_rtz[R * 8 + k] = r->res_toZeroF[k];
            R++;
        }
        P++;
    }
    *F_out = fs.frames.size(); *P_out = P; *R_out = R;
}

// timing helper for bench.py's cpu_baseline leg: run `reps` optimize(niters) calls from a fresh copy of
// the state each time; returns seconds per call (median-free mean, the caller repeats).
double orc_time_optimize(void *h, int niters) {
    FullSystem &fs = ((OrcWindow *) h)->fs;
    auto t0 = std::chrono::steady_clock::now();
    fs.optimize(niters);
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// Eigen::LDLT<Mat88>::solve as the tracker uses it (`Hl.ldlt().solve(-b)`, CoarseTracker.cc:120-128): the checker of the device tracker's 8 x 8 solve
// (tests/test_tracker_gpu.py: ldso_tr_debug_solve8).  A row-major 8 x 8, b, x: 8.
void orc_ldlt_solve8(const double *A, const double *b, double *x) {
    Mat<double, 8, 8> M; Mat<double, 8, 1> v;
    for (int i = 0; i < 8; i++) { v[i] = b[i]; for (int j = 0; j < 8; j++) M(i, j) = A[i * 8 + j]; }
    const Mat<double, 8, 1> r = ldlt_solve<8>(M, v);
    for (int i = 0; i < 8; i++) x[i] = r[i];
}

}  // extern "C"

#include "tracker_capi.inc"
