// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h / backend.h headers).  Pinned to reference-compiled code by tests/test_ref_pin.py (see linalg.h).
#include "backend.h"

namespace orc {

// ===================================================================================================
// ResidualProjections.h:24-33 and :57-84
// ===================================================================================================
static inline bool projectPoint(const float &u_pt, const float &v_pt, const float &idepth, const Mat33f &KRKi, const Vec3f &Kt,
                                float &Ku, float &Kv, const Globals &g) {
    Vec3f p; p[0] = u_pt; p[1] = v_pt; p[2] = 1;
    Vec3f ptp = KRKi * p + Kt * idepth;
    Ku = ptp[0] / ptp[2];
    Kv = ptp[1] / ptp[2];
    return Ku > 1.1f && Kv > 1.1f && Ku < g.wM3G && Kv < g.hM3G;
}

static inline bool projectPoint(const float &u_pt, const float &v_pt, const float &idepth, const int &dx, const int &dy,
                                CalibHessian *HCalib, const Mat33f &R, const Vec3f &t, float &drescale, float &u, float &v,
                                float &Ku, float &Kv, Vec3f &KliP, float &new_idepth, const Globals &g) {
    KliP[0] = (u_pt + dx - HCalib->cxl()) * HCalib->fxli();
    KliP[1] = (v_pt + dy - HCalib->cyl()) * HCalib->fyli();
    KliP[2] = 1;
    Vec3f ptp = R * KliP + t * idepth;
    drescale = 1.0f / ptp[2];
    new_idepth = idepth * drescale;
    if (!(drescale > 0)) return false;
    u = ptp[0] * drescale;
    v = ptp[1] * drescale;
    Ku = u * HCalib->fxl() + HCalib->cxl();
    Kv = v * HCalib->fyl() + HCalib->cyl();
    return Ku > 1.1f && Kv > 1.1f && Ku < g.wM3G && Kv < g.hM3G;
}

// ===================================================================================================
// FrameFramePrecalc.cc:6-35
// ===================================================================================================
void FrameFramePrecalc::Set(FrameHessian *host, FrameHessian *target, CalibHessian *HCalib) {
    SE3 leftToLeft_0 = target->worldToCam_evalPT * host->worldToCam_evalPT.inverse();
    PRE_RTll_0 = leftToLeft_0.rotationMatrix().cast<float>();
    PRE_tTll_0 = leftToLeft_0.translation().cast<float>();

    SE3 leftToLeft = target->PRE_worldToCam * host->PRE_camToWorld;
    PRE_RTll = leftToLeft.rotationMatrix().cast<float>();
    PRE_tTll = leftToLeft.translation().cast<float>();
    distanceLL = (float) leftToLeft.translation().norm();

    Mat33f K;
    K(0, 0) = HCalib->fxl(); K(1, 1) = HCalib->fyl(); K(0, 2) = HCalib->cxl(); K(1, 2) = HCalib->cyl(); K(2, 2) = 1;
    Mat33f Ki = inverse3(K);
    PRE_KRKiTll = K * PRE_RTll * Ki;
    PRE_RKiTll = PRE_RTll * Ki;
    PRE_KtTll = K * PRE_tTll;

    PRE_aff_mode = AffLight::fromToVecExposure(host->ab_exposure, target->ab_exposure, host->aff_g2l(), target->aff_g2l()).cast<float>();
    PRE_b0_mode = host->aff_g2l_0().b;
}

// ===================================================================================================
// FrameHessian.cc:12-42
// ===================================================================================================
void FrameHessian::setStateZero(const Vec10 &sz) {
    state_zero = sz;
    for (int i = 0; i < 6; i++) {
        Vec6 eps; eps[i] = 1e-3;
        SE3 EepsP = SE3::exp(eps);
        SE3 EepsM = SE3::exp(-eps);
        SE3 w2c_leftEps_P_x0 = (worldToCam_evalPT * EepsP) * worldToCam_evalPT.inverse();
        SE3 w2c_leftEps_M_x0 = (worldToCam_evalPT * EepsM) * worldToCam_evalPT.inverse();
        Vec6 c = (w2c_leftEps_P_x0.log() - w2c_leftEps_M_x0.log()) * (1.0 / (2e-3));
        for (int r = 0; r < 6; r++) nullspaces_pose(r, i) = c[r];
    }
    SE3 w2c_leftEps_P_x0 = worldToCam_evalPT;
    w2c_leftEps_P_x0.translation() *= 1.00001;
    w2c_leftEps_P_x0 = w2c_leftEps_P_x0 * worldToCam_evalPT.inverse();
    SE3 w2c_leftEps_M_x0 = worldToCam_evalPT;
    w2c_leftEps_M_x0.translation() *= (1.0 / 1.00001);
    w2c_leftEps_M_x0 = w2c_leftEps_M_x0 * worldToCam_evalPT.inverse();
    nullspaces_scale = (w2c_leftEps_P_x0.log() - w2c_leftEps_M_x0.log()) * (1.0 / (2e-3));
    nullspaces_affine.setZero();
    nullspaces_affine(0, 0) = 1; nullspaces_affine(1, 0) = 0;
    nullspaces_affine(0, 1) = 0; nullspaces_affine(1, 1) = expf(aff_g2l_0().a) * ab_exposure;
}

// ===================================================================================================
// PointHessian.h:53-73
// ===================================================================================================
bool PointHessian::isOOB(const std::vector<FrameHessian *> &toMarg, const Globals &g) const {
    int visInToMarg = 0;
    for (PointFrameResidual *r : residuals) {
        if (r->state_state != IN) continue;
        for (FrameHessian *k : toMarg) if (r->target == k) visInToMarg++;
    }
    if ((int) residuals.size() >= g.setting_minGoodActiveResForMarg && numGoodResiduals > g.setting_minGoodResForMarg + 10 &&
        (int) residuals.size() - visInToMarg < g.setting_minGoodActiveResForMarg)
        return true;
    if (lastResiduals[0].second == OOB) return true;
    if (residuals.size() < 2) return false;
    if (lastResiduals[0].second == OUTLIER && lastResiduals[1].second == OUTLIER) return true;
    return false;
}

// ===================================================================================================
// Residuals.cc:13-214
// ===================================================================================================
double PointFrameResidual::linearize(CalibHessian *HCalib, const Globals &g) {
    state_NewEnergyWithOutlier = -1;
    if (state_state == OOB) { state_NewState = OOB; return state_energy; }

    FrameHessian *f = host;
    FrameHessian *ftarget = target;
    PointHessian *fPoint = point;
    FrameFramePrecalc *precalc = &(f->targetPrecalc[ftarget->idx]);

    float energyLeft = 0;
    const float *dIl = ftarget->dI;
    const Mat33f &PRE_KRKiTll = precalc->PRE_KRKiTll;
    const Vec3f &PRE_KtTll = precalc->PRE_KtTll;
    const Mat33f &PRE_RTll_0 = precalc->PRE_RTll_0;
    const Vec3f &PRE_tTll_0 = precalc->PRE_tTll_0;
    const float *const color = fPoint->color;
    const float *const weights = fPoint->weights;
    Vec2f affLL = precalc->PRE_aff_mode;
    float b0 = precalc->PRE_b0_mode;

    Vec6f d_xi_x, d_xi_y;
    VecCf d_C_x, d_C_y;
    float d_d_x, d_d_y;
    {
        float drescale, u, v, new_idepth;
        float Ku, Kv;
        Vec3f KliP;
        PointHessian *p = point;
        if (!projectPoint(p->u, p->v, p->idepth_zero_scaled, 0, 0, HCalib, PRE_RTll_0, PRE_tTll_0, drescale, u, v, Ku, Kv, KliP, new_idepth, g)) {
            state_NewState = OOB;
            return state_energy;
        }
        centerProjectedTo[0] = Ku; centerProjectedTo[1] = Kv; centerProjectedTo[2] = new_idepth;

        d_d_x = drescale * (PRE_tTll_0[0] - PRE_tTll_0[2] * u) * SCALE_IDEPTH * HCalib->fxl();
        d_d_y = drescale * (PRE_tTll_0[1] - PRE_tTll_0[2] * v) * SCALE_IDEPTH * HCalib->fyl();

        d_C_x[2] = drescale * (PRE_RTll_0(2, 0) * u - PRE_RTll_0(0, 0));
        d_C_x[3] = HCalib->fxl() * drescale * (PRE_RTll_0(2, 1) * u - PRE_RTll_0(0, 1)) * HCalib->fyli();
        d_C_x[0] = KliP[0] * d_C_x[2];
        d_C_x[1] = KliP[1] * d_C_x[3];

        d_C_y[2] = HCalib->fyl() * drescale * (PRE_RTll_0(2, 0) * v - PRE_RTll_0(1, 0)) * HCalib->fxli();
        d_C_y[3] = drescale * (PRE_RTll_0(2, 1) * v - PRE_RTll_0(1, 1));
        d_C_y[0] = KliP[0] * d_C_y[2];
        d_C_y[1] = KliP[1] * d_C_y[3];

        d_C_x[0] = (d_C_x[0] + u) * SCALE_F;
        d_C_x[1] *= SCALE_F;
        d_C_x[2] = (d_C_x[2] + 1) * SCALE_C;
        d_C_x[3] *= SCALE_C;

        d_C_y[0] *= SCALE_F;
        d_C_y[1] = (d_C_y[1] + v) * SCALE_F;
        d_C_y[2] *= SCALE_C;
        d_C_y[3] = (d_C_y[3] + 1) * SCALE_C;

        d_xi_x[0] = new_idepth * HCalib->fxl();
        d_xi_x[1] = 0;
        d_xi_x[2] = -new_idepth * u * HCalib->fxl();
        d_xi_x[3] = -u * v * HCalib->fxl();
        d_xi_x[4] = (1 + u * u) * HCalib->fxl();
        d_xi_x[5] = -v * HCalib->fxl();

        d_xi_y[0] = 0;
        d_xi_y[1] = new_idepth * HCalib->fyl();
        d_xi_y[2] = -new_idepth * v * HCalib->fyl();
        d_xi_y[3] = -(1 + v * v) * HCalib->fyl();
        d_xi_y[4] = u * v * HCalib->fyl();
        d_xi_y[5] = u * HCalib->fyl();
    }
    {
        J.Jpdxi[0] = d_xi_x; J.Jpdxi[1] = d_xi_y;
        J.Jpdc[0] = d_C_x; J.Jpdc[1] = d_C_y;
        J.Jpdd[0] = d_d_x; J.Jpdd[1] = d_d_y;
    }

    float JIdxJIdx_00 = 0, JIdxJIdx_11 = 0, JIdxJIdx_10 = 0;
    float JabJIdx_00 = 0, JabJIdx_01 = 0, JabJIdx_10 = 0, JabJIdx_11 = 0;
    float JabJab_00 = 0, JabJab_01 = 0, JabJab_11 = 0;
    float wJI2_sum = 0;

    for (int idx = 0; idx < patternNum; idx++) {
        float Ku, Kv;
        PointHessian *p = point;
        if (!projectPoint(p->u + patternP[idx][0], p->v + patternP[idx][1], p->idepth_scaled, PRE_KRKiTll, PRE_KtTll, Ku, Kv, g)) {
            state_NewState = OOB;
            return state_energy;
        }
        projectedTo[idx][0] = Ku;
        projectedTo[idx][1] = Kv;

        Vec3f hitColor = getInterpolatedElement33(dIl, Ku, Kv, g.wG[0]);
        float residual = hitColor[0] - (float) (affLL[0] * color[idx] + affLL[1]);
        float drdA = (color[idx] - b0);
        if (!std::isfinite((float) hitColor[0])) { state_NewState = OOB; return state_energy; }

        float w = sqrtf(g.s.outlierTHSumComponent / (g.s.outlierTHSumComponent + (hitColor[1] * hitColor[1] + hitColor[2] * hitColor[2])));
        w = 0.5f * (w + weights[idx]);

        float hw = fabsf(residual) < g.s.huberTH ? 1 : g.s.huberTH / fabsf(residual);
        energyLeft += w * w * hw * residual * residual * (2 - hw);
        {
            if (hw < 1) hw = sqrtf(hw);
            hw = hw * w;
            hitColor[1] *= hw;
            hitColor[2] *= hw;

            J.resF[idx] = residual * hw;
            J.JIdx[0][idx] = hitColor[1];
            J.JIdx[1][idx] = hitColor[2];
            J.JabF[0][idx] = drdA * hw;
            J.JabF[1][idx] = hw;

            JIdxJIdx_00 += hitColor[1] * hitColor[1];
            JIdxJIdx_11 += hitColor[2] * hitColor[2];
            JIdxJIdx_10 += hitColor[1] * hitColor[2];

            JabJIdx_00 += drdA * hw * hitColor[1];
            JabJIdx_01 += drdA * hw * hitColor[2];
            JabJIdx_10 += hw * hitColor[1];
            JabJIdx_11 += hw * hitColor[2];

            JabJab_00 += drdA * drdA * hw * hw;
            JabJab_01 += drdA * hw * hw;
            JabJab_11 += hw * hw;

            wJI2_sum += hw * hw * (hitColor[1] * hitColor[1] + hitColor[2] * hitColor[2]);

            if (g.s.affineOptModeA < 0) J.JabF[0][idx] = 0;
            if (g.s.affineOptModeB < 0) J.JabF[1][idx] = 0;
        }
    }

    J.JIdx2(0, 0) = JIdxJIdx_00; J.JIdx2(0, 1) = JIdxJIdx_10; J.JIdx2(1, 0) = JIdxJIdx_10; J.JIdx2(1, 1) = JIdxJIdx_11;
    J.JabJIdx(0, 0) = JabJIdx_00; J.JabJIdx(0, 1) = JabJIdx_01; J.JabJIdx(1, 0) = JabJIdx_10; J.JabJIdx(1, 1) = JabJIdx_11;
    J.Jab2(0, 0) = JabJab_00; J.Jab2(0, 1) = JabJab_01; J.Jab2(1, 0) = JabJab_01; J.Jab2(1, 1) = JabJab_11;

    state_NewEnergyWithOutlier = energyLeft;

    if (energyLeft > std::max<float>(f->frameEnergyTH, ftarget->frameEnergyTH) || wJI2_sum < 2) {
        energyLeft = std::max<float>(f->frameEnergyTH, ftarget->frameEnergyTH);
        state_NewState = OUTLIER;
    } else {
        state_NewState = IN;
    }
    state_NewEnergy = energyLeft;
    return energyLeft;
}

// Residuals.cc:216-242
void PointFrameResidual::fixLinearizationF(EnergyFunctional *ef) {
    Mat18f dp = ef->adHTdeltaF[hostIDX + ef->nFrames * targetIDX];
    float dpx = 0, dpy = 0;
    for (int i = 0; i < 6; i++) { dpx += J.Jpdxi[0][i] * dp[i]; dpy += J.Jpdxi[1][i] * dp[i]; }
    __m128 Jp_delta_x = _mm_set1_ps(dpx + J.Jpdc[0].dot(ef->cDeltaF) + J.Jpdd[0] * point->deltaF);
    __m128 Jp_delta_y = _mm_set1_ps(dpy + J.Jpdc[1].dot(ef->cDeltaF) + J.Jpdd[1] * point->deltaF);
    __m128 delta_a = _mm_set1_ps((float) (dp[6]));
    __m128 delta_b = _mm_set1_ps((float) (dp[7]));
    for (int i = 0; i < patternNum; i += 4) {
        __m128 rtz = _mm_loadu_ps(J.resF.d + i);
        rtz = _mm_sub_ps(rtz, _mm_mul_ps(_mm_loadu_ps(J.JIdx[0].d + i), Jp_delta_x));
        rtz = _mm_sub_ps(rtz, _mm_mul_ps(_mm_loadu_ps(J.JIdx[1].d + i), Jp_delta_y));
        rtz = _mm_sub_ps(rtz, _mm_mul_ps(_mm_loadu_ps(J.JabF[0].d + i), delta_a));
        rtz = _mm_sub_ps(rtz, _mm_mul_ps(_mm_loadu_ps(J.JabF[1].d + i), delta_b));
        _mm_storeu_ps(res_toZeroF.d + i, rtz);
    }
    isLinearized = true;
}

// ===================================================================================================
// AccumulatedTopHessian.cc:8-118
// ===================================================================================================
template <int mode>
void AccumulatedTopHessianSSE::addPoint(PointHessian *p, EnergyFunctional const *const ef, int tid) {
    VecCf dc = ef->cDeltaF;
    float dd = p->deltaF;
    float bd_acc = 0;
    float Hdd_acc = 0;
    VecCf Hcd_acc = VecCf::Zero();

    for (PointFrameResidual *r : p->residuals) {
        if (mode == 0) { if (r->isLinearized || !r->isActive()) continue; }
        if (mode == 1) { if (!r->isLinearized || !r->isActive()) continue; }
        if (mode == 2) { if (!r->isActive()) continue; }

        RawResidualJacobian *rJ = &r->J;
        int htIDX = r->hostIDX + r->targetIDX * nframes[tid];
        Mat18f dp = ef->adHTdeltaF[htIDX];

        VecNRf resApprox;
        if (mode == 0) resApprox = rJ->resF;
        if (mode == 2) resApprox = r->res_toZeroF;
        if (mode == 1) {
            float dpx = 0, dpy = 0;
            for (int i = 0; i < 6; i++) { dpx += rJ->Jpdxi[0][i] * dp[i]; dpy += rJ->Jpdxi[1][i] * dp[i]; }
            __m128 Jp_delta_x = _mm_set1_ps(dpx + rJ->Jpdc[0].dot(dc) + rJ->Jpdd[0] * dd);
            __m128 Jp_delta_y = _mm_set1_ps(dpy + rJ->Jpdc[1].dot(dc) + rJ->Jpdd[1] * dd);
            __m128 delta_a = _mm_set1_ps((float) (dp[6]));
            __m128 delta_b = _mm_set1_ps((float) (dp[7]));
            for (int i = 0; i < patternNum; i += 4) {
                __m128 rtz = _mm_loadu_ps(r->res_toZeroF.d + i);
                rtz = _mm_add_ps(rtz, _mm_mul_ps(_mm_loadu_ps(rJ->JIdx[0].d + i), Jp_delta_x));
                rtz = _mm_add_ps(rtz, _mm_mul_ps(_mm_loadu_ps(rJ->JIdx[1].d + i), Jp_delta_y));
                rtz = _mm_add_ps(rtz, _mm_mul_ps(_mm_loadu_ps(rJ->JabF[0].d + i), delta_a));
                rtz = _mm_add_ps(rtz, _mm_mul_ps(_mm_loadu_ps(rJ->JabF[1].d + i), delta_b));
                _mm_storeu_ps(resApprox.d + i, rtz);
            }
        }

        Vec2f JI_r, Jab_r;
        float rr = 0;
        for (int i = 0; i < patternNum; i++) {
            JI_r[0] += resApprox[i] * rJ->JIdx[0][i];
            JI_r[1] += resApprox[i] * rJ->JIdx[1][i];
            Jab_r[0] += resApprox[i] * rJ->JabF[0][i];
            Jab_r[1] += resApprox[i] * rJ->JabF[1][i];
            rr += resApprox[i] * resApprox[i];
        }

        acc[tid][htIDX].update(rJ->Jpdc[0].d, rJ->Jpdxi[0].d, rJ->Jpdc[1].d, rJ->Jpdxi[1].d, rJ->JIdx2(0, 0), rJ->JIdx2(0, 1), rJ->JIdx2(1, 1));
        acc[tid][htIDX].updateBotRight(rJ->Jab2(0, 0), rJ->Jab2(0, 1), Jab_r[0], rJ->Jab2(1, 1), Jab_r[1], rr);
        acc[tid][htIDX].updateTopRight(rJ->Jpdc[0].d, rJ->Jpdxi[0].d, rJ->Jpdc[1].d, rJ->Jpdxi[1].d,
                                       rJ->JabJIdx(0, 0), rJ->JabJIdx(0, 1), rJ->JabJIdx(1, 0), rJ->JabJIdx(1, 1), JI_r[0], JI_r[1]);

        Vec2f Ji2_Jpdd = rJ->JIdx2 * rJ->Jpdd;
        bd_acc += JI_r[0] * rJ->Jpdd[0] + JI_r[1] * rJ->Jpdd[1];
        Hdd_acc += Ji2_Jpdd.dot(rJ->Jpdd);
        Hcd_acc += rJ->Jpdc[0] * Ji2_Jpdd[0] + rJ->Jpdc[1] * Ji2_Jpdd[1];

        nres[tid]++;
    }

    if (mode == 0) { p->Hdd_accAF = Hdd_acc; p->bd_accAF = bd_acc; p->Hcd_accAF = Hcd_acc; }
    if (mode == 1 || mode == 2) { p->Hdd_accLF = Hdd_acc; p->bd_accLF = bd_acc; p->Hcd_accLF = Hcd_acc; }
    if (mode == 2) { p->Hcd_accAF.setZero(); p->Hdd_accAF = 0; p->bd_accAF = 0; }
}
template void AccumulatedTopHessianSSE::addPoint<0>(PointHessian *, EnergyFunctional const *const, int);
template void AccumulatedTopHessianSSE::addPoint<1>(PointHessian *, EnergyFunctional const *const, int);
template void AccumulatedTopHessianSSE::addPoint<2>(PointHessian *, EnergyFunctional const *const, int);

static inline Mat88 blk88(const MatPCPC &a) { Mat88 m; for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) m(i, j) = a(CPARS + i, CPARS + j); return m; }
static inline Mat8C blk8C(const MatPCPC &a) { Mat8C m; for (int i = 0; i < 8; i++) for (int j = 0; j < CPARS; j++) m(i, j) = a(CPARS + i, j); return m; }
static inline Vec8 blk81(const MatPCPC &a) { Vec8 m; for (int i = 0; i < 8; i++) m[i] = a(CPARS + i, 8 + CPARS); return m; }

static inline void top_stitch_block(MatXX &H, VecX &b, const MatPCPC &accH, const Mat88 &AH, const Mat88 &AT, int hIdx, int tIdx) {
    Mat88 A88 = blk88(accH);
    Mat8C A8C = blk8C(accH);
    Vec8 A81 = blk81(accH);
    H.addBlock<8, 8>(hIdx, hIdx, AH * A88 * AH.transpose());
    H.addBlock<8, 8>(tIdx, tIdx, AT * A88 * AT.transpose());
    H.addBlock<8, 8>(hIdx, tIdx, AH * A88 * AT.transpose());
    H.addBlock<8, CPARS>(hIdx, 0, AH * A8C);
    H.addBlock<8, CPARS>(tIdx, 0, AT * A8C);
    for (int i = 0; i < CPARS; i++) for (int j = 0; j < CPARS; j++) H(i, j) += accH(i, j);
    Vec8 bh = AH * A81, bt = AT * A81;
    for (int i = 0; i < 8; i++) { b[hIdx + i] += bh[i]; b[tIdx + i] += bt[i]; }
    for (int i = 0; i < CPARS; i++) b[i] += accH(i, 8 + CPARS);
}

static inline void top_copy_transposed(MatXX &H, int nf) {   // AccumulatedTopHessian.h:95-104 / .cc:169-180
    for (int h = 0; h < nf; h++) {
        int hIdx = CPARS + h * 8;
        for (int i = 0; i < CPARS; i++) for (int j = 0; j < 8; j++) H(i, hIdx + j) = H(hIdx + j, i);
        for (int t = h + 1; t < nf; t++) {
            int tIdx = CPARS + t * 8;
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(hIdx + i, tIdx + j) += H(tIdx + j, hIdx + i);
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H(tIdx + j, hIdx + i) = H(hIdx + i, tIdx + j);
        }
    }
}

static inline void top_add_prior(MatXX &H, VecX &b, EnergyFunctional const *const EF, int nf) {   // .cc:183-190 / :246-254
    for (int i = 0; i < CPARS; i++) { H(i, i) += EF->cPrior[i]; b[i] += EF->cPrior[i] * (double) EF->cDeltaF[i]; }
    for (int h = 0; h < nf; h++)
        for (int i = 0; i < 8; i++) {
            H(CPARS + h * 8 + i, CPARS + h * 8 + i) += EF->frames[h]->prior[i];
            b[CPARS + h * 8 + i] += EF->frames[h]->prior[i] * EF->frames[h]->delta_prior[i];
        }
}

// AccumulatedTopHessian.cc:129-191
void AccumulatedTopHessianSSE::stitchDouble(MatXX &H, VecX &b, EnergyFunctional const *const EF, bool usePrior, bool useDelta, int tid) {
    int nf = nframes[tid];
    H = MatXX::Zero(nf * 8 + CPARS, nf * 8 + CPARS);
    b = VecX::Zero(nf * 8 + CPARS);
    for (int h = 0; h < nf; h++)
        for (int t = 0; t < nf; t++) {
            int hIdx = CPARS + h * 8, tIdx = CPARS + t * 8, aidx = h + nf * t;
            acc[tid][aidx].finish();
            if (acc[tid][aidx].num == 0) continue;
            MatPCPC accH = acc[tid][aidx].H.cast<double>();
            top_stitch_block(H, b, accH, EF->adHost[aidx], EF->adTarget[aidx], hIdx, tIdx);
        }
    top_copy_transposed(H, nf);
    if (usePrior) top_add_prior(H, b, EF, nf);
}

// AccumulatedTopHessian.cc:193-255
void AccumulatedTopHessianSSE::stitchDoubleInternal(MatXX *H, VecX *b, EnergyFunctional const *const EF, bool usePrior, int min, int max, Vec10 *stats, int tid) {
    int toAggregate = NUM_THREADS;
    if (tid == -1) { toAggregate = 1; tid = 0; }
    if (min == max) return;
    for (int k = min; k < max; k++) {
        int h = k % nframes[0], t = k / nframes[0];
        int hIdx = CPARS + h * 8, tIdx = CPARS + t * 8, aidx = h + nframes[0] * t;
        MatPCPC accH = MatPCPC::Zero();
        for (int tid2 = 0; tid2 < toAggregate; tid2++) {
            acc[tid2][aidx].finish();
            if (acc[tid2][aidx].num == 0) continue;
            accH += acc[tid2][aidx].H.cast<double>();
        }
        top_stitch_block(H[tid], b[tid], accH, EF->adHost[aidx], EF->adTarget[aidx], hIdx, tIdx);
    }
    if (min == 0 && usePrior) top_add_prior(H[tid], b[tid], EF, nframes[tid]);
}

// AccumulatedTopHessian.h:64-105
void AccumulatedTopHessianSSE::stitchDoubleMT(IndexThreadReduce *red, MatXX &H, VecX &b, EnergyFunctional const *const EF, bool usePrior, bool MT) {
    int n = nframes[0] * 8 + CPARS;
    if (MT) {
        MatXX Hs[NUM_THREADS];
        VecX bs[NUM_THREADS];
        for (int i = 0; i < NUM_THREADS; i++) { Hs[i] = MatXX::Zero(n, n); bs[i] = VecX::Zero(n); }
        using namespace std::placeholders;
        red->reduce(std::bind(&AccumulatedTopHessianSSE::stitchDoubleInternal, this, Hs, bs, EF, usePrior, _1, _2, _3, _4), 0, nframes[0] * nframes[0], 0);
        H = Hs[0]; b = bs[0];
        for (int i = 1; i < NUM_THREADS; i++) { H += Hs[i]; b += bs[i]; nres[0] += nres[i]; }
    } else {
        H = MatXX::Zero(n, n); b = VecX::Zero(n);
        stitchDoubleInternal(&H, &b, EF, usePrior, 0, nframes[0] * nframes[0], 0, -1);
    }
    top_copy_transposed(H, nframes[0]);
}

// ===================================================================================================
// AccumulatedSCHessian.cc:9-51
// ===================================================================================================
void AccumulatedSCHessianSSE::addPoint(PointHessian *p, bool shiftPriorToZero, int tid) {
    int ngoodres = 0;
    for (auto r : p->residuals) if (r->isActive()) ngoodres++;
    if (ngoodres == 0) { p->HdiF = 0; p->bdSumF = 0; p->idepth_hessian = 0; p->maxRelBaseline = 0; return; }

    float H = p->Hdd_accAF + p->Hdd_accLF + p->priorF;
    if (H < 1e-10) H = 1e-10;
    p->idepth_hessian = H;
    p->HdiF = 1.0 / H;
    p->bdSumF = p->bd_accAF + p->bd_accLF;
    if (shiftPriorToZero) p->bdSumF += p->priorF * p->deltaF;
    VecCf Hcd = p->Hcd_accAF + p->Hcd_accLF;
    accHcc[tid].update(Hcd, Hcd, p->HdiF);
    accbc[tid].update(Hcd, p->bdSumF * p->HdiF);

    int nFrames2 = nframes[tid] * nframes[tid];
    for (auto r1 : p->residuals) {
        if (!r1->isActive()) continue;
        int r1ht = r1->hostIDX + r1->targetIDX * nframes[tid];
        for (auto r2 : p->residuals) {
            if (!r2->isActive()) continue;
            accD[tid][r1ht + r2->targetIDX * nFrames2].update(r1->JpJdF, r2->JpJdF, p->HdiF);
        }
        accE[tid][r1ht].update(r1->JpJdF, Hcd, p->HdiF);
        accEB[tid][r1ht].update(r1->JpJdF, p->HdiF * p->bdSumF);
    }
}

static inline void sc_stitch_pair(MatXX &H, VecX &b, const Mat8C &Hpc, const Vec8 &bp, const Mat88 &AH, const Mat88 &AT, int iIdx, int jIdx) {
    H.addBlock<8, CPARS>(iIdx, 0, AH * Hpc);
    H.addBlock<8, CPARS>(jIdx, 0, AT * Hpc);
    Vec8 bi = AH * bp, bj = AT * bp;
    for (int r = 0; r < 8; r++) { b[iIdx + r] += bi[r]; b[jIdx + r] += bj[r]; }
}

// AccumulatedSCHessian.cc:53-119
void AccumulatedSCHessianSSE::stitchDoubleInternal(MatXX *H, VecX *b, EnergyFunctional const *const EF, int min, int max, Vec10 *stats, int tid) {
    int toAggregate = NUM_THREADS;
    if (tid == -1) { toAggregate = 1; tid = 0; }
    if (min == max) return;
    int nf = nframes[0];
    int nframes2 = nf * nf;
    for (int k = min; k < max; k++) {
        int i = k % nf, j = k / nf;
        int iIdx = CPARS + i * 8, jIdx = CPARS + j * 8, ijIdx = i + nf * j;
        Mat8C Hpc = Mat8C::Zero();
        Vec8 bp = Vec8::Zero();
        for (int tid2 = 0; tid2 < toAggregate; tid2++) {
            accE[tid2][ijIdx].finish();
            accEB[tid2][ijIdx].finish();
            Hpc += accE[tid2][ijIdx].A1m.cast<double>();
            bp += accEB[tid2][ijIdx].A1m.cast<double>();
        }
        sc_stitch_pair(H[tid], b[tid], Hpc, bp, EF->adHost[ijIdx], EF->adTarget[ijIdx], iIdx, jIdx);
        for (int k2 = 0; k2 < nf; k2++) {
            int kIdx = CPARS + k2 * 8, ijkIdx = ijIdx + k2 * nframes2, ikIdx = i + nf * k2;
            Mat88 accDM = Mat88::Zero();
            for (int tid2 = 0; tid2 < toAggregate; tid2++) {
                accD[tid2][ijkIdx].finish();
                if (accD[tid2][ijkIdx].num == 0) continue;
                accDM += accD[tid2][ijkIdx].A1m.cast<double>();
            }
            H[tid].addBlock<8, 8>(iIdx, iIdx, EF->adHost[ijIdx] * accDM * EF->adHost[ikIdx].transpose());
            H[tid].addBlock<8, 8>(jIdx, kIdx, EF->adTarget[ijIdx] * accDM * EF->adTarget[ikIdx].transpose());
            H[tid].addBlock<8, 8>(jIdx, iIdx, EF->adTarget[ijIdx] * accDM * EF->adHost[ikIdx].transpose());
            H[tid].addBlock<8, 8>(iIdx, kIdx, EF->adHost[ijIdx] * accDM * EF->adTarget[ikIdx].transpose());
        }
    }
    if (min == 0) {
        for (int tid2 = 0; tid2 < toAggregate; tid2++) {
            accHcc[tid2].finish();
            accbc[tid2].finish();
            for (int i = 0; i < CPARS; i++) { for (int j = 0; j < CPARS; j++) H[tid](i, j) += (double) accHcc[tid2].A1m(i, j); b[tid][i] += (double) accbc[tid2].A1m[i]; }
        }
    }
}

// AccumulatedSCHessian.h:64-98
void AccumulatedSCHessianSSE::stitchDoubleMT(IndexThreadReduce *red, MatXX &H, VecX &b, EnergyFunctional const *const EF, bool MT) {
    int n = nframes[0] * 8 + CPARS;
    if (MT) {
        MatXX Hs[NUM_THREADS];
        VecX bs[NUM_THREADS];
        for (int i = 0; i < NUM_THREADS; i++) { Hs[i] = MatXX::Zero(n, n); bs[i] = VecX::Zero(n); }
        using namespace std::placeholders;
        red->reduce(std::bind(&AccumulatedSCHessianSSE::stitchDoubleInternal, this, Hs, bs, EF, _1, _2, _3, _4), 0, nframes[0] * nframes[0], 0);
        H = Hs[0]; b = bs[0];
        for (int i = 1; i < NUM_THREADS; i++) { H += Hs[i]; b += bs[i]; }
    } else {
        H = MatXX::Zero(n, n); b = VecX::Zero(n);
        stitchDoubleInternal(&H, &b, EF, 0, nframes[0] * nframes[0], 0, -1);
    }
    for (int h = 0; h < nframes[0]; h++) {
        int hIdx = CPARS + h * 8;
        for (int i = 0; i < CPARS; i++) for (int j = 0; j < 8; j++) H(i, hIdx + j) = H(hIdx + j, i);
    }
}

// AccumulatedSCHessian.cc:121-177
void AccumulatedSCHessianSSE::stitchDouble(MatXX &H, VecX &b, const EnergyFunctional *const EF, int tid) {
    int nf = nframes[0];
    int nframes2 = nf * nf;
    H = MatXX::Zero(nf * 8 + CPARS, nf * 8 + CPARS);
    b = VecX::Zero(nf * 8 + CPARS);
    for (int i = 0; i < nf; i++)
        for (int j = 0; j < nf; j++) {
            int iIdx = CPARS + i * 8, jIdx = CPARS + j * 8, ijIdx = i + nf * j;
            accE[tid][ijIdx].finish();
            accEB[tid][ijIdx].finish();
            Mat8C accEM = accE[tid][ijIdx].A1m.cast<double>();
            Vec8 accEBV = accEB[tid][ijIdx].A1m.cast<double>();
            sc_stitch_pair(H, b, accEM, accEBV, EF->adHost[ijIdx], EF->adTarget[ijIdx], iIdx, jIdx);
            for (int k = 0; k < nf; k++) {
                int kIdx = CPARS + k * 8, ijkIdx = ijIdx + k * nframes2, ikIdx = i + nf * k;
                accD[tid][ijkIdx].finish();
                if (accD[tid][ijkIdx].num == 0) continue;
                Mat88 accDM = accD[tid][ijkIdx].A1m.cast<double>();
                H.addBlock<8, 8>(iIdx, iIdx, EF->adHost[ijIdx] * accDM * EF->adHost[ikIdx].transpose());
                H.addBlock<8, 8>(jIdx, kIdx, EF->adTarget[ijIdx] * accDM * EF->adTarget[ikIdx].transpose());
                H.addBlock<8, 8>(jIdx, iIdx, EF->adTarget[ijIdx] * accDM * EF->adHost[ikIdx].transpose());
                H.addBlock<8, 8>(iIdx, kIdx, EF->adHost[ijIdx] * accDM * EF->adTarget[ikIdx].transpose());
            }
        }
    accHcc[tid].finish();
    accbc[tid].finish();
    for (int i = 0; i < CPARS; i++) { for (int j = 0; j < CPARS; j++) H(i, j) = (double) accHcc[tid].A1m(i, j); b[i] = (double) accbc[tid].A1m[i]; }
    for (int h = 0; h < nf; h++) {
        int hIdx = CPARS + h * 8;
        for (int i = 0; i < CPARS; i++) for (int j = 0; j < 8; j++) H(i, hIdx + j) = H(hIdx + j, i);
    }
}

// ===================================================================================================
// EnergyFunctional.cc
// ===================================================================================================
void EnergyFunctional::insertFrame(FrameHessian *fh, CalibHessian *Hcalib) {   // :32-61
    fh->takeData();
    frames.push_back(fh);
    fh->idx = frames.size();
    nFrames++;
    bM.conservativeResize(8 * nFrames + CPARS);
    HM.conservativeResize(8 * nFrames + CPARS, 8 * nFrames + CPARS);   // new rows/cols are zero-filled by our resize
    EFIndicesValid = false; EFAdjointsValid = false; EFDeltaValid = false;
    setAdjointsF(Hcalib);
    makeIDX();
}

void EnergyFunctional::dropResidual(PointFrameResidual *r) {   // :63-70
    PointHessian *p = r->point;
    for (auto &t : p->residuals) if (t == r) { t = p->residuals.back(); p->residuals.pop_back(); break; }
    nResiduals--;
}

void EnergyFunctional::marginalizeFrame(FrameHessian *fh) {   // :72-151
    int ndim = nFrames * 8 + CPARS - 8;
    int odim = nFrames * 8 + CPARS;

    if ((int) fh->idx != (int) frames.size() - 1) {
        int io = fh->idx * 8 + CPARS;
        int ntail = 8 * (nFrames - fh->idx - 1);
        // move the frame's 8 rows/cols to the end, keeping the order of the others
        std::vector<int> perm;
        for (int i = 0; i < io; i++) perm.push_back(i);
        for (int i = io + 8; i < odim; i++) perm.push_back(i);
        for (int i = io; i < io + 8; i++) perm.push_back(i);
        (void) ntail;
        VecX b2(odim); MatXX H2(odim, odim);
        for (int i = 0; i < odim; i++) { b2[i] = bM[perm[i]]; for (int j = 0; j < odim; j++) H2(i, j) = HM(perm[i], perm[j]); }
        bM = b2; HM = H2;
    }

    for (int i = 0; i < 8; i++) { HM(ndim + i, ndim + i) += fh->prior[i]; bM[ndim + i] += fh->prior[i] * fh->delta_prior[i]; }

    VecX SVec(odim), SVecI(odim);
    for (int i = 0; i < odim; i++) { SVec[i] = std::sqrt(std::fabs(HM(i, i)) + 10); SVecI[i] = 1.0 / SVec[i]; }

    MatXX HMScaled(odim, odim);
    VecX bMScaled(odim);
    for (int i = 0; i < odim; i++) { bMScaled[i] = SVecI[i] * bM[i]; for (int j = 0; j < odim; j++) HMScaled(i, j) = SVecI[i] * HM(i, j) * SVecI[j]; }

    Mat88 hpi = HMScaled.block<8, 8>(ndim, ndim);
    hpi = (hpi + hpi) * 0.5;          // sic: 0.5f * (hpi + hpi) in the reference
    hpi = inverse_lu<8>(hpi);
    hpi = (hpi + hpi) * 0.5;

    // bli = bottomLeft(8,ndim)^T * hpi   (ndim x 8)
    MatXX bli(ndim, 8);
    for (int i = 0; i < ndim; i++) for (int j = 0; j < 8; j++) { double s = 0; for (int k = 0; k < 8; k++) s += HMScaled(ndim + k, i) * hpi(k, j); bli(i, j) = s; }
    for (int i = 0; i < ndim; i++) {
        for (int j = 0; j < ndim; j++) { double s = 0; for (int k = 0; k < 8; k++) s += bli(i, k) * HMScaled(ndim + k, j); HMScaled(i, j) -= s; }
        double s = 0; for (int k = 0; k < 8; k++) s += bli(i, k) * bMScaled[ndim + k];
        bMScaled[i] -= s;
    }

    for (int i = 0; i < odim; i++) { bMScaled[i] = SVec[i] * bMScaled[i]; for (int j = 0; j < odim; j++) HMScaled(i, j) = SVec[i] * HMScaled(i, j) * SVec[j]; }

    MatXX Hn(ndim, ndim); VecX bn(ndim);
    for (int i = 0; i < ndim; i++) { bn[i] = bMScaled[i]; for (int j = 0; j < ndim; j++) Hn(i, j) = 0.5 * (HMScaled(i, j) + HMScaled(j, i)); }
    HM = Hn; bM = bn;

    for (unsigned int i = fh->idx; i + 1 < frames.size(); i++) { frames[i] = frames[i + 1]; frames[i]->idx = i; }
    frames.pop_back();
    nFrames--;

    EFIndicesValid = false; EFAdjointsValid = false; EFDeltaValid = false;
    makeIDX();
}

void EnergyFunctional::removePoint(PointHessian *ph) {   // :153-163
    for (auto &r : ph->residuals) { (void) r; nResiduals--; }
    ph->residuals.clear();
    if (!ph->alreadyRemoved) nPoints--;
    EFIndicesValid = false;
}

void EnergyFunctional::marginalizePointsF() {   // :165-222
    allPointsToMarg.clear();
    for (auto f : frames)
        for (PointHessian *p : f->features)
            if (p->status == PS_MARGINALIZED && !p->alreadyRemoved) {
                p->priorF *= g->s.idepthFixPriorMargFac;
                allPointsToMarg.push_back(p);
            }
    accSSE_bot->setZero(nFrames);
    accSSE_top_A->setZero(nFrames);
    for (auto p : allPointsToMarg) {
        accSSE_top_A->addPoint<2>(p, this);
        accSSE_bot->addPoint(p, false);
        removePoint(p);
        p->alreadyRemoved = true;   // stands in for Point::ReleasePH() (Point.cc) so the point is skipped afterwards
    }
    MatXX M, Msc;
    VecX Mb, Mbsc;
    accSSE_top_A->stitchDouble(M, Mb, this, false, false);
    accSSE_bot->stitchDouble(Msc, Mbsc, this);
    resInM += accSSE_top_A->nres[0];
    MatXX H = M - Msc;
    VecX b = Mb - Mbsc;
    if (g->s.solverMode & LDSO_SOLVER_ORTHOGONALIZE_POINTMARG) {
        bool haveFirstFrame = false;
        for (auto f : frames) if (f->frameID == 0) haveFirstFrame = true;
        if (!haveFirstFrame) orthogonalize(&bM, &HM);
    }
    HM += H * (double) g->s.margWeightFac;
    bM += b * (double) g->s.margWeightFac;
    if (g->s.solverMode & LDSO_SOLVER_ORTHOGONALIZE_FULL) orthogonalize(&bM, &HM);
    EFIndicesValid = false;
    makeIDX();
}

void EnergyFunctional::dropPointsF() {   // :224-238
    for (auto f : frames)
        for (PointHessian *p : f->features)
            if ((p->status == PS_OUTLIER || p->status == PS_OUT) && p->alreadyRemoved == false) { removePoint(p); p->alreadyRemoved = true; }
    EFIndicesValid = false;
    makeIDX();
}

void EnergyFunctional::solveSystemF(int iteration, double lambda, CalibHessian *HCalib) {   // :240-351
    if (g->s.solverMode & LDSO_SOLVER_USE_GN) lambda = 0;
    if (g->s.solverMode & LDSO_SOLVER_FIX_LAMBDA) lambda = 1e-5;

    MatXX HL_top, HA_top, H_sc;
    VecX bL_top, bA_top, bM_top, b_sc;

    accumulateAF_MT(HA_top, bA_top, multiThreading);
    accumulateLF_MT(HL_top, bL_top, multiThreading);
    accumulateSCF_MT(H_sc, b_sc, multiThreading);

    bM_top = (bM + HM * getStitchedDeltaF());

    MatXX HFinal_top;
    VecX bFinal_top;
    const int n = 8 * nFrames + CPARS;

    if (g->s.solverMode & LDSO_SOLVER_ORTHOGONALIZE_SYSTEM) {
        bool haveFirstFrame = false;
        for (auto f : frames) if (f->frameID == 0) haveFirstFrame = true;
        MatXX HT_act = HL_top + HA_top - H_sc;
        VecX bT_act = bL_top + bA_top - b_sc;
        if (!haveFirstFrame) orthogonalize(&bT_act, &HT_act);
        HFinal_top = HT_act + HM;
        bFinal_top = bT_act + bM_top;
        lastHS = HFinal_top;
        lastbS = bFinal_top;
        for (int i = 0; i < n; i++) HFinal_top(i, i) *= (1 + lambda);
    } else {
        HFinal_top = HL_top + HM + HA_top;
        bFinal_top = bL_top + bM_top + bA_top - b_sc;
        lastHS = HFinal_top - H_sc;
        lastbS = bFinal_top;
        for (int i = 0; i < n; i++) HFinal_top(i, i) *= (1 + lambda);
        HFinal_top -= H_sc * (double) (1.0f / (1 + lambda));
    }
    last_HA = HA_top; last_bA = bA_top; last_HL = HL_top; last_bL = bL_top; last_Hsc = H_sc; last_bsc = b_sc;
    last_HFinal = HFinal_top; last_bFinal = bFinal_top;

    VecX x;
    if (g->s.solverMode & LDSO_SOLVER_SVD) {
        // EnergyFunctional.cc:296-322 — not the default mode; restated via the same thin SVD
        VecX SVecI(n);
        for (int i = 0; i < n; i++) SVecI[i] = 1.0 / std::sqrt(HFinal_top(i, i));
        MatXX HFinalScaled(n, n); VecX bFinalScaled(n);
        for (int i = 0; i < n; i++) { bFinalScaled[i] = SVecI[i] * bFinal_top[i]; for (int j = 0; j < n; j++) HFinalScaled(i, j) = SVecI[i] * HFinal_top(i, j) * SVecI[j]; }
        MatXX U, V; VecX S;
        jacobi_svd_thin(HFinalScaled, U, S, V);
        // sort descending like Eigen
        std::vector<int> order(n); for (int i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b2) { return S[a] > S[b2]; });
        double maxSv = 0; for (int i = 0; i < n; i++) maxSv = std::max(maxSv, S[i]);
        VecX Ub(n);
        for (int i = 0; i < n; i++) { double s = 0; for (int r = 0; r < n; r++) s += U(r, order[i]) * bFinalScaled[r]; Ub[i] = s; }
        for (int i = 0; i < n; i++) {
            if (S[order[i]] < g->s.solverModeDelta * maxSv) Ub[i] = 0;
            if ((g->s.solverMode & LDSO_SOLVER_SVD_CUT7) && (i >= n - 7)) Ub[i] = 0; else Ub[i] /= S[order[i]];
        }
        x = VecX(n);
        for (int r = 0; r < n; r++) { double s = 0; for (int i = 0; i < n; i++) s += V(r, order[i]) * Ub[i]; x[r] = SVecI[r] * s; }
    } else {
        VecX SVecI(n);
        for (int i = 0; i < n; i++) SVecI[i] = 1.0 / std::sqrt(HFinal_top(i, i) + 10);
        MatXX HFinalScaled(n, n);
        VecX bs(n);
        for (int i = 0; i < n; i++) { bs[i] = SVecI[i] * bFinal_top[i]; for (int j = 0; j < n; j++) HFinalScaled(i, j) = SVecI[i] * HFinal_top(i, j) * SVecI[j]; }
        x = LDLT(HFinalScaled).solve(bs);
        for (int i = 0; i < n; i++) x[i] = SVecI[i] * x[i];
    }

    if ((g->s.solverMode & LDSO_SOLVER_ORTHOGONALIZE_X) || (iteration >= 2 && (g->s.solverMode & LDSO_SOLVER_ORTHOGONALIZE_X_LATER))) {
        orthogonalize(&x, 0);
    }
    lastX = x;
    currentLambda = lambda;
    resubstituteF_MT(x, HCalib, multiThreading);
    currentLambda = 0;
}

double EnergyFunctional::calcMEnergyF() {   // :353-359
    VecX delta = getStitchedDeltaF();
    return delta.dot(bM * 2.0 + HM * delta);
}

double EnergyFunctional::calcLEnergyF_MT() {   // :361-378
    double E = 0;
    for (auto f : frames) for (int i = 0; i < 8; i++) E += f->delta_prior[i] * f->prior[i] * f->delta_prior[i];
    { float e = 0; for (int i = 0; i < CPARS; i++) e += cDeltaF[i] * cPriorF[i] * cDeltaF[i]; E += e; }
    if (multiThreading) {
        using namespace std::placeholders;
        red->reduce(std::bind(&EnergyFunctional::calcLEnergyPt, this, _1, _2, _3, _4), 0, allPoints.size(), 50);
        return E + red->stats[0];
    }
    Vec10 st; calcLEnergyPt(0, allPoints.size(), &st, 0);
    return E + st[0];
}

void EnergyFunctional::makeIDX() {   // :380-401
    for (unsigned int idx = 0; idx < frames.size(); idx++) frames[idx]->idx = idx;
    allPoints.clear();
    for (auto f : frames)
        for (PointHessian *p : f->features)
            if (p->status == PS_ACTIVE && !p->alreadyRemoved) {
                allPoints.push_back(p);
                for (auto &r : p->residuals) { r->hostIDX = r->host->idx; r->targetIDX = r->target->idx; }
            }
    EFIndicesValid = true;
}

void EnergyFunctional::setDeltaF(CalibHessian *HCalib) {   // :403-429
    adHTdeltaF.assign(nFrames * nFrames, Mat18f());
    for (int h = 0; h < nFrames; h++)
        for (int t = 0; t < nFrames; t++) {
            int idx = h + t * nFrames;
            Vec10 dh = frames[h]->get_state_minus_stateZero(), dt = frames[t]->get_state_minus_stateZero();
            Mat18f r;
            for (int c = 0; c < 8; c++) {
                float s = 0, s2 = 0;
                for (int k = 0; k < 8; k++) s += (float) dh[k] * adHostF[idx](k, c);
                for (int k = 0; k < 8; k++) s2 += (float) dt[k] * adTargetF[idx](k, c);
                r[c] = s + s2;
            }
            adHTdeltaF[idx] = r;
        }
    cDeltaF = HCalib->value_minus_value_zero.cast<float>();
    for (auto f : frames) {
        Vec10 d = f->get_state_minus_stateZero();
        for (int i = 0; i < 8; i++) { f->delta[i] = d[i]; f->delta_prior[i] = f->state[i]; }
        for (PointHessian *p : f->features) if (p->status == PS_ACTIVE && !p->alreadyRemoved) p->deltaF = p->idepth - p->idepth_zero;
    }
    EFDeltaValid = true;
}

void EnergyFunctional::setAdjointsF(CalibHessian *Hcalib) {   // :431-489
    adHost.assign(nFrames * nFrames, Mat88());
    adTarget.assign(nFrames * nFrames, Mat88());
    for (int h = 0; h < nFrames; h++)
        for (int t = 0; t < nFrames; t++) {
            FrameHessian *host = frames[h];
            FrameHessian *target = frames[t];
            SE3 hostToTarget = target->worldToCam_evalPT * host->worldToCam_evalPT.inverse();
            Mat88 AH = Mat88::Identity();
            Mat88 AT = Mat88::Identity();
            Mat66 adj = hostToTarget.Adj();
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) AH(i, j) = -adj(j, i);
            Vec2f affLL = AffLight::fromToVecExposure(host->ab_exposure, target->ab_exposure, host->aff_g2l_0(), target->aff_g2l_0()).cast<float>();
            AT(6, 6) = -affLL[0];
            AH(6, 6) = affLL[0];
            AT(7, 7) = -1;
            AH(7, 7) = affLL[0];
            for (int c = 0; c < 8; c++) {
                for (int r = 0; r < 3; r++) { AH(r, c) *= SCALE_XI_TRANS; AT(r, c) *= SCALE_XI_TRANS; }
                for (int r = 3; r < 6; r++) { AH(r, c) *= SCALE_XI_ROT; AT(r, c) *= SCALE_XI_ROT; }
                AH(6, c) *= SCALE_A; AT(6, c) *= SCALE_A;
                AH(7, c) *= SCALE_B; AT(7, c) *= SCALE_B;
            }
            adHost[h + t * nFrames] = AH;
            adTarget[h + t * nFrames] = AT;
        }
    cPrior = VecC::Constant(g->s.initialCalibHessian);
    adHostF.assign(nFrames * nFrames, Mat88f());
    adTargetF.assign(nFrames * nFrames, Mat88f());
    for (int i = 0; i < nFrames * nFrames; i++) { adHostF[i] = adHost[i].cast<float>(); adTargetF[i] = adTarget[i].cast<float>(); }
    cPriorF = cPrior.cast<float>();
    EFAdjointsValid = true;
}

void EnergyFunctional::resubstituteF_MT(const VecX &x, CalibHessian *HCalib, bool MT) {   // :491-516
    std::vector<float> xF(x.size());
    for (int i = 0; i < x.size(); i++) xF[i] = (float) x[i];
    for (int i = 0; i < CPARS; i++) HCalib->step[i] = -x[i];

    std::vector<Mat18f> xAd(nFrames * nFrames);
    VecCf cstep; for (int i = 0; i < CPARS; i++) cstep[i] = xF[i];
    for (auto h : frames) {
        for (int i = 0; i < 8; i++) h->step[i] = -x[CPARS + 8 * h->idx + i];
        h->step[8] = 0; h->step[9] = 0;
        for (auto t : frames) {
            Mat18f r;
            const Mat88f &AH = adHostF[h->idx + nFrames * t->idx];
            const Mat88f &AT = adTargetF[h->idx + nFrames * t->idx];
            for (int c = 0; c < 8; c++) {
                float s = 0, s2 = 0;
                for (int k = 0; k < 8; k++) s += xF[CPARS + 8 * h->idx + k] * AH(k, c);
                for (int k = 0; k < 8; k++) s2 += xF[CPARS + 8 * t->idx + k] * AT(k, c);
                r[c] = s + s2;
            }
            xAd[nFrames * h->idx + t->idx] = r;
        }
    }
    if (MT) {
        using namespace std::placeholders;
        red->reduce(std::bind(&EnergyFunctional::resubstituteFPt, this, cstep, xAd.data(), _1, _2, _3, _4), 0, allPoints.size(), 50);
    } else
        resubstituteFPt(cstep, xAd.data(), 0, allPoints.size(), 0, 0);
}

void EnergyFunctional::resubstituteFPt(const VecCf &xc, Mat18f *xAd, int min, int max, Vec10 *stats, int tid) {   // :518-547
    for (int k = min; k < max; k++) {
        auto p = allPoints[k];
        int ngoodres = 0;
        for (auto r : p->residuals) if (r->isActive()) ngoodres++;
        if (ngoodres == 0) { p->step = 0; continue; }
        float b = p->bdSumF;
        b -= xc.dot(p->Hcd_accAF + p->Hcd_accLF);
        for (auto r : p->residuals) {
            if (!r->isActive()) continue;
            const Mat18f &xa = xAd[r->hostIDX * nFrames + r->targetIDX];
            float s = 0; for (int i = 0; i < 8; i++) s += xa[i] * r->JpJdF[i];
            b -= s;
        }
        if (!std::isfinite(b) || std::isnan(b)) return;
        p->step = -b * p->HdiF;
    }
}

void EnergyFunctional::accumulateAF_MT(MatXX &H, VecX &b, bool MT) {   // :550-574
    using namespace std::placeholders;
    if (MT) {
        red->reduce(std::bind(&AccumulatedTopHessianSSE::setZero, accSSE_top_A, nFrames, _1, _2, _3, _4), 0, 0, 0);
        red->reduce(std::bind(&AccumulatedTopHessianSSE::addPointsInternal<0>, accSSE_top_A, &allPoints, this, _1, _2, _3, _4), 0, allPoints.size(), 50);
        accSSE_top_A->stitchDoubleMT(red, H, b, this, false, true);
        resInA = accSSE_top_A->nres[0];
    } else {
        accSSE_top_A->setZero(nFrames);
        for (auto f : frames) for (PointHessian *p : f->features) if (p->status == PS_ACTIVE && !p->alreadyRemoved) accSSE_top_A->addPoint<0>(p, this);
        accSSE_top_A->stitchDoubleMT(red, H, b, this, false, false);
        resInA = accSSE_top_A->nres[0];
    }
}

void EnergyFunctional::accumulateLF_MT(MatXX &H, VecX &b, bool MT) {   // :577-600
    using namespace std::placeholders;
    if (MT) {
        red->reduce(std::bind(&AccumulatedTopHessianSSE::setZero, accSSE_top_L, nFrames, _1, _2, _3, _4), 0, 0, 0);
        red->reduce(std::bind(&AccumulatedTopHessianSSE::addPointsInternal<1>, accSSE_top_L, &allPoints, this, _1, _2, _3, _4), 0, allPoints.size(), 50);
        accSSE_top_L->stitchDoubleMT(red, H, b, this, true, true);
        resInL = accSSE_top_L->nres[0];
    } else {
        accSSE_top_L->setZero(nFrames);
        for (auto f : frames) for (PointHessian *p : f->features) if (p->status == PS_ACTIVE && !p->alreadyRemoved) accSSE_top_L->addPoint<1>(p, this);
        accSSE_top_L->stitchDoubleMT(red, H, b, this, true, false);
        resInL = accSSE_top_L->nres[0];
    }
}

void EnergyFunctional::accumulateSCF_MT(MatXX &H, VecX &b, bool MT) {   // :603-624
    using namespace std::placeholders;
    if (MT) {
        red->reduce(std::bind(&AccumulatedSCHessianSSE::setZero, accSSE_bot, nFrames, _1, _2, _3, _4), 0, 0, 0);
        red->reduce(std::bind(&AccumulatedSCHessianSSE::addPointsInternal, accSSE_bot, &allPoints, true, _1, _2, _3, _4), 0, allPoints.size(), 50);
        accSSE_bot->stitchDoubleMT(red, H, b, this, true);
    } else {
        accSSE_bot->setZero(nFrames);
        for (auto f : frames) for (PointHessian *p : f->features) if (p->status == PS_ACTIVE && !p->alreadyRemoved) accSSE_bot->addPoint(p, true);
        accSSE_bot->stitchDoubleMT(red, H, b, this, false);
    }
}

void EnergyFunctional::calcLEnergyPt(int min, int max, Vec10 *stats, int tid) {   // :627-682
    Accumulator11 E;
    E.initialize();
    VecCf dc = cDeltaF;
    for (int i = min; i < max; i++) {
        auto p = allPoints[i];
        float dd = p->deltaF;
        for (auto r : p->residuals) {
            if (!r->isLinearized || !r->isActive()) continue;
            Mat18f dp = adHTdeltaF[r->hostIDX + nFrames * r->targetIDX];
            RawResidualJacobian *rJ = &r->J;
            float dpx = 0, dpy = 0;
            for (int k = 0; k < 6; k++) { dpx += rJ->Jpdxi[0][k] * dp[k]; dpy += rJ->Jpdxi[1][k] * dp[k]; }
            float Jp_delta_x_1 = dpx + rJ->Jpdc[0].dot(dc) + rJ->Jpdd[0] * dd;
            float Jp_delta_y_1 = dpy + rJ->Jpdc[1].dot(dc) + rJ->Jpdd[1] * dd;
            __m128 Jp_delta_x = _mm_set1_ps(Jp_delta_x_1);
            __m128 Jp_delta_y = _mm_set1_ps(Jp_delta_y_1);
            __m128 delta_a = _mm_set1_ps((float) (dp[6]));
            __m128 delta_b = _mm_set1_ps((float) (dp[7]));
            for (int k = 0; k + 3 < patternNum; k += 4) {
                __m128 Jdelta = _mm_mul_ps(_mm_loadu_ps(rJ->JIdx[0].d + k), Jp_delta_x);
                Jdelta = _mm_add_ps(Jdelta, _mm_mul_ps(_mm_loadu_ps(rJ->JIdx[1].d + k), Jp_delta_y));
                Jdelta = _mm_add_ps(Jdelta, _mm_mul_ps(_mm_loadu_ps(rJ->JabF[0].d + k), delta_a));
                Jdelta = _mm_add_ps(Jdelta, _mm_mul_ps(_mm_loadu_ps(rJ->JabF[1].d + k), delta_b));
                __m128 r0 = _mm_loadu_ps(r->res_toZeroF.d + k);
                r0 = _mm_add_ps(r0, r0);
                r0 = _mm_add_ps(r0, Jdelta);
                Jdelta = _mm_mul_ps(Jdelta, r0);
                E.updateSSENoShift(Jdelta);
            }
        }
        E.updateSingle(p->deltaF * p->deltaF * p->priorF);
    }
    E.finish();
    (*stats)[0] += E.A;
}

void EnergyFunctional::orthogonalize(VecX *b, MatXX *H) {   // :685-717
    std::vector<VecX> ns;
    ns.insert(ns.end(), lastNullspaces_pose.begin(), lastNullspaces_pose.end());
    ns.insert(ns.end(), lastNullspaces_scale.begin(), lastNullspaces_scale.end());
    int dim = ns[0].size(), m = ns.size();
    MatXX N(dim, m);
    for (int i = 0; i < m; i++) { double nn = ns[i].norm(); for (int r = 0; r < dim; r++) N(r, i) = ns[i][r] / nn; }
    MatXX U, V; VecX SNN;
    jacobi_svd_thin(N, U, SNN, V);
    double maxSv = 0;
    for (int i = 0; i < m; i++) if (SNN[i] > maxSv) maxSv = SNN[i];
    for (int i = 0; i < m; i++) { if (SNN[i] > g->s.solverModeDelta * maxSv) SNN[i] = 1.0 / SNN[i]; else SNN[i] = 0; }
    // Npi = U * diag(SNN) * V^T  (dim x m);  NNpiT = N * Npi^T;  NNpiTS = 0.5 (NNpiT + NNpiT^T)
    MatXX Npi(dim, m);
    for (int r = 0; r < dim; r++) for (int c = 0; c < m; c++) { double s = 0; for (int k = 0; k < m; k++) s += U(r, k) * SNN[k] * V(c, k); Npi(r, c) = s; }
    MatXX NNpiT = N * Npi.transpose();
    MatXX NNpiTS(dim, dim);
    for (int r = 0; r < dim; r++) for (int c = 0; c < dim; c++) NNpiTS(r, c) = 0.5 * (NNpiT(r, c) + NNpiT(c, r));
    if (b != 0) *b -= NNpiTS * *b;
    if (H != 0) *H -= NNpiTS * *H * NNpiTS;
}

// ===================================================================================================
// FullSystem.cc optimisation slice
// ===================================================================================================
void FullSystem::collectActiveResiduals() {   // :735-755
    activeResiduals.clear();
    for (FrameHessian *fr : frames)
        for (PointHessian *ph : fr->features)
            if (ph->status == PS_ACTIVE && !ph->alreadyRemoved)
                for (auto &r : ph->residuals)
                    if (!r->isLinearized) { activeResiduals.push_back(r); r->resetOOB(); }
}

float FullSystem::optimize(int mnumOptIts) {   // :725-864
    if (frames.size() < 2) return 0;
    if (!forceAllIterations) {
        if (frames.size() < 3) mnumOptIts = 20;
        if (frames.size() < 4) mnumOptIts = 15;
    }
    energyLog.clear();
    collectActiveResiduals();

    Vec3 lastEnergy = linearizeAll(false);
    double lastEnergyL = calcLEnergy();
    double lastEnergyM = calcMEnergy();
    energyLog.push_back(lastEnergy[0]);

    using namespace std::placeholders;
    if (multiThreading) threadReduce->reduce(std::bind(&FullSystem::applyRes_Reductor, this, true, _1, _2, _3, _4), 0, activeResiduals.size(), 50);
    else applyRes_Reductor(true, 0, activeResiduals.size(), 0, 0);

    double lambda = 1e-1;
    float stepsize = 1;
    VecX previousX = VecX::Constant(CPARS + 8 * frames.size(), NAN);

    for (int iteration = 0; iteration < mnumOptIts; iteration++) {
        backupState(iteration != 0);
        solveSystem(iteration, lambda);
        double incDirChange = (1e-20 + previousX.dot(ef->lastX)) / (1e-20 + previousX.norm() * ef->lastX.norm());
        previousX = ef->lastX;
        if (std::isfinite(incDirChange) && (g.s.solverMode & LDSO_SOLVER_STEPMOMENTUM)) {
            float newStepsize = exp(incDirChange * 1.4);
            if (incDirChange < 0 && stepsize > 1) stepsize = 1;
            stepsize = sqrtf(sqrtf(newStepsize * stepsize * stepsize * stepsize));
            if (stepsize > 2) stepsize = 2;
            if (stepsize < 0.25) stepsize = 0.25;
        }
        bool canbreak = doStepFromBackup(stepsize, stepsize, stepsize, stepsize, stepsize);

        Vec3 newEnergy = linearizeAll(false);
        double newEnergyL = calcLEnergy();
        double newEnergyM = calcMEnergy();
        energyLog.push_back(newEnergy[0]);

        if (g.s.forceAcceptStep || (newEnergy[0] + newEnergy[1] + newEnergyL + newEnergyM < lastEnergy[0] + lastEnergy[1] + lastEnergyL + lastEnergyM)) {
            if (multiThreading) threadReduce->reduce(std::bind(&FullSystem::applyRes_Reductor, this, true, _1, _2, _3, _4), 0, activeResiduals.size(), 50);
            else applyRes_Reductor(true, 0, activeResiduals.size(), 0, 0);
            lastEnergy = newEnergy; lastEnergyL = newEnergyL; lastEnergyM = newEnergyM;
            lambda *= 0.25;
        } else {
            loadSateBackup();
            lastEnergy = linearizeAll(false);
            lastEnergyL = calcLEnergy();
            lastEnergyM = calcMEnergy();
            lambda *= 1e2;
        }
        if (canbreak && iteration >= g.s.minOptIterations && !forceAllIterations) break;
    }

    Vec10 newStateZero = Vec10::Zero();
    newStateZero[6] = frames.back()->state[6];
    newStateZero[7] = frames.back()->state[7];
    frames.back()->setEvalPT(frames.back()->PRE_worldToCam, newStateZero);

    ef->EFDeltaValid = false;
    ef->EFAdjointsValid = false;
    ef->setAdjointsF(&Hcalib);
    setPrecalcValues();

    lastEnergy = linearizeAll(true);
    energyLog.push_back(lastEnergy[0]);

    if (!std::isfinite((double) lastEnergy[0]) || !std::isfinite((double) lastEnergy[1]) || !std::isfinite((double) lastEnergy[2])) isLost = true;

    return sqrtf((float) (lastEnergy[0] / (patternNum * ef->resInA)));
}

void FullSystem::setPrecalcValues() {   // :1423-1431
    for (auto &fr : frames) {
        fr->targetPrecalc.resize(frames.size());
        for (size_t i = 0; i < frames.size(); i++) fr->targetPrecalc[i].Set(fr, frames[i], &Hcalib);
    }
    ef->setDeltaF(&Hcalib);
}

void FullSystem::solveSystem(int iteration, double lambda) {   // :1433-1440
    ef->lastNullspaces_forLogging = getNullspaces(ef->lastNullspaces_pose, ef->lastNullspaces_scale, ef->lastNullspaces_affA, ef->lastNullspaces_affB);
    ef->solveSystemF(iteration, lambda, &Hcalib);
}

Vec3 FullSystem::linearizeAll(bool fixLinearization) {   // :1442-1492
    double lastEnergyP = 0, lastEnergyR = 0, num = 0;
    std::vector<PointFrameResidual *> toRemove[NUM_THREADS];
    if (multiThreading) {
        using namespace std::placeholders;
        threadReduce->reduce(std::bind(&FullSystem::linearizeAll_Reductor, this, fixLinearization, toRemove, _1, _2, _3, _4), 0, activeResiduals.size(), 0);
        lastEnergyP = threadReduce->stats[0];
    } else {
        Vec10 stats;   // zero-initialised here; the reference leaves it uninitialised (FullSystem.cc:1459, latent bug)
        linearizeAll_Reductor(fixLinearization, toRemove, 0, activeResiduals.size(), &stats, 0);
        lastEnergyP = stats[0];
    }
    setNewFrameEnergyTH();
    if (fixLinearization) {
        for (auto r : activeResiduals) {
            PointHessian *ph = r->point;
            if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].second = r->state_state;
            else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].second = r->state_state;
        }
        for (int i = 0; i < NUM_THREADS; i++)
            for (auto r : toRemove[i]) {
                PointHessian *ph = r->point;
                if (ph->lastResiduals[0].first == r) ph->lastResiduals[0].first = 0;
                else if (ph->lastResiduals[1].first == r) ph->lastResiduals[1].first = 0;
                ef->dropResidual(r);
            }
    }
    Vec3 r; r[0] = lastEnergyP; r[1] = lastEnergyR; r[2] = num;
    return r;
}

void FullSystem::linearizeAll_Reductor(bool fixLinearization, std::vector<PointFrameResidual *> *toRemove, int min, int max, Vec10 *stats, int tid) {   // :1494-1543
    for (int k = min; k < max; k++) {
        PointFrameResidual *r = activeResiduals[k];
        (*stats)[0] += r->linearize(&Hcalib, g);
        if (fixLinearization) {
            r->applyRes(true);
            if (r->isActive()) {
                if (r->isNew) {
                    PointHessian *p = r->point;
                    FrameHessian *host = r->host, *target = r->target;
                    Vec3f pv; pv[0] = p->u; pv[1] = p->v; pv[2] = 1;
                    Vec3f ptp_inf = host->targetPrecalc[target->idx].PRE_KRKiTll * pv;
                    Vec3f ptp = ptp_inf + host->targetPrecalc[target->idx].PRE_KtTll * p->idepth_scaled;
                    float ax = ptp_inf[0] / ptp_inf[2] - ptp[0] / ptp[2], ay = ptp_inf[1] / ptp_inf[2] - ptp[1] / ptp[2];
                    float relBS = 0.01 * std::sqrt(ax * ax + ay * ay);
                    if (relBS > p->maxRelBaseline) p->maxRelBaseline = relBS;
                    p->numGoodResiduals++;
                }
            } else {
                toRemove[tid].push_back(activeResiduals[k]);
            }
        }
    }
}

bool FullSystem::doStepFromBackup(float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD) {   // :1546-1623 (non-momentum branch)
    Vec10 pstepfac;
    for (int i = 0; i < 3; i++) pstepfac[i] = stepfacT;
    for (int i = 3; i < 6; i++) pstepfac[i] = stepfacR;
    for (int i = 6; i < 10; i++) pstepfac[i] = stepfacA;
    float sumA = 0, sumB = 0, sumT = 0, sumR = 0, sumID = 0, numID = 0;
    float sumNID = 0;

    Hcalib.setValue(Hcalib.value_backup + Hcalib.step * (double) stepfacC);
    for (auto &fh : frames) {
        Vec10 ns;
        for (int i = 0; i < 10; i++) ns[i] = fh->state_backup[i] + pstepfac[i] * fh->step[i];
        fh->setState(ns);
        sumA += fh->step[6] * fh->step[6];
        sumB += fh->step[7] * fh->step[7];
        sumT += fh->step[0] * fh->step[0] + fh->step[1] * fh->step[1] + fh->step[2] * fh->step[2];
        sumR += fh->step[3] * fh->step[3] + fh->step[4] * fh->step[4] + fh->step[5] * fh->step[5];
        for (PointHessian *ph : fh->features)
            if (ph->status == PS_ACTIVE && !ph->alreadyRemoved) {
                ph->setIdepth(ph->idepth_backup + stepfacD * ph->step);
                sumID += ph->step * ph->step;
                sumNID += fabsf(ph->idepth_backup);
                numID++;
                ph->setIdepthZero(ph->idepth_backup + stepfacD * ph->step);
            }
    }
    sumA /= frames.size(); sumB /= frames.size(); sumR /= frames.size(); sumT /= frames.size();
    sumID /= numID; sumNID /= numID;

    ef->EFDeltaValid = false;
    setPrecalcValues();

    return sqrtf(sumA) < 0.0005 * g.s.thOptIterations && sqrtf(sumB) < 0.00005 * g.s.thOptIterations &&
           sqrtf(sumR) < 0.00005 * g.s.thOptIterations && sqrtf(sumT) * sumNID < 0.00005 * g.s.thOptIterations;
}

void FullSystem::backupState(bool backupLastStep) {   // :1625-1673 (non-momentum branch)
    Hcalib.value_backup = Hcalib.value;
    for (auto &fh : frames) {
        fh->state_backup = fh->state;
        for (PointHessian *ph : fh->features) if (ph->status == PS_ACTIVE && !ph->alreadyRemoved) ph->idepth_backup = ph->idepth;
    }
}

void FullSystem::loadSateBackup() {   // :1675-1692
    Hcalib.setValue(Hcalib.value_backup);
    for (auto fh : frames) {
        fh->setState(fh->state_backup);
        for (PointHessian *ph : fh->features) if (ph->status == PS_ACTIVE && !ph->alreadyRemoved) { ph->setIdepth(ph->idepth_backup); ph->setIdepthZero(ph->idepth_backup); }
    }
    ef->EFDeltaValid = false;
    setPrecalcValues();
}

std::vector<VecX> FullSystem::getNullspaces(std::vector<VecX> &nullspaces_pose, std::vector<VecX> &nullspaces_scale,
                                            std::vector<VecX> &nullspaces_affA, std::vector<VecX> &nullspaces_affB) {   // :1711-1760
    nullspaces_pose.clear(); nullspaces_scale.clear(); nullspaces_affA.clear(); nullspaces_affB.clear();
    int n = CPARS + frames.size() * 8;
    std::vector<VecX> nullspaces_x0_pre;
    for (int i = 0; i < 6; i++) {
        VecX nullspace_x0(n);
        for (auto fh : frames) {
            for (int r = 0; r < 6; r++) nullspace_x0[CPARS + fh->idx * 8 + r] = fh->nullspaces_pose(r, i);
            for (int r = 0; r < 3; r++) nullspace_x0[CPARS + fh->idx * 8 + r] *= SCALE_XI_TRANS_INVERSE;
            for (int r = 3; r < 6; r++) nullspace_x0[CPARS + fh->idx * 8 + r] *= SCALE_XI_ROT_INVERSE;
        }
        nullspaces_x0_pre.push_back(nullspace_x0);
        nullspaces_pose.push_back(nullspace_x0);
    }
    for (int i = 0; i < 2; i++) {
        VecX nullspace_x0(n);
        for (auto fh : frames) {
            nullspace_x0[CPARS + fh->idx * 8 + 6] = fh->nullspaces_affine(0, i);
            nullspace_x0[CPARS + fh->idx * 8 + 7] = fh->nullspaces_affine(1, i);
            nullspace_x0[CPARS + fh->idx * 8 + 6] *= SCALE_A_INVERSE;
            nullspace_x0[CPARS + fh->idx * 8 + 7] *= SCALE_B_INVERSE;
        }
        nullspaces_x0_pre.push_back(nullspace_x0);
        if (i == 0) nullspaces_affA.push_back(nullspace_x0);
        if (i == 1) nullspaces_affB.push_back(nullspace_x0);
    }
    VecX nullspace_x0(n);
    for (auto fh : frames) {
        for (int r = 0; r < 6; r++) nullspace_x0[CPARS + fh->idx * 8 + r] = fh->nullspaces_scale[r];
        for (int r = 0; r < 3; r++) nullspace_x0[CPARS + fh->idx * 8 + r] *= SCALE_XI_TRANS_INVERSE;
        for (int r = 3; r < 6; r++) nullspace_x0[CPARS + fh->idx * 8 + r] *= SCALE_XI_ROT_INVERSE;
    }
    nullspaces_x0_pre.push_back(nullspace_x0);
    nullspaces_scale.push_back(nullspace_x0);
    return nullspaces_x0_pre;
}

void FullSystem::setNewFrameEnergyTH() {   // :1762-1793
    allResVec.clear();
    allResVec.reserve(activeResiduals.size() * 2);
    FrameHessian *newFrame = frames.back();
    for (auto &r : activeResiduals)
        if (r->state_NewEnergyWithOutlier >= 0 && r->target == newFrame) allResVec.push_back(r->state_NewEnergyWithOutlier);
    if (allResVec.size() == 0) { newFrame->frameEnergyTH = 12 * 12 * patternNum; return; }
    int nthIdx = g.s.frameEnergyTHN * allResVec.size();
    std::nth_element(allResVec.begin(), allResVec.begin() + nthIdx, allResVec.end());
    float nthElement = sqrtf(allResVec[nthIdx]);
    newFrame->frameEnergyTH = nthElement * g.s.frameEnergyTHFacMedian;
    newFrame->frameEnergyTH = 26.0f * g.s.frameEnergyTHConstWeight + newFrame->frameEnergyTH * (1 - g.s.frameEnergyTHConstWeight);
    newFrame->frameEnergyTH = newFrame->frameEnergyTH * newFrame->frameEnergyTH;
    newFrame->frameEnergyTH *= g.s.overallEnergyTHWeight * g.s.overallEnergyTHWeight;
}

void FullSystem::flagPointsForRemoval() {   // :1208-1270
    std::vector<FrameHessian *> fhsToMargPoints;
    for (auto fh : frames) if (fh->flaggedForMarginalization) fhsToMargPoints.push_back(fh);
    for (auto &host : frames)
        for (PointHessian *ph : host->features) {
            if (!(ph->status == PS_ACTIVE && !ph->alreadyRemoved)) continue;
            if (ph->idepth_scaled < 0 || ph->residuals.size() == 0) {
                ph->status = PS_OUTLIER;
            } else if (ph->isOOB(fhsToMargPoints, g) || host->flaggedForMarginalization) {
                if (ph->isInlierNew(g)) {
                    for (auto r : ph->residuals) {
                        r->resetOOB();
                        r->linearize(&Hcalib, g);
                        r->isLinearized = false;
                        r->applyRes(true);
                        if (r->isActive()) r->fixLinearizationF(ef);
                    }
                    if (ph->idepth_hessian > g.setting_minIdepthH_marg) ph->status = PS_MARGINALIZED;
                    else ph->status = PS_OUT;
                } else {
                    ph->status = PS_OUT;
                }
            }
        }
}

}  // namespace orc
