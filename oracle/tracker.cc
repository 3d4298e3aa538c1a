// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h header).  Pinned BIT FOR BIT to the reference-compiled CoarseTracker.cc by tests/test_ref_pin.py (see linalg.h).
// tracker.cc — restatement of src/frontend/CoarseTracker.cc:61-632.
#include "tracker.h"

namespace orc {

void CoarseTracker::makeK(CalibHessian *HCalib) {   // :219-246
    w[0] = g->wG[0];
    h[0] = g->hG[0];
    fx[0] = HCalib->fxl(); fy[0] = HCalib->fyl(); cx[0] = HCalib->cxl(); cy[0] = HCalib->cyl();
    for (int level = 1; level < pyrLevelsUsed; ++level) {
        w[level] = w[0] >> level;
        h[level] = h[0] >> level;
        fx[level] = fx[level - 1] * 0.5;
        fy[level] = fy[level - 1] * 0.5;
        cx[level] = (cx[0] + 0.5) / ((int) 1 << level) - 0.5;
        cy[level] = (cy[0] + 0.5) / ((int) 1 << level) - 0.5;
    }
    for (int level = 0; level < pyrLevelsUsed; ++level) {
        K[level].setZero();
        K[level](0, 0) = fx[level]; K[level](0, 2) = cx[level]; K[level](1, 1) = fy[level]; K[level](1, 2) = cy[level]; K[level](2, 2) = 1;
        Ki[level] = inverse3(K[level]);
        fxi[level] = Ki[level](0, 0); fyi[level] = Ki[level](1, 1); cxi[level] = Ki[level](0, 2); cyi[level] = Ki[level](1, 2);
    }
}

void CoarseTracker::makeCoarseDepthL0(const float *pts, int n) {   // :258-438
    memset(idepth[0].data(), 0, sizeof(float) * w[0] * h[0]);
    memset(weightSums[0].data(), 0, sizeof(float) * w[0] * h[0]);
    for (int i = 0; i < n; i++) {
        int u = pts[4 * i + 0] + 0.5f;
        int v = pts[4 * i + 1] + 0.5f;
        float new_idepth = pts[4 * i + 2];
        float weight = sqrtf(1e-3 / (pts[4 * i + 3] + 1e-12));
        idepth[0][u + w[0] * v] += new_idepth * weight;
        weightSums[0][u + w[0] * v] += weight;
    }
    for (int lvl = 1; lvl < pyrLevelsUsed; lvl++) {
        int lvlm1 = lvl - 1;
        int wl = w[lvl], hl = h[lvl], wlm1 = w[lvlm1];
        float *idepth_l = idepth[lvl].data(), *weightSums_l = weightSums[lvl].data();
        float *idepth_lm = idepth[lvlm1].data(), *weightSums_lm = weightSums[lvlm1].data();
        for (int y = 0; y < hl; y++)
            for (int x = 0; x < wl; x++) {
                int bidx = 2 * x + 2 * y * wlm1;
                idepth_l[x + y * wl] = idepth_lm[bidx] + idepth_lm[bidx + 1] + idepth_lm[bidx + wlm1] + idepth_lm[bidx + wlm1 + 1];
                weightSums_l[x + y * wl] = weightSums_lm[bidx] + weightSums_lm[bidx + 1] + weightSums_lm[bidx + wlm1] + weightSums_lm[bidx + wlm1 + 1];
            }
    }
    // dilate idepth by 1 (diagonal neighbours) on levels 0,1
    for (int lvl = 0; lvl < 2; lvl++) {
        int wh = w[lvl] * h[lvl] - w[lvl];
        int wl = w[lvl];
        float *weightSumsl = weightSums[lvl].data();
        float *weightSumsl_bak = weightSums_bak[lvl].data();
        memcpy(weightSumsl_bak, weightSumsl, w[lvl] * h[lvl] * sizeof(float));
        float *idepthl = idepth[lvl].data();
        for (int i = w[lvl]; i < wh; i++) {
            if (weightSumsl_bak[i] <= 0) {
                float sum = 0, num = 0, numn = 0;
                if (weightSumsl_bak[i + 1 + wl] > 0) { sum += idepthl[i + 1 + wl]; num += weightSumsl_bak[i + 1 + wl]; numn++; }
                if (weightSumsl_bak[i - 1 - wl] > 0) { sum += idepthl[i - 1 - wl]; num += weightSumsl_bak[i - 1 - wl]; numn++; }
                if (weightSumsl_bak[i + wl - 1] > 0) { sum += idepthl[i + wl - 1]; num += weightSumsl_bak[i + wl - 1]; numn++; }
                if (weightSumsl_bak[i - wl + 1] > 0) { sum += idepthl[i - wl + 1]; num += weightSumsl_bak[i - wl + 1]; numn++; }
                if (numn > 0) { idepthl[i] = sum / numn; weightSumsl[i] = num / numn; }
            }
        }
    }
    // dilate by 1 (4-neighbours) on levels >= 2
    for (int lvl = 2; lvl < pyrLevelsUsed; lvl++) {
        int wh = w[lvl] * h[lvl] - w[lvl];
        int wl = w[lvl];
        float *weightSumsl = weightSums[lvl].data();
        float *weightSumsl_bak = weightSums_bak[lvl].data();
        memcpy(weightSumsl_bak, weightSumsl, w[lvl] * h[lvl] * sizeof(float));
        float *idepthl = idepth[lvl].data();
        for (int i = w[lvl]; i < wh; i++) {
            if (weightSumsl_bak[i] <= 0) {
                float sum = 0, num = 0, numn = 0;
                if (weightSumsl_bak[i + 1] > 0) { sum += idepthl[i + 1]; num += weightSumsl_bak[i + 1]; numn++; }
                if (weightSumsl_bak[i - 1] > 0) { sum += idepthl[i - 1]; num += weightSumsl_bak[i - 1]; numn++; }
                if (weightSumsl_bak[i + wl] > 0) { sum += idepthl[i + wl]; num += weightSumsl_bak[i + wl]; numn++; }
                if (weightSumsl_bak[i - wl] > 0) { sum += idepthl[i - wl]; num += weightSumsl_bak[i - wl]; numn++; }
                if (numn > 0) { idepthl[i] = sum / numn; weightSumsl[i] = num / numn; }
            }
        }
    }
    // normalise idepths and weights, compact
    for (int lvl = 0; lvl < pyrLevelsUsed; lvl++) {
        float *weightSumsl = weightSums[lvl].data();
        float *idepthl = idepth[lvl].data();
        const float *dIRefl = lastRef_dIp[lvl];
        int wl = w[lvl], hl = h[lvl];
        int lpc_n = 0;
        float *lpc_u = pc_u[lvl].data(), *lpc_v = pc_v[lvl].data(), *lpc_idepth = pc_idepth[lvl].data(), *lpc_color = pc_color[lvl].data();
        for (int y = 2; y < hl - 2; y++)
            for (int x = 2; x < wl - 2; x++) {
                int i = x + y * wl;
                if (weightSumsl[i] > 0) {
                    idepthl[i] /= weightSumsl[i];
                    lpc_u[lpc_n] = x;
                    lpc_v[lpc_n] = y;
                    lpc_idepth[lpc_n] = idepthl[i];
                    lpc_color[lpc_n] = dIRefl[3 * i + 0];
                    if (!std::isfinite(lpc_color[lpc_n]) || !(idepthl[i] > 0)) { idepthl[i] = -1; continue; }
                    lpc_n++;
                } else
                    idepthl[i] = -1;
                weightSumsl[i] = 1;
            }
        pc_n[lvl] = lpc_n;
    }
}

Vec6 CoarseTracker::calcRes(int lvl, const SE3 &refToNew, AffLight aff_g2l, float cutoffTH) {   // :440-572
    float E = 0;
    int numTermsInE = 0, numTermsInWarped = 0, numSaturated = 0;
    int wl = w[lvl], hl = h[lvl];
    const float *dINewl = newFrame_dIp[lvl];
    float fxl = fx[lvl], fyl = fy[lvl], cxl = cx[lvl], cyl = cy[lvl];

    Mat33f RKi = (refToNew.rotationMatrix().cast<float>() * Ki[lvl]);
    Vec3f t = (refToNew.translation()).cast<float>();
    Vec2f affLL = AffLight::fromToVecExposure(lastRef_ab_exposure, newFrame_ab_exposure, lastRef_aff_g2l, aff_g2l).cast<float>();

    float sumSquaredShiftT = 0, sumSquaredShiftRT = 0, sumSquaredShiftNum = 0;
    float maxEnergy = 2 * g->s.huberTH * cutoffTH - g->s.huberTH * g->s.huberTH;

    int nl = pc_n[lvl];
    float *lpc_u = pc_u[lvl].data(), *lpc_v = pc_v[lvl].data(), *lpc_idepth = pc_idepth[lvl].data(), *lpc_color = pc_color[lvl].data();

    for (int i = 0; i < nl; i++) {
        float id = lpc_idepth[i];
        float x = lpc_u[i];
        float y = lpc_v[i];
        Vec3f xy1; xy1[0] = x; xy1[1] = y; xy1[2] = 1;
        Vec3f pt = RKi * xy1 + t * id;
        float u = pt[0] / pt[2];
        float v = pt[1] / pt[2];
        float Ku = fxl * u + cxl;
        float Kv = fyl * v + cyl;
        float new_idepth = id / pt[2];

        if (lvl == 0 && i % 32 == 0) {
            Vec3f ptT = Ki[lvl] * xy1 + t * id;
            float uT = ptT[0] / ptT[2], vT = ptT[1] / ptT[2];
            float KuT = fxl * uT + cxl, KvT = fyl * vT + cyl;
            Vec3f ptT2 = Ki[lvl] * xy1 - t * id;
            float uT2 = ptT2[0] / ptT2[2], vT2 = ptT2[1] / ptT2[2];
            float KuT2 = fxl * uT2 + cxl, KvT2 = fyl * vT2 + cyl;
            Vec3f pt3 = RKi * xy1 - t * id;
            float u3 = pt3[0] / pt3[2], v3 = pt3[1] / pt3[2];
            float Ku3 = fxl * u3 + cxl, Kv3 = fyl * v3 + cyl;
            sumSquaredShiftT += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
            sumSquaredShiftT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
            sumSquaredShiftRT += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
            sumSquaredShiftRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
            sumSquaredShiftNum += 2;
        }

        if (!(Ku > 2 && Kv > 2 && Ku < wl - 3 && Kv < hl - 3 && new_idepth > 0)) continue;

        float refColor = lpc_color[i];
        Vec3f hitColor = getInterpolatedElement33(dINewl, Ku, Kv, wl);
        if (!std::isfinite((float) hitColor[0])) continue;
        float residual = hitColor[0] - (float) (affLL[0] * refColor + affLL[1]);
        float hw = fabs(residual) < g->s.huberTH ? 1 : g->s.huberTH / fabs(residual);

        if (fabs(residual) > cutoffTH) {
            E += maxEnergy;
            numTermsInE++;
            numSaturated++;
        } else {
            E += hw * residual * residual * (2 - hw);
            numTermsInE++;
            buf_warped_idepth[numTermsInWarped] = new_idepth;
            buf_warped_u[numTermsInWarped] = u;
            buf_warped_v[numTermsInWarped] = v;
            buf_warped_dx[numTermsInWarped] = hitColor[1];
            buf_warped_dy[numTermsInWarped] = hitColor[2];
            buf_warped_residual[numTermsInWarped] = residual;
            buf_warped_weight[numTermsInWarped] = hw;
            buf_warped_refColor[numTermsInWarped] = lpc_color[i];
            numTermsInWarped++;
        }
    }
    while (numTermsInWarped % 4 != 0) {
        buf_warped_idepth[numTermsInWarped] = 0; buf_warped_u[numTermsInWarped] = 0; buf_warped_v[numTermsInWarped] = 0;
        buf_warped_dx[numTermsInWarped] = 0; buf_warped_dy[numTermsInWarped] = 0; buf_warped_residual[numTermsInWarped] = 0;
        buf_warped_weight[numTermsInWarped] = 0; buf_warped_refColor[numTermsInWarped] = 0;
        numTermsInWarped++;
    }
    buf_warped_n = numTermsInWarped;

    Vec6 rs;
    rs[0] = E;
    rs[1] = numTermsInE;
    rs[2] = sumSquaredShiftT / (sumSquaredShiftNum + 0.1);
    rs[3] = 0;
    rs[4] = sumSquaredShiftRT / (sumSquaredShiftNum + 0.1);
    rs[5] = numSaturated / (float) numTermsInE;
    return rs;
}

void CoarseTracker::calcGSSSE(int lvl, Mat88 &H_out, Vec8 &b_out, const SE3 &refToNew, AffLight aff_g2l) {   // :574-632
    acc.initialize();
    __m128 fxl = _mm_set1_ps(fx[lvl]);
    __m128 fyl = _mm_set1_ps(fy[lvl]);
    __m128 b0 = _mm_set1_ps(lastRef_aff_g2l.b);
    __m128 a = _mm_set1_ps((float) (AffLight::fromToVecExposure(lastRef_ab_exposure, newFrame_ab_exposure, lastRef_aff_g2l, aff_g2l)[0]));
    __m128 one = _mm_set1_ps(1);
    __m128 minusOne = _mm_set1_ps(-1);
    __m128 zero = _mm_set1_ps(0);
    int n = buf_warped_n;
    for (int i = 0; i < n; i += 4) {
        __m128 dx = _mm_mul_ps(_mm_loadu_ps(buf_warped_dx.data() + i), fxl);
        __m128 dy = _mm_mul_ps(_mm_loadu_ps(buf_warped_dy.data() + i), fyl);
        __m128 u = _mm_loadu_ps(buf_warped_u.data() + i);
        __m128 v = _mm_loadu_ps(buf_warped_v.data() + i);
        __m128 id = _mm_loadu_ps(buf_warped_idepth.data() + i);
        __m128 J[9];
        J[0] = _mm_mul_ps(id, dx);
        J[1] = _mm_mul_ps(id, dy);
        J[2] = _mm_sub_ps(zero, _mm_mul_ps(id, _mm_add_ps(_mm_mul_ps(u, dx), _mm_mul_ps(v, dy))));
        J[3] = _mm_sub_ps(zero, _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dx), _mm_mul_ps(dy, _mm_add_ps(one, _mm_mul_ps(v, v)))));
        J[4] = _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dy), _mm_mul_ps(dx, _mm_add_ps(one, _mm_mul_ps(u, u))));
        J[5] = _mm_sub_ps(_mm_mul_ps(u, dy), _mm_mul_ps(v, dx));
        J[6] = _mm_mul_ps(a, _mm_sub_ps(b0, _mm_loadu_ps(buf_warped_refColor.data() + i)));
        J[7] = minusOne;
        J[8] = _mm_loadu_ps(buf_warped_residual.data() + i);
        acc.updateSSE_eighted(J, _mm_loadu_ps(buf_warped_weight.data() + i));
    }
    acc.finish();
    for (int i = 0; i < 8; i++) {
        for (int j = 0; j < 8; j++) H_out(i, j) = (double) acc.H(i, j) * (double) (1.0f / n);
        b_out[i] = (double) acc.H(i, 8) * (double) (1.0f / n);
    }
    const double colScale[8] = {SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_A, SCALE_B};
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H_out(i, j) *= colScale[j];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) H_out(i, j) *= colScale[i];
    for (int i = 0; i < 8; i++) b_out[i] *= colScale[i];
}

bool CoarseTracker::trackNewestCoarse(SE3 &lastToNew_out, AffLight &aff_g2l_out, int coarsestLvl, Vec5 minResForAbort, int *iterations_out) {   // :61-217
    for (int i = 0; i < 5; i++) lastResiduals[i] = NAN;
    for (int i = 0; i < 3; i++) lastFlowIndicators[i] = 1000;
    int maxIterations[] = {10, 20, 50, 50, 50};
    float lambdaExtrapolationLimit = 0.001;
    SE3 refToNew_current = lastToNew_out;
    AffLight aff_g2l_current = aff_g2l_out;
    bool haveRepeated = false;
    int itCount = 0;

    for (int lvl = coarsestLvl; lvl >= 0; lvl--) {
        Mat88 H;
        Vec8 b;
        float levelCutoffRepeat = 1;
        Vec6 resOld = calcRes(lvl, refToNew_current, aff_g2l_current, g->s.coarseCutoffTH * levelCutoffRepeat);
        while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
            levelCutoffRepeat *= 2;
            resOld = calcRes(lvl, refToNew_current, aff_g2l_current, g->s.coarseCutoffTH * levelCutoffRepeat);
        }
        calcGSSSE(lvl, H, b, refToNew_current, aff_g2l_current);
        float lambda = 0.01;

        for (int iteration = 0; iteration < maxIterations[lvl]; iteration++) {
            itCount++;
            Mat88 Hl = H;
            for (int i = 0; i < 8; i++) Hl(i, i) *= (1 + lambda);
            Vec8 inc = ldlt_solve<8>(Hl, -b);

            if (g->s.affineOptModeA < 0 && g->s.affineOptModeB < 0) {
                Mat<double, 6, 6> H6; Mat<double, 6, 1> b6;
                for (int i = 0; i < 6; i++) { b6[i] = -b[i]; for (int j = 0; j < 6; j++) H6(i, j) = Hl(i, j); }
                auto x6 = ldlt_solve<6>(H6, b6);
                for (int i = 0; i < 6; i++) inc[i] = x6[i];
                inc[6] = 0; inc[7] = 0;
            }
            if (!(g->s.affineOptModeA < 0) && g->s.affineOptModeB < 0) {
                Mat<double, 7, 7> H7; Mat<double, 7, 1> b7;
                for (int i = 0; i < 7; i++) { b7[i] = -b[i]; for (int j = 0; j < 7; j++) H7(i, j) = Hl(i, j); }
                auto x7 = ldlt_solve<7>(H7, b7);
                for (int i = 0; i < 7; i++) inc[i] = x7[i];
                inc[7] = 0;
            }
            if (g->s.affineOptModeA < 0 && !(g->s.affineOptModeB < 0)) {
                Mat88 HlStitch = Hl;
                Vec8 bStitch = b;
                for (int i = 0; i < 8; i++) HlStitch(i, 6) = HlStitch(i, 7);
                for (int j = 0; j < 8; j++) HlStitch(6, j) = HlStitch(7, j);
                bStitch[6] = bStitch[7];
                Mat<double, 7, 7> H7; Mat<double, 7, 1> b7;
                for (int i = 0; i < 7; i++) { b7[i] = -bStitch[i]; for (int j = 0; j < 7; j++) H7(i, j) = HlStitch(i, j); }
                auto x7 = ldlt_solve<7>(H7, b7);
                inc.setZero();
                for (int i = 0; i < 6; i++) inc[i] = x7[i];
                inc[6] = 0;
                inc[7] = x7[6];
            }

            float extrapFac = 1;
            if (lambda < lambdaExtrapolationLimit) extrapFac = sqrtf(sqrt(lambdaExtrapolationLimit / lambda));
            inc *= extrapFac;

            Vec8 incScaled = inc;
            for (int i = 0; i < 3; i++) incScaled[i] *= SCALE_XI_ROT;
            for (int i = 3; i < 6; i++) incScaled[i] *= SCALE_XI_TRANS;
            incScaled[6] *= SCALE_A;
            incScaled[7] *= SCALE_B;
            if (!std::isfinite(incScaled.sum())) incScaled.setZero();

            Vec6 inc6; for (int i = 0; i < 6; i++) inc6[i] = incScaled[i];
            SE3 refToNew_new = SE3::exp(inc6) * refToNew_current;
            AffLight aff_g2l_new = aff_g2l_current;
            aff_g2l_new.a += incScaled[6];
            aff_g2l_new.b += incScaled[7];

            Vec6 resNew = calcRes(lvl, refToNew_new, aff_g2l_new, g->s.coarseCutoffTH * levelCutoffRepeat);
            bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
            if (accept) {
                calcGSSSE(lvl, H, b, refToNew_new, aff_g2l_new);
                resOld = resNew;
                aff_g2l_current = aff_g2l_new;
                refToNew_current = refToNew_new;
                lambda *= 0.5;
            } else {
                lambda *= 4;
                if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit;
            }
            if (!(inc.norm() > 1e-3)) break;
        }

        lastResiduals[lvl] = sqrtf((float) (resOld[0] / resOld[1]));
        for (int i = 0; i < 3; i++) lastFlowIndicators[i] = resOld[2 + i];
        if (lastResiduals[lvl] > 1.5 * minResForAbort[lvl]) { if (iterations_out) *iterations_out = itCount; return false; }
        if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    }
    if (iterations_out) *iterations_out = itCount;

    lastToNew_out = refToNew_current;
    aff_g2l_out = aff_g2l_current;

    if ((g->s.affineOptModeA != 0 && (fabsf(aff_g2l_out.a) > 1.2)) || (g->s.affineOptModeB != 0 && (fabsf(aff_g2l_out.b) > 200))) return false;
    Vec2f relAff = AffLight::fromToVecExposure(lastRef_ab_exposure, newFrame_ab_exposure, lastRef_aff_g2l, aff_g2l_out).cast<float>();
    if ((g->s.affineOptModeA == 0 && (fabsf(logf((float) relAff[0])) > 1.5)) || (g->s.affineOptModeB == 0 && (fabsf((float) relAff[1]) > 200))) return false;
    if (g->s.affineOptModeA < 0) aff_g2l_out.a = 0;
    if (g->s.affineOptModeB < 0) aff_g2l_out.b = 0;
    return true;
}

}  // namespace orc
