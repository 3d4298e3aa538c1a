// ORACLE — TEST INFRASTRUCTURE ONLY (not part of the product path; see linalg.h header).  Pinned to the reference-compiled translation unit by tests/test_ref_pin.py (traceOn byte-identical; trackFrame to 1e-6). The reference ships
// no tests or fixtures for the initialiser; the restatement is validated by tests/test_init_oracle.py (finite differences of the
// energy against b, explicit Schur complement, pose/depth recovery on synthetic scenes).
//
// CPU restatement of CoarseInitializer (reference src/frontend/CoarseInitializer.cc, include/frontend/CoarseInitializer.h):
//   trackFrame :40-178, calcResAndGS :181-405, calcEC :412-428, optReg :430-459, propagateUp :462-496, propagateDown :498-522,
//   resetPoints :621-643, doStep :645-671, applyStep :673-687, makeK :689-715, and the point set-up of setFirst :567-618.
// Pixel selection (PixelSelector::makeMaps, makePixelStatus) and the kd-tree queries of makeNN (:717-783, nanoflann) are upstream
// of this path: their results (positions, 10 neighbours, parent) arrive in the ldso_init_point_t records.
// Arithmetic is fp32 in the reference's operation order; poses in double (Sophus restatement of lie.h).
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include "linalg.h"
#include "lie.h"
#include "accumulators.h"
#include "../include/ldso_window.h"

namespace orc {

typedef Mat<float, 8, 8> Mat88f;
typedef Mat<float, 8, 1> Vec8f;
typedef Mat<float, 10, 1> Vec10f;

namespace {
const int kPat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};   // Setting.cc:221
const int kPatNum = 8;

inline float interp31(const float *img, float x, float y, int w) {   // GlobalFuncs.h:146-159
    int ix = (int) x, iy = (int) y;
    float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float *bp = img + 3 * (ix + iy * w);
    return dxdy * bp[3 + 3 * w] + (dy - dxdy) * bp[3 * w] + (dx - dxdy) * bp[3] + (1 - dx - dy + dxdy) * bp[0];
}
inline void interp33(const float *img, float x, float y, int w, float out[3]) {   // GlobalFuncs.h:89-103
    int ix = (int) x, iy = (int) y;
    float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float *bp = img + 3 * (ix + iy * w);
    for (int c = 0; c < 3; c++)
        out[c] = dxdy * bp[3 + 3 * w + c] + (dy - dxdy) * bp[3 * w + c] + (dx - dxdy) * bp[3 + c] + (1 - dx - dy + dxdy) * bp[c];
}

// Eigen::LDLT<Matrix<float,N,N>>::solve restated in float (same algorithm as linalg.h's double LDLT)
template <int N>
void ldlt_solve_f(const float *Ain, const float *rhs, float *x) {
    float m[N * N];
    int tr[N];
    float temp[N];
    memcpy(m, Ain, sizeof(m));
    bool zero = false;
    for (int k = 0; k < N; k++) {
        int idx = k;
        float biggest = std::fabs(m[k * N + k]);
        for (int i = k + 1; i < N; i++) if (std::fabs(m[i * N + i]) > biggest) { biggest = std::fabs(m[i * N + i]); idx = i; }
        tr[k] = idx;
        if (k != idx) {
            for (int j = 0; j < k; j++) std::swap(m[k * N + j], m[idx * N + j]);
            for (int i = idx + 1; i < N; i++) std::swap(m[i * N + k], m[i * N + idx]);
            std::swap(m[k * N + k], m[idx * N + idx]);
            for (int i = k + 1; i < idx; i++) std::swap(m[i * N + k], m[idx * N + i]);
        }
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = m[j * N + j] * m[k * N + j];
            float s = 0;
            for (int j = 0; j < k; j++) s += m[k * N + j] * temp[j];
            m[k * N + k] -= s;
            for (int i = k + 1; i < N; i++) {
                float t = 0;
                for (int j = 0; j < k; j++) t += m[i * N + j] * temp[j];
                m[i * N + k] -= t;
            }
        }
        float akk = m[k * N + k];
        bool valid = std::fabs(akk) > 0.0f;
        if (k == 0 && !valid) { for (int j = 0; j < N; j++) tr[j] = j; zero = true; break; }
        if (valid) for (int i = k + 1; i < N; i++) m[i * N + k] /= akk;
    }
    (void) zero;
    for (int i = 0; i < N; i++) x[i] = rhs[i];
    for (int k = 0; k < N; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    for (int i = 0; i < N; i++) { float s = x[i]; for (int j = 0; j < i; j++) s -= m[i * N + j] * x[j]; x[i] = s; }
    const float tol = (std::numeric_limits<float>::min)();
    for (int i = 0; i < N; i++) { if (std::fabs(m[i * N + i]) > tol) x[i] /= m[i * N + i]; else x[i] = 0; }
    for (int i = N - 1; i >= 0; i--) { float s = x[i]; for (int j = i + 1; j < N; j++) s -= m[j * N + i] * x[j]; x[i] = s; }
    for (int k = N - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
}
}  // namespace

struct CoarseInitializer {
    typedef ldso_init_point_t Pnt;
    int pyrLevelsUsed = 0;
    int w[LDSO_PYR_LEVELS], h[LDSO_PYR_LEVELS];
    double fx[LDSO_PYR_LEVELS], fy[LDSO_PYR_LEVELS], cx[LDSO_PYR_LEVELS], cy[LDSO_PYR_LEVELS];
    Mat33 K[LDSO_PYR_LEVELS], Ki[LDSO_PYR_LEVELS];
    std::vector<Pnt> points[LDSO_PYR_LEVELS];
    int numPoints[LDSO_PYR_LEVELS];
    std::vector<float> firstI[LDSO_PYR_LEVELS], newI[LDSO_PYR_LEVELS];   // dIp[lvl] of the two frames
    float first_exposure = 1, new_exposure = 1;
    SE3 thisToNext;
    float aff_a = 0, aff_b = 0;   // thisToNext_aff
    bool snapped = false;
    int snappedAt = 0, frameID = -1;
    bool fixAffine = true;
    std::vector<Vec10f> JbA, JbB;
    Vec10f *JbBuffer = nullptr, *JbBuffer_new = nullptr;
    Accumulator9 acc9, acc9SC;
    float alphaK, alphaW, regWeight, couplingWeight;
    float wM[8];
    float setting_huberTH = 9;
    int evals = 0;

    CoarseInitializer(int ww, int hh, int levels) : pyrLevelsUsed(levels) {   // :14-29
        for (int l = 0; l < LDSO_PYR_LEVELS; l++) { numPoints[l] = 0; w[l] = ww >> l; h[l] = hh >> l; }
        JbA.assign((size_t) ww * hh, Vec10f()); JbB.assign((size_t) ww * hh, Vec10f());
        JbBuffer = JbA.data(); JbBuffer_new = JbB.data();
        wM[0] = wM[1] = wM[2] = 1.0f;      // SCALE_XI_ROT
        wM[3] = wM[4] = wM[5] = 0.5f;      // SCALE_XI_TRANS
        wM[6] = 10.0f; wM[7] = 1000.0f;    // SCALE_A, SCALE_B
    }

    void makeK(float fxl, float fyl, float cxl, float cyl) {   // :689-715
        fx[0] = fxl; fy[0] = fyl; cx[0] = cxl; cy[0] = cyl;
        for (int level = 1; level < pyrLevelsUsed; ++level) {
            fx[level] = fx[level - 1] * 0.5;
            fy[level] = fy[level - 1] * 0.5;
            cx[level] = (cx[0] + 0.5) / ((int) 1 << level) - 0.5;
            cy[level] = (cy[0] + 0.5) / ((int) 1 << level) - 0.5;
        }
        for (int level = 0; level < pyrLevelsUsed; ++level) {
            K[level].setZero();
            K[level](0, 0) = fx[level]; K[level](0, 2) = cx[level]; K[level](1, 1) = fy[level]; K[level](1, 2) = cy[level]; K[level](2, 2) = 1.0;
            Ki[level] = inverse3(K[level]);
        }
    }

    // the state part of setFirst (:610-617)
    void setFirstState() {
        thisToNext = SE3(Quat(1, 0, 0, 0), Vec3());
        snapped = false;
        frameID = snappedAt = 0;
    }

    Mat<float, 3, 1> calcResAndGS(int lvl, Mat88f &H_out, Vec8f &b_out, Mat88f &H_out_sc, Vec8f &b_out_sc, const SE3 &refToNew, float ra, float rb) {   // :181-405
        evals++;
        int wl = w[lvl], hl = h[lvl];
        const float *colorRef = firstI[lvl].data();
        const float *colorNew = newI[lvl].data();
        Mat33f RKi = (refToNew.rotationMatrix() * Ki[lvl]).cast<float>();
        Mat<float, 3, 1> t = refToNew.translation().cast<float>();
        float r2new_aff[2] = {std::exp(ra), rb};
        float fxl = fx[lvl], fyl = fy[lvl], cxl = cx[lvl], cyl = cy[lvl];

        Accumulator11 E;
        acc9.initialize();
        E.initialize();
        int npts = numPoints[lvl];
        Pnt *ptsl = points[lvl].data();
        for (int i = 0; i < npts; i++) {
            Pnt *point = ptsl + i;
            point->maxstep = 1e10;
            if (!point->isGood) {
                E.updateSingle((float) (point->energy[0]));
                point->energy_new[0] = point->energy[0]; point->energy_new[1] = point->energy[1];
                point->isGood_new = false;
                continue;
            }
            alignas(16) float dp[8][8], dd[8], r[8];   // dp[k][idx]
            JbBuffer_new[i].setZero();
            bool isGood = true;
            float energy = 0;
            for (int idx = 0; idx < kPatNum; idx++) {
                int dx = kPat[idx][0], dy = kPat[idx][1];
                float px = point->u + dx, py = point->v + dy;
                float pt[3];
                for (int q = 0; q < 3; q++) pt[q] = ((RKi(q, 0) * px + RKi(q, 1) * py) + RKi(q, 2) * 1.0f) + t[q] * point->idepth_new;
                float u = pt[0] / pt[2];
                float v = pt[1] / pt[2];
                float Ku = fxl * u + cxl;
                float Kv = fyl * v + cyl;
                float new_idepth = point->idepth_new / pt[2];
                if (!(Ku > 1 && Kv > 1 && Ku < wl - 2 && Kv < hl - 2 && new_idepth > 0)) { isGood = false; break; }
                float hitColor[3];
                interp33(colorNew, Ku, Kv, wl, hitColor);
                float rlR = interp31(colorRef, point->u + dx, point->v + dy, wl);
                if (!std::isfinite(rlR) || !std::isfinite((float) hitColor[0])) { isGood = false; break; }
                float residual = hitColor[0] - r2new_aff[0] * rlR - r2new_aff[1];
                float hw = std::fabs(residual) < setting_huberTH ? 1 : setting_huberTH / std::fabs(residual);
                energy += hw * residual * residual * (2 - hw);
                float dxdd = (t[0] - t[2] * u) / pt[2];
                float dydd = (t[1] - t[2] * v) / pt[2];
                if (hw < 1) hw = sqrtf(hw);
                float dxInterp = hw * hitColor[1] * fxl;
                float dyInterp = hw * hitColor[2] * fyl;
                dp[0][idx] = new_idepth * dxInterp;
                dp[1][idx] = new_idepth * dyInterp;
                dp[2][idx] = -new_idepth * (u * dxInterp + v * dyInterp);
                dp[3][idx] = -u * v * dxInterp - (1 + v * v) * dyInterp;
                dp[4][idx] = (1 + u * u) * dxInterp + u * v * dyInterp;
                dp[5][idx] = -v * dxInterp + u * dyInterp;
                dp[6][idx] = -hw * r2new_aff[0] * rlR;
                dp[7][idx] = -hw * 1;
                dd[idx] = dxInterp * dxdd + dyInterp * dydd;
                r[idx] = hw * residual;
                float nx = dxdd * fxl, ny = dydd * fyl;
                float maxstep = 1.0f / std::sqrt(nx * nx + ny * ny);
                if (maxstep < point->maxstep) point->maxstep = maxstep;
                for (int k = 0; k < 8; k++) JbBuffer_new[i][k] += dp[k][idx] * dd[idx];
                JbBuffer_new[i][8] += r[idx] * dd[idx];
                JbBuffer_new[i][9] += dd[idx] * dd[idx];
            }
            if (!isGood || energy > point->outlierTH * 20) {
                E.updateSingle((float) (point->energy[0]));
                point->isGood_new = false;
                point->energy_new[0] = point->energy[0]; point->energy_new[1] = point->energy[1];
                continue;
            }
            E.updateSingle(energy);
            point->isGood_new = true;
            point->energy_new[0] = energy;
            for (int i4 = 0; i4 + 3 < kPatNum; i4 += 4) {
                __m128 J[9];
                for (int k = 0; k < 8; k++) J[k] = _mm_load_ps(&dp[k][i4]);
                J[8] = _mm_load_ps(&r[i4]);
                acc9.updateSSE(J);
            }
        }
        E.finish();
        acc9.finish();

        // :339-351 — the loop adds into E AFTER E.finish(): E.A keeps its value, E.num grows, EAlpha stays empty
        Accumulator11 EAlpha;
        EAlpha.initialize();
        for (int i = 0; i < npts; i++) {
            Pnt *point = ptsl + i;
            if (!point->isGood_new) {
                E.updateSingle((float) (point->energy[1]));
            } else {
                point->energy_new[1] = (point->idepth_new - 1) * (point->idepth_new - 1);
                E.updateSingle((float) (point->energy_new[1]));
            }
        }
        EAlpha.finish();
        float alphaEnergy = (float) (alphaW * (EAlpha.A + refToNew.translation().squaredNorm() * npts));
        float alphaOpt;
        if (alphaEnergy > alphaK * npts) { alphaOpt = 0; alphaEnergy = alphaK * npts; }
        else alphaOpt = alphaW;

        acc9SC.initialize();
        for (int i = 0; i < npts; i++) {
            Pnt *point = ptsl + i;
            if (!point->isGood_new) continue;
            point->lastHessian_new = JbBuffer_new[i][9];
            JbBuffer_new[i][8] += alphaOpt * (point->idepth_new - 1);
            JbBuffer_new[i][9] += alphaOpt;
            if (alphaOpt == 0) {
                JbBuffer_new[i][8] += couplingWeight * (point->idepth_new - point->iR);
                JbBuffer_new[i][9] += couplingWeight;
            }
            JbBuffer_new[i][9] = 1 / (1 + JbBuffer_new[i][9]);
            acc9SC.updateSingleWeighted(JbBuffer_new[i].d, JbBuffer_new[i][9]);
        }
        acc9SC.finish();

        for (int r_ = 0; r_ < 8; r_++) {
            for (int c_ = 0; c_ < 8; c_++) { H_out(r_, c_) = acc9.H(r_, c_); H_out_sc(r_, c_) = acc9SC.H(r_, c_); }
            b_out[r_] = acc9.H(r_, 8); b_out_sc[r_] = acc9SC.H(r_, 8);
        }
        H_out(0, 0) += alphaOpt * npts;
        H_out(1, 1) += alphaOpt * npts;
        H_out(2, 2) += alphaOpt * npts;
        Vec6 lg = refToNew.log();
        float tlog[3] = {(float) lg[0], (float) lg[1], (float) lg[2]};
        b_out[0] += tlog[0] * alphaOpt * npts;
        b_out[1] += tlog[1] * alphaOpt * npts;
        b_out[2] += tlog[2] * alphaOpt * npts;
        Mat<float, 3, 1> res; res[0] = E.A; res[1] = alphaEnergy; res[2] = (float) E.num;
        return res;
    }

    Mat<float, 3, 1> calcEC(int lvl) {   // :412-428
        Mat<float, 3, 1> res;
        if (!snapped) { res[0] = 0; res[1] = 0; res[2] = (float) numPoints[lvl]; return res; }
        AccumulatorX<2> E;
        E.initialize();
        int npts = numPoints[lvl];
        for (int i = 0; i < npts; i++) {
            Pnt *point = points[lvl].data() + i;
            if (!point->isGood_new) continue;
            float rOld = (point->idepth - point->iR);
            float rNew = (point->idepth_new - point->iR);
            Mat<float, 2, 1> v; v[0] = rOld * rOld; v[1] = rNew * rNew;
            E.updateNoWeight(v);
        }
        E.finish();
        res[0] = couplingWeight * E.A1m[0]; res[1] = couplingWeight * E.A1m[1]; res[2] = (float) E.num;
        return res;
    }

    void optReg(int lvl) {   // :430-459
        int npts = numPoints[lvl];
        Pnt *ptsl = points[lvl].data();
        if (!snapped) { for (int i = 0; i < npts; i++) ptsl[i].iR = 1; return; }
        for (int i = 0; i < npts; i++) {
            Pnt *point = ptsl + i;
            if (!point->isGood) continue;
            float idnn[10];
            int nnn = 0;
            for (int j = 0; j < 10; j++) {
                if (point->neighbours[j] == -1) continue;
                Pnt *other = ptsl + point->neighbours[j];
                if (!other->isGood) continue;
                idnn[nnn] = other->iR;
                nnn++;
            }
            if (nnn > 2) {
                std::nth_element(idnn, idnn + nnn / 2, idnn + nnn);
                point->iR = (1 - regWeight) * point->idepth + regWeight * idnn[nnn / 2];
            }
        }
    }

    void propagateUp(int srcLvl) {   // :462-496
        int nptss = numPoints[srcLvl], nptst = numPoints[srcLvl + 1];
        Pnt *ptss = points[srcLvl].data(), *ptst = points[srcLvl + 1].data();
        for (int i = 0; i < nptst; i++) { ptst[i].iR = 0; ptst[i].iRSumNum = 0; }
        for (int i = 0; i < nptss; i++) {
            Pnt *point = ptss + i;
            if (!point->isGood) continue;
            Pnt *parent = ptst + point->parent;
            parent->iR += point->iR * point->lastHessian;
            parent->iRSumNum += point->lastHessian;
        }
        for (int i = 0; i < nptst; i++) {
            Pnt *parent = ptst + i;
            if (parent->iRSumNum > 0) {
                parent->idepth = parent->iR = (parent->iR / parent->iRSumNum);
                parent->isGood = true;
            }
        }
        optReg(srcLvl + 1);
    }

    void propagateDown(int srcLvl) {   // :498-522
        int nptst = numPoints[srcLvl - 1];
        Pnt *ptss = points[srcLvl].data(), *ptst = points[srcLvl - 1].data();
        for (int i = 0; i < nptst; i++) {
            Pnt *point = ptst + i;
            Pnt *parent = ptss + point->parent;
            if (!parent->isGood || parent->lastHessian < 0.1) continue;
            if (!point->isGood) {
                point->iR = point->idepth = point->idepth_new = parent->iR;
                point->isGood = true;
                point->lastHessian = 0;
            } else {
                float newiR = (point->iR * point->lastHessian * 2 + parent->iR * parent->lastHessian) / (point->lastHessian * 2 + parent->lastHessian);
                point->iR = point->idepth = point->idepth_new = newiR;
            }
        }
        optReg(srcLvl - 1);
    }

    void resetPoints(int lvl) {   // :621-643
        Pnt *pts = points[lvl].data();
        int npts = numPoints[lvl];
        for (int i = 0; i < npts; i++) {
            pts[i].energy[0] = pts[i].energy[1] = 0;
            pts[i].idepth_new = pts[i].idepth;
            if (lvl == pyrLevelsUsed - 1 && !pts[i].isGood) {
                float snd = 0, sn = 0;
                for (int n = 0; n < 10; n++) {
                    if (pts[i].neighbours[n] == -1 || !pts[pts[i].neighbours[n]].isGood) continue;
                    snd += pts[pts[i].neighbours[n]].iR;
                    sn += 1;
                }
                if (sn > 0) {
                    pts[i].isGood = true;
                    pts[i].iR = pts[i].idepth = pts[i].idepth_new = snd / sn;
                }
            }
        }
    }

    void doStep(int lvl, float lambda, const Vec8f &inc) {   // :645-671
        const float maxPixelStep = 0.25;
        const float idMaxStep = 1e10;
        Pnt *pts = points[lvl].data();
        int npts = numPoints[lvl];
        for (int i = 0; i < npts; i++) {
            if (!pts[i].isGood) continue;
            // Eigen's vectorised 8-float dot: two 4-lane products added, then the (0+2)+(1+3) horizontal sum
            float p4[4];
            for (int k = 0; k < 4; k++) p4[k] = JbBuffer[i][k] * inc[k] + JbBuffer[i][k + 4] * inc[k + 4];
            float dot = (p4[0] + p4[2]) + (p4[1] + p4[3]);
            float b = JbBuffer[i][8] + dot;
            float step = -b * JbBuffer[i][9] / (1 + lambda);
            float maxstep = maxPixelStep * pts[i].maxstep;
            if (maxstep > idMaxStep) maxstep = idMaxStep;
            if (step > maxstep) step = maxstep;
            if (step < -maxstep) step = -maxstep;
            float newIdepth = pts[i].idepth + step;
            if (newIdepth < 1e-3) newIdepth = 1e-3;
            if (newIdepth > 50) newIdepth = 50;
            pts[i].idepth_new = newIdepth;
        }
    }

    void applyStep(int lvl) {   // :673-687
        Pnt *pts = points[lvl].data();
        int npts = numPoints[lvl];
        for (int i = 0; i < npts; i++) {
            if (!pts[i].isGood) { pts[i].idepth = pts[i].idepth_new = pts[i].iR; continue; }
            pts[i].energy[0] = pts[i].energy_new[0]; pts[i].energy[1] = pts[i].energy_new[1];
            pts[i].isGood = pts[i].isGood_new;
            pts[i].idepth = pts[i].idepth_new;
            pts[i].lastHessian = pts[i].lastHessian_new;
        }
        std::swap(JbBuffer, JbBuffer_new);
    }

    bool trackFrame() {   // :40-178
        int maxIterations[] = {5, 5, 10, 30, 50};
        alphaK = 2.5 * 2.5;
        alphaW = 150 * 150;
        regWeight = 0.8;
        couplingWeight = 1;
        evals = 0;
        if (!snapped) {
            thisToNext.translation().setZero();
            for (int lvl = 0; lvl < pyrLevelsUsed; lvl++) {
                Pnt *ptsl = points[lvl].data();
                for (int i = 0; i < numPoints[lvl]; i++) { ptsl[i].iR = 1; ptsl[i].idepth_new = 1; ptsl[i].lastHessian = 0; }
            }
        }
        SE3 refToNew_current = thisToNext;
        float cur_a = aff_a, cur_b = aff_b;
        if (first_exposure > 0 && new_exposure > 0) { cur_a = logf(new_exposure / first_exposure); cur_b = 0; }

        for (int lvl = pyrLevelsUsed - 1; lvl >= 0; lvl--) {
            if (lvl < pyrLevelsUsed - 1) propagateDown(lvl + 1);
            Mat88f H, Hsc;
            Vec8f b, bsc;
            resetPoints(lvl);
            Mat<float, 3, 1> resOld = calcResAndGS(lvl, H, b, Hsc, bsc, refToNew_current, cur_a, cur_b);
            applyStep(lvl);
            float lambda = 0.1;
            float eps = 1e-4;
            int fails = 0;
            int iteration = 0;
            while (true) {
                Mat88f Hl = H;
                for (int i = 0; i < 8; i++) Hl(i, i) *= (1 + lambda);
                Hl -= Hsc * (1 / (1 + lambda));
                Vec8f bl = b - bsc * (1 / (1 + lambda));
                const float sc = (0.01f / (w[lvl] * h[lvl]));
                for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) Hl(i, j) = ((wM[i] * Hl(i, j)) * wM[j]) * sc;
                for (int i = 0; i < 8; i++) bl[i] = (wM[i] * bl[i]) * sc;
                Vec8f inc;
                if (fixAffine) {
                    float A6[36], b6[6], x6[6];
                    for (int i = 0; i < 6; i++) { b6[i] = bl[i]; for (int j = 0; j < 6; j++) A6[i * 6 + j] = Hl(i, j); }
                    ldlt_solve_f<6>(A6, b6, x6);
                    for (int i = 0; i < 6; i++) inc[i] = -(wM[i] * x6[i]);
                    inc[6] = inc[7] = 0;
                } else {
                    float x8[8];
                    ldlt_solve_f<8>(Hl.d, bl.d, x8);
                    for (int i = 0; i < 8; i++) inc[i] = -(wM[i] * x8[i]);
                }
                Vec6 inc6; for (int i = 0; i < 6; i++) inc6[i] = (double) inc[i];
                SE3 refToNew_new = SE3::exp(inc6) * refToNew_current;
                float new_a = cur_a, new_b = cur_b;
                new_a += inc[6];
                new_b += inc[7];
                doStep(lvl, lambda, inc);

                Mat88f H_new, Hsc_new;
                Vec8f b_new, bsc_new;
                Mat<float, 3, 1> resNew = calcResAndGS(lvl, H_new, b_new, Hsc_new, bsc_new, refToNew_new, new_a, new_b);
                Mat<float, 3, 1> regEnergy = calcEC(lvl);
                float eTotalNew = (resNew[0] + resNew[1] + regEnergy[1]);
                float eTotalOld = (resOld[0] + resOld[1] + regEnergy[0]);
                bool accept = eTotalOld > eTotalNew;
                if (accept) {
                    if (resNew[1] == alphaK * numPoints[lvl]) snapped = true;
                    H = H_new; b = b_new; Hsc = Hsc_new; bsc = bsc_new;
                    resOld = resNew;
                    cur_a = new_a; cur_b = new_b;
                    refToNew_current = refToNew_new;
                    applyStep(lvl);
                    optReg(lvl);
                    lambda *= 0.5;
                    fails = 0;
                    if (lambda < 0.0001) lambda = 0.0001;
                } else {
                    fails++;
                    lambda *= 4;
                    if (lambda > 10000) lambda = 10000;
                }
                bool quitOpt = false;
                if (!(inc.norm() > eps) || iteration >= maxIterations[lvl] || fails >= 2) quitOpt = true;
                if (quitOpt) break;
                iteration++;
            }
        }
        thisToNext = refToNew_current;
        aff_a = cur_a; aff_b = cur_b;
        for (int i = 0; i < pyrLevelsUsed - 1; i++) propagateUp(i);
        frameID++;
        if (!snapped) snappedAt = 0;
        if (snapped && snappedAt == 0) snappedAt = frameID;
        return snapped && frameID > snappedAt + 5;
    }
};

}  // namespace orc

using orc::CoarseInitializer;

extern "C" {

void *orc_init_create(int w, int h, int levels) { return new CoarseInitializer(w, h, levels); }
void orc_init_destroy(void *p) { delete (CoarseInitializer *) p; }

// dIp: `levels` pointers to (w>>l)*(h>>l)*3 floats (FrameHessian::dIp)
void orc_init_set_first(void *p, const float *calib, const float *const *dIp, float exposure, const ldso_init_point_t *const *points, const int *n_points,
                        float huberTH, int fixAffine) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    c->makeK(calib[0], calib[1], calib[2], calib[3]);
    c->setting_huberTH = huberTH;
    c->fixAffine = fixAffine != 0;
    c->first_exposure = exposure;
    for (int l = 0; l < c->pyrLevelsUsed; l++) {
        c->firstI[l].assign(dIp[l], dIp[l] + (size_t) c->w[l] * c->h[l] * 3);
        c->points[l].assign(points[l], points[l] + n_points[l]);
        c->numPoints[l] = n_points[l];
    }
    c->setFirstState();
}
void orc_init_set_new_frame(void *p, const float *const *dIp, float exposure) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    c->new_exposure = exposure;
    for (int l = 0; l < c->pyrLevelsUsed; l++) c->newI[l].assign(dIp[l], dIp[l] + (size_t) c->w[l] * c->h[l] * 3);
}
static void fill_state(CoarseInitializer *c, ldso_init_state_t *s, int ready) {
    orc::Mat33 R = c->thisToNext.rotationMatrix();
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) s->thisToNext[i * 4 + j] = R(i, j); s->thisToNext[i * 4 + 3] = c->thisToNext.translation()[i]; }
    s->aff_a = c->aff_a; s->aff_b = c->aff_b;
    s->snapped = c->snapped; s->snappedAt = c->snappedAt; s->frameID = c->frameID; s->ready = ready; s->evals = c->evals; s->pad_ = 0;
}
int orc_init_track_frame(void *p, ldso_init_state_t *state_out) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    int r = c->trackFrame() ? 1 : 0;
    if (state_out) fill_state(c, state_out, r);
    return r;
}
void orc_init_get_state(void *p, ldso_init_state_t *s) { CoarseInitializer *c = (CoarseInitializer *) p; fill_state(c, s, c->snapped && c->frameID > c->snappedAt + 5); }
void orc_init_set_state(void *p, const ldso_init_state_t *s) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    double m[12]; memcpy(m, s->thisToNext, sizeof(m));
    c->thisToNext = orc::SE3::fromMatrix34(m);
    c->aff_a = (float) s->aff_a; c->aff_b = (float) s->aff_b;
    c->snapped = s->snapped != 0; c->snappedAt = s->snappedAt; c->frameID = s->frameID;
}
void orc_init_get_points(void *p, int lvl, ldso_init_point_t *out) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    memcpy(out, c->points[lvl].data(), sizeof(ldso_init_point_t) * c->numPoints[lvl]);
}
void orc_init_set_points(void *p, int lvl, const ldso_init_point_t *in) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    memcpy(c->points[lvl].data(), in, sizeof(ldso_init_point_t) * c->numPoints[lvl]);
}
// one calcResAndGS + calcEC on the current point state (stage parity)
void orc_init_calc_res_and_gs(void *p, int lvl, const double *refToNew, double a, double b, float *H, float *bo, float *Hsc, float *bsc, float *res, float *ec) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    c->alphaK = 2.5 * 2.5; c->alphaW = 150 * 150; c->regWeight = 0.8; c->couplingWeight = 1;
    orc::Mat88f Hm, Hs; orc::Vec8f bm, bs;
    orc::Mat<float, 3, 1> r = c->calcResAndGS(lvl, Hm, bm, Hs, bs, orc::SE3::fromMatrix34(refToNew), (float) a, (float) b);
    orc::Mat<float, 3, 1> e = c->calcEC(lvl);
    for (int i = 0; i < 64; i++) { H[i] = Hm.d[i]; Hsc[i] = Hs.d[i]; }
    for (int i = 0; i < 8; i++) { bo[i] = bm[i]; bsc[i] = bs[i]; }
    for (int i = 0; i < 3; i++) { res[i] = r[i]; ec[i] = e[i]; }
}
void orc_init_get_jb(void *p, int lvl, float *out /* n x 10, JbBuffer_new */) {
    CoarseInitializer *c = (CoarseInitializer *) p;
    for (int i = 0; i < c->numPoints[lvl]; i++) for (int k = 0; k < 10; k++) out[i * 10 + k] = c->JbBuffer_new[i][k];
}

}  // extern "C"
