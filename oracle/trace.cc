// ORACLE — TEST INFRASTRUCTURE ONLY (not part of the product path).  Pinned to the reference-compiled translation unit by tests/test_ref_pin.py (traceOn byte-identical; trackFrame to 1e-6). The reference ships no tests or
// fixtures for this function; the restatement is validated by tests/test_trace_oracle.py (depth recovery on synthetic scenes).
//
// CPU restatement of ImmaturePoint::traceOn (reference src/internal/ImmaturePoint.cc:47-310) and of the loop of
// FullSystem::traceNewCoarse over the immature points of the window (src/frontend/FullSystem.cc:1012-1050), on the
// plain-C records of include/ldso_window.h.  Arithmetic is fp32 in the reference's operation order.
#include <cmath>
#include <cstring>
#include "../include/ldso_window.h"

namespace {

// staticPattern[8] (Setting.cc:221)
const int kPat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

// getInterpolatedElement31 (GlobalFuncs.h:146-159): channel 0, bilinear
inline float interp31(const float *img, float x, float y, int w) {
    int ix = (int) x, iy = (int) y;
    float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float *bp = img + 3 * (ix + iy * w);
    return dxdy * bp[3 + 3 * w] + (dy - dxdy) * bp[3 * w] + (dx - dxdy) * bp[3] + (1 - dx - dy + dxdy) * bp[0];
}
// getInterpolatedElement33 (GlobalFuncs.h:89-103): all three channels
inline void interp33(const float *img, float x, float y, int w, float out[3]) {
    int ix = (int) x, iy = (int) y;
    float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float *bp = img + 3 * (ix + iy * w);
    for (int c = 0; c < 3; c++)
        out[c] = dxdy * bp[3 + 3 * w + c] + (dy - dxdy) * bp[3 * w + c] + (dx - dxdy) * bp[3 + c] + (1 - dx - dy + dxdy) * bp[c];
}

inline int fail(ldso_immature_t &p, int status) { p.lastTraceUV[0] = -1; p.lastTraceUV[1] = -1; p.lastTracePixelInterval = 0; return p.lastTraceStatus = status; }

// ImmaturePoint.cc:47-310
int trace_on(ldso_immature_t &p, const float *dI, int w, int h, const float *KRKi, const float *Kt, const float *aff, const ldso_trace_settings_t &s) {
    if (p.lastTraceStatus == LDSO_IPS_OOB) return p.lastTraceStatus;                                   // :53
    const float maxPixSearch = (w + h) * s.maxPixSearch;                                             // :54
    // ---- project idepth_min / idepth_max (:59-124) ----
    const float pr[3] = {(KRKi[0] * p.u + KRKi[1] * p.v) + KRKi[2] * 1.0f, (KRKi[3] * p.u + KRKi[4] * p.v) + KRKi[5] * 1.0f,
                         (KRKi[6] * p.u + KRKi[7] * p.v) + KRKi[8] * 1.0f};
    const float ptpMin[3] = {pr[0] + Kt[0] * p.idepth_min, pr[1] + Kt[1] * p.idepth_min, pr[2] + Kt[2] * p.idepth_min};
    const float uMin = ptpMin[0] / ptpMin[2], vMin = ptpMin[1] / ptpMin[2];
    if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) return fail(p, LDSO_IPS_OOB);
    float dist, uMax, vMax;
    if (std::isfinite(p.idepth_max)) {
        const float ptpMax[3] = {pr[0] + Kt[0] * p.idepth_max, pr[1] + Kt[1] * p.idepth_max, pr[2] + Kt[2] * p.idepth_max};
        uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) return fail(p, LDSO_IPS_OOB);
        dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
        dist = sqrtf(dist);
        if (dist < s.trace_slackInterval) {                                                          // :91-96
            p.lastTraceUV[0] = (uMax + uMin) * 0.5f; p.lastTraceUV[1] = (vMax + vMin) * 0.5f;
            p.lastTracePixelInterval = dist;
            return p.lastTraceStatus = LDSO_IPS_SKIPPED;
        }
    } else {
        dist = maxPixSearch;
        const float q[3] = {pr[0] + Kt[0] * 0.01f, pr[1] + Kt[1] * 0.01f, pr[2] + Kt[2] * 0.01f};     // direction from an arbitrary depth
        uMax = q[0] / q[2]; vMax = q[1] / q[2];
        float dx = uMax - uMin, dy = vMax - vMin;
        float d = 1.0f / sqrtf(dx * dx + dy * dy);
        uMax = uMin + dist * dx * d; vMax = vMin + dist * dy * d;
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) return fail(p, LDSO_IPS_OOB);
    }
    if (!(p.idepth_min < 0 || (ptpMin[2] > 0.75f && ptpMin[2] < 1.5f))) return fail(p, LDSO_IPS_OOB);    // scale change (:127-131)
    // ---- error bound in pixels (:134-147) ----
    float dx = s.trace_stepsize * (uMax - uMin), dy = s.trace_stepsize * (vMax - vMin);
    const float *g = p.gradH;
    // v^T gradH v evaluated as (v^T gradH) v, v = (dx,dy) resp. (dy,-dx)  (gradH row-major)
    float a = (dx * g[0] + dy * g[2]) * dx + (dx * g[1] + dy * g[3]) * dy;
    float b = (dy * g[0] + -dx * g[2]) * dy + (dy * g[1] + -dx * g[3]) * -dx;
    float errorInPixel = 0.2f + 0.2f * (a + b) / a;
    if (errorInPixel * s.trace_minImprovementFactor > dist && std::isfinite(p.idepth_max)) {
        p.lastTraceUV[0] = (uMax + uMin) * 0.5f; p.lastTraceUV[1] = (vMax + vMin) * 0.5f;
        p.lastTracePixelInterval = dist;
        return p.lastTraceStatus = LDSO_IPS_BADCONDITION;
    }
    if (errorInPixel > 10) errorInPixel = 10;
    // ---- discrete search (:150-205) ----
    dx /= dist; dy /= dist;
    if (dist > maxPixSearch) { uMax = uMin + maxPixSearch * dx; vMax = vMin + maxPixSearch * dy; dist = maxPixSearch; }
    int numSteps = (int) (1.9999f + dist / s.trace_stepsize);
    float randShift = uMin * 1000 - floorf(uMin * 1000);
    float ptx = uMin - randShift * dx, pty = vMin - randShift * dy;
    float rot[8][2];
    for (int i = 0; i < 8; i++) { rot[i][0] = KRKi[0] * kPat[i][0] + KRKi[1] * kPat[i][1]; rot[i][1] = KRKi[3] * kPat[i][0] + KRKi[4] * kPat[i][1]; }
    if (!std::isfinite(dx) || !std::isfinite(dy)) return fail(p, LDSO_IPS_OOB);
    float errors[100];
    float bestU = 0, bestV = 0, bestEnergy = 1e10f;
    int bestIdx = -1;
    if (numSteps >= 100) numSteps = 99;
    for (int i = 0; i < numSteps; i++) {
        float energy = 0;
        for (int k = 0; k < 8; k++) {
            float hit = interp31(dI, ptx + rot[k][0], pty + rot[k][1], w);
            if (!std::isfinite(hit)) { energy += 1e5f; continue; }
            float r = hit - (float) (aff[0] * p.color[k] + aff[1]);
            float hw = fabsf(r) < s.huberTH ? 1 : s.huberTH / fabsf(r);
            energy += hw * r * r * (2 - hw);
        }
        errors[i] = energy;
        if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = i; }
        ptx += dx; pty += dy;
    }
    float secondBest = 1e10f;                                                                        // best score outside +-radius (:208-215)
    for (int i = 0; i < numSteps; i++)
        if ((i < bestIdx - s.minTraceTestRadius || i > bestIdx + s.minTraceTestRadius) && errors[i] < secondBest) secondBest = errors[i];
    float newQuality = secondBest / bestEnergy;
    if (newQuality < p.quality || numSteps > 10) p.quality = newQuality;
    // ---- Gauss-Newton refinement along the line (:218-268) ----
    float uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;
    if (s.trace_GNIterations > 0) bestEnergy = 1e5f;
    for (int it = 0; it < s.trace_GNIterations; it++) {
        float H = 1, bb = 0, energy = 0;
        for (int k = 0; k < 8; k++) {
            float hit[3];
            interp33(dI, bestU + rot[k][0], bestV + rot[k][1], w, hit);
            if (!std::isfinite(hit[0])) { energy += 1e5f; continue; }
            float r = hit[0] - (aff[0] * p.color[k] + aff[1]);
            float dResdDist = dx * hit[1] + dy * hit[2];
            float hw = fabsf(r) < s.huberTH ? 1 : s.huberTH / fabsf(r);
            H += hw * dResdDist * dResdDist;
            bb += hw * r * dResdDist;
            energy += p.weights[k] * p.weights[k] * hw * r * r * (2 - hw);
        }
        if (energy > bestEnergy) {
            stepBack *= 0.5f;
            bestU = uBak + stepBack * dx; bestV = vBak + stepBack * dy;
        } else {
            float step = -gnstepsize * bb / H;
            if (step < -0.5f) step = -0.5f; else if (step > 0.5f) step = 0.5f;
            if (!std::isfinite(step)) step = 0;
            uBak = bestU; vBak = bestV; stepBack = step;
            bestU += step * dx; bestV += step * dy;
            bestEnergy = energy;
        }
        if (fabsf(stepBack) < s.trace_GNThreshold) break;
    }
    // ---- energy-based outlier (:271-278) ----
    if (!(bestEnergy < p.energyTH * s.trace_extraSlackOnTH)) {
        int prev = p.lastTraceStatus;
        return fail(p, prev == LDSO_IPS_OUTLIER ? LDSO_IPS_OOB : LDSO_IPS_OUTLIER);
    }
    // ---- new interval (:281-300) ----
    if (dx * dx > dy * dy) {
        p.idepth_min = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
        p.idepth_max = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
    } else {
        p.idepth_min = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
        p.idepth_max = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
    }
    if (p.idepth_min > p.idepth_max) { float t = p.idepth_min; p.idepth_min = p.idepth_max; p.idepth_max = t; }
    if (!std::isfinite(p.idepth_min) || !std::isfinite(p.idepth_max) || (p.idepth_max < 0)) return fail(p, LDSO_IPS_OUTLIER);
    p.lastTracePixelInterval = 2 * errorInPixel;
    p.lastTraceUV[0] = bestU; p.lastTraceUV[1] = bestV;
    return p.lastTraceStatus = LDSO_IPS_GOOD;
}

}  // namespace

extern "C" {

void orc_trace_settings_default(ldso_trace_settings_t *s) {
    memset(s, 0, sizeof(*s));
    s->maxPixSearch = 0.027f; s->trace_stepsize = 1.0f; s->trace_GNThreshold = 0.1f; s->trace_extraSlackOnTH = 1.2f;
    s->trace_slackInterval = 1.5f; s->trace_minImprovementFactor = 2; s->huberTH = 9; s->trace_GNIterations = 3; s->minTraceTestRadius = 2;
}

// FullSystem::traceNewCoarse (FullSystem.cc:1012-1050): every immature point against the new frame; counts[6] per status
void orc_trace_on(int n, ldso_immature_t *pts, const float *dI, int w, int h, int n_hosts, const float *KRKi, const float *Kt, const float *aff,
                  const ldso_trace_settings_t *s, int *counts) {
    if (counts) for (int i = 0; i < 6; i++) counts[i] = 0;
    for (int i = 0; i < n; i++) {
        int hst = pts[i].host;
        if (hst < 0 || hst >= n_hosts) continue;
        int st = trace_on(pts[i], dI, w, h, KRKi + 9 * hst, Kt + 3 * hst, aff + 2 * hst, *s);
        if (counts && st >= 0 && st < 6) counts[st]++;
    }
}

}  // extern "C"
static_assert(sizeof(ldso_immature_t) == 128 && sizeof(ldso_trace_settings_t) == 40, "layout");

// =====================================================================================================================
// FullSystem::optimizeImmaturePoint (src/frontend/FullSystem.cc:892-1010) with ImmaturePoint::linearizeResidual
// (src/internal/ImmaturePoint.cc:312-381), projectPoint (include/internal/ResidualProjections.h:57-84) and derive_idepth
// (:12-18) on plain arrays.  pairs[host*F + target] = {R 9 (PRE_RTll), t 3 (PRE_tTll), aff 2 (PRE_aff_mode)}.
// =====================================================================================================================
namespace {

struct TmpRes { int state_state; double state_energy; int state_NewState; double state_NewEnergy; int target; };
enum { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };

double linearize_residual(const ldso_immature_t &p, const float *const *dI, int w, int h, const float *K4 /*fx fy cx cy*/, const float *pair,
                          float huberTH, float outlierTHSlack, TmpRes &tr, float &Hdd, float &bd, float idepth) {
    if (tr.state_state == RS_OOB) { tr.state_NewState = RS_OOB; return tr.state_energy; }                      // :317-320
    const float *R = pair, *t = pair + 9, *aff = pair + 12;
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3], fxi = 1.0f / fx, fyi = 1.0f / fy;
    const float *img = dI[tr.target];
    float energyLeft = 0;
    for (int k = 0; k < 8; k++) {
        const int dx = kPat[k][0], dy = kPat[k][1];
        // projectPoint (ResidualProjections.h:57-84)
        const float Kl0 = (p.u + dx - cx) * fxi, Kl1 = (p.v + dy - cy) * fyi;
        const float q0 = (R[0] * Kl0 + R[1] * Kl1) + R[2] * 1.0f + t[0] * idepth, q1 = (R[3] * Kl0 + R[4] * Kl1) + R[5] * 1.0f + t[1] * idepth,
                    q2 = (R[6] * Kl0 + R[7] * Kl1) + R[8] * 1.0f + t[2] * idepth;
        const float drescale = 1.0f / q2;
        bool ok = drescale > 0;
        float u = 0, v = 0, Ku = 0, Kv = 0;
        if (ok) { u = q0 * drescale; v = q1 * drescale; Ku = u * fx + cx; Kv = v * fy + cy; ok = Ku > 1.1f && Kv > 1.1f && Ku < w - 3 && Kv < h - 3; }
        if (!ok) { tr.state_NewState = RS_OOB; return tr.state_energy; }
        float hit[3];
        interp33(img, Ku, Kv, w, hit);
        if (!std::isfinite(hit[0])) { tr.state_NewState = RS_OOB; return tr.state_energy; }
        const float residual = hit[0] - (aff[0] * p.color[k] + aff[1]);
        float hw = fabsf(residual) < huberTH ? 1 : huberTH / fabsf(residual);
        energyLeft += p.weights[k] * p.weights[k] * hw * residual * residual * (2 - hw);
        const float dxInterp = hit[1] * fx, dyInterp = hit[2] * fy;
        const float d_idepth = (dxInterp * drescale * (t[0] - t[2] * u) + dyInterp * drescale * (t[1] - t[2] * v)) * 1.0f;      // SCALE_IDEPTH = 1
        hw *= p.weights[k] * p.weights[k];
        Hdd += (hw * d_idepth) * d_idepth;
        bd += (hw * residual) * d_idepth;
    }
    if (energyLeft > p.energyTH * outlierTHSlack) { energyLeft = p.energyTH * outlierTHSlack; tr.state_NewState = RS_OUTLIER; }
    else tr.state_NewState = RS_IN;
    tr.state_NewEnergy = energyLeft;
    return energyLeft;
}

}  // namespace

extern "C" void orc_activate_points(int n, const ldso_immature_t *pts, int F, const float *const *dI, int w, int h, const float *K4, const float *pairs,
                                    float huberTH, float minIdepthH_act, int GNIts, int minObs, ldso_activation_t *out) {
    for (int i = 0; i < n; i++) {
        const ldso_immature_t &p = pts[i];
        ldso_activation_t &o = out[i];
        memset(&o, 0, sizeof(o));
        for (int f = 0; f < LDSO_MAX_FRAMES; f++) o.res_state[f] = -1;
        TmpRes tr[LDSO_MAX_FRAMES];
        int nres = 0;
        for (int f = 0; f < F; f++) if (f != p.host) { tr[nres].state_NewEnergy = tr[nres].state_energy = 0; tr[nres].state_NewState = RS_OUTLIER; tr[nres].state_state = RS_IN; tr[nres].target = f; nres++; }
        float lastEnergy = 0, lastHdd = 0, lastbd = 0;
        float currentIdepth = (p.idepth_max + p.idepth_min) * 0.5f;
        for (int r = 0; r < nres; r++) {
            lastEnergy += linearize_residual(p, dI, w, h, K4, pairs + (size_t) (p.host * F + tr[r].target) * 14, huberTH, 1000, tr[r], lastHdd, lastbd, currentIdepth);
            tr[r].state_state = tr[r].state_NewState; tr[r].state_energy = tr[r].state_NewEnergy;
        }
        bool failed = !std::isfinite(lastEnergy) || lastHdd < minIdepthH_act;                                       // return 0 (:924-926)
        float lambda = 0.1f;
        int its = 0;
        for (int iteration = 0; !failed && iteration < GNIts; iteration++) {
            its++;
            float H = lastHdd;
            H *= 1 + lambda;
            float step = (1.0 / H) * lastbd;
            float newIdepth = currentIdepth - step;
            float newHdd = 0, newbd = 0, newEnergy = 0;
            for (int r = 0; r < nres; r++)
                newEnergy += linearize_residual(p, dI, w, h, K4, pairs + (size_t) (p.host * F + tr[r].target) * 14, huberTH, 1, tr[r], newHdd, newbd, newIdepth);
            if (!std::isfinite(lastEnergy) || newHdd < minIdepthH_act) { failed = true; break; }                       // :945-947
            if (newEnergy < lastEnergy) {
                currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
                for (int r = 0; r < nres; r++) { tr[r].state_state = tr[r].state_NewState; tr[r].state_energy = tr[r].state_NewEnergy; }
                lambda *= 0.5f;
            } else lambda *= 5;
            if (fabsf(step) < 0.0001 * currentIdepth) break;
        }
        // outputs: the loop state at exit, whatever the verdict
        o.idepth = currentIdepth; o.energy = lastEnergy; o.Hdd = lastHdd; o.bd = lastbd; o.iterations = its;
        int good = 0;
        for (int r = 0; r < nres; r++) { o.res_state[tr[r].target] = tr[r].state_state; if (tr[r].state_state == RS_IN) good++; }
        o.numGoodRes = good;
        o.ok = (!failed && std::isfinite(currentIdepth) && good >= minObs) ? 1 : 0;                                     // :968-980
    }
}
static_assert(sizeof(ldso_activation_t) == 96, "layout");
