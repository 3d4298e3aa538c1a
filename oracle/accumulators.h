// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h header).
// accumulators.h — the fp32 "shift-up" accumulators of
// include/internal/OptimizationBackend/MatrixAccumulators.h, restated with the same three-level
// (1 / 1k / 1M) flush schedule on float counters:
//   AccumulatorXX<i,j> (:20-66), Accumulator11 (:68-142), AccumulatorX<i> (:145-197),
//   AccumulatorApprox (:749-1101), Accumulator9 (:1104-1645; updateSSE_eighted for the tracker, updateSSE / updateSingleWeighted for the initialiser).
#pragma once
#include <xmmintrin.h>
#include "linalg.h"

namespace orc {

template <int I, int J>
struct AccumulatorXX {
    Mat<float, I, J> A, A1k, A1m;
    size_t num;
    float numIn1, numIn1k, numIn1m;
    void initialize() { A.setZero(); A1k.setZero(); A1m.setZero(); num = 0; numIn1 = numIn1k = numIn1m = 0; }
    void finish() { shiftUp(true); num = (size_t) (numIn1 + numIn1k + numIn1m); }
    // A += w * L * R^T  (MatrixAccumulators.h:43-47)
    void update(const Mat<float, I, 1> &L, const Mat<float, J, 1> &R, float w) {
        for (int i = 0; i < I; i++) { float wl = w * L[i]; for (int j = 0; j < J; j++) A(i, j) += wl * R[j]; }
        numIn1++;
        shiftUp(false);
    }
    void shiftUp(bool force) {
        if (numIn1 > 1000 || force) { A1k += A; A.setZero(); numIn1k += numIn1; numIn1 = 0; }
        if (numIn1k > 1000 || force) { A1m += A1k; A1k.setZero(); numIn1m += numIn1k; numIn1k = 0; }
    }
};

template <int I>
struct AccumulatorX {
    Mat<float, I, 1> A, A1k, A1m;
    size_t num;
    float numIn1, numIn1k, numIn1m;
    void initialize() { A.setZero(); A1k.setZero(); A1m.setZero(); num = 0; numIn1 = numIn1k = numIn1m = 0; }
    void finish() { shiftUp(true); num = (size_t) (numIn1 + numIn1k + numIn1m); }
    void update(const Mat<float, I, 1> &L, float w) {   // :170-174
        for (int i = 0; i < I; i++) A[i] += w * L[i];
        numIn1++;
        shiftUp(false);
    }
    void updateNoWeight(const Mat<float, I, 1> &L) {   // :174-178
        for (int i = 0; i < I; i++) A[i] += L[i];
        numIn1++;
        shiftUp(false);
    }
    void shiftUp(bool force) {
        if (numIn1 > 1000 || force) { A1k += A; A.setZero(); numIn1k += numIn1; numIn1 = 0; }
        if (numIn1k > 1000 || force) { A1m += A1k; A1k.setZero(); numIn1m += numIn1k; numIn1k = 0; }
    }
};

struct Accumulator11 {
    float A;
    size_t num;
    alignas(16) float SSEData[4], SSEData1k[4], SSEData1m[4];
    float numIn1, numIn1k, numIn1m;
    void initialize() {
        A = 0; memset(SSEData, 0, sizeof(SSEData)); memset(SSEData1k, 0, sizeof(SSEData1k)); memset(SSEData1m, 0, sizeof(SSEData1m));
        num = 0; numIn1 = numIn1k = numIn1m = 0;
    }
    void finish() { shiftUp(true); A = SSEData1m[0] + SSEData1m[1] + SSEData1m[2] + SSEData1m[3]; }
    void updateSingle(const float val) { SSEData[0] += val; num++; numIn1++; shiftUp(false); }
    void updateSingleNoShift(const float val) { SSEData[0] += val; num++; numIn1++; }
    void updateSSENoShift(const __m128 val) { _mm_store_ps(SSEData, _mm_add_ps(_mm_load_ps(SSEData), val)); num += 4; numIn1++; }
    void shiftUp(bool force) {
        if (numIn1 > 1000 || force) {
            _mm_store_ps(SSEData1k, _mm_add_ps(_mm_load_ps(SSEData), _mm_load_ps(SSEData1k)));
            numIn1k += numIn1; numIn1 = 0; memset(SSEData, 0, sizeof(SSEData));
        }
        if (numIn1k > 1000 || force) {
            _mm_store_ps(SSEData1m, _mm_add_ps(_mm_load_ps(SSEData1k), _mm_load_ps(SSEData1m)));
            numIn1m += numIn1k; numIn1k = 0; memset(SSEData1k, 0, sizeof(SSEData1k));
        }
    }
};

// 13x13 symmetric "approx" accumulator: rows/cols 0-3 calib, 4-9 pose, 10-11 affine, 12 residual.
struct AccumulatorApprox {
    Mat1313f H;
    size_t num;
    alignas(16) float Data[60], Data1k[60], Data1m[60];
    alignas(16) float TopRight_Data[32], TopRight_Data1k[32], TopRight_Data1m[32];
    alignas(16) float BotRight_Data[8], BotRight_Data1k[8], BotRight_Data1m[8];
    float numIn1, numIn1k, numIn1m;

    void initialize() {
        memset(Data, 0, sizeof(Data)); memset(Data1k, 0, sizeof(Data1k)); memset(Data1m, 0, sizeof(Data1m));
        memset(TopRight_Data, 0, sizeof(TopRight_Data)); memset(TopRight_Data1k, 0, sizeof(TopRight_Data1k)); memset(TopRight_Data1m, 0, sizeof(TopRight_Data1m));
        memset(BotRight_Data, 0, sizeof(BotRight_Data)); memset(BotRight_Data1k, 0, sizeof(BotRight_Data1k)); memset(BotRight_Data1m, 0, sizeof(BotRight_Data1m));
        num = 0; numIn1 = numIn1k = numIn1m = 0;
    }

    void finish() {   // :771-800
        H.setZero();
        shiftUp(true);
        int idx = 0;
        for (int r = 0; r < 10; r++) for (int c = r; c < 10; c++) { H(r, c) = H(c, r) = Data1m[idx]; idx++; }
        idx = 0;
        for (int r = 0; r < 10; r++) for (int c = 0; c < 3; c++) { H(r, c + 10) = H(c + 10, r) = TopRight_Data1m[idx]; idx++; }
        H(10, 10) = BotRight_Data1m[0];
        H(10, 11) = H(11, 10) = BotRight_Data1m[1];
        H(10, 12) = H(12, 10) = BotRight_Data1m[2];
        H(11, 11) = BotRight_Data1m[3];
        H(11, 12) = H(12, 11) = BotRight_Data1m[4];
        H(12, 12) = BotRight_Data1m[5];
        num = (size_t) (numIn1 + numIn1k + numIn1m);
    }

    // H[0:10,0:10] += a x x^T + c y y^T + b (x y^T + y x^T), x=[x4;x6], y=[y4;y6]   (:893-979)
    void update(const float *x4, const float *x6, const float *y4, const float *y6, const float a, const float b, const float c) {
        float x[10], y[10];
        for (int i = 0; i < 4; i++) { x[i] = x4[i]; y[i] = y4[i]; }
        for (int i = 0; i < 6; i++) { x[4 + i] = x6[i]; y[4 + i] = y6[i]; }
        int idx = 0;
        for (int r = 0; r < 10; r++)
            for (int cc = r; cc < 10; cc++) {
                // reference term order: a*x[cc]*x[r] + c*y[cc]*y[r] + b*(x[cc]*y[r] + y[cc]*x[r])
                Data[idx] += a * x[cc] * x[r] + c * y[cc] * y[r] + b * (x[cc] * y[r] + y[cc] * x[r]);
                idx++;
            }
        num++;
        numIn1++;
        shiftUp(false);
    }

    void updateTopRight(const float *x4, const float *x6, const float *y4, const float *y6,
                        const float TR00, const float TR10, const float TR01, const float TR11, const float TR02, const float TR12) {   // :982-1030
        for (int i = 0; i < 4; i++) {
            TopRight_Data[3 * i + 0] += x4[i] * TR00 + y4[i] * TR10;
            TopRight_Data[3 * i + 1] += x4[i] * TR01 + y4[i] * TR11;
            TopRight_Data[3 * i + 2] += x4[i] * TR02 + y4[i] * TR12;
        }
        for (int i = 0; i < 6; i++) {
            TopRight_Data[12 + 3 * i + 0] += x6[i] * TR00 + y6[i] * TR10;
            TopRight_Data[12 + 3 * i + 1] += x6[i] * TR01 + y6[i] * TR11;
            TopRight_Data[12 + 3 * i + 2] += x6[i] * TR02 + y6[i] * TR12;
        }
    }

    void updateBotRight(const float a00, const float a01, const float a02, const float a11, const float a12, const float a22) {   // :1032-1045
        BotRight_Data[0] += a00; BotRight_Data[1] += a01; BotRight_Data[2] += a02;
        BotRight_Data[3] += a11; BotRight_Data[4] += a12; BotRight_Data[5] += a22;
    }

    void shiftUp(bool force) {   // :1065-1100
        if (numIn1 > 1000 || force) {
            for (int i = 0; i < 60; i++) Data1k[i] = Data[i] + Data1k[i];
            for (int i = 0; i < 32; i++) TopRight_Data1k[i] = TopRight_Data[i] + TopRight_Data1k[i];
            for (int i = 0; i < 8; i++) BotRight_Data1k[i] = BotRight_Data[i] + BotRight_Data1k[i];
            numIn1k += numIn1; numIn1 = 0;
            memset(Data, 0, sizeof(Data)); memset(TopRight_Data, 0, sizeof(TopRight_Data)); memset(BotRight_Data, 0, sizeof(BotRight_Data));
        }
        if (numIn1k > 1000 || force) {
            for (int i = 0; i < 60; i++) Data1m[i] = Data1k[i] + Data1m[i];
            for (int i = 0; i < 32; i++) TopRight_Data1m[i] = TopRight_Data1k[i] + TopRight_Data1m[i];
            for (int i = 0; i < 8; i++) BotRight_Data1m[i] = BotRight_Data1k[i] + BotRight_Data1m[i];
            numIn1m += numIn1k; numIn1k = 0;
            memset(Data1k, 0, sizeof(Data1k)); memset(TopRight_Data1k, 0, sizeof(TopRight_Data1k)); memset(BotRight_Data1k, 0, sizeof(BotRight_Data1k));
        }
    }
};

// 9x9 symmetric accumulator, 4 SSE lanes x 45 unique entries (tracker).
struct Accumulator9 {
    Mat99f H;
    size_t num;
    alignas(16) float SSEData[4 * 45], SSEData1k[4 * 45], SSEData1m[4 * 45];
    float numIn1, numIn1k, numIn1m;

    void initialize() {
        H.setZero();
        memset(SSEData, 0, sizeof(SSEData)); memset(SSEData1k, 0, sizeof(SSEData1k)); memset(SSEData1m, 0, sizeof(SSEData1m));
        num = 0; numIn1 = numIn1k = numIn1m = 0;
    }
    void finish() {   // :1120-1134
        H.setZero();
        shiftUp(true);
        int idx = 0;
        for (int r = 0; r < 9; r++)
            for (int c = r; c < 9; c++) {
                float d = SSEData1m[idx + 0] + SSEData1m[idx + 1] + SSEData1m[idx + 2] + SSEData1m[idx + 3];
                H(r, c) = H(c, r) = d;
                idx += 4;
            }
    }
    // entry (r,c>=r) += (J_r * w) * J_c per lane   (:1250-1369)
    void updateSSE_eighted(const __m128 *J, const __m128 w) {
        float *pt = SSEData;
        for (int r = 0; r < 9; r++) {
            __m128 Jrw = _mm_mul_ps(J[r], w);
            for (int c = r; c < 9; c++) {
                _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(Jrw, J[c])));
                pt += 4;
            }
        }
        num += 4;
        numIn1++;
        shiftUp(false);
    }
    // entry (r,c>=r) += J_r * J_c per lane   (:1138-1247)
    void updateSSE(const __m128 *J) {
        float *pt = SSEData;
        for (int r = 0; r < 9; r++)
            for (int c = r; c < 9; c++) {
                _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(J[r], J[c])));
                pt += 4;
            }
        num += 4;
        numIn1++;
        shiftUp(false);
    }
    // lane `off` only: diagonal += J_r*J_r*w, then J_r *= w and (r,c>r) += J_c*J_r   (:1489-1614)
    void updateSingleWeighted(const float *Jin, float w, int off = 0) {
        float J[9];
        for (int i = 0; i < 9; i++) J[i] = Jin[i];
        float *pt = SSEData + off;
        for (int r = 0; r < 9; r++) {
            *pt += J[r] * J[r] * w;
            pt += 4;
            J[r] *= w;
            for (int c = r + 1; c < 9; c++) { *pt += J[c] * J[r]; pt += 4; }
        }
        num++;
        numIn1++;
        shiftUp(false);
    }
    void shiftUp(bool force) {   // :1624-1643
        if (numIn1 > 1000 || force) {
            for (int i = 0; i < 45; i++) _mm_store_ps(SSEData1k + 4 * i, _mm_add_ps(_mm_load_ps(SSEData + 4 * i), _mm_load_ps(SSEData1k + 4 * i)));
            numIn1k += numIn1; numIn1 = 0; memset(SSEData, 0, sizeof(SSEData));
        }
        if (numIn1k > 1000 || force) {
            for (int i = 0; i < 45; i++) _mm_store_ps(SSEData1m + 4 * i, _mm_add_ps(_mm_load_ps(SSEData1k + 4 * i), _mm_load_ps(SSEData1m + 4 * i)));
            numIn1m += numIn1k; numIn1k = 0; memset(SSEData1k, 0, sizeof(SSEData1k));
        }
    }
};

}  // namespace orc
