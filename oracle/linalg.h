// ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference (tum-vision/LDSO) hot path.
// Nothing under oracle/ may be imported, linked or executed by the product path (ldso_amd/);
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
// PARITY PINNED for the BA hot path and makeImages: the reference ships no tests / golden vectors and its own build needs
// Eigen3 / OpenCV / glog / Pangolin (absent), but its hot-path translation units compile unmodified against the header shim
// oracle/ref_shim (oracle/Makefile `ref` -> oracle/_ref/libldso_ref.so); tests/test_ref_pin.py holds this restatement to that
// library BIT FOR BIT and tests/golden/ref_*.npz are its outputs.  Not pinned: what the shim itself replaces (Eigen's product /
// reduction order, LDLT / SVD / inverse, Sophus) and the restated FullSystem loops; see DESIGN.md section 3.
//
// linalg.h — the small dense linear algebra the reference gets from Eigen3 (system dependency, version
// unpinned; cmake/FindEigen3.cmake:18-25).  Restated from Eigen's published algorithms:
//   * fixed-size row-major matrices with coefficient-wise products,
//   * LDLT  = Eigen::LDLT (robust Cholesky with diagonal pivoting; call sites
//             EnergyFunctional.cc:334, CoarseTracker.cc:109),
//   * inverse3f = Eigen 3x3 cofactor inverse (FrameFramePrecalc.cc:28, CoarseTracker.cc:240),
//   * inverse_lu = partial-pivot LU inverse (Mat88::inverse(), EnergyFunctional.cc:117),
//   * jacobi_svd_thin = thin SVD by one-sided Jacobi (Eigen::JacobiSVD, EnergyFunctional.cc:697).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cassert>
#include <limits>

namespace orc {

template <class T, int R, int C>
struct Mat {
    T d[R * C];
    Mat() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    T &operator()(int r, int c) { return d[r * C + c]; }
    const T &operator()(int r, int c) const { return d[r * C + c]; }
    T &operator[](int i) { return d[i]; }
    const T &operator[](int i) const { return d[i]; }
    static Mat Identity() { Mat m; for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = T(1); return m; }
    static Mat Zero() { return Mat(); }
    static Mat Constant(T v) { Mat m; for (int i = 0; i < R * C; i++) m.d[i] = v; return m; }
    void setZero() { for (int i = 0; i < R * C; i++) d[i] = T(0); }
    Mat<T, C, R> transpose() const { Mat<T, C, R> t; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) t(c, r) = (*this)(r, c); return t; }
    template <class U> Mat<U, R, C> cast() const { Mat<U, R, C> m; for (int i = 0; i < R * C; i++) m.d[i] = (U) d[i]; return m; }
    Mat operator+(const Mat &o) const { Mat m; for (int i = 0; i < R * C; i++) m.d[i] = d[i] + o.d[i]; return m; }
    Mat operator-(const Mat &o) const { Mat m; for (int i = 0; i < R * C; i++) m.d[i] = d[i] - o.d[i]; return m; }
    Mat operator-() const { Mat m; for (int i = 0; i < R * C; i++) m.d[i] = -d[i]; return m; }
    Mat operator*(T s) const { Mat m; for (int i = 0; i < R * C; i++) m.d[i] = d[i] * s; return m; }
    Mat &operator+=(const Mat &o) { for (int i = 0; i < R * C; i++) d[i] += o.d[i]; return *this; }
    Mat &operator-=(const Mat &o) { for (int i = 0; i < R * C; i++) d[i] -= o.d[i]; return *this; }
    Mat &operator*=(T s) { for (int i = 0; i < R * C; i++) d[i] *= s; return *this; }
    T dot(const Mat &o) const { T s = T(0); for (int i = 0; i < R * C; i++) s += d[i] * o.d[i]; return s; }
    T squaredNorm() const { return dot(*this); }
    T norm() const { return std::sqrt(squaredNorm()); }
    T sum() const { T s = T(0); for (int i = 0; i < R * C; i++) s += d[i]; return s; }
};

template <class T, int R, int K, int C>
inline Mat<T, R, C> operator*(const Mat<T, R, K> &a, const Mat<T, K, C> &b) {
    Mat<T, R, C> m;
    for (int r = 0; r < R; r++)
        for (int c = 0; c < C; c++) {
            T s = a(r, 0) * b(0, c);
            for (int k = 1; k < K; k++) s += a(r, k) * b(k, c);
            m(r, c) = s;
        }
    return m;
}

typedef Mat<float, 2, 1> Vec2f;
typedef Mat<float, 3, 1> Vec3f;
typedef Mat<float, 4, 1> VecCf;
typedef Mat<float, 6, 1> Vec6f;
typedef Mat<float, 8, 1> Vec8f;
typedef Mat<float, 8, 1> VecNRf;
typedef Mat<float, 1, 8> Mat18f;
typedef Mat<float, 2, 2> Mat22f;
typedef Mat<float, 3, 3> Mat33f;
typedef Mat<float, 8, 8> Mat88f;
typedef Mat<float, 13, 13> Mat1313f;
typedef Mat<float, 9, 9> Mat99f;
typedef Mat<double, 2, 1> Vec2;
typedef Mat<double, 3, 1> Vec3;
typedef Mat<double, 4, 1> VecC;
typedef Mat<double, 5, 1> Vec5;
typedef Mat<double, 6, 1> Vec6;
typedef Mat<double, 8, 1> Vec8;
typedef Mat<double, 10, 1> Vec10;
typedef Mat<double, 3, 3> Mat33;
typedef Mat<double, 4, 2> Mat42;
typedef Mat<double, 6, 6> Mat66;
typedef Mat<double, 8, 8> Mat88;
typedef Mat<double, 8, 4> Mat8C;
typedef Mat<double, 13, 13> MatPCPC;

// Eigen fixed-size 3x3 inverse by cofactors (Eigen/src/LU/InverseImpl.h, compute_inverse<.,.,3>).
template <class T>
inline Mat<T, 3, 3> inverse3(const Mat<T, 3, 3> &m) {
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
    };
    T c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    T det = c0 * m(0, 0) + c1 * m(1, 0) + c2 * m(2, 0);
    T invdet = T(1) / det;
    Mat<T, 3, 3> r;
    r(0, 0) = c0 * invdet; r(0, 1) = c1 * invdet; r(0, 2) = c2 * invdet;
    r(1, 0) = cof(0, 1) * invdet; r(1, 1) = cof(1, 1) * invdet; r(1, 2) = cof(2, 1) * invdet;
    r(2, 0) = cof(0, 2) * invdet; r(2, 1) = cof(1, 2) * invdet; r(2, 2) = cof(2, 2) * invdet;
    return r;
}

// ---------------------------------------------------------------------------------------------
// dynamic double matrices (MatXX / VecX of the reference)
// ---------------------------------------------------------------------------------------------
struct VecX {
    std::vector<double> d;
    VecX() {}
    explicit VecX(int n) : d(n, 0.0) {}
    static VecX Zero(int n) { return VecX(n); }
    static VecX Constant(int n, double v) { VecX x(n); std::fill(x.d.begin(), x.d.end(), v); return x; }
    int size() const { return (int) d.size(); }
    double &operator[](int i) { return d[i]; }
    const double &operator[](int i) const { return d[i]; }
    VecX operator+(const VecX &o) const { VecX r(size()); for (int i = 0; i < size(); i++) r[i] = d[i] + o[i]; return r; }
    VecX operator-(const VecX &o) const { VecX r(size()); for (int i = 0; i < size(); i++) r[i] = d[i] - o[i]; return r; }
    VecX operator-() const { VecX r(size()); for (int i = 0; i < size(); i++) r[i] = -d[i]; return r; }
    VecX operator*(double s) const { VecX r(size()); for (int i = 0; i < size(); i++) r[i] = d[i] * s; return r; }
    VecX &operator+=(const VecX &o) { for (int i = 0; i < size(); i++) d[i] += o[i]; return *this; }
    VecX &operator-=(const VecX &o) { for (int i = 0; i < size(); i++) d[i] -= o[i]; return *this; }
    double dot(const VecX &o) const { double s = 0; for (int i = 0; i < size(); i++) s += d[i] * o[i]; return s; }
    double norm() const { return std::sqrt(dot(*this)); }
    void conservativeResize(int n) { d.resize(n, 0.0); }
};

struct MatXX {
    int r = 0, c = 0;
    std::vector<double> d;   // row-major
    MatXX() {}
    MatXX(int r_, int c_) : r(r_), c(c_), d((size_t) r_ * c_, 0.0) {}
    static MatXX Zero(int r, int c) { return MatXX(r, c); }
    int rows() const { return r; }
    int cols() const { return c; }
    double &operator()(int i, int j) { return d[(size_t) i * c + j]; }
    const double &operator()(int i, int j) const { return d[(size_t) i * c + j]; }
    MatXX operator+(const MatXX &o) const { MatXX m(r, c); for (size_t i = 0; i < d.size(); i++) m.d[i] = d[i] + o.d[i]; return m; }
    MatXX operator-(const MatXX &o) const { MatXX m(r, c); for (size_t i = 0; i < d.size(); i++) m.d[i] = d[i] - o.d[i]; return m; }
    MatXX operator*(double s) const { MatXX m(r, c); for (size_t i = 0; i < d.size(); i++) m.d[i] = d[i] * s; return m; }
    MatXX &operator+=(const MatXX &o) { for (size_t i = 0; i < d.size(); i++) d[i] += o.d[i]; return *this; }
    MatXX &operator-=(const MatXX &o) { for (size_t i = 0; i < d.size(); i++) d[i] -= o.d[i]; return *this; }
    MatXX transpose() const { MatXX t(c, r); for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) t(j, i) = (*this)(i, j); return t; }
    MatXX operator*(const MatXX &o) const {
        MatXX m(r, o.c);
        for (int i = 0; i < r; i++)
            for (int k = 0; k < c; k++) {
                double a = (*this)(i, k);
                if (a == 0) continue;
                for (int j = 0; j < o.c; j++) m(i, j) += a * o(k, j);
            }
        return m;
    }
    VecX operator*(const VecX &v) const {
        VecX x(r);
        for (int i = 0; i < r; i++) { double s = 0; for (int j = 0; j < c; j++) s += (*this)(i, j) * v[j]; x[i] = s; }
        return x;
    }
    // Eigen conservativeResize: keep the top-left block, new entries uninitialised (caller zeroes them).
    void conservativeResize(int nr, int nc) {
        MatXX m(nr, nc);
        for (int i = 0; i < std::min(r, nr); i++) for (int j = 0; j < std::min(c, nc); j++) m(i, j) = (*this)(i, j);
        *this = m;
    }
    template <int R, int C> Mat<double, R, C> block(int i0, int j0) const {
        Mat<double, R, C> b; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) b(i, j) = (*this)(i0 + i, j0 + j); return b;
    }
    template <int R, int C> void addBlock(int i0, int j0, const Mat<double, R, C> &b) {
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) (*this)(i0 + i, j0 + j) += b(i, j);
    }
    template <int R, int C> void setBlock(int i0, int j0, const Mat<double, R, C> &b) {
        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) (*this)(i0 + i, j0 + j) = b(i, j);
    }
};

// Eigen::LDLT<MatXX, Lower>: in-place unblocked factorisation with symmetric diagonal pivoting
// (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked) and solve (_solve_impl).
struct LDLT {
    int n = 0;
    MatXX m;                 // lower triangle holds L (unit diagonal implied), diagonal holds D
    std::vector<int> tr;     // transpositions

    explicit LDLT(const MatXX &A) { compute(A); }

    void compute(const MatXX &A) {
        n = A.rows();
        m = A;
        tr.assign(n, 0);
        std::vector<double> temp(n, 0.0);
        for (int k = 0; k < n; k++) {
            int idx = k;
            double biggest = std::fabs(m(k, k));
            for (int i = k + 1; i < n; i++) if (std::fabs(m(i, i)) > biggest) { biggest = std::fabs(m(i, i)); idx = i; }
            tr[k] = idx;
            if (k != idx) {
                int s = n - idx - 1;
                for (int j = 0; j < k; j++) std::swap(m(k, j), m(idx, j));
                for (int i = 0; i < s; i++) std::swap(m(idx + 1 + i, k), m(idx + 1 + i, idx));
                std::swap(m(k, k), m(idx, idx));
                for (int i = k + 1; i < idx; i++) std::swap(m(i, k), m(idx, i));
            }
            int rs = n - k - 1;
            if (k > 0) {
                for (int j = 0; j < k; j++) temp[j] = m(j, j) * m(k, j);
                double s = 0;
                for (int j = 0; j < k; j++) s += m(k, j) * temp[j];
                m(k, k) -= s;
                for (int i = 0; i < rs; i++) {
                    double t = 0;
                    for (int j = 0; j < k; j++) t += m(k + 1 + i, j) * temp[j];
                    m(k + 1 + i, k) -= t;
                }
            }
            double akk = m(k, k);
            bool pivot_is_valid = std::fabs(akk) > 0.0;
            if (k == 0 && !pivot_is_valid) {
                for (int j = 0; j < n; j++) tr[j] = j;
                return;
            }
            if (rs > 0 && pivot_is_valid) for (int i = 0; i < rs; i++) m(k + 1 + i, k) /= akk;
        }
    }

    VecX solve(const VecX &rhs) const {
        VecX x = rhs;
        for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
        for (int i = 0; i < n; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= m(i, j) * x[j]; x[i] = s; }
        const double tolerance = (std::numeric_limits<double>::min)();
        for (int i = 0; i < n; i++) { if (std::fabs(m(i, i)) > tolerance) x[i] /= m(i, i); else x[i] = 0; }
        for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < n; j++) s -= m(j, i) * x[j]; x[i] = s; }
        for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
        return x;
    }
};

template <int N>
inline Mat<double, N, 1> ldlt_solve(const Mat<double, N, N> &A, const Mat<double, N, 1> &b) {
    MatXX M(N, N); VecX v(N);
    for (int i = 0; i < N; i++) { v[i] = b[i]; for (int j = 0; j < N; j++) M(i, j) = A(i, j); }
    VecX x = LDLT(M).solve(v);
    Mat<double, N, 1> r; for (int i = 0; i < N; i++) r[i] = x[i];
    return r;
}

// PartialPivLU inverse for fixed size N (Eigen uses this for fixed sizes > 4).
template <int N>
inline Mat<double, N, N> inverse_lu(const Mat<double, N, N> &A) {
    Mat<double, N, N> lu = A, inv = Mat<double, N, N>::Identity();
    int perm[N];
    for (int i = 0; i < N; i++) perm[i] = i;
    for (int k = 0; k < N; k++) {
        int p = k; double best = std::fabs(lu(k, k));
        for (int i = k + 1; i < N; i++) if (std::fabs(lu(i, k)) > best) { best = std::fabs(lu(i, k)); p = i; }
        if (p != k) { for (int j = 0; j < N; j++) { std::swap(lu(k, j), lu(p, j)); std::swap(inv(k, j), inv(p, j)); } }
        for (int i = k + 1; i < N; i++) {
            lu(i, k) /= lu(k, k);
            for (int j = k + 1; j < N; j++) lu(i, j) -= lu(i, k) * lu(k, j);
        }
    }
    // inv currently = P; solve L y = P, U x = y per column
    for (int c = 0; c < N; c++) {
        for (int i = 0; i < N; i++) { double s = inv(i, c); for (int j = 0; j < i; j++) s -= lu(i, j) * inv(j, c); inv(i, c) = s; }
        for (int i = N - 1; i >= 0; i--) { double s = inv(i, c); for (int j = i + 1; j < N; j++) s -= lu(i, j) * inv(j, c); inv(i, c) = s / lu(i, i); }
    }
    return inv;
}

// Thin SVD A = U diag(S) V^T by one-sided (Hestenes) Jacobi; A is rows x cols with rows >= cols.
inline void jacobi_svd_thin(const MatXX &A, MatXX &U, VecX &S, MatXX &V) {
    int m = A.rows(), n = A.cols();
    U = A;
    V = MatXX(n, n);
    for (int i = 0; i < n; i++) V(i, i) = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < m; i++) { alpha += U(i, p) * U(i, p); beta += U(i, q) * U(i, q); gamma += U(i, p) * U(i, q); }
                if (gamma == 0) continue;
                off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
                double zeta = (beta - alpha) / (2 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
                double cs = 1 / std::sqrt(1 + t * t), sn = cs * t;
                for (int i = 0; i < m; i++) { double up = U(i, p), uq = U(i, q); U(i, p) = cs * up - sn * uq; U(i, q) = sn * up + cs * uq; }
                for (int i = 0; i < n; i++) { double vp = V(i, p), vq = V(i, q); V(i, p) = cs * vp - sn * vq; V(i, q) = sn * vp + cs * vq; }
            }
        if (off < 1e-15) break;
    }
    S = VecX(n);
    for (int j = 0; j < n; j++) {
        double s = 0; for (int i = 0; i < m; i++) s += U(i, j) * U(i, j);
        s = std::sqrt(s); S[j] = s;
        if (s > 0) for (int i = 0; i < m; i++) U(i, j) /= s;
    }
}

}  // namespace orc
