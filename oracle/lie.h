// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h header).
// lie.h — the pieces of vendored Sophus the hot path uses, restated:
//   SO3 expAndTheta (thirdparty/sophus/so3.hpp:343-371), logAndTheta (:491-531), hat (:431-438),
//   SE3 exp (thirdparty/sophus/se3.hpp:407-428), log (:560-585), Adj (:131-139), inverse (:169-173),
//   operator* = fastMultiply + normalize (:160-163, :268-271), quaternion -> rotation matrix (Eigen).
#pragma once
#include "linalg.h"

namespace orc {

struct Quat {
    double w = 1, x = 0, y = 0, z = 0;
    Quat() {}
    Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
    Quat operator*(const Quat &b) const {   // Eigen quaternion product
        return Quat(w * b.w - x * b.x - y * b.y - z * b.z,
                    w * b.x + x * b.w + y * b.z - z * b.y,
                    w * b.y + y * b.w + z * b.x - x * b.z,
                    w * b.z + z * b.w + x * b.y - y * b.x);
    }
    Quat conjugate() const { return Quat(w, -x, -y, -z); }
    double norm() const { return std::sqrt(w * w + x * x + y * y + z * z); }
    void normalize() { double l = norm(); w /= l; x /= l; y /= l; z /= l; }
    // Eigen::QuaternionBase::toRotationMatrix
    Mat33 toRotationMatrix() const {
        Mat33 res;
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        res(0, 0) = 1 - (tyy + tzz); res(0, 1) = txy - twz; res(0, 2) = txz + twy;
        res(1, 0) = txy + twz; res(1, 1) = 1 - (txx + tzz); res(1, 2) = tyz - twx;
        res(2, 0) = txz - twy; res(2, 1) = tyz + twx; res(2, 2) = 1 - (txx + tyy);
        return res;
    }
    // Eigen::QuaternionBase::_transformVector: v + w*uv*2 + u x uv * 2 with uv = u x v
    Vec3 transformVector(const Vec3 &v) const {
        Vec3 uv; uv[0] = y * v[2] - z * v[1]; uv[1] = z * v[0] - x * v[2]; uv[2] = x * v[1] - y * v[0];
        uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
        Vec3 r;
        r[0] = v[0] + w * uv[0] + (y * uv[2] - z * uv[1]);
        r[1] = v[1] + w * uv[1] + (z * uv[0] - x * uv[2]);
        r[2] = v[2] + w * uv[2] + (x * uv[1] - y * uv[0]);
        return r;
    }
    // Eigen: Quaternion from rotation matrix (quaternionbase_assign_impl<Mat,3,3>)
    static Quat fromRotationMatrix(const Mat33 &m) {
        Quat q;
        double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0) {
            t = std::sqrt(t + 1.0);
            q.w = 0.5 * t; t = 0.5 / t;
            q.x = (m(2, 1) - m(1, 2)) * t; q.y = (m(0, 2) - m(2, 0)) * t; q.z = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
            double v[3];
            v[i] = 0.5 * t; t = 0.5 / t;
            q.w = (m(k, j) - m(j, k)) * t;
            v[j] = (m(j, i) + m(i, j)) * t; v[k] = (m(k, i) + m(i, k)) * t;
            q.x = v[0]; q.y = v[1]; q.z = v[2];
        }
        return q;
    }
};

inline Mat33 hat(const Vec3 &o) {
    Mat33 O;
    O(0, 0) = 0; O(0, 1) = -o[2]; O(0, 2) = o[1];
    O(1, 0) = o[2]; O(1, 1) = 0; O(1, 2) = -o[0];
    O(2, 0) = -o[1]; O(2, 1) = o[0]; O(2, 2) = 0;
    return O;
}

static const double SOPHUS_EPS = 1e-10;   // SophusConstants<double>::epsilon()

struct SE3 {
    Quat q;
    Vec3 t;
    SE3() {}
    SE3(const Quat &q_, const Vec3 &t_) : q(q_), t(t_) {}
    static SE3 fromMatrix34(const double *m) {   // row-major [R|t]
        Mat33 R; Vec3 tt;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = m[i * 4 + j]; tt[i] = m[i * 4 + 3]; }
        Quat qq = Quat::fromRotationMatrix(R); qq.normalize();
        return SE3(qq, tt);
    }
    Mat33 rotationMatrix() const { return q.toRotationMatrix(); }
    const Vec3 &translation() const { return t; }
    Vec3 &translation() { return t; }
    SE3 inverse() const {
        Quat qi = q.conjugate();
        return SE3(qi, qi.transformVector(t * -1.0));
    }
    SE3 operator*(const SE3 &o) const {
        SE3 r(*this);
        r.t += q.transformVector(o.t);
        r.q = q * o.q;
        r.q.normalize();
        return r;
    }
    static Quat so3_expAndTheta(const Vec3 &omega, double *theta) {
        const double theta_sq = omega.squaredNorm();
        *theta = std::sqrt(theta_sq);
        const double half_theta = 0.5 * (*theta);
        double imag_factor, real_factor;
        if ((*theta) < SOPHUS_EPS) {
            const double theta_po4 = theta_sq * theta_sq;
            imag_factor = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
            real_factor = 1 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
        } else {
            const double sin_half_theta = std::sin(half_theta);
            imag_factor = sin_half_theta / (*theta);
            real_factor = std::cos(half_theta);
        }
        return Quat(real_factor, imag_factor * omega[0], imag_factor * omega[1], imag_factor * omega[2]);
    }
    static SE3 exp(const Vec6 &a) {
        Vec3 omega; omega[0] = a[3]; omega[1] = a[4]; omega[2] = a[5];
        Vec3 ups; ups[0] = a[0]; ups[1] = a[1]; ups[2] = a[2];
        double theta;
        Quat so3 = so3_expAndTheta(omega, &theta);
        Mat33 Omega = hat(omega);
        Mat33 Omega_sq = Omega * Omega;
        Mat33 V;
        if (theta < SOPHUS_EPS) {
            V = so3.toRotationMatrix();
        } else {
            double theta_sq = theta * theta;
            V = Mat33::Identity() + Omega * ((1 - std::cos(theta)) / theta_sq) + Omega_sq * ((theta - std::sin(theta)) / (theta_sq * theta));
        }
        return SE3(so3, V * ups);
    }
    static Vec3 so3_logAndTheta(const Quat &q, double *theta) {
        const double squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
        const double n = std::sqrt(squared_n);
        const double w = q.w;
        double two_atan_nbyw_by_n;
        if (n < SOPHUS_EPS) {
            const double squared_w = w * w;
            two_atan_nbyw_by_n = 2.0 / w - 2.0 * (squared_n) / (w * squared_w);
        } else {
            if (std::fabs(w) < SOPHUS_EPS) two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
            else two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
        }
        *theta = two_atan_nbyw_by_n * n;
        Vec3 r; r[0] = two_atan_nbyw_by_n * q.x; r[1] = two_atan_nbyw_by_n * q.y; r[2] = two_atan_nbyw_by_n * q.z;
        return r;
    }
    Vec6 log() const {
        double theta;
        Vec3 om = so3_logAndTheta(q, &theta);
        Mat33 Omega = hat(om);
        Mat33 V_inv;
        if (std::fabs(theta) < SOPHUS_EPS) {
            V_inv = Mat33::Identity() - Omega * 0.5 + (Omega * Omega) * (1. / 12.);
        } else {
            V_inv = Mat33::Identity() - Omega * 0.5 + (Omega * Omega) * ((1 - theta / (2 * std::tan(theta / 2))) / (theta * theta));
        }
        Vec3 u = V_inv * t;
        Vec6 r; r[0] = u[0]; r[1] = u[1]; r[2] = u[2]; r[3] = om[0]; r[4] = om[1]; r[5] = om[2];
        return r;
    }
    Mat66 Adj() const {
        Mat33 R = rotationMatrix();
        Mat33 tR = hat(t) * R;
        Mat66 res;
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { res(i, j) = R(i, j); res(3 + i, 3 + j) = R(i, j); res(i, 3 + j) = tR(i, j); res(3 + i, j) = 0; }
        return res;
    }
};

}  // namespace orc
