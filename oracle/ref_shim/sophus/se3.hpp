// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for the vendored Sophus headers (thirdparty/sophus), which cannot be compiled without
// Eigen's Quaternion / Map machinery.  SE3d / SO3d forward to the oracle's restatement of the Sophus functions the hot path uses
// (../../lie.h: unit-quaternion storage, exp / log / Adj / inverse / operator* with renormalisation, each citing the Sophus lines it
// follows), so poses are NOT pinned by the reference-compiled library; everything downstream of them is.
#pragma once
#include <Eigen/Core>
#include "../../lie.h"

namespace Sophus {

// Eigen::Quaterniond as FullSystem.cc uses it: Sophus::Quaterniond(w, x, y, z) in the motion-hypothesis list of trackNewCoarse (:228-303)
// and the unit_quaternion() accessors of printResult (:1942-1945)
struct Quaterniond {
    double w_, x_, y_, z_;
    Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
    double w() const { return w_; } double x() const { return x_; } double y() const { return y_; } double z() const { return z_; }
};

class SO3d {
public:
    orc::Quat q;
    SO3d() {}
    explicit SO3d(const orc::Quat &q_) : q(q_) {}
    Quaterniond unit_quaternion() const { return Quaterniond(q.w, q.x, q.y, q.z); }
    Eigen::Matrix3d matrix() const { orc::Mat33 R = q.toRotationMatrix(); Eigen::Matrix3d m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = R(i, j); return m; }
    SO3d inverse() const { return SO3d(q.conjugate()); }
};

class SE3d {
public:
    orc::SE3 T;
    // translation() hands out an Eigen vector that aliases the oracle's storage (both are three contiguous doubles)
    SE3d() {}
    explicit SE3d(const orc::SE3 &t) : T(t) {}
    template <class DR, class DT>
    SE3d(const Eigen::Base<DR, double, 3, 3> &R, const Eigen::Base<DT, double, 3, 1> &t) {
        orc::Mat33 Rm; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rm(i, j) = R(i, j);
        T.q = orc::Quat::fromRotationMatrix(Rm); T.q.normalize();
        for (int i = 0; i < 3; i++) T.t[i] = t[i];
    }
    // SE3(Matrix4): map I/O only (Frame.cc:163)
    explicit SE3d(const Eigen::Matrix4d &m) {
        orc::Mat33 Rm; for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Rm(i, j) = m(i, j); T.t[i] = m(i, 3); }
        T.q = orc::Quat::fromRotationMatrix(Rm); T.q.normalize();
    }
    // SE3(Quaternion, Point) -> SO3(Quaternion): the quaternion is normalised (thirdparty/sophus/so3.hpp: SO3Group(QuaternionBase) ctor)
    template <class DT>
    SE3d(const Quaterniond &q, const Eigen::Base<DT, double, 3, 1> &t) {
        T.q = orc::Quat(q.w_, q.x_, q.y_, q.z_); T.q.normalize();
        for (int i = 0; i < 3; i++) T.t[i] = t[i];
    }
    static SE3d exp(const Eigen::Matrix<double, 6, 1> &a) { orc::Vec6 v; for (int i = 0; i < 6; i++) v[i] = a[i]; return SE3d(orc::SE3::exp(v)); }
    Eigen::Matrix<double, 6, 1> log() const { orc::Vec6 v = T.log(); Eigen::Matrix<double, 6, 1> r; for (int i = 0; i < 6; i++) r[i] = v[i]; return r; }
    SE3d inverse() const { return SE3d(T.inverse()); }
    SE3d operator*(const SE3d &o) const { return SE3d(T * o.T); }
    SE3d &operator*=(const SE3d &o) { T = T * o.T; return *this; }
    Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { orc::Vec3 v; for (int i = 0; i < 3; i++) v[i] = p[i]; orc::Vec3 r = T.q.transformVector(v); Eigen::Vector3d o; for (int i = 0; i < 3; i++) o[i] = r[i] + T.t[i]; return o; }
    Eigen::Matrix3d rotationMatrix() const { orc::Mat33 R = T.rotationMatrix(); Eigen::Matrix3d m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = R(i, j); return m; }
    Eigen::View<double, 3, 1> translation() { return Eigen::View<double, 3, 1>(&T.t[0], 3, 1, 1, 3); }
    Eigen::View<const double, 3, 1> translation() const { return Eigen::View<const double, 3, 1>(&T.t[0], 3, 1, 1, 3); }
    SO3d so3() const { return SO3d(T.q); }
    Eigen::Matrix<double, 6, 6> Adj() const { orc::Mat66 A = T.Adj(); Eigen::Matrix<double, 6, 6> m; for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) m(i, j) = A(i, j); return m; }
    Eigen::Matrix4d matrix() const { Eigen::Matrix4d m = Eigen::Matrix4d::Identity(); orc::Mat33 R = T.rotationMatrix(); for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m(i, j) = R(i, j); m(i, 3) = T.t[i]; } return m; }
    Eigen::Matrix<double, 3, 4> matrix3x4() const { Eigen::Matrix<double, 3, 4> m; orc::Mat33 R = T.rotationMatrix(); for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) m(i, j) = R(i, j); m(i, 3) = T.t[i]; } return m; }
};

}  // namespace Sophus
