// ORACLE — TEST INFRASTRUCTURE ONLY.  Sim3d is a data member of ldso::Frame (loop closing, map output).  FullSystem.cc builds one from a
// 4x4 matrix when it hands a pose to the map / viewer (:418, :545-564, :857) and inverts it in printResult (:1935, :1969); with scale 1 (no
// loop closing in the pin) that is the SE3.  Never on the arithmetic path the pin checks.
#pragma once
#include "se3.hpp"
namespace Sophus {
class Sim3d {
public:
    SE3d se3; double s = 1.0;
    Sim3d() {}
    explicit Sim3d(const Eigen::Matrix4d &m) {
        Eigen::Matrix3d R; Eigen::Vector3d t;
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = m(i, j); t[i] = m(i, 3); }
        se3 = SE3d(R, t);
    }
    Eigen::Matrix4d matrix() const { return se3.matrix(); }
    Sim3d inverse() const { Sim3d r; r.se3 = se3.inverse(); r.s = 1.0 / s; return r; }
    Eigen::Matrix3d rotationMatrix() const { return se3.rotationMatrix(); }
    Eigen::Vector3d translation() const { Eigen::Vector3d t; for (int i = 0; i < 3; i++) t[i] = se3.T.t[i]; return t; }
    double scale() const { return s; }
    Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { Eigen::Vector3d q = se3 * p; for (int i = 0; i < 3; i++) q[i] = s * (q[i] - se3.T.t[i]) + se3.T.t[i]; return q; }      // Point::ComputeWorldPos (map output only)
};
}  // namespace Sophus
