// ORACLE — TEST INFRASTRUCTURE ONLY.  Sim3d is only a data member of ldso::Frame on the hot path (loop closing uses it): an opaque stand-in.
#pragma once
#include "se3.hpp"
namespace Sophus {
class Sim3d {
public:
    SE3d se3; double s = 1.0;
    Sim3d() {}
    Eigen::Matrix4d matrix() const { return se3.matrix(); }
};
}  // namespace Sophus
