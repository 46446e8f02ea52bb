// harness/camera.h -- host-side project / unproject of the four camera models of
// include/visnav/camera_models.h (pinhole :75-117, eucm :158-213, ds :246-302, kb4 :341-438), in the
// reference's operation order.  The device kernels have their own project(); the host needs
// unproject() for the bearing vectors of PnP / triangulation / the epipolar test, which the reference
// evaluates on the CPU as well.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "geometry.h"

namespace visnav {
namespace harness {

enum CameraKind { kDS = 0, kPinhole = 1, kEUCM = 2, kKB4 = 3 };  // = VSL_CAM_* of include/vslam_hip.h

inline int camera_kind(const std::string& name) {
  if (name == "ds") return kDS;
  if (name == "pinhole") return kPinhole;
  if (name == "eucm") return kEUCM;
  if (name == "kb4") return kKB4;
  std::fprintf(stderr, "Camera model %s is not implemented.\n", name.c_str());  // camera_models.h:493-495
  std::abort();
}

inline void project(int kind, const double* param, const Vec3& p, double* uv) {
  const double fx = param[0], fy = param[1], cx = param[2], cy = param[3];
  const double x = p.x, y = p.y, z = p.z;
  switch (kind) {
    case kPinhole:
      uv[0] = fx * x / z + cx;
      uv[1] = fy * y / z + cy;
      return;
    case kEUCM: {
      const double alpha = param[4], beta = param[5];
      const double d = std::sqrt(beta * (x * x + y * y) + z * z);
      uv[0] = fx * x / (alpha * d + (1.0 - alpha) * z) + cx;
      uv[1] = fy * y / (alpha * d + (1.0 - alpha) * z) + cy;
      return;
    }
    case kDS: {
      const double xi = param[4], alpha = param[5];
      const double d1 = std::sqrt(x * x + y * y + z * z);
      const double d2 = std::sqrt(x * x + y * y + (xi * d1 + z) * (xi * d1 + z));
      uv[0] = fx * x / (alpha * d2 + (1.0 - alpha) * (xi * d1 + z)) + cx;
      uv[1] = fy * y / (alpha * d2 + (1.0 - alpha) * (xi * d1 + z)) + cy;
      return;
    }
    default: {  // kb4
      const double k1 = param[4], k2 = param[5], k3 = param[6], k4 = param[7];
      const double r = std::sqrt(x * x + y * y);
      const double th = std::atan2(r, z);
      const double th2 = th * th;
      const double d = th + k1 * th * th2 + k2 * th * th2 * th2 + k3 * th * th2 * th2 * th2 + k4 * th * th2 * th2 * th2 * th2;
      if (r == 0.0) {
        uv[0] = cx;
        uv[1] = cy;
      } else {
        uv[0] = fx * d * x / r + cx;
        uv[1] = fy * d * y / r + cy;
      }
      return;
    }
  }
}

inline Vec3 unproject(int kind, const double* param, double u, double v) {
  const double fx = param[0], fy = param[1], cx = param[2], cy = param[3];
  const double mx = (u - cx) / fx, my = (v - cy) / fy;
  switch (kind) {
    case kPinhole: {
      const double s = 1.0 / std::sqrt(mx * mx + my * my + 1.0);
      return {mx * s, my * s, s};
    }
    case kEUCM: {
      const double alpha = param[4], beta = param[5];
      const double rr = mx * mx + my * my;
      const double mz = (1.0 - beta * alpha * alpha * rr) / (alpha * std::sqrt(1.0 - (2.0 * alpha - 1.0) * beta * rr) + (1.0 - alpha));
      const double s = 1.0 / std::sqrt(mx * mx + my * my + mz * mz);
      return {mx * s, my * s, mz * s};
    }
    case kDS: {
      const double xi = param[4], alpha = param[5];
      const double rr = mx * mx + my * my;
      const double mz = (1.0 - alpha * alpha * rr) / (alpha * std::sqrt(1.0 - (2.0 * alpha - 1.0) * rr) + 1.0 - alpha);
      const double s = (mz * xi + std::sqrt(mz * mz + (1.0 - xi * xi) * rr)) / (mz * mz + rr);
      return {mx * s, my * s, mz * s - xi};
    }
    default: {  // kb4: five Newton steps from theta = 0 (camera_models.h:404-424)
      const double k1 = param[4], k2 = param[5], k3 = param[6], k4 = param[7];
      const double ru = std::sqrt(mx * mx + my * my);
      double th = 0.0;
      for (int it = 0; it < 5; it++) {
        const double t2 = th * th;
        const double f = th + k1 * th * t2 + k2 * th * t2 * t2 + k3 * th * t2 * t2 * t2 + k4 * th * t2 * t2 * t2 * t2 - ru;
        const double df = 1.0 + 3.0 * k1 * t2 + 5.0 * k2 * t2 * t2 + 7.0 * k3 * t2 * t2 * t2 + 9.0 * k4 * t2 * t2 * t2 * t2;
        th = th - f / df;
      }
      if (ru == 0.0) return {0.0, 0.0, std::cos(th)};
      return {std::sin(th) * mx / ru, std::sin(th) * my / ru, std::cos(th)};
    }
  }
}

}  // namespace harness
}  // namespace visnav
