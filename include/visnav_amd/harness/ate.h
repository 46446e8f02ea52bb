// harness/ate.h -- absolute trajectory error of the reference (alignSVD, src/slam.cpp:1618-1710): associate
// every estimated position with the linearly interpolated ground truth, align rigidly (Kabsch / Horn via
// a 3x3 SVD, reflection guard), return the RMSE of the aligned positions.  Host only.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "geometry.h"

namespace visnav {
namespace harness {

// src/slam.cpp:1618-1710.  Returns the RMSE after rigid alignment; n_assoc = number of associations.
inline double align_svd(const std::vector<int64_t>& est_t_ns, const std::vector<Vec3>& est_t_w_i,
                        const std::vector<int64_t>& gt_t_ns, const std::vector<Vec3>& gt_t_w_i, int* n_assoc = nullptr) {
  std::vector<Vec3> est, gt;
  for (size_t i = 0; i < est_t_w_i.size(); i++) {
    const int64_t t_ns = est_t_ns[i];
    size_t j;
    for (j = 0; j < gt_t_ns.size(); j++)
      if (gt_t_ns[j] > t_ns) break;
    j--;  // wraps to SIZE_MAX when t_ns precedes the first ground-truth sample, caught by the next test
    if (gt_t_ns.empty() || j >= gt_t_ns.size() - 1) continue;
    const double dt_ns = (double)(t_ns - gt_t_ns[j]);
    const double int_t_ns = (double)(gt_t_ns[j + 1] - gt_t_ns[j]);
    if (int_t_ns > 1.1e8) continue;  // skip if the ground-truth gap is larger than 100 ms
    const double ratio = dt_ns / int_t_ns;
    gt.push_back((1 - ratio) * gt_t_w_i[j] + ratio * gt_t_w_i[j + 1]);
    est.push_back(est_t_w_i[i]);
  }
  const int n = (int)est.size();
  if (n_assoc) *n_assoc = n;
  if (n == 0) return std::nan("");
  Vec3 mean_gt, mean_est;
  for (int i = 0; i < n; i++) {
    mean_gt = mean_gt + gt[i];
    mean_est = mean_est + est[i];
  }
  mean_gt = (1.0 / n) * mean_gt;
  mean_est = (1.0 / n) * mean_est;
  Mat3 cov = Mat3::zero();
  for (int i = 0; i < n; i++) {
    const Vec3 g = gt[i] - mean_gt, e = est[i] - mean_est;
    const double gv[3] = {g.x, g.y, g.z}, ev[3] = {e.x, e.y, e.z};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) cov.m[a][b] += gv[a] * ev[b];
  }
  Mat3 U, V;
  double s[3];
  svd3(cov, U, s, V);
  Mat3 S;
  if (det(U) * det(V) < 0) S.m[2][2] = -1;
  const Mat3 rot_gt_est = U * S * transpose(V);
  const Vec3 trans = mean_gt - rot_gt_est * mean_est;
  double error = 0;
  for (int i = 0; i < n; i++) {
    const Vec3 res = (rot_gt_est * est[i] + trans) - gt[i];
    error += dot(res, res);
  }
  return std::sqrt(error / n);
}

}  // namespace harness
}  // namespace visnav
