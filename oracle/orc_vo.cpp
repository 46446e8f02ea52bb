// orc_vo.cpp -- CPU restatement of the per-frame landmark projection / guided matching of the
// reference (include/visnav/vo_utils.h:48-167).  TEST INFRASTRUCTURE ONLY (see vslam_oracle.h).
//
// find_matches_landmarks keeps the reference's std::partial_sort call literally: which of two equally
// distant landmarks ends up first is libstdc++ heap-select behaviour, and this file is compiled against
// the same libstdc++ the reference would use.
#include <algorithm>
#include <bitset>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "vslam_oracle.h"

extern "C" void orc_project(int model, const double* intr8, const double* p3, double* uv2);

namespace {
// [upstream] Sophus: SE3::inverse() * p  with a unit quaternion (x, y, z, w) and translation t:
//   q^-1 = conjugate, t' = q^-1 * (t * -1),  p_c = q^-1 * p + t'   (same operation order as orc_ba.cpp)
void quat_rotate(const double q[4], const double p[3], double out[3]) {
  double uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) out[i] = p[i] + q[3] * uv[i] + c[i];
}
}  // namespace

extern "C" {

// vo_utils.h:48-81.  points are visited in the given order (the reference iterates its unordered_map).
// Returns the number of projected points; proj_idx[i] = index of the landmark in `points`.
int orc_project_landmarks(const double* pose7, int model, const double* intr8, int width, int height,
                          const double* points, int n, double cam_z_threshold, double* proj_uv, int32_t* proj_idx) {
  const double qi[4] = {-pose7[0], -pose7[1], -pose7[2], pose7[3]};
  const double nt[3] = {pose7[4] * -1.0, pose7[5] * -1.0, pose7[6] * -1.0};
  double ti[3];
  quat_rotate(qi, nt, ti);
  int m = 0;
  for (int i = 0; i < n; i++) {
    double rp[3], pc[3];
    quat_rotate(qi, points + 3 * (size_t)i, rp);
    for (int k = 0; k < 3; k++) pc[k] = rp[k] + ti[k];
    if (pc[2] < cam_z_threshold) continue;
    double uv[2];
    orc_project(model, intr8, pc, uv);
    if (uv[0] > width || uv[1] > height || uv[0] < 0 || uv[1] < 0) continue;
    proj_uv[2 * (size_t)m] = uv[0];
    proj_uv[2 * (size_t)m + 1] = uv[1];
    proj_idx[m] = i;
    m++;
  }
  return m;
}

// vo_utils.h:83-167.  Landmark l's observation descriptors are obs_desc[lm_obs_start[l] .. lm_obs_start[l+1])
// (the reference walks landmarks.at(t_id).all_obs and looks each descriptor up in feature_corners).
// pairs: (keypoint index, landmark index), capacity 2*n_kp.  Returns the number of matches.
int orc_find_matches_landmarks(const double* kp_xy, const uint64_t* kp_desc, int n_kp, const double* proj_uv,
                               const int32_t* proj_lm, int n_proj, const int32_t* lm_obs_start, const uint64_t* obs_desc,
                               double match_max_dist_2d, int feature_match_threshold, double feature_match_dist_2_best,
                               int32_t* pairs) {
  typedef std::bitset<256> Desc;
  int nm = 0;
  for (int k = 0; k < n_kp; k++) {
    Desc dk;
    std::memcpy((void*)&dk, kp_desc + 4 * (size_t)k, 32);
    std::vector<std::pair<int64_t, int>> landmark_distances;
    for (int j = 0; j < n_proj; j++) {
      const double dx = kp_xy[2 * (size_t)k] - proj_uv[2 * (size_t)j];
      const double dy = kp_xy[2 * (size_t)k + 1] - proj_uv[2 * (size_t)j + 1];
      const double dist_2d = std::sqrt(dx * dx + dy * dy);  // Eigen (a - b).norm()
      if (dist_2d < match_max_dist_2d) {
        const int l = proj_lm[j];
        int minimal_dist = 256;
        for (int o = lm_obs_start[l]; o < lm_obs_start[l + 1]; o++) {
          Desc d;
          std::memcpy((void*)&d, obs_desc + 4 * (size_t)o, 32);
          const int dist = (int)(d ^ dk).count();
          if (dist < minimal_dist) minimal_dist = dist;
        }
        landmark_distances.push_back(std::make_pair((int64_t)l, minimal_dist));
      }
    }
    std::partial_sort(landmark_distances.begin(),
                      landmark_distances.begin() + std::min((int)landmark_distances.size(), 2),
                      landmark_distances.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
    if (landmark_distances.size() == 0) continue;
    if (landmark_distances[0].second >= feature_match_threshold) continue;
    if (landmark_distances.size() < 2) {
      if (256 < landmark_distances[0].second * feature_match_dist_2_best) continue;
    } else {
      if (landmark_distances[1].second < landmark_distances[0].second * feature_match_dist_2_best) continue;
    }
    pairs[2 * nm] = k;
    pairs[2 * nm + 1] = (int32_t)landmark_distances[0].first;
    nm++;
  }
  return nm;
}

}  // extern "C"
