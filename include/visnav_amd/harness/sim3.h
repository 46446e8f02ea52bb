// harness/sim3.h -- compute_sim3 of include/visnav/sim3.h:228-361: the relative pose between the current keyframe and
// a loop candidate from 3-D / bearing correspondences gathered through matchDescriptors (the MI355X drop-in) against
// the candidate and its covisible neighbours, then PnP-RANSAC + refinement (harness/pnp.h stands in for OpenGV:
// parity unpinned).  Despite its name the result is an SE(3) (the reference never estimates scale: stereo).
#pragma once
#include "tracking.h"

namespace visnav {
namespace harness {

inline bool compute_sim3(const Calibration& calib_cam, const FrameCamId& fcid1, const FrameCamId& fcid2, const Corners& feature_corners,
                         const Cameras& keyframes, const Landmarks& landmarks, const CovisibilityGraph& graph,
                         double reprojection_error_pnp_inlier_threshold_pixel, Sophus::SE3d& sim3, XorShift& rng) {
  std::vector<Vec3> points, bearings;
  correspondences_with_candidate(fcid1, fcid2, graph.at(fcid2), calib_cam, feature_corners, keyframes, landmarks, points, bearings, nullptr);
  if (points.size() < 5) return false;
  int total_iteration = 0;
  const int max_iteration = 10;
  while (true) {
    PnpRound r = pnp_round(bearings, points, reprojection_error_pnp_inlier_threshold_pixel, rng);
    if (r.ok) {
      sim3 = se3_mul(se3_inv(keyframes.at(fcid2).T_w_c), r.T_w_c);
      double l[6];
      amd::rt_log(amd::rt_of(sim3), l);
      if (std::fabs(l[0]) + std::fabs(l[1]) + std::fabs(l[2]) <= 5) return true;  // sim3.h:348-354
    }
    if (++total_iteration > max_iteration) return false;
  }
}

}  // namespace harness
}  // namespace visnav
