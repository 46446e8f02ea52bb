// visnav_amd/bundle_adjustment.h -- drop-in for visnav::bundle_adjustment
// (include/visnav/map_utils.h:319-421) and visnav::global_bundle_adjustment
// (include/visnav/loop_closure_utils.h:651-748).  The wrappers flatten the reference's node-based
// containers (Cameras = std::map, Landmarks = unordered_map, Corners = concurrent map) into the SoA
// arrays of vsl_ba_problem, call the MI355X solver, and write poses / landmark positions back in
// place -- the contract of the Ceres version (parameter blocks are the containers' own storage).
#pragma once
#include <stdexcept>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "keypoints.h"  // context holder + types

namespace visnav {

// include/visnav/map_utils.h:319-334
struct BundleAdjustmentOptions {
  int verbosity_level = 1;
  bool optimize_intrinsics = false;
  bool use_huber = true;
  double huber_parameter = 1.0;
  int max_num_iterations = 20;
};
// include/visnav/loop_closure_utils.h:651-663
struct GlobalBundleAdjustmentOptions {
  int verbosity_level = 1;
  bool use_huber = true;
  double huber_parameter = 1.0;
  int max_num_iterations = 20;
};

namespace amd {
inline int camera_model_id(const std::string& name) {
  if (name == "ds") return VSL_CAM_DS;
  if (name == "pinhole") return VSL_CAM_PINHOLE;
  if (name == "eucm") return VSL_CAM_EUCM;
  if (name == "kb4") return VSL_CAM_KB4;
  std::fprintf(stderr, "Camera model %s is not implemented.\n", name.c_str());  // camera_models.h:493-495
  std::abort();
}

template <bool kAllObs>
inline void run_ba(const Corners& feature_corners, bool use_huber, double huber_parameter, int max_num_iterations,
                   int verbosity_level, const std::set<FrameCamId>& fixed_cameras, Calibration& calib_cam,
                   Cameras& cameras, Landmarks& landmarks, bool optimize_intrinsics = false) {
  if (cameras.empty() || landmarks.empty()) return;
  std::vector<double> poses, points, uv, intr(16, 0.0);
  std::vector<uint8_t> fixed;
  std::vector<int32_t> cam_intr, obs_cam, obs_lm;
  std::vector<Camera*> cam_ptr;
  std::vector<Landmark*> lm_ptr;
  // cameras in std::map order; an observation list (std::map<FrameCamId, FeatureId>) is walked against it with two
  // cursors instead of two tree lookups per observation (20 k observations per local window)
  std::vector<FrameCamId> cam_id;
  std::vector<const KeypointsData*> cam_kd;
  for (auto& kv : cameras) {  // std::map order == the order Ceres receives the blocks (map_utils.h:359)
    cam_id.push_back(kv.first);
    const auto kd = feature_corners.find(kv.first);
    cam_kd.push_back(kd == feature_corners.end() ? nullptr : &kd->second);
    cam_ptr.push_back(&kv.second);
    const double* d = kv.second.T_w_c.data();
    poses.insert(poses.end(), d, d + 7);
    fixed.push_back(fixed_cameras.count(kv.first) ? 1 : 0);
    cam_intr.push_back((int32_t)kv.first.cam_id);
  }
  for (auto& kv : landmarks) {
    Landmark& lm = kv.second;
    const auto& track = kAllObs ? lm.all_obs : lm.obs;  // loop_closure_utils.h:706 vs map_utils.h:373
    const int li = (int)lm_ptr.size();
    lm_ptr.push_back(&lm);
    points.insert(points.end(), lm.p.data(), lm.p.data() + 3);
    size_t ci = 0;
    for (const auto& ob : track) {
      while (ci < cam_id.size() && cam_id[ci] < ob.first) ci++;
      if (ci == cam_id.size() || ob.first < cam_id[ci] || !cam_kd[ci])  // .at(): std::out_of_range like the reference
        throw std::out_of_range("bundle_adjustment: an observation refers to a camera / keypoint set that is not there");
      const auto& p_2d = cam_kd[ci]->corners[ob.second];
      obs_cam.push_back((int32_t)ci);
      obs_lm.push_back(li);
      uv.push_back(p_2d[0]);
      uv.push_back(p_2d[1]);
    }
  }
  if (obs_cam.empty()) return;
  vsl_ba_problem prob;
  prob.n_cams = (int32_t)cam_ptr.size();
  prob.n_lms = (int32_t)lm_ptr.size();
  prob.n_obs = (int32_t)obs_cam.size();
  for (int k = 0; k < 2; k++) {
    prob.cam_model[k] = camera_model_id(calib_cam.intrinsics[k]->name());
    const double* p = calib_cam.intrinsics[k]->data();
    for (int j = 0; j < 8; j++) intr[8 * k + j] = p[j];
  }
  prob.poses = poses.data();
  prob.cam_fixed = fixed.data();
  prob.cam_intr = cam_intr.data();
  prob.intr = intr.data();
  prob.points = points.data();
  prob.obs_cam = obs_cam.data();
  prob.obs_lm = obs_lm.data();
  prob.obs_uv = uv.data();
  vsl_ba_options opt;
  opt.use_huber = use_huber ? 1 : 0;
  opt.huber_parameter = huber_parameter;
  opt.max_num_iterations = max_num_iterations;
  opt.verbosity = 0;
  vsl_ba_summary sum;
#ifdef VISNAV_AMD_HAVE_RCCL
  // multi-GPU global bundle adjustment: every rank runs this call on the same map (rccl_world.h)
  if (kAllObs && RcclWorld::instance().enabled()) {
    RcclWorld& w = RcclWorld::instance();
    check(vsl_global_bundle_adjust(ctx(), &prob, &opt, &RcclWorld::allreduce, &w, w.rank(), w.world(), &sum),
          "global_bundle_adjustment (RCCL)");
  } else
#endif
  if (optimize_intrinsics) {  // map_utils.h:397-403: the intrinsics blocks stay variable and are written back
    check(vsl_bundle_adjust_intrinsics(ctx(), &prob, &opt, intr.data(), &sum), "bundle_adjustment (optimize_intrinsics)");
    for (int k = 0; k < 2; k++)
      for (int j = 0; j < 8; j++) calib_cam.intrinsics[k]->data()[j] = intr[8 * k + j];
  } else if (std::getenv("VISNAV_AMD_BA_SELFCHECK")) {  // diagnostic: the same problem solved twice must give the same bits
    const std::vector<double> poses0 = poses, points0 = points;
    check(vsl_bundle_adjust(ctx(), &prob, &opt, &sum), "bundle_adjustment");
    const std::vector<double> poses1 = poses, points1 = points;
    const int it1 = sum.iterations;
    poses = poses0;
    points = points0;
    prob.poses = poses.data();
    prob.points = points.data();
    check(vsl_bundle_adjust(ctx(), &prob, &opt, &sum), "bundle_adjustment");
    double dmax = 0;
    size_t nd = 0;
    for (size_t i = 0; i < poses.size(); i++) if (poses[i] != poses1[i]) { nd++; dmax = std::max(dmax, std::fabs(poses[i] - poses1[i])); }
    for (size_t i = 0; i < points.size(); i++) if (points[i] != points1[i]) { nd++; dmax = std::max(dmax, std::fabs(points[i] - points1[i])); }
    std::fprintf(stderr, "  BA selfcheck: %d cameras %d landmarks %d observations, iterations %d / %d, %zu values differ (max %.3g)\n", prob.n_cams, prob.n_lms,
                 prob.n_obs, it1, (int)sum.iterations, nd, dmax);
  } else
    check(vsl_bundle_adjust(ctx(), &prob, &opt, &sum), "bundle_adjustment");
  for (size_t c = 0; c < cam_ptr.size(); c++) {
    double* d = cam_ptr[c]->T_w_c.data();
    for (int j = 0; j < 7; j++) d[j] = poses[7 * c + j];
  }
  for (size_t l = 0; l < lm_ptr.size(); l++)
    for (int j = 0; j < 3; j++) lm_ptr[l]->p.data()[j] = points[3 * l + j];
  if (verbosity_level >= 1)  // stands in for summary.BriefReport() (map_utils.h:414-415)
    std::printf("vslam_hip BA: iterations %d, initial cost %.6e, final cost %.6e, termination %d, %.3f ms\n",
                sum.iterations, sum.initial_cost, sum.final_cost, sum.termination, sum.total_ms);
}
}  // namespace amd

// include/visnav/map_utils.h:337-421, including options.optimize_intrinsics (:397-403; the reference wires it to a
// hidden GUI variable that defaults to false, src/slam.cpp:304, :1545): the two intrinsics blocks are then optimised
// with the poses and landmarks and calib_cam.intrinsics is updated.
inline void bundle_adjustment(const Corners& feature_corners, const BundleAdjustmentOptions& options,
                              const std::set<FrameCamId>& fixed_cameras, Calibration& calib_cam, Cameras& cameras,
                              Landmarks& landmarks) {
  amd::run_ba<false>(feature_corners, options.use_huber, options.huber_parameter, options.max_num_iterations,
                     options.verbosity_level, fixed_cameras, calib_cam, cameras, landmarks, options.optimize_intrinsics);
}

// include/visnav/loop_closure_utils.h:672-748.  With rccl_world.h included first and VISNAV_AMD_WORLD > 1 (one process
// per GPU, every rank calling this on the same map) the solve is partitioned over the ranks and all-reduced through
// RCCL; otherwise it is the single-GPU solver.
inline void global_bundle_adjustment(const Corners& feature_corners, const GlobalBundleAdjustmentOptions& options,
                                     const std::set<FrameCamId>& fixed_cameras, Calibration& calib_cam, Cameras& cameras,
                                     Landmarks& landmarks) {
  amd::run_ba<true>(feature_corners, options.use_huber, options.huber_parameter, options.max_num_iterations,
                    options.verbosity_level, fixed_cameras, calib_cam, cameras, landmarks);
}

}  // namespace visnav
