// harness/tracking.h -- the relocalisation branch of the reference's next_step (src/slam.cpp:1167-1191, :1348-1372):
// include/visnav/tracking.h restated against the MI355X operators -- matchDescriptors and ORBVocabulary::score /
// compute_bow_vector are the drop-ins of include/visnav_amd/, the PnP pieces the reference takes from OpenGV are the
// harness's own (harness/pnp.h; OpenGV is an empty submodule: parity unpinned).  Same names, argument meaning and
// control flow; the cv::Ptr<cv::ORB> and image-path arguments become the decoded image.
//   track_camera                      tracking.h:57-160
//   detect_relocalization_candidate   tracking.h:169-222
//   relocalize_camera                 tracking.h:241-419
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>

#include "../bow.h"
#include "../keypoints.h"
#include "../loop_closure.h"
#include "camera.h"
#include "geometry.h"
#include "pnp.h"

namespace visnav {
namespace harness {

inline Vec3 unproject_cam(const std::shared_ptr<AmdCameraD>& cam, const Eigen::Vector2d& p) {
  return harness::unproject(camera_kind(cam->name()), cam->data(), p[0], p[1]);
}
inline Vec3 to_vec3_e(const Eigen::Vector3d& p) { return {p[0], p[1], p[2]}; }
inline Pose se3_to_pose(const Sophus::SE3d& T) { return pose_from7(T.data()); }
inline Sophus::SE3d pose_to_se3(const Pose& T) {
  Sophus::SE3d r;
  pose_to7(T, r.data());
  return r;
}
// sum |upsilon| of (log(a^-1 b) - log(vel)): the motion-model check of tracking.h:133-135
inline double motion_model_error(const Sophus::SE3d& current_pose, const Sophus::SE3d& T_w_c, const Sophus::SE3d& vel) {
  double l1[6], l2[6];
  amd::rt_log(amd::rt_mul(amd::rt_inv(amd::rt_of(current_pose)), amd::rt_of(T_w_c)), l1);
  amd::rt_log(amd::rt_of(vel), l2);
  return std::fabs(l1[0] - l2[0]) + std::fabs(l1[1] - l2[1]) + std::fabs(l1[2] - l2[2]);
}
inline Sophus::SE3d se3_mul(const Sophus::SE3d& a, const Sophus::SE3d& b) { return amd::se3_of(amd::rt_mul(amd::rt_of(a), amd::rt_of(b))); }
inline Sophus::SE3d se3_inv(const Sophus::SE3d& a) { return amd::se3_of(amd::rt_inv(amd::rt_of(a))); }

// one RANSAC(KNEIP) + nonlinear refinement + selectWithinDistance round, the block tracking.h:96-129 repeats
struct PnpRound {
  bool ok = false;
  Sophus::SE3d T_w_c;
  std::vector<int> inliers;
};
inline PnpRound pnp_round(const std::vector<Vec3>& bearings, const std::vector<Vec3>& points, double pixel_threshold, XorShift& rng) {
  PnpRound out;
  const double threshold = 1.0 - std::cos(std::atan(pixel_threshold / 500.0));
  RansacResult rr = ransac_p3p(bearings, points, threshold, rng);
  if (!rr.ok) return out;
  const Pose refined = refine_pose(rr.T_w_c, bearings, points, rr.inliers);
  out.T_w_c = pose_to_se3(refined);
  select_within(refined, bearings, points, threshold, out.inliers);
  out.ok = true;
  return out;
}

// tracking.h:57-160
inline bool track_camera(const Sophus::SE3d& current_pose, const std::shared_ptr<AmdCameraD>& cam, const KeypointsData& kdl,
                         const Landmarks& landmarks, const double reprojection_error_pnp_inlier_threshold_pixel,
                         LandmarkMatchData& md, const Sophus::SE3d& vel, double motion_threshold, bool last_tracking_successful,
                         XorShift& rng) {
  md.inliers.clear();
  md.T_w_c = current_pose;
  if (md.matches.size() < 10) {
    if (last_tracking_successful) md.T_w_c = se3_mul(current_pose, vel);  // constant-motion prediction
    return false;
  }
  const int max_iteration = 5;
  int total_iteration = 0;
  std::vector<Vec3> points, bearings;
  for (const auto& kv : md.matches) {
    points.push_back(to_vec3_e(landmarks.at(kv.second).p));
    bearings.push_back(unproject_cam(cam, kdl.corners[kv.first]));
  }
  while (true) {
    PnpRound r = pnp_round(bearings, points, reprojection_error_pnp_inlier_threshold_pixel, rng);
    if (r.ok) md.T_w_c = r.T_w_c;
    const double err = r.ok ? motion_model_error(current_pose, md.T_w_c, vel) : 1e30;
    if (err > motion_threshold) {
      md.T_w_c = last_tracking_successful ? se3_mul(current_pose, vel) : current_pose;
      total_iteration++;
    } else {
      for (int i : r.inliers) md.inliers.push_back(md.matches[(size_t)i]);
      return true;
    }
    if (total_iteration > max_iteration) break;
  }
  return false;
}

// tracking.h:169-222 (the scores of the surviving keyframes in ONE batched launch)
inline bool detect_relocalization_candidate(const ORBVocabularyAmd* voc, const DBoWInvertedFile& recognition_database,
                                            const DBoW2::BowVector& bow_vector, const Cameras& keyframes,
                                            std::vector<FrameCamId>& candidate_kf_fcids) {
  std::unordered_map<FrameCamId, int, FrameCamIdHash> num_sharing_words;
  std::vector<FrameCamId> first_seen;  // deterministic iteration order (the reference walks an unordered_map)
  bool has_any_sharing_words = false;
  for (const auto& wv : bow_vector) {
    if (wv.first >= recognition_database.size()) continue;
    has_any_sharing_words = true;
    for (const auto& rkf : recognition_database[wv.first]) {
      auto it = num_sharing_words.find(rkf);
      if (it != num_sharing_words.end()) {
        it->second += 1;
      } else {
        num_sharing_words[rkf] = 0;  // sic: the first shared word counts 0 (tracking.h:186)
        first_seen.push_back(rkf);
      }
    }
  }
  if (!has_any_sharing_words || num_sharing_words.empty()) return false;
  int max_num_sharing_words = 0;
  for (const auto& kv : num_sharing_words) max_num_sharing_words = std::max(max_num_sharing_words, kv.second);
  const int sharing_words_threshold = (int)(max_num_sharing_words * 0.8f);
  std::vector<FrameCamId> cand;
  std::vector<const DBoW2::BowVector*> bows;
  for (const auto& f : first_seen)
    if (num_sharing_words.at(f) > sharing_words_threshold) {
      cand.push_back(f);
      bows.push_back(&keyframes.at(f).bow_vector);
    }
  const std::vector<double> scores = voc->score_batch(bow_vector, bows);
  std::vector<std::pair<double, FrameCamId>> score_and_match;
  for (size_t i = 0; i < cand.size(); i++) score_and_match.emplace_back(scores[i], cand[i]);
  const int top = std::min((int)score_and_match.size(), 5);
  std::partial_sort(score_and_match.begin(), score_and_match.begin() + top, score_and_match.end(),
                    [](const std::pair<double, FrameCamId>& a, const std::pair<double, FrameCamId>& b) { return a.first > b.first; });
  for (int i = 0; i < top; i++) candidate_kf_fcids.push_back(score_and_match[(size_t)i].second);
  return true;
}

// 3-D / bearing correspondences between the map points seen by `ref_fcid` (and its covisible neighbours) and the
// keypoints of `fcid`: the block tracking.h:291-346 and sim3.h:244-292 share.  matchDescriptors puts the CANDIDATE's
// descriptors first (tracking.h:283).
inline void correspondences_with_candidate(const FrameCamId& fcid, const FrameCamId& ref_fcid, const std::set<FrameCamId>& neighbours,
                                           const Calibration& calib_cam, const Corners& feature_corners, const Cameras& keyframes,
                                           const Landmarks& landmarks, std::vector<Vec3>& points, std::vector<Vec3>& bearings,
                                           std::vector<std::pair<FeatureId, TrackId>>* first_matches) {
  std::set<TrackId> already_added_landmarks;
  std::set<FeatureId> already_added_features;
  const KeypointsData& cur = feature_corners.at(fcid);
  auto harvest = [&](const FrameCamId& other, bool record) {
    MatchData md;
    matchDescriptors(feature_corners.at(other).corner_descriptors, cur.corner_descriptors, md.matches, 70, 1.2);
    std::map<FeatureId, FeatureId> matches;
    for (const auto& m : md.matches) matches.emplace(m);
    for (const auto& kv : keyframes.at(other).map_points) {
      auto it = matches.find(kv.second);
      if (it == matches.end()) continue;
      if (already_added_landmarks.count(kv.first) || already_added_features.count(it->second)) continue;
      if (record && first_matches) first_matches->emplace_back(it->second, kv.first);
      points.push_back(to_vec3_e(landmarks.at(kv.first).p));
      bearings.push_back(unproject_cam(calib_cam.intrinsics[0], cur.corners[(size_t)it->second]));
      already_added_landmarks.insert(kv.first);
      already_added_features.insert(it->second);
    }
  };
  harvest(ref_fcid, true);
  for (const auto& nb : neighbours) {
    if (nb == fcid) continue;
    harvest(nb, false);
  }
}

// tracking.h:241-419.  `img` = the decoded left image of frame `fcid` (the reference re-reads it from img_path).
inline bool relocalize_camera(const FrameCamId& fcid, const pangolin::ManagedImage<uint8_t>& img, const Calibration& calib_cam,
                              const CovisibilityGraph& graph, const ORBVocabularyAmd* voc, const DBoWInvertedFile& recognition_database,
                              const Cameras& keyframes, const Sophus::SE3d vel, const Sophus::SE3d& current_pose,
                              const Corners& feature_corners, const Landmarks& landmarks, double motion_threshold,
                              double reprojection_error_pnp_inlier_threshold_pixel, LandmarkMatchData& lm_match_data, XorShift& rng) {
  lm_match_data.matches.clear();
  lm_match_data.inliers.clear();
  DBoW2::BowVector bow_vector;
  DBoW2::FeatureVector feature_vector;
  compute_bow_vector(img, 1500, voc, bow_vector, feature_vector);
  std::vector<FrameCamId> reloc_fcids;
  bool reloc_pose_good = false;
  const bool trace = std::getenv("VISNAV_AMD_TRACE") != nullptr;
  if (!detect_relocalization_candidate(voc, recognition_database, bow_vector, keyframes, reloc_fcids)) {
    if (trace) std::fprintf(stderr, "relocalize frame %lld: no BoW candidate\n", (long long)fcid.frame_id);
    return false;
  }
  for (const auto& reloc_fcid : reloc_fcids) {
    int total_iteration = 0;
    const int max_iteration = 5;
    const std::set<FrameCamId>& candidates = graph.at(reloc_fcid);
    while (!reloc_pose_good) {
      std::vector<Vec3> points, bearings;
      correspondences_with_candidate(fcid, reloc_fcid, candidates, calib_cam, feature_corners, keyframes, landmarks, points, bearings,
                                     &lm_match_data.matches);
      if (trace) std::fprintf(stderr, "relocalize frame %lld vs keyframe %lld: %zu correspondences\n", (long long)fcid.frame_id, (long long)reloc_fcid.frame_id, points.size());
      if (points.size() < 5) return false;
      PnpRound r = pnp_round(bearings, points, reprojection_error_pnp_inlier_threshold_pixel, rng);
      if (r.ok) lm_match_data.T_w_c = r.T_w_c;
      if (trace) std::fprintf(stderr, "   pnp ok %d inliers %zu motion error %.3f\n", (int)r.ok, r.inliers.size(), r.ok ? motion_model_error(current_pose, lm_match_data.T_w_c, vel) : -1.0);
      if (!r.ok || r.inliers.size() < 10) {
        total_iteration++;
      } else if (motion_model_error(current_pose, lm_match_data.T_w_c, vel) > motion_threshold) {
        total_iteration++;
      } else {
        reloc_pose_good = true;
        for (int idx : r.inliers)
          if ((size_t)idx < lm_match_data.matches.size()) lm_match_data.inliers.push_back(lm_match_data.matches[(size_t)idx]);
        break;
      }
      if (total_iteration > max_iteration) break;
    }
    if (reloc_pose_good) break;
  }
  return reloc_pose_good;
}

}  // namespace harness
}  // namespace visnav
