// harness/odometry.h -- headless restatement of the reference's per-frame pipeline, i.e. the call order
// of next_step() (src/slam.cpp:1087-1458) without the GUI.  The relocalisation branch (track_camera /
// relocalize_camera, src/slam.cpp:1167-1191) and the loop-closure branch (detect_loop_closure -> compute_sim3 ->
// loop_closure -> global_ba, :1219-1258, :1287) are options like the reference's GUI switches (default off here: they
// need a vocabulary; the reference's defaults are on).  Written against the SAME operator names the
// reference calls: detectKeypointsAndDescriptors, matchDescriptors, project_landmarks,
// find_matches_landmarks, bundle_adjustment -- here the MI355X drop-ins of include/visnav_amd/ -- plus the
// host-side pieces the reference takes from OpenGV (harness/pnp.h) and its own small helpers:
//   computeEssential / findInliersEssential   include/visnav/matching_utils.h:56-88
//   localize_camera                           include/visnav/vo_utils.h:170-226
//   add_new_landmarks                         include/visnav/vo_utils.h:228-322
//   remove_old_keyframes                      include/visnav/vo_utils.h:324-380
//   optimize() + the merge-back of its result src/slam.cpp:1510-1571, :1379-1412
//   alignSVD / align_svd (ATE)                src/slam.cpp:1618-1722
#pragma once
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <vector>

#include "../bow.h"
#include "../bundle_adjustment.h"
#include "../keypoints.h"
#include "../loop_closure.h"
#include "../vo_utils.h"
#include "ate.h"
#include "camera.h"
#include "geometry.h"
#include "io.h"
#include "pnp.h"
#include "sim3.h"
#include "tracking.h"

namespace visnav {
namespace harness {

struct OdometryOptions {  // defaults = the pangolin::Var defaults of src/slam.cpp:258-309
  int num_features_per_image = 1500;
  bool rotate_features = true;
  int feature_match_max_dist = 70;
  double feature_match_test_next_best = 1.2;
  double match_max_dist_2d = 20.0;
  int new_kf_min_inliers = 80;
  int max_num_kfs = 10;
  double cam_z_threshold = 0.1;
  double reprojection_error_pnp_inlier_threshold_pixel = 3.0;
  double reprojection_error_huber_pixel = 1.0;
  int ba_max_iterations = 20;
  int ba_verbose = 0;
  bool async_ba = false;  // true: optimize() runs in its own thread like the reference (result depends on timing)
  // async_ba with a DETERMINISTIC hand-over: the optimised window is merged back exactly this many frames after the
  // keyframe that started it (the main thread waits there if the worker is not done), and no keyframe is taken while a
  // result is outstanding -- the reference's rule (!opt_running && !opt_finished, src/slam.cpp:1374-1377) with the
  // timing taken out.  1 = the hand-over of the synchronous mode (merge at the end of the next frame: identical
  // trajectories); larger values hide the optimisation under the tracking of the following frames.  0 = off.
  int ba_merge_after = 0;
  // true: the per-frame device work goes through the device-resident frame store + map (include/vslam_hip.h
  // "device-resident map"): the image is uploaded once, keypoints and descriptors stay in HBM, landmark
  // projection + guided matching is ONE call, stereo matching runs slot against slot, and keyframe
  // descriptors are copied device-to-device into the map's pool.  Same kernels, same matches, same
  // trajectory as the operator-by-operator sequence (tests/test_headless_gpu.py); fewer PCIe round trips.
  bool fused_tracking = false;
  // src/slam.cpp:244-247, :274-294 (the reference's defaults are true / true / true; they need --voc-path)
  bool enable_relocalization = false;
  bool enable_loop_closure = false;
  bool enable_global_ba_after_loop_closure = true;
  double motion_threshold = 0.5;
  int num_cov_threshold = 10;
  int num_ess_threshold = 30;
  int num_consistency = 3;
  int loop_closing_time_threshold = 500;  // frames between a keyframe and a loop candidate
  bool use_sim3 = true;
  bool fixed_current_kf = true;
  int gba_max_iterations = 20;
  // TEST HOOK (-1 = off): from keyframe frame id `force_loop_from` on, keyframe `force_loop_candidate` is handed to
  // the loop-closing stage as a consistent candidate once, whatever detect_loop_closure says -- the rendered box room
  // is not distinctive enough for the BoW consistency test to fire (every keyframe pair scores 0.27-0.38 with a random
  // vocabulary), so the stages behind detection (compute_sim3, loop_closure, pose graph, global BA) are exercised this way;
  // detection itself is tested on constructed vectors (tests/cpp/loop_closure_test.cpp)
  int force_loop_from = -1, force_loop_candidate = 0;
};

struct StageClock {
  double detect_ms = 0, stereo_match_ms = 0, project_match_ms = 0, localize_ms = 0, map_ms = 0, ba_ms = 0, bow_ms = 0;
  int ba_runs = 0;
};

// non-owning pangolin::ManagedImage over a decoded image
struct ImageRef {
  pangolin::ManagedImage<uint8_t> img;
  explicit ImageRef(const GreyImage& g) {
    img.ptr = const_cast<uint8_t*>(g.px.data());
    img.w = (size_t)g.w;
    img.h = (size_t)g.h;
    img.pitch = (size_t)g.w;
  }
  ~ImageRef() { img.ptr = nullptr; }
};

inline Vec3 unproject(const std::shared_ptr<AmdCameraD>& cam, const Eigen::Vector2d& p) {
  return harness::unproject(camera_kind(cam->name()), cam->data(), p[0], p[1]);
}
inline Vec3 to_vec3(const Eigen::Vector3d& p) { return {p[0], p[1], p[2]}; }
inline Eigen::Vector3d to_eigen(const Vec3& p) { return Eigen::Vector3d(p.x, p.y, p.z); }
inline Pose to_pose(const Sophus::SE3d& T) { return pose_from7(T.data()); }
inline Sophus::SE3d to_se3(const Pose& T) {
  Sophus::SE3d r;
  pose_to7(T, r.data());
  return r;
}

// matching_utils.h:56-62
inline Mat3 compute_essential(const Pose& T_0_1) { return skew(normalized(T_0_1.t)) * T_0_1.R; }

// matching_utils.h:64-88
inline void find_inliers_essential(const KeypointsData& kd1, const KeypointsData& kd2, const std::shared_ptr<AmdCameraD>& cam1,
                                   const std::shared_ptr<AmdCameraD>& cam2, const Mat3& E, double epipolar_error_threshold,
                                   MatchData& md) {
  md.inliers.clear();
  for (const auto& m : md.matches) {
    const Vec3 p0 = unproject(cam1, kd1.corners[m.first]);
    const Vec3 p1 = unproject(cam2, kd2.corners[m.second]);
    const double err = dot(p0, E * p1);
    if (!(std::fabs(err) > epipolar_error_threshold)) md.inliers.push_back(m);
  }
}

// vo_utils.h:170-226
inline void localize_camera(const Sophus::SE3d& current_pose, const std::shared_ptr<AmdCameraD>& cam, const KeypointsData& kdl,
                            const Landmarks& landmarks, double reprojection_error_pnp_inlier_threshold_pixel,
                            LandmarkMatchData& md, XorShift& rng) {
  md.inliers.clear();
  md.T_w_c = current_pose;  // default to previous pose if not enough inliers
  if (md.matches.size() < 10) return;
  std::vector<Vec3> points, bearings;
  for (const auto& kv : md.matches) {
    points.push_back(to_vec3(landmarks.at(kv.second).p));
    bearings.push_back(unproject(cam, kdl.corners[kv.first]));
  }
  const double threshold = 1.0 - std::cos(std::atan(reprojection_error_pnp_inlier_threshold_pixel / 500.0));
  RansacResult rr = ransac_p3p(bearings, points, threshold, rng);
  if (!rr.ok) return;
  const Pose refined = refine_pose(rr.T_w_c, bearings, points, rr.inliers);
  md.T_w_c = to_se3(refined);
  std::vector<int> inl;
  select_within(refined, bearings, points, threshold, inl);
  for (int i : inl) md.inliers.push_back(md.matches[i]);
}

// vo_utils.h:228-322
inline void add_new_landmarks(const FrameCamId fcidl, const FrameCamId fcidr, const KeypointsData& kdl, const KeypointsData& kdr,
                              const Calibration& calib_cam, const MatchData& md_stereo, const LandmarkMatchData& md,
                              Landmarks& landmarks, TrackId& next_landmark_id) {
  const Pose T_0_1 = inverse(to_pose(calib_cam.T_i_c[0])) * to_pose(calib_cam.T_i_c[1]);
  const Pose T_w_c = to_pose(md.T_w_c);
  std::map<FeatureId, FeatureId> stereo_of;  // left feature -> right feature (first inlier pair wins, like the linear scan)
  for (const auto& s : md_stereo.inliers) stereo_of.emplace(s.first, s.second);
  std::set<FeatureId> localized;
  for (const auto& kv : md.inliers) {
    const FeatureId f_id = kv.first;
    const TrackId t_id = kv.second;
    localized.insert(f_id);
    auto it = landmarks.find(t_id);
    if (it == landmarks.end()) continue;
    Landmark& lm = it->second;
    lm.modified = true;
    lm.obs.emplace(fcidl, f_id);
    lm.all_obs.emplace(fcidl, f_id);
    auto st = stereo_of.find(f_id);
    if (st != stereo_of.end()) {
      lm.obs.emplace(fcidr, st->second);
      lm.all_obs.emplace(fcidr, st->second);
    }
  }
  for (const auto& kv : md_stereo.inliers) {
    const FeatureId f_idl = kv.first, f_idr = kv.second;
    if (localized.count(f_idl)) continue;  // already attached to an existing landmark
    const Vec3 b1 = unproject(calib_cam.intrinsics[fcidl.cam_id], kdl.corners.at(f_idl));
    const Vec3 b2 = unproject(calib_cam.intrinsics[fcidr.cam_id], kdr.corners.at(f_idr));
    const Vec3 p_c = triangulate_midpoint(b1, b2, T_0_1.R, T_0_1.t);
    Landmark l;
    l.p = to_eigen(T_w_c * p_c);
    l.p_c = to_eigen(inverse(T_w_c) * (T_w_c * p_c));
    l.from_fcid = fcidl;
    l.active = true;
    l.obs.emplace(fcidl, f_idl);
    l.obs.emplace(fcidr, f_idr);
    l.all_obs.emplace(fcidl, f_idl);
    l.all_obs.emplace(fcidr, f_idr);
    landmarks.emplace(next_landmark_id, l);
    next_landmark_id++;
  }
}

// vo_utils.h:324-380
inline void remove_old_keyframes(const FrameCamId fcidl, const int max_num_kfs, Cameras& cameras, Landmarks& landmarks,
                                 std::set<FrameId>& kf_frames) {
  kf_frames.emplace(fcidl.frame_id);
  while ((int)kf_frames.size() > max_num_kfs) {
    const FrameId kf_id = *kf_frames.begin();
    kf_frames.erase(kf_frames.begin());
    const FrameCamId removed[2] = {FrameCamId(kf_id, 0), FrameCamId(kf_id, 1)};
    for (const auto& fcid : removed) {
      cameras.at(fcid).modified = true;
      cameras.at(fcid).active = false;
    }
    for (auto& tid_lm : landmarks)
      for (const auto& fcid : removed)
        if (tid_lm.second.obs.count(fcid)) {
          tid_lm.second.modified = true;
          tid_lm.second.obs.erase(fcid);
        }
  }
  for (auto& tid_lm : landmarks) {
    if (tid_lm.second.obs.size() == 0) {
      tid_lm.second.modified = true;
      tid_lm.second.active = false;
    } else {
      tid_lm.second.active = true;
    }
  }
}

class Odometry {
 public:
  Odometry(const Calibration& calib, const OdometryOptions& options) : calib_cam(calib), opt(options) {
    T_0_1 = inverse(to_pose(calib_cam.T_i_c[0])) * to_pose(calib_cam.T_i_c[1]);
  }
  ~Odometry() {
    stop_worker();
    release_device();
  }
  // The device objects belong to the calling thread's vsl_ctx (include/visnav_amd/keypoints.h: one context per
  // host thread): a thread that ran next_step must call this before it exits.
  void release_device() {
    if (dev_map) vsl_map_destroy(dev_map);
    if (dev_frames) vsl_frames_destroy(dev_frames);
    dev_map = nullptr;
    dev_frames = nullptr;
  }

  // ---- state, with the names of src/slam.cpp
  Calibration calib_cam;
  OdometryOptions opt;
  Corners feature_corners;
  Cameras cameras;
  Landmarks landmarks;
  std::set<FrameId> kf_frames;
  TrackId next_landmark_id = 0;
  Sophus::SE3d current_pose;
  bool take_keyframe = true;
  int current_frame = 0;
  FrameCamId last_kf_fcid = FrameCamId(-1, 0);
  std::vector<Sophus::SE3d> frame_poses;  // T_w_c of every processed frame (for inspection)
  // optional: the vocabulary of the reference's --voc-path.  With it every keyframe gets its BowVector /
  // FeatureVector like current_cam_left.bow_vector in src/slam.cpp:1206-1208 (the loop detector that would
  // consume them is outside this harness).
  const ORBVocabularyAmd* orb_voc = nullptr;
  std::map<FrameCamId, DBoW2::BowVector> bow_vectors;
  std::map<FrameCamId, DBoW2::FeatureVector> feature_vectors;
  StageClock clock;
  int last_inliers = 0, last_matches = 0;
  // relocalisation / loop closure state, with the names of src/slam.cpp:127-190
  Sophus::SE3d vel, last_pose;
  bool tracking_successful = true;
  CovisibilityGraph graph;
  DBoWInvertedFile orb_db;  // resized to the vocabulary size (src/slam.cpp:380)
  ConsistentGroups consistent_groups;
  std::vector<FrameCamId> enough_consistent_candidates;
  std::vector<std::pair<FrameCamId, FrameCamId>> loop_edges;
  bool pose_graph_opt_done = false;
  int n_tracking_lost = 0, n_relocalized = 0, n_loops_closed = 0, n_global_ba = 0;
  double loop_ms = 0, gba_ms = 0;

  // One step of the pipeline on the stereo pair of frame `current_frame` (the right image is only
  // looked at on keyframes).
  // `next_left` (fused mode only, may be null): the left image of the FOLLOWING frame.  Its upload and
  // detect / describe kernels are enqueued as soon as this frame's device results are on the host, so
  // they run on the GPU while the host does P3P-RANSAC, triangulation and map bookkeeping for this frame.
  // src/slam.cpp:1099-1114: with relocalisation on, the guided search projects with the constant-motion prediction
  Sophus::SE3d projection_pose() const {
    return (opt.enable_relocalization && tracking_successful) ? se3_mul(current_pose, vel) : current_pose;
  }
  // src/slam.cpp:1167-1191 / :1348-1372
  void localize(const FrameCamId& fcidl, const GreyImage& img_left, const KeypointsData& kdl, LandmarkMatchData& md) {
    if (!opt.enable_relocalization) {
      localize_camera(current_pose, calib_cam.intrinsics[0], kdl, landmarks, opt.reprojection_error_pnp_inlier_threshold_pixel, md, rng);
      current_pose = md.T_w_c;
      return;
    }
    tracking_successful = track_camera(current_pose, calib_cam.intrinsics[0], kdl, landmarks,
                                       opt.reprojection_error_pnp_inlier_threshold_pixel, md, vel, opt.motion_threshold,
                                       tracking_successful, rng);
    if (tracking_successful) {
      current_pose = md.T_w_c;
      return;
    }
    n_tracking_lost++;
    const Sophus::SE3d tracking_result = md.T_w_c;
    if (opt.fused_tracking && orb_voc) {  // lazily: this frame's descriptors for relocalize_camera's matching
      KeypointsData& kd = feature_corners[fcidl];
      if (kd.corner_descriptors.size() != kd.corners.size()) fused_download_corners(cur_base, kd);
    }
    ImageRef l(img_left);
    if (orb_voc && relocalize_camera(fcidl, l.img, calib_cam, graph, orb_voc, orb_db, cameras, vel, current_pose, feature_corners,
                                     landmarks, opt.motion_threshold, opt.reprojection_error_pnp_inlier_threshold_pixel, md, rng)) {
      current_pose = md.T_w_c;
      tracking_successful = true;
      n_relocalized++;
    } else {
      current_pose = tracking_result;
    }
  }

  void next_step(const GreyImage& img_left, const GreyImage& img_right, const GreyImage* next_left = nullptr) {
    typedef std::chrono::steady_clock Clk;
    auto ms = [](Clk::time_point a, Clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const FrameCamId fcidl(current_frame, 0), fcidr(current_frame, 1);
    if (orb_voc && orb_db.empty()) orb_db.resize(orb_voc->size());
    std::vector<Eigen::Vector2d, Eigen::aligned_allocator<Eigen::Vector2d>> projected_points;
    std::vector<TrackId> projected_track_ids;
    LandmarkMatchData md;
    if (take_keyframe) {
      take_keyframe = false;
      auto t0 = Clk::now();
      if (!opt.fused_tracking)
        project_landmarks(projection_pose(), calib_cam.intrinsics[0], landmarks, opt.cam_z_threshold, projected_points,
                          projected_track_ids);
      auto t1 = Clk::now();
      MatchData md_stereo;
      KeypointsData kdl, kdr;
      if (opt.fused_tracking) {
        fused_detect(img_left, &img_right);  // the left image is skipped when the previous step prefetched it
      } else {
        ImageRef l(img_left), r(img_right);
        detectKeypointsAndDescriptors(l.img, kdl, opt.num_features_per_image, opt.rotate_features);
        detectKeypointsAndDescriptors(r.img, kdr, opt.num_features_per_image, opt.rotate_features);
      }
      auto t2 = Clk::now();
      md_stereo.T_i_j = to_se3(T_0_1);
      const Mat3 E = compute_essential(T_0_1);
      if (opt.fused_tracking) {
        fused_track(md);  // before the downloads below: they synchronise the stream
        fused_stereo(kdl, kdr, md_stereo);
        fused_prefetch(next_left);
      } else {
        matchDescriptors(kdl.corner_descriptors, kdr.corner_descriptors, md_stereo.matches, opt.feature_match_max_dist,
                         opt.feature_match_test_next_best);
      }
      find_inliers_essential(kdl, kdr, calib_cam.intrinsics[0], calib_cam.intrinsics[1], E, 1e-3, md_stereo);
      auto t3 = Clk::now();
      feature_corners[fcidl] = kdl;
      feature_corners[fcidr] = kdr;
      if (!opt.fused_tracking)
        find_matches_landmarks(kdl, landmarks, feature_corners, projected_points, projected_track_ids, opt.match_max_dist_2d,
                               opt.feature_match_max_dist, opt.feature_match_test_next_best, md);
      auto t4 = Clk::now();
      localize(fcidl, img_left, kdl, md);
      auto t5 = Clk::now();
      const size_t lm_before = landmarks.size();
      add_new_landmarks(fcidl, fcidr, kdl, kdr, calib_cam, md_stereo, md, landmarks, next_landmark_id);
      if (std::getenv("VISNAV_AMD_TRACE_FRAMES")) {
        unsigned long long hl = 1469598103934665603ull, hr = hl, hm = hl;
        for (const auto& d : kdl.corner_descriptors) hl = (hl ^ (unsigned long long)(d.to_string().substr(0, 64).size() + d.count())) * 1099511628211ull;
        for (const auto& d : kdr.corner_descriptors) hr = (hr ^ (unsigned long long)d.count()) * 1099511628211ull;
        for (const auto& m : md_stereo.matches) hm = (hm ^ (unsigned long long)(m.first * 4099 + m.second)) * 1099511628211ull;
        if (std::getenv("VISNAV_AMD_TRACE_DESC") && (int)fcidl.frame_id == std::atoi(std::getenv("VISNAV_AMD_TRACE_DESC")))
          for (size_t i = 0; i < kdl.corner_descriptors.size(); i++)
            std::fprintf(stderr, "DESC %zu %.1f %.1f %.17g %s\n", i, kdl.corners[i][0], kdl.corners[i][1], kdl.corner_angles[i], kdl.corner_descriptors[i].to_string().c_str());
        std::fprintf(stderr, "  keyframe %d: kp L %zu R %zu (hash %llx %llx) stereo matches %zu (hash %llx) inliers %zu, landmarks %zu -> %zu\n", (int)fcidl.frame_id,
                     kdl.corners.size(), kdr.corners.size(), hl, hr, md_stereo.matches.size(), hm, md_stereo.inliers.size(), lm_before, landmarks.size());
      }
      Camera cam_left, cam_right;
      cam_left.T_w_c = current_pose;
      const bool graph_needed = opt.enable_relocalization || opt.enable_loop_closure;
      if (graph_needed) construct_visibility_graph(fcidl, cameras, landmarks, cam_left, graph, opt.num_cov_threshold);  // :1198
      cam_left.active = true;
      cam_left.last_fcid = last_kf_fcid;
      if (orb_voc) {  // src/slam.cpp:1205-1208
        auto tb = Clk::now();
        ImageRef l(img_left);
        compute_bow_vector(l.img, opt.num_features_per_image, orb_voc, cam_left.bow_vector, cam_left.feature_vector);
        bow_vectors[fcidl] = cam_left.bow_vector;
        feature_vectors[fcidl] = cam_left.feature_vector;
        clock.bow_ms += ms(tb, Clk::now());
      }
      cam_right.T_w_c = to_se3(to_pose(current_pose) * T_0_1);
      cam_right.active = true;
      if (opt.enable_loop_closure && orb_voc) {  // src/slam.cpp:1219-1258
        auto tl = Clk::now();
        bool loop_detected = detect_loop_closure(fcidl, cam_left, cameras, orb_db, orb_voc, graph, consistent_groups,
                                                 enough_consistent_candidates, opt.num_cov_threshold * 2, opt.num_consistency);
        if (opt.force_loop_from >= 0 && fcidl.frame_id >= opt.force_loop_from && n_loops_closed == 0 &&
            cameras.count(FrameCamId(opt.force_loop_candidate, 0))) {
          enough_consistent_candidates.assign(1, FrameCamId(opt.force_loop_candidate, 0));
          loop_detected = true;
        }
        if (loop_detected) {
          for (size_t i = 0; i < enough_consistent_candidates.size(); i++) {
            const FrameCamId cand = enough_consistent_candidates[i];
            if (fcidl.frame_id - cand.frame_id <= opt.loop_closing_time_threshold) continue;
            Sophus::SE3d sim3;
            if (!compute_sim3(calib_cam, fcidl, cand, feature_corners, cameras, landmarks, graph,
                              opt.reprojection_error_pnp_inlier_threshold_pixel, sim3, rng))
              continue;
            loop_edges.emplace_back(fcidl, cand);
            if (!opt.use_sim3) sim3 = Sophus::SE3d();
            LoopClosureOptions lco;
            lco.verbosity_level = opt.ba_verbose;
            lco.set_current_kf_fixed = opt.fixed_current_kf;
            loop_closure(fcidl, cam_left, cand, to_se3(T_0_1), sim3, cameras, landmarks, opt.num_ess_threshold, lco);
            n_loops_closed++;
            map_dirty = true;
            if (opt.enable_global_ba_after_loop_closure) pose_graph_opt_done = true;
          }
        }
        loop_ms += ms(tl, Clk::now());
      } else if (orb_voc && graph_needed) {
        insert_new_kf_to_db(fcidl, cam_left, orb_db);  // relocalisation alone still needs the inverted file
      }
      cameras[fcidl] = cam_left;
      cameras[fcidr] = cam_right;
      remove_old_keyframes(fcidl, opt.max_num_kfs, cameras, landmarks, kf_frames);
      if (opt.fused_tracking) fused_register_observations(fcidl, fcidr);
      auto t6 = Clk::now();
      optimize();
      if (pose_graph_opt_done) {  // src/slam.cpp:1285-1288
        auto tg = Clk::now();
        global_ba();
        gba_ms += ms(tg, Clk::now());
      }
      auto t7 = Clk::now();
      current_pose = cameras[fcidl].T_w_c;
      last_kf_fcid = fcidl;
      clock.project_match_ms += ms(t0, t1) + ms(t3, t4);
      clock.detect_ms += ms(t1, t2);
      clock.stereo_match_ms += ms(t2, t3);
      clock.localize_ms += ms(t4, t5);
      clock.map_ms += ms(t5, t6);
      clock.ba_ms += ms(t6, t7);
    } else {
      auto t0 = Clk::now();
      if (!opt.fused_tracking)
        project_landmarks(projection_pose(), calib_cam.intrinsics[0], landmarks, opt.cam_z_threshold, projected_points,
                          projected_track_ids);
      auto t1 = Clk::now();
      KeypointsData kdl;
      if (opt.fused_tracking) {
        fused_detect(img_left, nullptr);
      } else {
        ImageRef l(img_left);
        detectKeypointsAndDescriptors(l.img, kdl, opt.num_features_per_image, opt.rotate_features);
      }
      auto t2 = Clk::now();
      if (opt.fused_tracking) {
        // the positions ride along with the matches (one round trip per tracked frame).  The DESCRIPTORS of a frame that
        // is not a keyframe are needed on the host only when tracking is lost and relocalize_camera matches them against
        // keyframes (tracking.h:283): they are fetched then (localize()), not on every frame
        fused_track(md, &kdl);
        fused_prefetch(next_left);
        feature_corners[fcidl] = kdl;
      } else {
        feature_corners[fcidl] = kdl;
        find_matches_landmarks(kdl, landmarks, feature_corners, projected_points, projected_track_ids, opt.match_max_dist_2d,
                               opt.feature_match_max_dist, opt.feature_match_test_next_best, md);
      }
      auto t3 = Clk::now();
      localize(fcidl, img_left, kdl, md);
      auto t4 = Clk::now();
      if (opt.async_ba && opt.ba_merge_after > 0) {  // deterministic hand-over
        if (ba_pending) ba_age++;
        if ((int)md.inliers.size() < opt.new_kf_min_inliers && !ba_pending) take_keyframe = true;
        if (ba_pending && ba_age >= opt.ba_merge_after) merge_optimized();  // waits for the worker
      } else {
        if ((int)md.inliers.size() < opt.new_kf_min_inliers && !opt_running && !opt_finished) take_keyframe = true;
        if (!opt_running && opt_finished) merge_optimized();
      }
      auto t5 = Clk::now();
      clock.project_match_ms += ms(t0, t1) + ms(t2, t3);
      clock.detect_ms += ms(t1, t2);
      clock.localize_ms += ms(t3, t4);
      clock.map_ms += ms(t4, t5);
    }
    last_inliers = (int)md.inliers.size();
    last_matches = (int)md.matches.size();
    if (std::getenv("VISNAV_AMD_TRACE_FRAMES")) {  // one line per frame: where two runs part ways
      const double* t = current_pose.data() + 4;
      std::fprintf(stderr, "frame %d matches %d inliers %d landmarks %zu pose %.17g %.17g %.17g\n", (int)current_frame, last_matches,
                   last_inliers, landmarks.size(), t[0], t[1], t[2]);
    }
    frame_poses.push_back(current_pose);
    current_frame++;
    vel = se3_mul(se3_inv(last_pose), current_pose);  // src/slam.cpp:1300-1301, :1454-1455
    last_pose = current_pose;
  }

  // Wait for a running optimisation and merge it (end of sequence).
  void finish() {
    wait_worker();
    if (opt_finished || ba_pending) merge_optimized();
  }

  // src/slam.cpp:1712-1722: keyframe positions of the left camera in the body frame vs ground truth
  double ate(const std::vector<int64_t>& timestamps, const std::vector<int64_t>& gt_t_ns, const std::vector<Vec3>& gt_t_w_i,
             int* n_assoc = nullptr) const {
    std::vector<int64_t> est_t_ns;
    std::vector<Vec3> est_t_w_i;
    const Pose T_c_i = inverse(to_pose(calib_cam.T_i_c[0]));
    for (const auto& kv : cameras)
      if (kv.first.cam_id == 0) {
        est_t_w_i.push_back((to_pose(kv.second.T_w_c) * T_c_i).t);
        est_t_ns.push_back(timestamps[(size_t)kv.first.frame_id]);
      }
    return align_svd(est_t_ns, est_t_w_i, gt_t_ns, gt_t_w_i, n_assoc);
  }

 private:
  Pose T_0_1;
  XorShift rng;
  Cameras cameras_opt;
  Landmarks landmarks_opt;
  Corners corners_opt;  // keypoint positions of the active cameras, private to the optimisation thread
  Calibration calib_cam_opt;
  std::atomic<bool> opt_running{false}, opt_finished{false};
  bool ba_pending = false;  // main thread only: an optimisation was started and its result is not merged yet
  int ba_age = 0;           // frames since it was started
  // the reference starts a new std::thread per optimisation (src/slam.cpp:1557); here ONE worker thread lives as
  // long as the object, so its thread-local vsl_ctx (stream, scratch, code objects) is created once
  std::unique_ptr<std::thread> opt_thread;
  std::mutex job_mutex;
  std::condition_variable job_cv;
  std::function<void()> job;
  bool job_pending = false, worker_exit = false;

  void worker_loop() {
    while (true) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(job_mutex);
        job_cv.wait(lk, [this] { return job_pending || worker_exit; });
        if (worker_exit && !job_pending) {
          amd::release_thread_ctx();  // this thread's context goes now, not in a thread-local destructor
          return;
        }
        f = job;
        job_pending = false;
      }
      f();
      job_cv.notify_all();
    }
  }
  void submit_job(std::function<void()> f) {
    if (!opt_thread) opt_thread.reset(new std::thread([this] { worker_loop(); }));
    {
      std::lock_guard<std::mutex> lk(job_mutex);
      job = std::move(f);
      job_pending = true;
    }
    job_cv.notify_all();
  }
  void wait_worker() {  // returns when no optimisation is queued or running
    if (!opt_thread) return;
    std::unique_lock<std::mutex> lk(job_mutex);
    job_cv.wait(lk, [this] { return !job_pending && !opt_running; });
  }
  void stop_worker() {
    if (!opt_thread) return;
    wait_worker();
    {
      std::lock_guard<std::mutex> lk(job_mutex);
      worker_exit = true;
    }
    job_cv.notify_all();
    opt_thread->join();
    opt_thread.reset();
  }

  // src/slam.cpp:1510-1571
  void optimize() {
    if (opt.async_ba) {  // a forced keyframe (tracking lost, a test hook) may arrive while a result is outstanding:
      wait_worker();     // the worker owns cameras_opt / landmarks_opt until it is done
      if (opt_finished || ba_pending) merge_optimized();
    }
    cameras_opt.clear();
    landmarks_opt.clear();
    corners_opt.clear();
    // The snapshot the optimisation works on (src/slam.cpp:1511-1553 copies whole Camera / Landmark objects).  Only
    // what bundle_adjustment reads and merge_optimized writes back is copied here -- the pose, the position, the
    // window's observations: a Landmark's all_obs map and a Camera's map_points / BoW / covisibility members cost a
    // millisecond of node allocations per keyframe and are never touched by the optimisation.
    for (const auto& kv : landmarks)
      if (kv.second.active) {
        Landmark& l = landmarks_opt[kv.first];
        l.p = kv.second.p;
        l.obs = kv.second.obs;
        l.active = true;
      }
    for (const auto& kv : cameras)
      if (kv.second.active) {
        Camera& c = cameras_opt[kv.first];
        c.T_w_c = kv.second.T_w_c;
        c.active = true;
        KeypointsData kd;
        kd.corners = feature_corners.at(kv.first).corners;
        corners_opt[kv.first] = kd;
      }
    const FrameId fid = *kf_frames.begin();
    BundleAdjustmentOptions ba_options;
    ba_options.optimize_intrinsics = false;
    ba_options.use_huber = true;
    ba_options.huber_parameter = opt.reprojection_error_huber_pixel;
    ba_options.max_num_iterations = opt.ba_max_iterations;
    ba_options.verbosity_level = opt.ba_verbose;
    calib_cam_opt = calib_cam;
    opt_running = true;
    ba_pending = true;
    ba_age = 0;
    clock.ba_runs++;
    auto work = [this, fid, ba_options] {
      const std::set<FrameCamId> fixed_cameras = {FrameCamId(fid, 0), FrameCamId(fid, 1)};
      bundle_adjustment(corners_opt, ba_options, fixed_cameras, calib_cam_opt, cameras_opt, landmarks_opt);
      if (std::getenv("VISNAV_AMD_TRACE_FRAMES")) {
        double h = 0;
        for (const auto& kv : cameras_opt) h += kv.second.T_w_c.data()[4] * 1.7 + kv.second.T_w_c.data()[5];
        for (const auto& kv : landmarks_opt) h += kv.second.p[0] + kv.second.p[2] * 0.3;
        std::fprintf(stderr, "  local BA after frame %d: %zu cameras %zu landmarks checksum %.15g\n", (int)fid, cameras_opt.size(), landmarks_opt.size(), h);
      }
      opt_finished = true;
      opt_running = false;
    };
    if (opt.async_ba)
      submit_job(work);
    else
      work();
  }

  // src/slam.cpp:1379-1412
  void merge_optimized() {
    wait_worker();
    for (const auto& kv : landmarks_opt) {
      Landmark& lm = landmarks.at(kv.first);
      lm.p = kv.second.p;  // the one member the optimisation changes (the reference assigns the whole copy back)
      lm.modified = true;
      const Camera& from = cameras_opt.count(lm.from_fcid) ? cameras_opt.at(lm.from_fcid) : cameras.at(lm.from_fcid);
      lm.p_c = to_eigen(inverse(to_pose(from.T_w_c)) * to_vec3(lm.p));
    }
    for (const auto& kv : cameras_opt) {
      cameras.at(kv.first).T_w_c = kv.second.T_w_c;
      cameras.at(kv.first).modified = true;
    }
    calib_cam = calib_cam_opt;
    opt_finished = false;
    ba_pending = false;
    map_dirty = true;  // landmark positions moved
  }

  // src/slam.cpp:1741-1788 + the merge-back of :1410-1447.  The reference runs it in global_ba_thread and merges at a
  // later non-keyframe step, skipping what the local BA modified meanwhile; here it runs synchronously (reproducible
  // runs), so nothing is modified in between and every landmark / camera is taken over.
  void global_ba() {
    pose_graph_opt_done = false;
    Cameras cameras_gba = cameras;
    Landmarks landmarks_gba = landmarks;
    Calibration calib_cam_gba = calib_cam;
    Corners corners_gba;
    for (const auto& kv : cameras_gba) {
      KeypointsData kd;
      kd.corners = feature_corners.at(kv.first).corners;
      corners_gba[kv.first] = kd;
    }
    GlobalBundleAdjustmentOptions ba_options;
    ba_options.use_huber = true;
    ba_options.huber_parameter = opt.reprojection_error_huber_pixel;
    ba_options.max_num_iterations = opt.gba_max_iterations;
    ba_options.verbosity_level = opt.ba_verbose;
    const std::set<FrameCamId> fixed_cameras = {FrameCamId(0, 0), FrameCamId(0, 1)};
    global_bundle_adjustment(corners_gba, ba_options, fixed_cameras, calib_cam_gba, cameras_gba, landmarks_gba);
    n_global_ba++;
    for (const auto& kv : landmarks_gba) {
      Landmark& lm = landmarks.at(kv.first);
      lm.p = kv.second.p;
      lm.p_c = to_eigen(inverse(to_pose(cameras_gba.at(lm.from_fcid).T_w_c)) * to_vec3(lm.p));
    }
    for (const auto& kv : cameras_gba) {
      Camera& cam = cameras.at(kv.first);
      cam.T_w_c = kv.second.T_w_c;
      std::map<FrameCamId, Sophus::SE3d> rel;
      for (const auto& fr : cam.covisible_rel_poses) rel.emplace(fr.first, se3_mul(se3_inv(cam.T_w_c), cameras_gba.at(fr.first).T_w_c));
      cam.covisible_rel_poses = rel;
    }
    map_dirty = true;
  }

  // ---- fused device path (OdometryOptions::fused_tracking)
  vsl_frames* dev_frames = nullptr;  // slots cur_base / cur_base + 1 = left / right image of the current frame,
                                     // the other pair receives the look-ahead frame
  int cur_base = 0;
  bool left_prefetched = false;
  vsl_map* dev_map = nullptr;
  bool map_dirty = true;
  std::vector<TrackId> table_ids;    // landmark table row -> TrackId
  std::unordered_map<TrackId, std::vector<int32_t>> lm_pool;  // landmark -> pool entries of its observation descriptors

  void fused_init(const GreyImage& img) {
    if (dev_frames) return;
    amd::check(vsl_frames_create(amd::ctx(), 4, img.w, img.h, opt.num_features_per_image, 1, &dev_frames), "vsl_frames_create");
    amd::check(vsl_map_create(amd::ctx(), 8192, 32768, &dev_map), "vsl_map_create");
  }

  void fused_detect(const GreyImage& left, const GreyImage* right) {
    fused_init(left);
    vsl_ctx* c = amd::ctx();
    if (left_prefetched) {  // the previous step already put this frame's left image through detect / describe
      cur_base ^= 2;
      left_prefetched = false;
      if (right) {
        amd::check(vsl_frames_upload(c, dev_frames, cur_base + 1, 1, right->px.data(), (size_t)right->w, (size_t)right->w * right->h),
                   "vsl_frames_upload");
        amd::check(vsl_frames_detect_describe(c, dev_frames, cur_base + 1, 1, opt.num_features_per_image, opt.rotate_features ? 1 : 0),
                   "vsl_frames_detect_describe");
      }
      return;
    }
    amd::check(vsl_frames_upload(c, dev_frames, cur_base, 1, left.px.data(), (size_t)left.w, (size_t)left.w * left.h), "vsl_frames_upload");
    if (right)
      amd::check(vsl_frames_upload(c, dev_frames, cur_base + 1, 1, right->px.data(), (size_t)right->w, (size_t)right->w * right->h),
                 "vsl_frames_upload");
    amd::check(vsl_frames_detect_describe(c, dev_frames, cur_base, right ? 2 : 1, opt.num_features_per_image, opt.rotate_features ? 1 : 0),
               "vsl_frames_detect_describe");
  }

  // enqueue (asynchronously) the next frame's left image into the other slot pair
  void fused_prefetch(const GreyImage* next_left) {
    if (!next_left) return;
    vsl_ctx* c = amd::ctx();
    const int nb = cur_base ^ 2;
    amd::check(vsl_frames_upload(c, dev_frames, nb, 1, next_left->px.data(), (size_t)next_left->w, (size_t)next_left->w * next_left->h),
               "vsl_frames_upload");
    amd::check(vsl_frames_detect_describe(c, dev_frames, nb, 1, opt.num_features_per_image, opt.rotate_features ? 1 : 0),
               "vsl_frames_detect_describe");
    left_prefetched = true;
  }

  // project_landmarks + find_matches_landmarks of the reference against the device-resident table
  // `kd_out` (frames that are not keyframes, descriptors not needed on the host): the keypoint positions come back in
  // the same round trip as the matches
  void fused_track(LandmarkMatchData& md, KeypointsData* kd_out = nullptr) {
    md.matches.clear();
    if (map_dirty) {
      std::vector<double> pts;
      std::vector<int32_t> start(1, 0), idx;
      table_ids.clear();
      for (const auto& kv : landmarks) {  // the reference's iteration order defines the candidate order (vo_utils.h:60)
        table_ids.push_back(kv.first);
        pts.insert(pts.end(), kv.second.p.data(), kv.second.p.data() + 3);
        auto it = lm_pool.find(kv.first);  // one pool entry per all_obs element (order is irrelevant: a minimum is taken)
        if (it != lm_pool.end()) idx.insert(idx.end(), it->second.begin(), it->second.end());
        start.push_back((int32_t)idx.size());
      }
      amd::check(vsl_map_set_landmarks(dev_map, (int)table_ids.size(), pts.data(), start.data(), idx.data()), "vsl_map_set_landmarks");
      map_dirty = false;
    }
    std::vector<int32_t> pairs(2 * (size_t)opt.num_features_per_image);
    int n = 0, n_proj = 0;
    const auto& cam = calib_cam.intrinsics[0];
    // the same pose the operator path projects with (src/slam.cpp:1099-1114: the constant-motion prediction while
    // relocalisation is enabled and tracking is healthy)
    const Sophus::SE3d T_proj = projection_pose();
    std::vector<double> xy;
    int n_xy = 0;
    if (kd_out) xy.resize(2 * (size_t)opt.num_features_per_image);
    amd::check(vsl_map_track_corners(dev_map, dev_frames, cur_base, T_proj.data(), amd::camera_model_id(cam->name()), cam->data(),
                                     cam->width(), cam->height(), opt.cam_z_threshold, opt.match_max_dist_2d,
                                     opt.feature_match_max_dist, opt.feature_match_test_next_best, pairs.data(), &n, &n_proj,
                                     kd_out ? xy.data() : nullptr, kd_out ? &n_xy : nullptr),
               "vsl_map_track");
    for (int i = 0; i < n; i++) md.matches.emplace_back(pairs[2 * i], table_ids[(size_t)pairs[2 * i + 1]]);
    if (kd_out) {
      kd_out->corners.clear();
      for (int i = 0; i < n_xy; i++) kd_out->corners.emplace_back(xy[2 * i], xy[2 * i + 1]);
    }
  }

  void fused_download_corners(int slot, KeypointsData& kd) {
    // relocalisation / loop closure match keyframe descriptors on demand (tracking.h:283, sim3.h:252): then the
    // descriptors come along; otherwise only the positions leave the device
    const bool with_desc = opt.enable_relocalization || opt.enable_loop_closure;
    std::vector<double> xy(2 * (size_t)opt.num_features_per_image);
    if (with_desc) kd.corner_descriptors.resize((size_t)opt.num_features_per_image);
    int n = 0;
    amd::check(vsl_frames_download_keypoints(amd::ctx(), dev_frames, slot, opt.num_features_per_image, xy.data(), nullptr,
                                             with_desc ? reinterpret_cast<uint64_t*>(kd.corner_descriptors.data()) : nullptr, &n),
               "vsl_frames_download_keypoints");
    kd.corners.clear();
    for (int i = 0; i < n; i++) kd.corners.emplace_back(xy[2 * i], xy[2 * i + 1]);
    if (with_desc) kd.corner_descriptors.resize((size_t)n);
  }

  void fused_stereo(KeypointsData& kdl, KeypointsData& kdr, MatchData& md_stereo) {
    const int32_t sp[2] = {cur_base, cur_base + 1};
    vsl_ctx* c = amd::ctx();
    amd::check(vsl_frames_match(c, dev_frames, sp, 1, opt.feature_match_max_dist, opt.feature_match_test_next_best), "vsl_frames_match");
    std::vector<int32_t> out(2 * (size_t)opt.num_features_per_image);
    int n = 0;
    amd::check(vsl_frames_download_matches(c, dev_frames, 0, opt.num_features_per_image, out.data(), &n), "vsl_frames_download_matches");
    md_stereo.matches.clear();
    for (int i = 0; i < n; i++) md_stereo.matches.emplace_back(out[2 * i], out[2 * i + 1]);
    fused_download_corners(cur_base, kdl);
    fused_download_corners(cur_base + 1, kdr);
  }

  // the observations add_new_landmarks attached to this keyframe: copy their descriptors from the frame
  // store into the map's pool (device to device) and remember where they went
  void fused_register_observations(const FrameCamId fcidl, const FrameCamId fcidr) {
    const FrameCamId fc[2] = {fcidl, fcidr};
    for (int slot = 0; slot < 2; slot++) {
      std::vector<int32_t> ids;
      std::vector<TrackId> owner;
      for (const auto& kv : landmarks) {
        auto it = kv.second.all_obs.find(fc[slot]);
        if (it != kv.second.all_obs.end()) {
          ids.push_back(it->second);
          owner.push_back(kv.first);
        }
      }
      if (ids.empty()) continue;
      int first = 0;
      amd::check(vsl_map_append_descriptors_from_frame(dev_map, dev_frames, cur_base + slot, (int)ids.size(), ids.data(), &first),
                 "vsl_map_append_descriptors_from_frame");
      for (size_t k = 0; k < ids.size(); k++) lm_pool[owner[k]].push_back(first + (int32_t)k);
    }
    map_dirty = true;
  }
};

}  // namespace harness
}  // namespace visnav
