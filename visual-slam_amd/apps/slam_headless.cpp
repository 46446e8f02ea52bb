// slam_headless -- the reference's odometry loop without its GUI: load an EuRoC-layout dataset and a
// calibration file (src/slam.cpp:1006-1079), call next_step for every frame (src/slam.cpp:1087-1458 via
// include/visnav_amd/harness/odometry.h), report frames/s and the ATE against the dataset's ground truth
// (src/slam.cpp:1618-1722) as one JSON line.  Every hot-path operator runs on the MI355X through
// libvslam_hip.so; there is no CPU fallback.
//
//   slam_headless --dataset-path <dir with cam0/ cam1/ ...> --cam-calib <calib.json>
//                 [--voc-path ORBvoc.txt] [--replicas N] [--frames N] [--async-ba] [--fused] [--traj out.csv] [--kf-min-inliers N] [--max-kfs N]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include <algorithm>
#include <cmath>
#include <memory>

#include "visnav_amd/harness/odometry.h"

using namespace visnav;
using namespace visnav::harness;

int main(int argc, char** argv) {
  std::string dataset, calib_path, traj_path, voc_path;
  int max_frames = -1;
  bool lookahead = true;  // fused mode: detect of frame t+1 enqueued under the host work of frame t
  int replicas = 1;       // BASELINE configs[3] on one GPU: that many independent streams, one host thread each
  std::string drop_spec;  // "a-b": frames a..b are replaced by a blank image (forces tracking loss -> relocalisation)
  bool trace = false;      // one stderr line per frame (matches, inliers, tracking state)
  int reloc_check = -1;    // >= 0: after the run, relocalise frame F from scratch (relocalize_camera with a displaced pose
                           // prior) against the finished map and report how far the result is from the tracked pose
  std::string drift_spec;  // "F:dx,dy,dz": before frame F the pose estimate is displaced by (dx, dy, dz) metres -- a test hook
                           // that stands in for accumulated drift (the rendered room is too small to drift by itself)
  OdometryOptions opt;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto need = [&](const char* what) {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "missing value for %s\n", what);
        std::exit(2);
      }
      return std::string(argv[++i]);
    };
    if (a == "--dataset-path") dataset = need("--dataset-path");
    else if (a == "--cam-calib") calib_path = need("--cam-calib");
    else if (a == "--frames") max_frames = std::atoi(need("--frames").c_str());
    else if (a == "--traj") traj_path = need("--traj");
    else if (a == "--voc-path") voc_path = need("--voc-path");
    else if (a == "--async-ba") opt.async_ba = true;
    else if (a == "--ba-merge-after") opt.ba_merge_after = std::atoi(need("--ba-merge-after").c_str());  // with --async-ba: deterministic hand-over
    else if (a == "--fused") opt.fused_tracking = true;
    else if (a == "--no-lookahead") lookahead = false;
    else if (a == "--replicas") replicas = std::atoi(need("--replicas").c_str());
    else if (a == "--kf-min-inliers") opt.new_kf_min_inliers = std::atoi(need("--kf-min-inliers").c_str());
    else if (a == "--max-kfs") opt.max_num_kfs = std::atoi(need("--max-kfs").c_str());
    else if (a == "--num-features") opt.num_features_per_image = std::atoi(need("--num-features").c_str());
    else if (a == "--ba-verbose") opt.ba_verbose = 1;
    else if (a == "--ba-iterations") opt.ba_max_iterations = std::atoi(need("--ba-iterations").c_str());
    else if (a == "--relocalization") opt.enable_relocalization = true;   // the reference's ui.relocalization (default on there)
    else if (a == "--loop-closure") opt.enable_loop_closure = true;       // ui.loop_closure
    else if (a == "--no-gba-after-loop") opt.enable_global_ba_after_loop_closure = false;
    else if (a == "--loop-time") opt.loop_closing_time_threshold = std::atoi(need("--loop-time").c_str());
    else if (a == "--num-consistency") opt.num_consistency = std::atoi(need("--num-consistency").c_str());
    else if (a == "--motion-threshold") opt.motion_threshold = std::atof(need("--motion-threshold").c_str());
    else if (a == "--drop-frames") drop_spec = need("--drop-frames");
    else if (a == "--inject-drift") drift_spec = need("--inject-drift");
    else if (a == "--trace") trace = true;
    else if (a == "--reloc-check") reloc_check = std::atoi(need("--reloc-check").c_str());
    else if (a == "--force-loop") {  // "F:C" test hook, see OdometryOptions::force_loop_from
      const std::string v = need("--force-loop");
      std::sscanf(v.c_str(), "%d:%d", &opt.force_loop_from, &opt.force_loop_candidate);
    }
    else {
      std::fprintf(stderr, "unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  if (dataset.empty() || calib_path.empty()) {
    std::fprintf(stderr, "usage: slam_headless --dataset-path DIR --cam-calib FILE [--frames N] [--async-ba] [--traj FILE]\n");
    return 2;
  }
  Calibration calib;
  if (!load_calibration(calib_path, calib)) {
    std::fprintf(stderr, "could not load camera calibration %s\n", calib_path.c_str());  // src/slam.cpp:1056-1059
    return 1;
  }
  EurocDataset ds;
  if (!load_euroc(dataset, ds)) {
    std::fprintf(stderr, "No dataset found in %s\n", dataset.c_str());
    return 1;
  }
  int n_frames = (int)ds.timestamps.size();
  if (max_frames > 0 && max_frames < n_frames) n_frames = max_frames;

  typedef std::chrono::steady_clock Clk;
  // decode every image up front so that the pipeline rate below is not a PNG-decoder benchmark; the
  // decode time is reported separately
  std::vector<GreyImage> left(n_frames), right(n_frames);
  const auto d0 = Clk::now();
  for (int i = 0; i < n_frames; i++) {
    if (!load_image(ds.images.at(FrameCamId(i, 0)), left[i]) || !load_image(ds.images.at(FrameCamId(i, 1)), right[i])) {
      std::fprintf(stderr, "could not read the images of frame %d (%s)\n", i, ds.images.at(FrameCamId(i, 0)).c_str());
      return 1;
    }
  }
  const double decode_s = std::chrono::duration<double>(Clk::now() - d0).count();
  // page-lock the decoded images: the per-frame upload is then an asynchronous DMA transfer (the look-ahead of the
  // device-resident path enqueues it under the host work of the previous frame); failures are not fatal
  int n_pinned = 0;
  for (int i = 0; i < n_frames; i++) {
    n_pinned += vsl_host_register(amd::ctx(), left[i].px.data(), left[i].px.size()) == VSL_OK;
    n_pinned += vsl_host_register(amd::ctx(), right[i].px.data(), right[i].px.size()) == VSL_OK;
  }
  if (!drop_spec.empty()) {
    int a = 0, b = -1;
    if (std::sscanf(drop_spec.c_str(), "%d-%d", &a, &b) == 2)
      for (int i = a; i <= b && i < n_frames; i++) {
        std::fill(left[i].px.begin(), left[i].px.end(), (uint8_t)90);
        std::fill(right[i].px.begin(), right[i].px.end(), (uint8_t)90);
      }
  }

  Odometry odo(calib, opt);
  ORBVocabularyAmd voc;
  if (!voc_path.empty()) {
    if (!voc.loadFromTextFile(voc_path)) {
      std::fprintf(stderr, "could not load the vocabulary %s\n", voc_path.c_str());
      return 1;
    }
    odo.orb_voc = &voc;
  }
  // replicas - 1 further streams run the same sequence in their own threads (own HIP context, frame store
  // and map through the thread-local vsl_ctx); stream 0 below is the one that is reported in detail
  std::vector<std::unique_ptr<Odometry>> others;
  for (int r = 1; r < replicas; r++) {
    others.emplace_back(new Odometry(calib, opt));
    others.back()->orb_voc = odo.orb_voc;
  }
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  auto run_stream = [&](Odometry& o, int* kf_count) {
    {  // per-thread warm-up outside the timed region: context creation, code objects of every kernel of a keyframe
       // step and of a tracking step (a throw-away odometry object runs the first frames), pinned buffers
      KeypointsData kd;
      ImageRef l(left[0]);
      detectKeypointsAndDescriptors(l.img, kd, opt.num_features_per_image, opt.rotate_features);
      Odometry warm(calib, opt);
      warm.orb_voc = o.orb_voc;
      // ... and enough frames (~50 ms of work) for the chip's clock to have ramped up from idle
      const int n_warm = n_frames < 60 ? n_frames : 60;
      for (int i = 0; i < n_warm; i++) warm.next_step(left[i], right[i], (lookahead && i + 1 < n_warm) ? &left[i + 1] : nullptr);
      warm.finish();
      warm.release_device();
    }
    ready++;
    while (!go) std::this_thread::yield();
    int drift_frame = -1;
    double dd[4] = {0, 0, 0, 0};  // dx, dy, dz [m], yaw [deg] about the camera's y axis
    if (!drift_spec.empty()) std::sscanf(drift_spec.c_str(), "%d:%lf,%lf,%lf,%lf", &drift_frame, &dd[0], &dd[1], &dd[2], &dd[3]);
    for (int i = 0; i < n_frames; i++) {
      if (i == drift_frame) {
        const double h = 0.5 * dd[3] * 3.14159265358979323846 / 180.0;
        Sophus::SE3d yaw;  // identity
        yaw.data()[1] = std::sin(h);
        yaw.data()[3] = std::cos(h);
        o.current_pose = se3_mul(o.current_pose, yaw);
        o.last_pose = se3_mul(o.last_pose, yaw);
        for (int c = 0; c < 3; c++) {
          o.current_pose.data()[4 + c] += dd[c];
          o.last_pose.data()[4 + c] += dd[c];
        }
        o.take_keyframe = true;
      }
      const bool kf = o.take_keyframe;
      o.next_step(left[i], right[i], (lookahead && i + 1 < n_frames) ? &left[i + 1] : nullptr);
      if (kf_count) *kf_count += kf ? 1 : 0;
      if (kf_count && trace)
        std::fprintf(stderr, "frame %d kf %d matches %d inliers %d tracking %d lost %d reloc %d loops %d t = %.3f %.3f %.3f\n", i, (int)kf,
                     o.last_matches, o.last_inliers, (int)o.tracking_successful, o.n_tracking_lost, o.n_relocalized, o.n_loops_closed,
                     o.current_pose.data()[4], o.current_pose.data()[5], o.current_pose.data()[6]);
    }
    o.finish();
    o.release_device();
    amd::release_thread_ctx();  // explicitly, not in the thread-local destructor (rocprofv3 is torn down by then)
  };
  int n_kf = 0;
  std::vector<std::thread> threads;
  for (auto& o : others) threads.emplace_back([&run_stream, &o] { run_stream(*o, nullptr); });
  std::thread main_stream([&] { run_stream(odo, &n_kf); });
  while (ready < replicas) std::this_thread::yield();
  const auto t0 = Clk::now();
  go = true;
  main_stream.join();
  for (auto& t : threads) t.join();
  const double run_s = std::chrono::duration<double>(Clk::now() - t0).count();
  bool replicas_agree = true;
  for (auto& o : others)
    replicas_agree = replicas_agree && o->frame_poses.size() == odo.frame_poses.size() &&
                     std::memcmp(o->frame_poses.back().data(), odo.frame_poses.back().data(), 7 * sizeof(double)) == 0;

  // --reloc-check F: the relocalisation operator on its own (tracking.h:241-419): the frame's keypoints under a fresh
  // id, a pose prior 0.2 m off, no motion model; success = BoW candidate found + PnP against the candidate's map points
  int reloc_ok = -1;
  double reloc_err_m = -1.0;
  if (reloc_check >= 0 && reloc_check < n_frames && odo.orb_voc) {
    std::thread t([&] {
      const FrameCamId probe((FrameId)n_frames + 1000, 0);
      KeypointsData kd;
      ImageRef l(left[reloc_check]);
      detectKeypointsAndDescriptors(l.img, kd, opt.num_features_per_image, opt.rotate_features);
      odo.feature_corners[probe] = kd;
      Sophus::SE3d prior = odo.frame_poses[(size_t)reloc_check];
      prior.data()[4] += 0.2;
      LandmarkMatchData md;
      XorShift rng;
      const bool ok = relocalize_camera(probe, l.img, odo.calib_cam, odo.graph, odo.orb_voc, odo.orb_db, odo.cameras, Sophus::SE3d(), prior,
                                        odo.feature_corners, odo.landmarks, 1.0, opt.reprojection_error_pnp_inlier_threshold_pixel, md, rng);
      reloc_ok = ok ? 1 : 0;
      if (ok) {
        double e = 0;
        for (int c = 0; c < 3; c++) {
          const double d = md.T_w_c.data()[4 + c] - odo.frame_poses[(size_t)reloc_check].data()[4 + c];
          e += d * d;
        }
        reloc_err_m = std::sqrt(e);
      }
      amd::release_thread_ctx();
    });
    t.join();
  }
  int n_assoc = 0;
  double ate = odo.ate(ds.timestamps, ds.gt_t_ns, ds.gt_t_w_i, &n_assoc);
  if (!std::isfinite(ate)) ate = -1.0;  // a dataset without ground truth (nan is not JSON)
  if (!traj_path.empty()) {
    FILE* f = std::fopen(traj_path.c_str(), "w");
    if (f) {
      std::fprintf(f, "#timestamp,tx,ty,tz,qx,qy,qz,qw,keyframe\n");
      for (int i = 0; i < (int)odo.frame_poses.size(); i++) {
        const double* d = odo.frame_poses[i].data();
        std::fprintf(f, "%lld,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%d\n", (long long)ds.timestamps[i], d[4], d[5], d[6], d[0], d[1],
                     d[2], d[3], odo.cameras.count(FrameCamId(i, 0)) ? 1 : 0);
      }
      std::fclose(f);
    }
  }
  if (n_pinned == 2 * n_frames)
    for (int i = 0; i < n_frames; i++) {
      vsl_host_unregister(amd::ctx(), left[i].px.data());
      vsl_host_unregister(amd::ctx(), right[i].px.data());
    }
  size_t n_active = 0;
  for (const auto& kv : odo.landmarks) n_active += kv.second.active ? 1 : 0;
  const StageClock& c = odo.clock;
  std::printf(
      "{\"frames\": %d, \"keyframes\": %d, \"streams\": %d, \"streams_agree\": %s, \"frames_per_s\": %.2f, \"ms_per_frame\": %.3f, \"image_decode_s\": %.3f, "
      "\"ate_rmse_m\": %.6f, \"ate_associations\": %d, \"landmarks\": %zu, \"active_landmarks\": %zu, \"async_ba\": %s, \"fused_tracking\": %s, "
      "\"stage_ms_total\": {\"detect\": %.1f, \"stereo_match\": %.1f, \"project_match\": %.1f, \"localize\": %.1f, \"map\": %.1f, "
      "\"ba\": %.1f, \"bow\": %.1f, \"loop\": %.1f, \"global_ba\": %.1f}, \"ba_runs\": %d, \"bow_vectors\": %zu, "
      "\"tracking_lost\": %d, \"relocalized\": %d, \"loops_closed\": %d, \"global_ba_runs\": %d, \"reloc_check_ok\": %d, \"reloc_check_err_m\": %.6f}\n",
      n_frames, n_kf, replicas, replicas_agree ? "true" : "false", replicas * n_frames / run_s, 1e3 * run_s / n_frames, decode_s, ate, n_assoc, odo.landmarks.size(), n_active,
      opt.async_ba ? "true" : "false", opt.fused_tracking ? "true" : "false", c.detect_ms, c.stereo_match_ms, c.project_match_ms, c.localize_ms, c.map_ms, c.ba_ms, c.bow_ms, odo.loop_ms, odo.gba_ms, c.ba_runs, odo.bow_vectors.size(),
      odo.n_tracking_lost, odo.n_relocalized, odo.n_loops_closed, odo.n_global_ba, reloc_ok, reloc_err_m);
  amd::release_thread_ctx();  // the main thread's context (image registration), before static / thread-local teardown
  return 0;
}
