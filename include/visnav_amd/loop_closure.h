// visnav_amd/loop_closure.h -- drop-ins for the two pieces of include/visnav/loop_closure_utils.h that SURVEY.md
// 8(f) ranks next after the per-frame path:
//   construct_visibility_graph (:52-96)   host graph bookkeeping, restated as is (it is a counting loop over the
//                                          landmarks' observation lists; no device work)
//   pose_graph_optimization    (:446-587)  the edge selection is the reference's (spanning-tree edges, covisibility
//                                          edges above the essential threshold, the loop constraint), the
//                                          optimisation itself runs on the MI355X (vsl_pose_graph_optimize)
// Same names, arguments and in-place update convention as the reference.
#pragma once
#include <cmath>
#include <map>
#include <set>
#include <vector>

#include "bundle_adjustment.h"

namespace visnav {

// loop_closure_utils.h:430-436
struct LoopClosureOptions {
  int verbosity_level = 1;
  bool set_current_kf_fixed = true;
};

namespace amd {
// rigid-transform helpers on the Sophus::SE3d storage order (qx qy qz qw tx ty tz) -- only what the edge
// assembly needs when the real Sophus headers are not available
struct Rt {
  double q[4], t[3];
};
inline Rt rt_of(const Sophus::SE3d& T) {
  Rt r;
  const double* d = T.data();
  for (int i = 0; i < 4; i++) r.q[i] = d[i];
  for (int i = 0; i < 3; i++) r.t[i] = d[4 + i];
  return r;
}
inline void qrot(const double* q, const double* p, double* o) {
  double uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int i = 0; i < 3; i++) uv[i] += uv[i];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) o[i] = p[i] + q[3] * uv[i] + c[i];
}
inline Rt rt_inv(const Rt& a) {
  Rt r;
  r.q[0] = -a.q[0];
  r.q[1] = -a.q[1];
  r.q[2] = -a.q[2];
  r.q[3] = a.q[3];
  double m[3] = {-a.t[0], -a.t[1], -a.t[2]};
  qrot(r.q, m, r.t);
  return r;
}
inline Rt rt_mul(const Rt& a, const Rt& b) {
  Rt r;
  r.q[0] = a.q[3] * b.q[0] + a.q[0] * b.q[3] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
  r.q[1] = a.q[3] * b.q[1] - a.q[0] * b.q[2] + a.q[1] * b.q[3] + a.q[2] * b.q[0];
  r.q[2] = a.q[3] * b.q[2] + a.q[0] * b.q[1] - a.q[1] * b.q[0] + a.q[2] * b.q[3];
  r.q[3] = a.q[3] * b.q[3] - a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2];
  double rb[3];
  qrot(a.q, b.t, rb);
  for (int i = 0; i < 3; i++) r.t[i] = rb[i] + a.t[i];
  return r;
}
inline Sophus::SE3d se3_of(const Rt& a) {
  Sophus::SE3d T;
  double* d = T.data();
  const double n = std::sqrt(a.q[0] * a.q[0] + a.q[1] * a.q[1] + a.q[2] * a.q[2] + a.q[3] * a.q[3]);
  for (int i = 0; i < 4; i++) d[i] = a.q[i] / n;
  for (int i = 0; i < 3; i++) d[4 + i] = a.t[i];
  return T;
}
// Sophus::SE3::log: (upsilon, omega)
inline void rt_log(const Rt& a, double* out) {
  const double* q = a.q;
  const double sq_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], w = q[3];
  double two_atan, c = 1.0 / 12.0;
  if (sq_n < 1e-20) {
    two_atan = 2.0 / w - 2.0 / 3.0 * sq_n / (w * w * w);
  } else {
    const double n = std::sqrt(sq_n);
    const double half = w < 0 ? std::atan2(-n, -w) : std::atan2(n, w);
    two_atan = 2.0 * half / n;
    const double theta = two_atan * n;
    if (std::fabs(theta) >= 1e-6) c = (1.0 - theta * std::cos(0.5 * theta) / (2.0 * std::sin(0.5 * theta))) / (theta * theta);
  }
  const double om[3] = {two_atan * q[0], two_atan * q[1], two_atan * q[2]};
  const double* t = a.t;
  const double x[3] = {om[1] * t[2] - om[2] * t[1], om[2] * t[0] - om[0] * t[2], om[0] * t[1] - om[1] * t[0]};
  const double y[3] = {om[1] * x[2] - om[2] * x[1], om[2] * x[0] - om[0] * x[2], om[0] * x[1] - om[1] * x[0]};
  for (int i = 0; i < 3; i++) {
    out[i] = t[i] - 0.5 * x[i] + c * y[i];
    out[3 + i] = om[i];
  }
}
}  // namespace amd

// loop_closure_utils.h:52-96
inline void construct_visibility_graph(const FrameCamId& new_fcid, const Cameras& cameras, const Landmarks& landmarks,
                                       Camera& new_camera, CovisibilityGraph& graph, int threshold) {
  std::map<FrameCamId, int> share_lm_count;
  for (const auto& tid_lm : landmarks) {
    const Landmark& lm = tid_lm.second;
    const auto it = lm.all_obs.find(new_fcid);
    if (it == lm.all_obs.end()) continue;
    new_camera.map_points.emplace(tid_lm.first, it->second);
    for (const auto& ob : lm.all_obs)  // every camera that also observes this landmark (and is in `cameras`)
      if (cameras.count(ob.first)) share_lm_count[ob.first] += 1;
  }
  std::set<FrameCamId> new_edges;
  const amd::Rt T_c_w = amd::rt_inv(amd::rt_of(new_camera.T_w_c));
  for (const auto& fcid_count : share_lm_count) {
    if (fcid_count.first.cam_id != 0 || fcid_count.second < threshold) continue;
    new_camera.covisible_weights.emplace(fcid_count.first, fcid_count.second);
    new_camera.covisible_rel_poses.emplace(fcid_count.first,
                                           amd::se3_of(amd::rt_mul(T_c_w, amd::rt_of(cameras.at(fcid_count.first).T_w_c))));
    new_edges.insert(fcid_count.first);
    graph[fcid_count.first].insert(new_fcid);
  }
  graph[new_fcid] = new_edges;
}

// loop_closure_utils.h:446-587.  cur_kf is not part of `keyframes` (the reference adds it as its own block).
inline void pose_graph_optimization(const FrameCamId& cur_kf_fcid, Camera& cur_kf, const FrameCamId& loop_candidate_fcid,
                                    const Sophus::SE3d& sim3, Cameras& keyframes, int essential_threshold,
                                    const LoopClosureOptions& options) {
  (void)cur_kf_fcid;
  // node 0 = the current keyframe, further nodes are created as edges touch them
  std::vector<Camera*> node_cam(1, &cur_kf);
  std::map<FrameCamId, int> node_of;
  std::vector<int32_t> ea, eb;
  std::vector<double> meas;
  auto node = [&](const FrameCamId& f) {
    auto it = node_of.find(f);
    if (it != node_of.end()) return it->second;
    node_cam.push_back(&keyframes.at(f));  // .at(): std::out_of_range like the reference
    node_of.emplace(f, (int)node_cam.size() - 1);
    return (int)node_cam.size() - 1;
  };
  auto add_edge = [&](int a, int b, const amd::Rt& rel) {
    double l[6];
    amd::rt_log(rel, l);
    ea.push_back(a);
    eb.push_back(b);
    meas.insert(meas.end(), l, l + 6);
  };
  auto tree_and_covisibility_edges = [&](int a, const Camera& K) {
    const bool strong = K.covisible_weights.count(K.last_fcid) && K.covisible_weights.at(K.last_fcid) > essential_threshold;
    if (!strong && K.last_fcid.frame_id != -1)  // the spanning-tree edge, unless the covisibility loop adds it anyway
      add_edge(a, node(K.last_fcid), amd::rt_mul(amd::rt_inv(amd::rt_of(K.T_w_c)), amd::rt_of(keyframes.at(K.last_fcid).T_w_c)));
    for (const auto& kv : K.covisible_weights)
      if (kv.second > essential_threshold) add_edge(a, node(kv.first), amd::rt_of(K.covisible_rel_poses.at(kv.first)));
  };
  tree_and_covisibility_edges(0, cur_kf);
  add_edge(0, node(loop_candidate_fcid), amd::rt_inv(amd::rt_of(sim3)));  // the Sim(3) constraint (:514-521)
  for (FrameCamId f = cur_kf.last_fcid; f.frame_id != -1; f = keyframes.at(f).last_fcid) tree_and_covisibility_edges(node(f), keyframes.at(f));

  const int N = (int)node_cam.size();
  std::vector<double> poses(7 * (size_t)N);
  std::vector<uint8_t> fixed(N, 0);
  for (int i = 0; i < N; i++)
    for (int c = 0; c < 7; c++) poses[7 * (size_t)i + c] = node_cam[i]->T_w_c.data()[c];
  fixed[0] = options.set_current_kf_fixed ? 1 : 0;
  vsl_pgo_problem prob;
  prob.n_nodes = N;
  prob.n_edges = (int32_t)ea.size();
  prob.poses = poses.data();
  prob.node_fixed = fixed.data();
  prob.edge_a = ea.data();
  prob.edge_b = eb.data();
  prob.edge_meas = meas.data();
  vsl_ba_options opt;
  opt.use_huber = 1;
  opt.huber_parameter = 1.0;
  opt.max_num_iterations = 20;
  opt.verbosity = 0;
  vsl_ba_summary sum;
  amd::check(vsl_pose_graph_optimize(amd::ctx(), &prob, &opt, &sum), "pose_graph_optimization");
  for (int i = 0; i < N; i++)
    for (int c = 0; c < 7; c++) node_cam[i]->T_w_c.data()[c] = poses[7 * (size_t)i + c];
  if (options.verbosity_level >= 1)  // stands in for summary.BriefReport() (:585)
    std::printf("vslam_hip PGO: %d nodes, %d edges, iterations %d, initial cost %.6e, final cost %.6e, termination %d\n", N,
                prob.n_edges, sum.iterations, sum.initial_cost, sum.final_cost, sum.termination);
}

}  // namespace visnav
