// Drives the loop-closure drop-ins of include/visnav_amd/loop_closure.h from a binary description written by
// tests/test_loop_closure_gpu.py and writes the results back (poses after pose_graph_optimization, the
// covisibility weights / graph edges of construct_visibility_graph).
#include <cstdio>
#include <fstream>
#include <vector>

#include "visnav_amd/loop_closure.h"

using namespace visnav;

template <class T>
static void get(std::ifstream& f, T* p, size_t n) {
  f.read(reinterpret_cast<char*>(p), sizeof(T) * n);
}
template <class T>
static void put(std::ofstream& f, const T* p, size_t n) {
  f.write(reinterpret_cast<const char*>(p), sizeof(T) * n);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::ifstream in(argv[1], std::ios::binary);
  std::ofstream out(argv[2], std::ios::binary);
  // ---- part 1: pose graph.  keyframes 0 .. K-1 (frame id = index, cam 0), the last one is the current keyframe
  int32_t K, essential, fixed_cur, loop_cand;
  get(in, &K, 1);
  get(in, &essential, 1);
  get(in, &fixed_cur, 1);
  get(in, &loop_cand, 1);
  std::vector<Camera> cams(K);
  for (int i = 0; i < K; i++) {
    get(in, cams[i].T_w_c.data(), 7);
    int32_t last, ncov;
    get(in, &last, 1);
    get(in, &ncov, 1);
    cams[i].last_fcid = FrameCamId(last, 0);
    for (int c = 0; c < ncov; c++) {
      int32_t other, weight;
      Sophus::SE3d rel;
      get(in, &other, 1);
      get(in, &weight, 1);
      get(in, rel.data(), 7);
      cams[i].covisible_weights[FrameCamId(other, 0)] = weight;
      cams[i].covisible_rel_poses[FrameCamId(other, 0)] = rel;
    }
  }
  Sophus::SE3d sim3;
  get(in, sim3.data(), 7);
  Cameras keyframes;
  for (int i = 0; i < K - 1; i++) keyframes[FrameCamId(i, 0)] = cams[i];
  Camera cur = cams[K - 1];
  LoopClosureOptions opt;
  opt.verbosity_level = 0;
  opt.set_current_kf_fixed = fixed_cur != 0;
  pose_graph_optimization(FrameCamId(K - 1, 0), cur, FrameCamId(loop_cand, 0), sim3, keyframes, essential, opt);
  for (int i = 0; i < K - 1; i++) put(out, keyframes[FrameCamId(i, 0)].T_w_c.data(), 7);
  put(out, cur.T_w_c.data(), 7);

  // ---- part 2: visibility graph.  L landmarks with observation lists (frame, cam, feature)
  int32_t L, C, threshold, new_frame;
  get(in, &L, 1);
  get(in, &C, 1);
  get(in, &threshold, 1);
  get(in, &new_frame, 1);
  Cameras cameras;
  for (int i = 0; i < C; i++) {
    int32_t fr, cm;
    get(in, &fr, 1);
    get(in, &cm, 1);
    Camera c;
    get(in, c.T_w_c.data(), 7);
    cameras[FrameCamId(fr, cm)] = c;
  }
  Landmarks lms;
  for (int l = 0; l < L; l++) {
    int32_t n;
    get(in, &n, 1);
    Landmark lm;
    for (int k = 0; k < n; k++) {
      int32_t o[3];
      get(in, o, 3);
      lm.all_obs[FrameCamId(o[0], o[1])] = o[2];
    }
    lms[l] = lm;
  }
  Camera nc;
  get(in, nc.T_w_c.data(), 7);
  CovisibilityGraph graph;
  construct_visibility_graph(FrameCamId(new_frame, 0), cameras, lms, nc, graph, threshold);
  int32_t n = (int32_t)nc.covisible_weights.size();
  put(out, &n, 1);
  for (const auto& kv : nc.covisible_weights) {
    int32_t rec[2] = {(int32_t)kv.first.frame_id, kv.second};
    put(out, rec, 2);
    put(out, nc.covisible_rel_poses.at(kv.first).data(), 7);
  }
  n = (int32_t)nc.map_points.size();
  put(out, &n, 1);
  n = (int32_t)graph[FrameCamId(new_frame, 0)].size();
  put(out, &n, 1);
  int32_t back = 0;
  for (const auto& kv : nc.covisible_weights) back += (int32_t)graph[kv.first].count(FrameCamId(new_frame, 0));
  put(out, &back, 1);
  return 0;
}
