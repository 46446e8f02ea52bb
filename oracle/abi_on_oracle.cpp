// abi_on_oracle.cpp -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.
//
// The subset of the C ABI of include/vslam_hip.h that the headless pipeline (include/visnav_amd/harness/*.h,
// visual-slam_amd/apps/slam_headless.cpp) reaches through the operator-by-operator path, implemented on the CPU
// oracle (vslam_oracle.h).  oracle/Makefile links the UNCHANGED application and drop-in headers against it:
//     oracle/_cpu/slam_headless_cpu   =  next_step on the CPU restatement of the reference's operators
// which is the end-to-end CPU baseline bench.py times next to the MI355X run on the same rendered sequence
// (cpu_baseline_end_to_end) and the trajectory-equality check of tests/test_headless_gpu.py.  The product
// (visual-slam_amd/) never links or loads this file; the device-resident entry points (vsl_frames_*, vsl_map_*)
// are not implemented here on purpose and report VSL_ERR_NO_DEVICE.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/vslam_hip.h"
#include "vslam_oracle.h"

struct vsl_ctx {
  char err[256];
};
struct vsl_voc {
  orc_voc* v;
};
static char g_err[256] = "";
static int fail(vsl_ctx* c, int code, const char* msg) {
  std::snprintf(c ? c->err : g_err, 256, "%s", msg);
  return code;
}

extern "C" {
const char* vsl_version(void) { return "vslam ABI on the CPU oracle (baseline only)"; }
const char* vsl_last_error(const vsl_ctx* c) { return c ? c->err : g_err; }
int vsl_device_count(void) { return 1; }  // one "device": the host
int vsl_host_register(vsl_ctx*, void*, size_t) { return VSL_OK; }  // nothing to pin on the CPU path
int vsl_host_unregister(vsl_ctx*, void*) { return VSL_OK; }
int vsl_ctx_create(int, vsl_ctx** out) {
  *out = new vsl_ctx();
  (*out)->err[0] = 0;
  return VSL_OK;
}
int vsl_ctx_destroy(vsl_ctx* c) {
  delete c;
  return VSL_OK;
}

int vsl_detect_describe(vsl_ctx* c, const uint8_t* img, int w, int h, size_t pitch, int num_features, int rotate_features, int cap,
                        double* corners_xy, double* angles, uint64_t* desc, int* n_out) {
  std::vector<double> xy(2 * (size_t)num_features), ang(num_features);
  std::vector<uint64_t> d(4 * (size_t)num_features);
  const int n = orc_detect_describe(img, w, h, pitch, num_features, rotate_features, xy.data(), ang.data(), d.data());
  *n_out = n;
  if (n > cap) return fail(c, VSL_ERR_CAPACITY, "keypoint capacity");
  if (corners_xy) std::memcpy(corners_xy, xy.data(), sizeof(double) * 2 * (size_t)n);
  if (angles) std::memcpy(angles, ang.data(), sizeof(double) * (size_t)n);
  if (desc) std::memcpy(desc, d.data(), 32 * (size_t)n);
  return VSL_OK;
}
int vsl_detect_keypoints(vsl_ctx* c, const uint8_t* img, int w, int h, size_t pitch, int num_features, int cap, double* corners_xy,
                         int* n_out) {
  std::vector<double> xy(2 * (size_t)(num_features > 0 ? num_features : 1));
  const int n = orc_detect_keypoints(img, w, h, pitch, num_features, xy.data());
  *n_out = n;
  if (n > cap) return fail(c, VSL_ERR_CAPACITY, "keypoint capacity");
  std::memcpy(corners_xy, xy.data(), sizeof(double) * 2 * (size_t)n);
  return VSL_OK;
}
int vsl_compute_angles(vsl_ctx*, const uint8_t* img, int w, int h, size_t pitch, const double* corners_xy, int n, int rotate_features,
                       double* angles) {
  orc_compute_angles(img, w, h, pitch, corners_xy, n, rotate_features, angles);
  return VSL_OK;
}
int vsl_compute_descriptors(vsl_ctx*, const uint8_t* img, int w, int h, size_t pitch, const double* corners_xy, const double* angles,
                            int n, uint64_t* desc) {
  orc_compute_descriptors(img, w, h, pitch, corners_xy, angles, n, desc);
  return VSL_OK;
}
int vsl_match_descriptors(vsl_ctx*, const uint64_t* d1, int n1, const uint64_t* d2, int n2, int threshold, double dist_2_best,
                          int32_t* pairs, int* n_out) {
  *n_out = 0;
  if (n1 == 0 || n2 == 0) return VSL_OK;
  std::vector<int32_t> p(2 * (size_t)n1);
  const int n = orc_match_descriptors(d1, n1, d2, n2, threshold, dist_2_best, p.data());
  std::memcpy(pairs, p.data(), sizeof(int32_t) * 2 * (size_t)n);
  *n_out = n;
  return VSL_OK;
}
int vsl_project_landmarks(vsl_ctx*, const double* pose7, int cam_model, const double* intr8, int width, int height, const double* points,
                          int n, double cam_z_threshold, double* proj_uv, int32_t* proj_idx, int* n_out) {
  *n_out = orc_project_landmarks(pose7, cam_model, intr8, width, height, points, n, cam_z_threshold, proj_uv, proj_idx);
  return VSL_OK;
}
int vsl_find_matches_landmarks(vsl_ctx*, const double* kp_xy, const uint64_t* kp_desc, int n_kp, const double* proj_uv,
                               const int32_t* proj_lm, int n_proj, const int32_t* lm_obs_start, int, const uint64_t* obs_desc,
                               double match_max_dist_2d, int feature_match_threshold, double feature_match_dist_2_best, int32_t* pairs,
                               int* n_out) {
  *n_out = orc_find_matches_landmarks(kp_xy, kp_desc, n_kp, proj_uv, proj_lm, n_proj, lm_obs_start, obs_desc, match_max_dist_2d,
                                      feature_match_threshold, feature_match_dist_2_best, pairs);
  return VSL_OK;
}

static int run_ba(vsl_ctx* c, const vsl_ba_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary, double* intr_io = nullptr) {
  orc_ba_problem p;
  p.n_cams = prob->n_cams;
  p.n_lms = prob->n_lms;
  p.n_obs = prob->n_obs;
  p.cam_model[0] = prob->cam_model[0];
  p.cam_model[1] = prob->cam_model[1];
  p.poses = prob->poses;
  p.cam_fixed = prob->cam_fixed;
  p.cam_intr = prob->cam_intr;
  p.intr = prob->intr;
  p.points = prob->points;
  p.obs_cam = prob->obs_cam;
  p.obs_lm = prob->obs_lm;
  p.obs_uv = prob->obs_uv;
  orc_ba_options o;
  o.use_huber = opt->use_huber;
  o.huber_parameter = opt->huber_parameter;
  o.max_num_iterations = opt->max_num_iterations;
  o.verbosity = opt->verbosity;
  o.num_threads = (int32_t)std::thread::hardware_concurrency();  // ceres_options.num_threads (map_utils.h:409)
  orc_ba_summary s;
  if ((intr_io ? orc_bundle_adjust_intrinsics(&p, &o, intr_io, &s) : orc_bundle_adjust(&p, &o, &s)) != 0)
    return fail(c, VSL_ERR_NUMERIC, "oracle bundle adjustment failed");
  if (summary) {
    std::memset(summary, 0, sizeof(*summary));
    summary->initial_cost = s.initial_cost;
    summary->final_cost = s.final_cost;
    summary->iterations = s.iterations;
    summary->successful_steps = s.successful_steps;
    summary->termination = s.termination;
    summary->total_ms = s.total_ms;
  }
  return VSL_OK;
}
int vsl_bundle_adjust(vsl_ctx* c, const vsl_ba_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary) {
  return run_ba(c, prob, opt, summary);
}
int vsl_bundle_adjust_intrinsics(vsl_ctx* c, const vsl_ba_problem* prob, const vsl_ba_options* opt, double* intr_io,
                                 vsl_ba_summary* summary) {
  return run_ba(c, prob, opt, summary, intr_io);
}
int vsl_global_bundle_adjust(vsl_ctx* c, const vsl_ba_problem* prob, const vsl_ba_options* opt, vsl_allreduce_fn, void*, int, int,
                             vsl_ba_summary* summary) {
  return run_ba(c, prob, opt, summary);
}
int vsl_pose_graph_optimize(vsl_ctx* c, const vsl_pgo_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary) {
  orc_pgo_problem p;
  p.n_nodes = prob->n_nodes;
  p.n_edges = prob->n_edges;
  p.poses = prob->poses;
  p.node_fixed = prob->node_fixed;
  p.edge_a = prob->edge_a;
  p.edge_b = prob->edge_b;
  p.edge_meas = prob->edge_meas;
  orc_ba_options o;
  o.use_huber = opt->use_huber;
  o.huber_parameter = opt->huber_parameter;
  o.max_num_iterations = opt->max_num_iterations;
  o.verbosity = opt->verbosity;
  o.num_threads = 1;
  orc_ba_summary s;
  if (orc_pose_graph_optimize(&p, &o, &s) != 0) return fail(c, VSL_ERR_NUMERIC, "oracle pose graph optimisation failed");
  if (summary) {
    std::memset(summary, 0, sizeof(*summary));
    summary->initial_cost = s.initial_cost;
    summary->final_cost = s.final_cost;
    summary->iterations = s.iterations;
    summary->successful_steps = s.successful_steps;
    summary->termination = s.termination;
  }
  return VSL_OK;
}

int vsl_voc_load_text(vsl_ctx* c, const char* path, vsl_voc** out) {
  orc_voc* v = orc_voc_load_text(path);
  if (!v) return fail(c, VSL_ERR_IO, "vocabulary could not be read");
  *out = new vsl_voc{v};
  return VSL_OK;
}
int vsl_voc_destroy(vsl_voc* voc) {
  if (voc) {
    orc_voc_free(voc->v);
    delete voc;
  }
  return VSL_OK;
}
int vsl_voc_info(const vsl_voc* voc, int* k, int* L, int* n_nodes, int* n_words) {
  orc_voc_info(voc->v, k, L, n_nodes, n_words);
  return VSL_OK;
}
int vsl_bow_transform(vsl_ctx*, const vsl_voc* voc, const uint8_t* desc32, int n, int levelsup, uint32_t* word_ids, double* word_vals,
                      int* nnz, uint32_t* fv_node, uint32_t* fv_feat, int* fv_n) {
  orc_bow_transform(voc->v, desc32, n, levelsup, word_ids, word_vals, nnz, fv_node, fv_feat, fv_n);
  return VSL_OK;
}
int vsl_compute_bow_vector(vsl_ctx* c, const vsl_voc* voc, const uint8_t* img, int w, int h, size_t pitch, int num_features, int levelsup,
                           int cap, uint32_t* word_ids, double* word_vals, int* nnz, uint32_t* fv_node, uint32_t* fv_feat, int* fv_n) {
  std::vector<float> kp(5 * (size_t)cap);
  std::vector<uint8_t> desc(32 * (size_t)cap);
  const int n = orc_orb_detect_describe(img, w, h, pitch, num_features, kp.data(), desc.data(), cap);
  if (n < 0 || n > cap) return fail(c, VSL_ERR_CAPACITY, "ORB capacity");
  orc_bow_transform(voc->v, desc.data(), n, levelsup, word_ids, word_vals, nnz, fv_node, fv_feat, fv_n);
  return VSL_OK;
}
int vsl_bow_score_batch(vsl_ctx*, const uint32_t* q_ids, const double* q_vals, int q_nnz, const uint32_t* c_ids, const double* c_vals,
                        const int32_t* c_offsets, int m, double* scores) {
  for (int i = 0; i < m; i++)
    scores[i] = orc_bow_score_l1(q_ids, q_vals, q_nnz, c_ids + c_offsets[i], c_vals + c_offsets[i], c_offsets[i + 1] - c_offsets[i]);
  return VSL_OK;
}
void vsl_desc_bitset_to_bytes(const uint64_t* desc, int n, uint8_t* desc32) {
  for (int i = 0; i < n; i++) orc_bitset_to_bytes(desc + 4 * (size_t)i, desc32 + 32 * (size_t)i);
}

// device-resident path: not part of the CPU baseline
int vsl_frames_create(vsl_ctx* c, int, int, int, int, int, vsl_frames**) { return fail(c, VSL_ERR_NO_DEVICE, "no frame store on the CPU baseline"); }
int vsl_frames_destroy(vsl_frames*) { return VSL_OK; }
int vsl_frames_upload(vsl_ctx* c, vsl_frames*, int, int, const uint8_t*, size_t, size_t) { return fail(c, VSL_ERR_NO_DEVICE, "no frame store"); }
int vsl_frames_detect_describe(vsl_ctx* c, vsl_frames*, int, int, int, int) { return fail(c, VSL_ERR_NO_DEVICE, "no frame store"); }
int vsl_frames_match(vsl_ctx* c, vsl_frames*, const int32_t*, int, int, double) { return fail(c, VSL_ERR_NO_DEVICE, "no frame store"); }
int vsl_frames_download_keypoints(vsl_ctx* c, vsl_frames*, int, int, double*, double*, uint64_t*, int*) { return fail(c, VSL_ERR_NO_DEVICE, "no frame store"); }
int vsl_frames_download_matches(vsl_ctx* c, vsl_frames*, int, int, int32_t*, int*) { return fail(c, VSL_ERR_NO_DEVICE, "no frame store"); }
int vsl_map_create(vsl_ctx* c, int, int, vsl_map**) { return fail(c, VSL_ERR_NO_DEVICE, "no device map"); }
void vsl_map_destroy(vsl_map*) {}
int vsl_map_append_descriptors_from_frame(vsl_map*, vsl_frames*, int, int, const int32_t*, int*) { return VSL_ERR_NO_DEVICE; }
int vsl_map_set_landmarks(vsl_map*, int, const double*, const int32_t*, const int32_t*) { return VSL_ERR_NO_DEVICE; }
int vsl_map_track(vsl_map*, vsl_frames*, int, const double*, int, const double*, int, int, double, double, int, double, int32_t*, int*, int*) {
  return VSL_ERR_NO_DEVICE;
}
int vsl_map_track_corners(vsl_map*, vsl_frames*, int, const double*, int, const double*, int, int, double, double, int, double, int32_t*,
                          int*, int*, double*, int*) {
  return VSL_ERR_NO_DEVICE;
}
}  // extern "C"
