// keypoints_api.hip -- C-ABI entry points of the keypoint path (host-buffer and device-resident).
#include <cmath>
#include <string>

#include "vsl_common.h"

extern "C" int vsl_ctx_set_tie_eps(vsl_ctx* ctx, double eps) {
  if (!ctx || !(eps >= 0.0) || eps > 0.5) return vsl_fail(ctx, VSL_ERR_INVALID, "tie eps must be in [0, 0.5]");
  ctx->tie_eps = eps;
  return VSL_OK;
}

extern "C" int vsl_ctx_set_diagnostic(vsl_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return VSL_ERR_INVALID;
  const std::string k(name);
  if (k == "match_use_valu") ctx->match_use_valu = value != 0;
  else if (k == "match_no_stagger") ctx->match_no_stagger = value != 0;
  else if (k == "match_use_i8") ctx->match_use_i8 = value != 0;
  else if (k == "force_generic_describe") ctx->force_generic_describe = value != 0;
  else if (k == "describe_tile_min_images") ctx->describe_tile_min_images = value;
  else if (k == "bow_keys64") ctx->bow_keys64 = value != 0;
  else if (k == "match_two_pass") ctx->match_two_pass = value != 0;
  else if (k == "k1_list_cap") ctx->k1_list_cap = value;
  else if (k == "exact_list_cap") ctx->exact_list_cap = value;
  else if (k == "vo_chain_ticket") ctx->vo_chain_ticket = value;
  else if (k == "pending_desc_max") ctx->pending_desc_max = value < 1 ? 1 : value;
  else if (k == "ba_schur_entries") ctx->ba_schur_entries = value != 0;
  else if (k == "bow_no_wg_score") ctx->bow_no_wg_score = value != 0;
  else if (k == "ba_no_fused") ctx->ba_no_fused = value != 0;
  else if (k == "ba_host_lm") ctx->ba_host_lm = value != 0;
  else if (k == "ba_no_cyclic") ctx->ba_no_cyclic = value != 0;
  else if (k == "ba_schur_atomics") ctx->ba_schur_atomics = value != 0;
  else if (k == "ba_force_dense") ctx->ba_force_dense = value != 0;
  else if (k == "chol_no_fused") ctx->chol_no_fused = value != 0;
  else if (k == "chol_no_bcr") ctx->chol_no_bcr = value != 0;
  else if (k == "chol_one_ended") ctx->chol_one_ended = value != 0;
  else if (k == "select_bucket_cap") ctx->select_bucket_cap = value < 0 ? 0 : value;
  else return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ctx_set_diagnostic: unknown knob '%s'", name);
  return VSL_OK;
}

extern "C" int vsl_frames_detect_describe(vsl_ctx* ctx, vsl_frames* f, int first, int n, int num_features,
                                          int rotate_features) {
  if (!ctx || !f || first < 0 || n < 0 || first + n > f->max_images)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_detect_describe: bad slot range [%d, %d)", first, first + n);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = vsl_launch_detect(ctx, f, first, n, num_features);
  if (rc) return rc;
  return vsl_launch_describe(ctx, f, first, n, rotate_features, 0);
}

extern "C" int vsl_frames_resolve_ties(vsl_ctx* ctx, vsl_frames* f, int* n_resolved) {
  if (!ctx || !f) return VSL_ERR_INVALID;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  return vsl_resolve_ties(ctx, f, n_resolved);
}

extern "C" int vsl_frames_exact_fallbacks(const vsl_frames* f) {
  return f ? f->exact_fallbacks : VSL_ERR_INVALID;
}

static int check_image(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch) {
  if (!ctx) return VSL_ERR_INVALID;
  if (!img || w < 40 || h < 40 || pitch < (size_t)w)
    return vsl_fail(ctx, VSL_ERR_INVALID, "bad image (ptr=%p w=%d h=%d pitch=%zu; w,h >= 40 required)", (const void*)img, w, h, pitch);
  return VSL_OK;
}

static int upload_image(vsl_ctx* ctx, vsl_frames* f, const uint8_t* img, size_t pitch) {
  VSL_HIP(ctx, hipMemcpy2DAsync(f->images, f->w, img, pitch, f->w, f->h, hipMemcpyHostToDevice, ctx->stream));
  return VSL_OK;
}

extern "C" int vsl_detect_describe(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch, int num_features,
                                   int rotate_features, int cap, double* corners_xy, double* angles, uint64_t* desc,
                                   int* n_out) {
  int rc = check_image(ctx, img, w, h, pitch);
  if (rc) return rc;
  if (!n_out || num_features < 1 || cap < 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_detect_describe: bad arguments");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_frames* f = nullptr;
  rc = vsl_ctx_scratch_frames(ctx, w, h, num_features > ctx->scratch_feat ? num_features : ctx->scratch_feat, &f);
  if (rc) return rc;
  if ((rc = upload_image(ctx, f, img, pitch))) return rc;
  if ((rc = vsl_launch_detect(ctx, f, 0, 1, num_features))) return rc;
  if (angles || desc) {
    if ((rc = vsl_launch_describe(ctx, f, 0, 1, rotate_features, 0))) return rc;
  }
  // (the download resolves flagged near-tie samples in the same round trip)
  return vsl_frames_download_keypoints(ctx, f, 0, cap, corners_xy, angles, desc, n_out);
}

extern "C" int vsl_detect_keypoints(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch, int num_features,
                                    int cap, double* corners_xy, int* n_out) {
  return vsl_detect_describe(ctx, img, w, h, pitch, num_features, 0, cap, corners_xy, nullptr, nullptr, n_out);
}

// corners (doubles, truncated like `const int cx = p[0]`, keypoints.h:159-160) -> kp_xy of slot 0
static int upload_corners(vsl_ctx* ctx, vsl_frames* f, const double* corners_xy, int n) {
  std::vector<int32_t> xy(2 * (size_t)n + 1);
  for (int i = 0; i < 2 * n; i++) xy[i] = (int32_t)corners_xy[i];
  const int32_t cnt = n;
  if (n > 0) VSL_HIP(ctx, hipMemcpy(f->kp_xy, xy.data(), sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice));
  VSL_HIP(ctx, hipMemcpy(f->kp_count, &cnt, sizeof(cnt), hipMemcpyHostToDevice));
  return VSL_OK;
}

extern "C" int vsl_compute_angles(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch, const double* corners_xy,
                                  int n, int rotate_features, double* angles) {
  int rc = check_image(ctx, img, w, h, pitch);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!corners_xy || !angles))) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_compute_angles: bad arguments");
  if (n == 0) return VSL_OK;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_frames* f = nullptr;
  rc = vsl_ctx_scratch_frames(ctx, w, h, n > ctx->scratch_feat ? n : ctx->scratch_feat, &f);
  if (rc) return rc;
  if ((rc = upload_image(ctx, f, img, pitch))) return rc;
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = upload_corners(ctx, f, corners_xy, n))) return rc;
  if ((rc = vsl_launch_describe(ctx, f, 0, 1, rotate_features, 0))) return rc;
  f->ties_settled();  // descriptors of this call are not returned
  VSL_HIP(ctx, hipMemsetAsync(f->tie_count, 0, sizeof(int32_t), ctx->stream));
  int n_out = 0;
  std::vector<double> xy(2 * (size_t)n);
  return vsl_frames_download_keypoints(ctx, f, 0, n, xy.data(), angles, nullptr, &n_out);
}

extern "C" int vsl_compute_descriptors(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch,
                                       const double* corners_xy, const double* angles, int n, uint64_t* desc) {
  int rc = check_image(ctx, img, w, h, pitch);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!corners_xy || !angles || !desc)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_compute_descriptors: bad arguments");
  if (n == 0) return VSL_OK;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_frames* f = nullptr;
  rc = vsl_ctx_scratch_frames(ctx, w, h, n > ctx->scratch_feat ? n : ctx->scratch_feat, &f);
  if (rc) return rc;
  if ((rc = upload_image(ctx, f, img, pitch))) return rc;
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = upload_corners(ctx, f, corners_xy, n))) return rc;
  VSL_HIP(ctx, hipMemcpy(f->kp_angle, angles, sizeof(double) * n, hipMemcpyHostToDevice));
  if ((rc = vsl_launch_describe(ctx, f, 0, 1, 1, 1))) return rc;
  if ((rc = vsl_resolve_ties(ctx, f, nullptr))) return rc;
  VSL_HIP(ctx, hipMemcpy(desc, f->kp_desc, sizeof(uint64_t) * 4 * n, hipMemcpyDeviceToHost));
  return VSL_OK;
}

extern "C" int vsl_min_eig_response(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch, float* response) {
  int rc = check_image(ctx, img, w, h, pitch);
  if (rc) return rc;
  if (!response) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_min_eig_response: response is null");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_frames* f = nullptr;
  rc = vsl_ctx_scratch_frames(ctx, w, h, ctx->scratch_feat > 0 ? ctx->scratch_feat : 2048, &f);
  if (rc) return rc;
  if ((rc = upload_image(ctx, f, img, pitch))) return rc;
  f->store_response = true;
  rc = vsl_launch_detect(ctx, f, 0, 1, 1);
  f->store_response = false;
  if (rc) return rc;
  VSL_HIP(ctx, hipMemcpyAsync(response, f->response, sizeof(float) * (size_t)w * h, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}
