// visnav_amd/keypoints.h -- drop-in for the keypoint / matcher operators of the reference's
// include/visnav/keypoints.h.  Same names, same argument meaning, same `void` convention, same
// namespace; the bodies marshal to the C ABI of libvslam_hip.so (include/vslam_hip.h), which runs
// the hand-written HIP kernels on an MI355X.  See INTEGRATION.md for the two-line patch to slam.cpp.
//
// Error behaviour: the reference's functions are `void` and abort on fatal errors
// (src/slam.cpp:1059, camera_models.h:495).  These wrappers do the same: a failing C-ABI call prints
// vsl_last_error() to stderr and calls std::abort() -- there is no silent CPU fallback.
#pragma once
#include <bitset>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../vslam_hip.h"
#include "device_select.h"

#if __has_include(<pangolin/image/managed_image.h>) && __has_include(<visnav/common_types.h>)
#include <pangolin/image/managed_image.h>
#include <visnav/common_types.h>
#else
#include "mirror_types.h"
#endif

namespace visnav {

static_assert(sizeof(std::bitset<256>) == 32, "std::bitset<256> must be four 64-bit words (libstdc++)");

namespace amd {
// One context per host thread: the reference calls this path from up to three threads
// (main, opt_thread, global_ba_thread -- src/slam.cpp:1557, :1780).
struct CtxHolder {
  vsl_ctx* c = nullptr;
  ~CtxHolder() {
    if (c) vsl_ctx_destroy(c);
  }
};
inline CtxHolder& ctx_holder() {
  static thread_local CtxHolder h;
  return h;
}
inline vsl_ctx* ctx() {
  CtxHolder& h = ctx_holder();
  if (!h.c) {
    // the same device rule as the RCCL communicator (device_select.h): VISNAV_AMD_DEVICE, LOCAL_RANK, RANK
    if (vsl_ctx_create(device_index_for(vsl_device_count()), &h.c) != VSL_OK) {
      std::fprintf(stderr, "visnav_amd: %s\n", vsl_last_error(nullptr));
      std::abort();
    }
  }
  return h.c;
}
// Destroys the calling thread's context NOW instead of in the thread-local destructor (a thread that is about to exit
// calls this when tools that hook the HIP runtime -- rocprofv3 -- are already tearing their own thread state down by
// the time thread-local destructors run).  A later ctx() on the same thread creates a fresh one.
inline void release_thread_ctx() {
  CtxHolder& h = ctx_holder();
  if (h.c) {
    vsl_ctx_destroy(h.c);
    h.c = nullptr;
  }
}
inline void check(int rc, const char* what) {
  if (rc != VSL_OK) {
    std::fprintf(stderr, "visnav_amd: %s failed (%d): %s\n", what, rc, vsl_last_error(ctx()));
    std::abort();
  }
}
}  // namespace amd

// include/visnav/keypoints.h:133-150
inline void detectKeypoints(const pangolin::ManagedImage<uint8_t>& img_raw, KeypointsData& kd, int num_features) {
  kd.corners.clear();
  kd.corner_angles.clear();
  kd.corner_descriptors.clear();
  if (num_features <= 0) return;
  std::vector<double> xy(2 * (size_t)num_features);
  int n = 0;
  amd::check(vsl_detect_keypoints(amd::ctx(), img_raw.ptr, (int)img_raw.w, (int)img_raw.h, img_raw.pitch, num_features,
                                  num_features, xy.data(), &n), "detectKeypoints");
  for (int i = 0; i < n; i++) kd.corners.emplace_back(xy[2 * i], xy[2 * i + 1]);
}

// include/visnav/keypoints.h:152-189
inline void computeAngles(const pangolin::ManagedImage<uint8_t>& img_raw, KeypointsData& kd, bool rotate_features) {
  kd.corner_angles.resize(kd.corners.size());
  if (kd.corners.empty()) return;
  static_assert(sizeof(kd.corners[0]) == 16, "Eigen::Vector2d must be two contiguous doubles");
  amd::check(vsl_compute_angles(amd::ctx(), img_raw.ptr, (int)img_raw.w, (int)img_raw.h, img_raw.pitch,
                                reinterpret_cast<const double*>(kd.corners.data()), (int)kd.corners.size(),
                                rotate_features ? 1 : 0, kd.corner_angles.data()), "computeAngles");
}

// include/visnav/keypoints.h:191-221
inline void computeDescriptors(const pangolin::ManagedImage<uint8_t>& img_raw, KeypointsData& kd) {
  kd.corner_descriptors.resize(kd.corners.size());
  if (kd.corners.empty()) return;
  amd::check(vsl_compute_descriptors(amd::ctx(), img_raw.ptr, (int)img_raw.w, (int)img_raw.h, img_raw.pitch,
                                     reinterpret_cast<const double*>(kd.corners.data()), kd.corner_angles.data(),
                                     (int)kd.corners.size(), reinterpret_cast<uint64_t*>(kd.corner_descriptors.data())),
             "computeDescriptors");
}

// include/visnav/keypoints.h:223-229 -- one fused device pass instead of three calls
inline void detectKeypointsAndDescriptors(const pangolin::ManagedImage<uint8_t>& img_raw, KeypointsData& kd,
                                          int num_features, bool rotate_features) {
  kd.corners.clear();
  kd.corner_angles.clear();
  kd.corner_descriptors.clear();
  if (num_features <= 0) return;
  std::vector<double> xy(2 * (size_t)num_features);
  kd.corner_angles.resize(num_features);
  kd.corner_descriptors.resize(num_features);
  int n = 0;
  amd::check(vsl_detect_describe(amd::ctx(), img_raw.ptr, (int)img_raw.w, (int)img_raw.h, img_raw.pitch, num_features,
                                 rotate_features ? 1 : 0, num_features, xy.data(), kd.corner_angles.data(),
                                 reinterpret_cast<uint64_t*>(kd.corner_descriptors.data()), &n),
             "detectKeypointsAndDescriptors");
  kd.corner_angles.resize(n);
  kd.corner_descriptors.resize(n);
  for (int i = 0; i < n; i++) kd.corners.emplace_back(xy[2 * i], xy[2 * i + 1]);
}

// include/visnav/keypoints.h:323-369
inline void matchDescriptors(const std::vector<std::bitset<256>>& corner_descriptors_1,
                             const std::vector<std::bitset<256>>& corner_descriptors_2,
                             std::vector<std::pair<int, int>>& matches, int threshold, double dist_2_best) {
  matches.clear();
  const int n1 = (int)corner_descriptors_1.size(), n2 = (int)corner_descriptors_2.size();
  if (n1 == 0 || n2 == 0) return;
  std::vector<int32_t> pairs(2 * (size_t)(n1 < n2 ? n1 : n2));
  int n = 0;
  amd::check(vsl_match_descriptors(amd::ctx(), reinterpret_cast<const uint64_t*>(corner_descriptors_1.data()), n1,
                                   reinterpret_cast<const uint64_t*>(corner_descriptors_2.data()), n2, threshold,
                                   dist_2_best, pairs.data(), &n), "matchDescriptors");
  matches.reserve(n);
  for (int i = 0; i < n; i++) matches.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
}

}  // namespace visnav
