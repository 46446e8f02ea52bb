// mirror_types.h -- minimal stand-ins for the third-party / reference types that appear in the
// reference's operator signatures, used ONLY when the real headers are not on the include path
// (this repository's own build and tests: Eigen, Sophus, Pangolin, TBB and the reference's
// common_types.h are not available offline).  Layouts match what the wrappers rely on:
//   Eigen::Vector2d / Vector3d : 2 / 3 contiguous doubles
//   Sophus::SE3d::data()       : qx qy qz qw tx ty tz            (include/visnav/serialization.h:153-162)
//   pangolin::ManagedImage<T>  : pitch, ptr, w, h                (pangolin/image/image.h)
//   visnav::KeypointsData      : include/visnav/common_types.h:111-122
//   visnav::FrameCamId, Camera, Landmark, Cameras, Landmarks, Corners, Calibration
//                              : include/visnav/common_types.h:64-96, :204-262, calibration.h:84-105
#pragma once
#include <bitset>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "dbow2_types.h"

namespace Eigen {
template <int N>
struct VectorNd {
  double v[N];
  VectorNd() { for (int i = 0; i < N; i++) v[i] = 0; }
  VectorNd(double a, double b) { static_assert(N == 2, ""); v[0] = a; v[1] = b; }
  VectorNd(double a, double b, double c) { static_assert(N == 3, ""); v[0] = a; v[1] = b; v[2] = c; }
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
  double* data() { return v; }
  const double* data() const { return v; }
};
typedef VectorNd<2> Vector2d;
typedef VectorNd<3> Vector3d;
template <class T>
using aligned_allocator = std::allocator<T>;
}  // namespace Eigen

namespace Sophus {
struct SE3d {
  double q_t[7] = {0, 0, 0, 1, 0, 0, 0};
  static constexpr int num_parameters = 7;
  double* data() { return q_t; }
  const double* data() const { return q_t; }
};
}  // namespace Sophus

namespace pangolin {
template <class T>
struct ManagedImage {
  size_t pitch = 0;
  T* ptr = nullptr;
  size_t w = 0, h = 0;
  ManagedImage() = default;
  ManagedImage(size_t w_, size_t h_) : pitch(w_ * sizeof(T)), ptr((T*)std::malloc(w_ * h_ * sizeof(T))), w(w_), h(h_) {}
  ManagedImage(const ManagedImage&) = delete;
  ~ManagedImage() { std::free(ptr); }
  T& operator()(size_t x, size_t y) { return ((T*)((unsigned char*)ptr + y * pitch))[x]; }
  const T& operator()(size_t x, size_t y) const { return ((const T*)((const unsigned char*)ptr + y * pitch))[x]; }
};
}  // namespace pangolin

namespace visnav {
using FrameId = int64_t;
using CamId = std::size_t;
using FeatureId = int;
using TrackId = int64_t;

struct FrameCamId {
  FrameId frame_id = 0;
  CamId cam_id = 0;
  FrameCamId() = default;
  FrameCamId(FrameId f, CamId c) : frame_id(f), cam_id(c) {}
  bool operator==(const FrameCamId& o) const { return frame_id == o.frame_id && cam_id == o.cam_id; }
  bool operator<(const FrameCamId& o) const { return frame_id == o.frame_id ? cam_id < o.cam_id : frame_id < o.frame_id; }
};
struct FrameCamIdHash {
  size_t operator()(const FrameCamId& f) const { return std::hash<int64_t>()(f.frame_id * 2 + (int64_t)f.cam_id); }
};

struct KeypointsData {
  std::vector<Eigen::Vector2d, Eigen::aligned_allocator<Eigen::Vector2d>> corners;
  std::vector<double> corner_angles;
  std::vector<std::bitset<256>> corner_descriptors;
};
using Corners = std::unordered_map<FrameCamId, KeypointsData, FrameCamIdHash>;  // tbb::concurrent_unordered_map upstream
using FeatureTrack = std::map<FrameCamId, FeatureId>;

// include/visnav/common_types.h:202
using DBoWInvertedFile = std::vector<std::vector<FrameCamId>>;
// include/visnav/common_types.h:204-222
struct Camera {
  Sophus::SE3d T_w_c;
  bool active = true;
  std::map<FrameCamId, int> covisible_weights;
  std::map<FrameCamId, Sophus::SE3d> covisible_rel_poses;
  FrameCamId last_fcid = FrameCamId(-1, 0);  // parent on the spanning tree; frame_id -1 = none (loop_closure_utils.h:522)
  DBoW2::BowVector bow_vector;
  DBoW2::FeatureVector feature_vector;
  std::map<TrackId, FeatureId> map_points;
  std::string img_path;
  bool modified = false;
};
using CovisibilityGraph = std::unordered_map<FrameCamId, std::set<FrameCamId>, FrameCamIdHash>;
// include/visnav/common_types.h:225-226
using ConsistentGroup = std::pair<std::set<FrameCamId>, int>;
using ConsistentGroups = std::vector<ConsistentGroup>;
// include/visnav/common_types.h:228-252
struct Landmark {
  Eigen::Vector3d p;
  Eigen::Vector3d p_c;   // position in the frame of the camera that created it
  FrameCamId from_fcid;
  FeatureTrack obs;      // inlier observations in the active window
  FeatureTrack all_obs;  // every observation (global BA)
  FeatureTrack outlier_obs;
  bool active = true;
  bool modified = false;
};
using Cameras = std::map<FrameCamId, Camera>;
using Landmarks = std::unordered_map<TrackId, Landmark>;

// AbstractCamera<double> of include/visnav/camera_models.h, reduced to what BA needs
struct AbstractCameraD {
  std::string model;  // "ds" | "pinhole" | "eucm" | "kb4"
  double param[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int width_ = 0, height_ = 0;
  std::string name() const { return model; }
  double* data() { return param; }
  const double* data() const { return param; }
  int width() const { return width_; }
  int height() const { return height_; }
};
// include/visnav/common_types.h:138-148
struct MatchData {
  Sophus::SE3d T_i_j;
  std::vector<std::pair<FeatureId, FeatureId>> matches;
  std::vector<std::pair<FeatureId, FeatureId>> inliers;
};
// include/visnav/common_types.h:150-160
struct LandmarkMatchData {
  Sophus::SE3d T_w_c;
  std::vector<std::pair<FeatureId, TrackId>> matches;
  std::vector<std::pair<FeatureId, TrackId>> inliers;
};
struct Calibration {
  std::vector<Sophus::SE3d> T_i_c;
  std::vector<std::shared_ptr<AbstractCameraD>> intrinsics;
};
}  // namespace visnav
