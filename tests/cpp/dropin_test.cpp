// Exercises the drop-in wrappers (include/visnav_amd/*.h) the way src/slam.cpp calls the reference:
// detectKeypointsAndDescriptors on a stereo pair, matchDescriptors(70, 1.2), bundle_adjustment on a
// tiny map, vocabulary transform + score.  Inputs come from raw files written by the pytest driver,
// outputs go back as raw files; the driver compares them with the CPU oracle.
//   dropin_test <left.raw> <right.raw> <w> <h> <ba.bin> <voc.txt> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "visnav_amd/bow.h"
#include "visnav_amd/bundle_adjustment.h"
#include "visnav_amd/keypoints.h"
#include "visnav_amd/vo_utils.h"

using namespace visnav;

static void load_raw(const char* path, pangolin::ManagedImage<uint8_t>& img) {
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(img.ptr, 1, img.w * img.h, f) != img.w * img.h) std::abort();
  std::fclose(f);
}
template <class T>
static void put(std::ofstream& o, const T* p, size_t n) { o.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }
template <class T>
static void get(std::ifstream& i, T* p, size_t n) { i.read(reinterpret_cast<char*>(p), sizeof(T) * n); }

int main(int argc, char** argv) {
  if (argc != 8) return 2;
  const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
  pangolin::ManagedImage<uint8_t> left(w, h), right(w, h);
  load_raw(argv[1], left);
  load_raw(argv[2], right);
  KeypointsData kdl, kdr;
  detectKeypointsAndDescriptors(left, kdl, 1500, true);
  // the three separate calls on the right image (the API allows both styles)
  detectKeypoints(right, kdr, 1500);
  computeAngles(right, kdr, true);
  computeDescriptors(right, kdr);
  std::vector<std::pair<int, int>> matches;
  matchDescriptors(kdl.corner_descriptors, kdr.corner_descriptors, matches, 70, 1.2);

  // ---- bundle adjustment: binary problem written by the driver
  std::ifstream in(argv[5], std::ios::binary);
  int32_t nc, nl, no;
  get(in, &nc, 1); get(in, &nl, 1); get(in, &no, 1);
  std::vector<double> poses(7 * nc), points(3 * nl), uv(2 * no), intr(16);
  std::vector<uint8_t> fixed(nc);
  std::vector<int32_t> ocam(no), olm(no);
  get(in, poses.data(), poses.size()); get(in, fixed.data(), fixed.size()); get(in, intr.data(), 16);
  get(in, points.data(), points.size()); get(in, ocam.data(), no); get(in, olm.data(), no); get(in, uv.data(), uv.size());
  Cameras cameras;
  Landmarks landmarks;
  Corners corners;
  std::set<FrameCamId> fixed_set;
  Calibration calib;
  for (int k = 0; k < 2; k++) {
    auto c = std::make_shared<AbstractCameraD>();
    c->model = "ds";
    for (int j = 0; j < 8; j++) c->param[j] = intr[8 * k + j];
    calib.intrinsics.push_back(c);
  }
  for (int c = 0; c < nc; c++) {
    FrameCamId fcid(c / 2, c % 2);
    for (int j = 0; j < 7; j++) cameras[fcid].T_w_c.data()[j] = poses[7 * c + j];
    if (fixed[c]) fixed_set.insert(fcid);
    corners[fcid];
  }
  for (int l = 0; l < nl; l++) landmarks[l].p = Eigen::Vector3d(points[3 * l], points[3 * l + 1], points[3 * l + 2]);
  for (int i = 0; i < no; i++) {
    FrameCamId fcid(ocam[i] / 2, ocam[i] % 2);
    auto& kd = corners[fcid];
    const int fid = (int)kd.corners.size();
    kd.corners.emplace_back(uv[2 * i], uv[2 * i + 1]);
    landmarks[olm[i]].obs[fcid] = fid;
  }
  BundleAdjustmentOptions opts;
  opts.verbosity_level = 0;
  bundle_adjustment(corners, opts, fixed_set, calib, cameras, landmarks);

  // ---- vocabulary
  ORBVocabularyAmd voc;
  if (!voc.loadFromTextFile(argv[6])) return 3;
  DBoW2::BowVector bl, br;
  DBoW2::FeatureVector fl, fr;
  voc.transform(kdl.corner_descriptors, bl, fl, 4);
  voc.transform(kdr.corner_descriptors, br, fr, 4);
  const double s_lr = voc.score(bl, br), s_ll = voc.score(bl, bl);

  // ---- per-frame guided matching (src/slam.cpp:1099-1114, :1159): landmarks = the BA map, every
  // observation carries the left image's descriptor of the same feature index modulo the count
  for (auto& kv : landmarks) kv.second.all_obs = kv.second.obs;
  for (auto& kv : corners) {
    auto& kd = kv.second;
    kd.corner_descriptors.resize(kd.corners.size());
    for (size_t i = 0; i < kd.corners.size(); i++) kd.corner_descriptors[i] = kdl.corner_descriptors[i % kdl.corner_descriptors.size()];
  }
  auto cam = std::make_shared<AbstractCameraD>(*calib.intrinsics[0]);
  cam->width_ = w;
  cam->height_ = h;
  std::vector<Eigen::Vector2d, Eigen::aligned_allocator<Eigen::Vector2d>> projected_points;
  std::vector<TrackId> projected_track_ids;
  project_landmarks(cameras.begin()->second.T_w_c, cam, landmarks, 0.1, projected_points, projected_track_ids);
  LandmarkMatchData md;
  find_matches_landmarks(kdl, landmarks, corners, projected_points, projected_track_ids, 20.0, 70, 1.2, md);

  std::ofstream out(argv[7], std::ios::binary);
  int32_t n;
  n = (int32_t)kdl.corners.size(); put(out, &n, 1);
  put(out, reinterpret_cast<const double*>(kdl.corners.data()), 2 * (size_t)n);
  put(out, kdl.corner_angles.data(), n);
  put(out, reinterpret_cast<const uint64_t*>(kdl.corner_descriptors.data()), 4 * (size_t)n);
  n = (int32_t)kdr.corners.size(); put(out, &n, 1);
  put(out, reinterpret_cast<const double*>(kdr.corners.data()), 2 * (size_t)n);
  put(out, kdr.corner_angles.data(), n);
  put(out, reinterpret_cast<const uint64_t*>(kdr.corner_descriptors.data()), 4 * (size_t)n);
  n = (int32_t)matches.size(); put(out, &n, 1);
  for (auto& m : matches) { int32_t p[2] = {m.first, m.second}; put(out, p, 2); }
  for (int c = 0; c < nc; c++) put(out, cameras[FrameCamId(c / 2, c % 2)].T_w_c.data(), 7);
  for (int l = 0; l < nl; l++) put(out, landmarks[l].p.data(), 3);
  n = (int32_t)bl.size(); put(out, &n, 1);
  for (auto& kv : bl) { uint32_t id = kv.first; put(out, &id, 1); put(out, &kv.second, 1); }
  put(out, &s_lr, 1); put(out, &s_ll, 1);
  n = (int32_t)projected_track_ids.size(); put(out, &n, 1);
  for (int i = 0; i < n; i++) { int64_t id = projected_track_ids[i]; put(out, &id, 1); put(out, projected_points[i].data(), 2); }
  n = (int32_t)md.matches.size(); put(out, &n, 1);
  for (auto& mm : md.matches) { int64_t p[2] = {mm.first, mm.second}; put(out, p, 2); }
  return 0;
}
