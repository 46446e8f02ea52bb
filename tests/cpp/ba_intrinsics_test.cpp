// visnav::bundle_adjustment with options.optimize_intrinsics = true through the drop-in wrapper
// (include/visnav_amd/bundle_adjustment.h; reference: include/visnav/map_utils.h:337-421, :397-403).
//   ba_intrinsics_test <ba.bin> <out.bin>      ba.bin as written by tests/test_dropin_cpp.py; out = poses, points, intrinsics
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "visnav_amd/bundle_adjustment.h"

using namespace visnav;

template <class T>
static void put(std::ofstream& o, const T* p, size_t n) { o.write(reinterpret_cast<const char*>(p), sizeof(T) * n); }
template <class T>
static void get(std::ifstream& i, T* p, size_t n) { i.read(reinterpret_cast<char*>(p), sizeof(T) * n); }

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  std::ifstream in(argv[1], std::ios::binary);
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
  opts.optimize_intrinsics = true;
  bundle_adjustment(corners, opts, fixed_set, calib, cameras, landmarks);
  std::ofstream out(argv[2], std::ios::binary);
  for (int c = 0; c < nc; c++) put(out, cameras[FrameCamId(c / 2, c % 2)].T_w_c.data(), 7);
  for (int l = 0; l < nl; l++) put(out, landmarks[l].p.data(), 3);
  for (int k = 0; k < 2; k++) put(out, calib.intrinsics[k]->data(), 8);
  return 0;
}
