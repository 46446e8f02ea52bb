// visnav::global_bundle_adjustment through the C++ RCCL path (include/visnav_amd/rccl_world.h): one process per GPU;
// on a one-GPU box VISNAV_AMD_FORCE_RCCL=1 builds a one-rank communicator so that ncclAllReduce on the solver's stream
// is exercised.  Problem: the binary layout of tests/test_dropin_cpp.py; output: poses, points (raw doubles).
//   global_ba_rccl_test <ba.bin> <out.bin> <max_iterations>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "visnav_amd/rccl_world.h"  // before bundle_adjustment.h: turns the multi-GPU path on
#include "visnav_amd/bundle_adjustment.h"

using namespace visnav;
template <class T>
static void get(std::ifstream& i, T* p, size_t n) { i.read(reinterpret_cast<char*>(p), sizeof(T) * n); }

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  std::ifstream in(argv[1], std::ios::binary);
  int32_t nc, nl, no;
  get(in, &nc, 1); get(in, &nl, 1); get(in, &no, 1);
  std::vector<double> poses(7 * (size_t)nc), points(3 * (size_t)nl), uv(2 * (size_t)no), intr(16);
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
    landmarks[olm[i]].all_obs[fcid] = fid;  // global BA iterates all_obs (loop_closure_utils.h:706)
  }
  GlobalBundleAdjustmentOptions opts;
  opts.verbosity_level = 1;
  opts.max_num_iterations = std::atoi(argv[3]);
  global_bundle_adjustment(corners, opts, fixed_set, calib, cameras, landmarks);
  if (amd::RcclWorld::instance().rank() == 0) {
    std::ofstream out(argv[2], std::ios::binary);
    for (int c = 0; c < nc; c++) out.write(reinterpret_cast<const char*>(cameras[FrameCamId(c / 2, c % 2)].T_w_c.data()), 56);
    for (int l = 0; l < nl; l++) out.write(reinterpret_cast<const char*>(landmarks[l].p.data()), 24);
  }
  std::printf("rccl %s, rank %d of %d\n", amd::RcclWorld::instance().enabled() ? "on" : "off", amd::RcclWorld::instance().rank(),
              amd::RcclWorld::instance().world());
  return 0;
}
