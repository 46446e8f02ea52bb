// Host-only checks of the headless pipeline's CPU pieces (no device call is made; the binary does not
// link libvslam_hip.so): camera models, P3P / RANSAC / refinement, triangulation, trajectory alignment,
// JSON / CSV / PNG / PGM readers.  Driven by tests/test_harness_cpu.py, which writes the input files.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "visnav_amd/harness/ate.h"
#include "visnav_amd/harness/camera.h"
#include "visnav_amd/harness/geometry.h"
#include "visnav_amd/harness/io.h"
#include "visnav_amd/harness/pnp.h"


using namespace visnav;
using namespace visnav::harness;

#define CHECK(c)                                                          \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

static double frand(uint64_t& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(s >> 11) / 9007199254740992.0;
}

static int test_cameras() {
  const double ds[8] = {351.04, 350.0, 365.89, 249.35, -0.2385, 0.5679, 0, 0};
  const double ph[8] = {402.5, 400.0, 505, 509, 0, 0, 0, 0};
  const double eu[8] = {402.5, 400.0, 505, 509, 0.6, 1.1, 0, 0};
  const double kb[8] = {402.5, 400.0, 505, 509, 0.01, -0.002, 0.0005, -0.0001};
  const double* par[4] = {ds, ph, eu, kb};
  uint64_t s = 7;
  for (int kind = 0; kind < 4; kind++)
    for (int it = 0; it < 200; it++) {
      const Vec3 p(2 * frand(s) - 1, 2 * frand(s) - 1, 1.0 + 3 * frand(s));
      double uv[2];
      project(kind, par[kind], p, uv);
      const Vec3 b = unproject(kind, par[kind], uv[0], uv[1]);
      CHECK(std::fabs(norm(b) - 1.0) < 1e-9);
      const Vec3 d = b - normalized(p);
      CHECK(norm(d) < (kind == kKB4 ? 1e-6 : 1e-9));  // kb4: five Newton steps, like the reference
    }
  return 0;
}

static Pose random_pose(uint64_t& s) {
  const Vec3 w(frand(s) - 0.5, frand(s) - 0.5, frand(s) - 0.5);
  return {exp_so3(2.0 * w), Vec3(frand(s) - 0.5, frand(s) - 0.5, frand(s) - 0.5)};
}

static int test_p3p_and_ransac() {
  uint64_t s = 11;
  int exact = 0;
  for (int it = 0; it < 200; it++) {
    const Pose T = random_pose(s);  // T_w_c
    Vec3 P[3], f[3];
    for (int i = 0; i < 3; i++) {
      const Vec3 pc(2 * frand(s) - 1, 2 * frand(s) - 1, 2 + 4 * frand(s));
      P[i] = T * pc;
      f[i] = normalized(pc);
    }
    Pose sol[4];
    const int n = p3p(f, P, sol);
    CHECK(n >= 1);
    double best = 1e9;
    for (int k = 0; k < n; k++) {
      double e = norm(sol[k].t - T.t);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) e += std::fabs(sol[k].R.m[a][b] - T.R.m[a][b]);
      best = std::min(best, e);
      // every returned solution must reproduce the three bearings
      for (int i = 0; i < 3; i++) CHECK(bearing_score(sol[k], f[i], P[i]) < 1e-8);
    }
    if (best < 1e-6) exact++;
  }
  CHECK(exact >= 197);  // the true pose is among the roots (near-degenerate samples may lose digits)

  // RANSAC with 30 % gross outliers and small bearing noise, then refinement
  const Pose T = random_pose(s);
  std::vector<Vec3> f, P;
  std::vector<bool> is_inlier;
  for (int i = 0; i < 300; i++) {
    const Vec3 pc(4 * frand(s) - 2, 3 * frand(s) - 1.5, 2 + 6 * frand(s));
    P.push_back(T * pc);
    const bool in = frand(s) > 0.3;
    Vec3 b = normalized(pc);
    if (in) b = normalized(b + 2e-4 * Vec3(frand(s) - 0.5, frand(s) - 0.5, frand(s) - 0.5));
    else b = normalized(Vec3(frand(s) - 0.5, frand(s) - 0.5, 1.0));
    f.push_back(b);
    is_inlier.push_back(in);
  }
  XorShift rng;
  const double thr = 1.0 - std::cos(std::atan(3.0 / 500.0));
  RansacResult rr = ransac_p3p(f, P, thr, rng);
  CHECK(rr.ok);
  const Pose Tr = refine_pose(rr.T_w_c, f, P, rr.inliers);
  CHECK(norm(Tr.t - T.t) < 2e-3);
  std::vector<int> inl;
  select_within(Tr, f, P, thr, inl);
  int true_in = 0;
  for (bool b : is_inlier) true_in += b ? 1 : 0;
  int hit = 0;
  for (int i : inl) hit += is_inlier[i] ? 1 : 0;
  CHECK(hit >= true_in - 2 && (int)inl.size() <= true_in + 6);
  // same seed, same result
  XorShift rng2;
  RansacResult rr2 = ransac_p3p(f, P, thr, rng2);
  CHECK(rr2.iterations == rr.iterations && rr2.inliers == rr.inliers);
  return 0;
}

static int test_triangulation_and_essential() {
  uint64_t s = 5;
  const Pose T_0_1 = {exp_so3(Vec3(0.01, -0.002, 0.003)), Vec3(0.11, -0.0003, 0.0002)};
  const Mat3 E = skew(normalized(T_0_1.t)) * T_0_1.R;
  for (int it = 0; it < 100; it++) {
    const Vec3 p0(2 * frand(s) - 1, 2 * frand(s) - 1, 1 + 5 * frand(s));
    const Vec3 p1 = inverse(T_0_1) * p0;
    const Vec3 f0 = normalized(p0), f1 = normalized(p1);
    CHECK(std::fabs(dot(f0, E * f1)) < 1e-12);  // matching_utils.h:80
    const Vec3 tri = triangulate_midpoint(f0, f1, T_0_1.R, T_0_1.t);
    CHECK(norm(tri - p0) < 1e-9);
  }
  return 0;
}

static int test_svd() {
  uint64_t s = 3;
  for (int it = 0; it < 100; it++) {
    Mat3 A;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) A.m[a][b] = frand(s) - 0.5;
    if (it % 10 == 0)  // rank 2
      for (int b = 0; b < 3; b++) A.m[2][b] = A.m[0][b] + A.m[1][b];
    Mat3 U, V;
    double sv[3];
    svd3(A, U, sv, V);
    Mat3 S = Mat3::zero();
    for (int k = 0; k < 3; k++) S.m[k][k] = sv[k];
    const Mat3 R = U * S * transpose(V);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) CHECK(std::fabs(R.m[a][b] - A.m[a][b]) < 1e-10);
    const Mat3 UtU = transpose(U) * U, VtV = transpose(V) * V;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        CHECK(std::fabs(UtU.m[a][b] - (a == b)) < 1e-10);
        CHECK(std::fabs(VtV.m[a][b] - (a == b)) < 1e-10);
      }
    CHECK(sv[0] >= sv[1] && sv[1] >= sv[2] && sv[2] >= 0);
  }
  return 0;
}

// files written by the Python test: argv[1] = directory
static int test_files(const std::string& dir) {
  Calibration calib;
  CHECK(load_calibration(dir + "/calib.json", calib));
  CHECK(calib.intrinsics.size() == 2 && calib.T_i_c.size() == 2);
  CHECK(calib.intrinsics[0]->name() == "ds" && calib.intrinsics[0]->width() == 752 && calib.intrinsics[0]->height() == 480);
  CHECK(std::fabs(calib.intrinsics[1]->data()[0] - 362.9532887030661) < 1e-12);
  CHECK(std::fabs(calib.T_i_c[1].data()[4] - 0.11002674958788125) < 1e-15);
  CHECK(std::fabs(calib.T_i_c[1].data()[3] - 0.9999738481299002) < 1e-15);

  EurocDataset ds;
  CHECK(load_euroc(dir + "/seq", ds));
  CHECK(ds.timestamps.size() == 3);
  CHECK(ds.timestamps[0] == 1403715273262142976ll && ds.timestamps[2] == 1403715273362142976ll);
  CHECK(ds.images.at(FrameCamId(1, 1)) == dir + "/seq/cam1/data/1403715273312143104.png");
  CHECK(ds.gt_t_ns.size() == 4 && std::fabs(ds.gt_t_w_i[2].y - 0.25) < 1e-15);

  // every PNG variant must decode to the pixels of the PGM next to it
  GreyImage ref;
  CHECK(load_image(dir + "/img.pgm", ref));
  CHECK(ref.w == 97 && ref.h == 61);
  const char* variants[] = {"img_l0.png", "img_l6.png", "img_l9.png", "img_filters.png", "img_rgb.png", "img_multi_idat.png"};
  for (const char* v : variants) {
    GreyImage g;
    if (!load_image(dir + "/" + v, g)) {
      std::fprintf(stderr, "decode failed: %s\n", v);
      return 1;
    }
    CHECK(g.w == ref.w && g.h == ref.h);
    if (std::string(v) == "img_rgb.png") {
      for (size_t i = 0; i < g.px.size(); i++) CHECK(std::abs((int)g.px[i] - (int)ref.px[i]) <= 1);
    } else {
      CHECK(g.px == ref.px);
    }
  }
  GreyImage bad;
  CHECK(!load_image(dir + "/truncated.png", bad));
  CHECK(!load_image(dir + "/missing.png", bad));

  // trajectory alignment: est = rigidly moved ground truth + known offsets
  FILE* f = std::fopen((dir + "/ate.txt").c_str(), "r");
  CHECK(f != nullptr);
  int n_est = 0, n_gt = 0;
  double expect = 0;
  CHECK(std::fscanf(f, "%d %d %lf", &n_est, &n_gt, &expect) == 3);
  std::vector<int64_t> te(n_est), tg(n_gt);
  std::vector<Vec3> pe(n_est), pg(n_gt);
  for (int i = 0; i < n_est; i++) {
    long long t;
    CHECK(std::fscanf(f, "%lld %lf %lf %lf", &t, &pe[i].x, &pe[i].y, &pe[i].z) == 4);
    te[i] = t;
  }
  for (int i = 0; i < n_gt; i++) {
    long long t;
    CHECK(std::fscanf(f, "%lld %lf %lf %lf", &t, &pg[i].x, &pg[i].y, &pg[i].z) == 4);
    tg[i] = t;
  }
  std::fclose(f);
  int n_assoc = 0;
  const double got = align_svd(te, pe, tg, pg, &n_assoc);
  std::printf("ate %.12f expect %.12f assoc %d\n", got, expect, n_assoc);
  CHECK(std::fabs(got - expect) < 1e-9);
  return 0;
}

int main(int argc, char** argv) {
  if (test_cameras()) return 1;
  if (test_p3p_and_ransac()) return 1;
  if (test_triangulation_and_essential()) return 1;
  if (test_svd()) return 1;
  if (argc > 1 && test_files(argv[1])) return 1;
  std::printf("harness host tests OK\n");
  return 0;
}
