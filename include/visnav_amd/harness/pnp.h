// harness/pnp.h -- host restatement of what the reference's per-frame pose estimation takes from OpenGV
// [upstream, not in /root/reference: thirdparty/opengv is an empty submodule]:
//   localize_camera  (include/visnav/vo_utils.h:170-226): AbsolutePoseSacProblem(KNEIP) + Ransac +
//                    optimize_nonlinear + selectWithinDistance
//   add_new_landmarks (:228-322): triangulation::triangulate for a calibrated stereo pair
// Parity with the OpenGV binaries is UNPINNED (no OpenGV, no fixtures).  The algorithms are restated from
// their published form: a minimal P3P solver (three law-of-cosine equations reduced to a quartic in the
// ratio of two depths (solved in closed form: Ferrari + Newton polish) -- Grunert's formulation; any exact P3P has the same solution set as Kneip's), a
// fourth point to pick the root, RANSAC with OpenGV's adaptive iteration count (p = 0.99, max 1000
// iterations) on the score 1 - f_meas . f_reproj, Gauss-Newton refinement on the bearing-vector
// residuals, and midpoint triangulation.  Sampling uses a fixed-seed generator so runs are reproducible
// (OpenGV seeds from the clock).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "geometry.h"

namespace visnav {
namespace harness {

// Largest real root of z^3 + A z^2 + B z + C (Cardano / trigonometric form, then Newton polish).
inline double cubic_largest_real_root(double A, double B, double C) {
  const double sh = A / 3.0;
  const double p = B - A * A / 3.0, q = 2.0 * A * A * A / 27.0 - A * B / 3.0 + C;  // t^3 + p t + q, z = t - A/3
  const double disc = q * q / 4.0 + p * p * p / 27.0;
  double t;
  if (disc > 0) {
    const double sq = std::sqrt(disc);
    t = std::cbrt(-q / 2.0 + sq) + std::cbrt(-q / 2.0 - sq);
  } else if (p < 0) {
    const double m = 2.0 * std::sqrt(-p / 3.0);
    double arg = 3.0 * q / (p * m);
    arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
    t = m * std::cos(std::acos(arg) / 3.0);  // k = 0 branch = the largest of the three real roots
  } else {
    t = 0.0;
  }
  double z = t - sh;
  for (int it = 0; it < 4; it++) {
    const double f = ((z + A) * z + B) * z + C, df = (3.0 * z + 2.0 * A) * z + B;
    if (std::fabs(df) < 1e-300) break;
    z -= f / df;
  }
  return z;
}

// Real roots of c4 x^4 + c3 x^3 + c2 x^2 + c1 x + c0: Ferrari's factorisation into two quadratics through
// the resolvent cubic, every root polished with Newton steps on the original polynomial.
inline int solve_quartic_real(const double c[5], double roots[4]) {
  const double scale = std::fabs(c[0]) + std::fabs(c[1]) + std::fabs(c[2]) + std::fabs(c[3]) + std::fabs(c[4]);
  if (!(scale > 0)) return 0;
  int n = 0;
  auto polish = [&](double x) {
    for (int it = 0; it < 3; it++) {
      const double f = (((c[4] * x + c[3]) * x + c[2]) * x + c[1]) * x + c[0];
      const double df = ((4.0 * c[4] * x + 3.0 * c[3]) * x + 2.0 * c[2]) * x + c[1];
      if (std::fabs(df) < 1e-300) break;
      x -= f / df;
    }
    return x;
  };
  auto quadratic = [&](double b1, double b0, double shift) {  // y^2 + b1 y + b0 = 0, x = y + shift
    const double d = b1 * b1 - 4.0 * b0;
    if (d < 0) {
      if (d > -1e-9 * (b1 * b1 + std::fabs(4.0 * b0) + 1e-300)) roots[n++] = polish(-0.5 * b1 + shift);  // double root
      return;
    }
    const double sq = std::sqrt(d);
    const double y1 = b1 >= 0 ? -0.5 * (b1 + sq) : -0.5 * (b1 - sq);  // the larger-magnitude root first
    roots[n++] = polish(y1 + shift);
    if (n < 4) roots[n++] = polish((y1 != 0.0 ? b0 / y1 : -b1 - y1) + shift);
  };
  if (std::fabs(c[4]) < 1e-14 * scale) {  // degenerate leading coefficient: cubic or lower -- bisection-free fallback
    if (std::fabs(c[3]) < 1e-14 * scale) {
      if (std::fabs(c[2]) < 1e-14 * scale) {
        if (std::fabs(c[1]) < 1e-14 * scale) return 0;
        roots[n++] = -c[0] / c[1];
        return n;
      }
      const double b1 = c[1] / c[2], b0 = c[0] / c[2];
      const double d = b1 * b1 - 4.0 * b0;
      if (d < 0) return 0;
      roots[n++] = 0.5 * (-b1 + std::sqrt(d));
      roots[n++] = 0.5 * (-b1 - std::sqrt(d));
      return n;
    }
    const double z = cubic_largest_real_root(c[2] / c[3], c[1] / c[3], c[0] / c[3]);
    roots[n++] = z;  // deflate to a quadratic
    const double A = c[2] / c[3], B = c[1] / c[3];
    const double q1 = A + z, q0 = B + z * q1;
    const double d = q1 * q1 - 4.0 * q0;
    if (d >= 0) {
      roots[n++] = 0.5 * (-q1 + std::sqrt(d));
      roots[n++] = 0.5 * (-q1 - std::sqrt(d));
    }
    return n;
  }
  const double a = c[3] / c[4], b = c[2] / c[4], cc = c[1] / c[4], d = c[0] / c[4];
  const double a2 = a * a;
  const double p = b - 0.375 * a2, q = cc - 0.5 * a * b + 0.125 * a2 * a;
  const double r = d - 0.25 * a * cc + a2 * b / 16.0 - 3.0 * a2 * a2 / 256.0;
  const double shift = -0.25 * a;
  const double mag = std::fabs(p) + std::sqrt(std::fabs(r)) + 1e-300;
  if (std::fabs(q) < 1e-12 * mag * std::sqrt(mag)) {  // biquadratic: y^4 + p y^2 + r
    const double dd = p * p - 4.0 * r;
    if (dd < 0) return 0;
    const double sq = std::sqrt(dd);
    const double u[2] = {0.5 * (-p + sq), 0.5 * (-p - sq)};
    for (int k = 0; k < 2; k++)
      if (u[k] >= 0 && n <= 2) {
        const double y = std::sqrt(u[k]);
        roots[n++] = polish(y + shift);
        roots[n++] = polish(-y + shift);
      }
    return n;
  }
  // resolvent cubic z^3 + 2p z^2 + (p^2 - 4r) z - q^2 = 0 has a positive root (value at 0 is -q^2 < 0)
  const double z = cubic_largest_real_root(2.0 * p, p * p - 4.0 * r, -q * q);
  if (!(z > 0)) return 0;
  const double sz = std::sqrt(z);
  quadratic(sz, 0.5 * (p + z - q / sz), shift);
  if (n <= 2) quadratic(-sz, 0.5 * (p + z + q / sz), shift);
  return n;
}

// P3P: bearing vectors f[i] (unit, camera frame) of world points P[i].  Returns up to 4 poses T_w_c.
inline int p3p(const Vec3 f[3], const Vec3 P[3], Pose out[4]) {
  const double a2 = dot(P[1] - P[2], P[1] - P[2]), b2 = dot(P[0] - P[2], P[0] - P[2]), c2 = dot(P[0] - P[1], P[0] - P[1]);
  if (a2 < 1e-18 || b2 < 1e-18 || c2 < 1e-18) return 0;
  const double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
  // s1, s2 = x s1, s3 = y s1 the depths along f1, f2, f3:
  //   b2 (x^2 + y^2 - 2 x y ca) = a2 (1 + y^2 - 2 y cb),   b2 (1 + x^2 - 2 x cg) = c2 (1 + y^2 - 2 y cb)
  // x is linear in their difference; substituting gives the quartic in y below (derived symbolically).
  double A[5];
  A[4] = a2 * a2 - 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 4 * b2 * c2 * ca * ca + 2 * b2 * c2 + c2 * c2;
  A[3] = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg - 2 * b2 * c2 * ca * ca * cb -
               b2 * c2 * ca * cg + b2 * c2 * cb + c2 * c2 * cb);
  A[2] = 2 * (2 * a2 * a2 * cb * cb + a2 * a2 - 4 * a2 * b2 * ca * cb * cg - 2 * a2 * b2 * cg * cg - 4 * a2 * c2 * cb * cb - 2 * a2 * c2 +
              2 * b2 * b2 * ca * ca + 2 * b2 * b2 * cg * cg - b2 * b2 - 2 * b2 * c2 * ca * ca - 4 * b2 * c2 * ca * cb * cg +
              2 * c2 * c2 * cb * cb + c2 * c2);
  A[1] = -4 * (a2 * a2 * cb - a2 * b2 * ca * cg - 2 * a2 * b2 * cb * cg * cg + a2 * b2 * cb - 2 * a2 * c2 * cb + b2 * b2 * ca * cg -
               b2 * c2 * ca * cg - b2 * c2 * cb + c2 * c2 * cb);
  A[0] = a2 * a2 - 4 * a2 * b2 * cg * cg + 2 * a2 * b2 - 2 * a2 * c2 + b2 * b2 - 2 * b2 * c2 + c2 * c2;
  double ys[4];
  const int nr = solve_quartic_real(A, ys);
  int n = 0;
  for (int r = 0; r < nr && n < 4; r++) {
    const double y = ys[r];
    if (!(y > 0)) continue;
    const double den = 2 * b2 * (ca * y - cg);
    if (std::fabs(den) < 1e-12 * b2) continue;
    const double x = (2 * a2 * cb * y - a2 * y * y - a2 + b2 * y * y - b2 - 2 * c2 * cb * y + c2 * y * y + c2) / den;
    if (!(x > 0)) continue;
    const double q = 1 + y * y - 2 * y * cb;
    if (!(q > 0)) continue;
    const double s1 = std::sqrt(b2 / q), s2 = x * s1, s3 = y * s1;
    const Vec3 X[3] = {s1 * f[0], s2 * f[1], s3 * f[2]};  // the three points in the camera frame
    // congruent triangles -> rotation from the two orthonormal frames they span
    const Vec3 w1 = normalized(P[1] - P[0]);
    const Vec3 w3 = normalized(cross(w1, P[2] - P[0]));
    const Vec3 w2 = cross(w3, w1);
    const Vec3 c1 = normalized(X[1] - X[0]);
    const Vec3 c3v = cross(c1, X[2] - X[0]);
    if (norm(c3v) < 1e-15) continue;
    const Vec3 c3 = normalized(c3v);
    const Vec3 c2v = cross(c3, c1);
    const Mat3 R = Mat3::from_cols(w1, w2, w3) * transpose(Mat3::from_cols(c1, c2v, c3));  // R_w_c
    out[n].R = R;
    out[n].t = P[0] - R * X[0];
    n++;
  }
  return n;
}

struct XorShift {  // fixed-seed sampler (OpenGV: rand() seeded from the clock)
  uint64_t s;
  explicit XorShift(uint64_t seed = 0x9E3779B97F4A7C15ull) : s(seed) {}
  uint32_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (uint32_t)(s >> 32);
  }
  int below(int n) { return (int)(next() % (uint32_t)n); }
};

// score of OpenGV's AbsolutePoseSacProblem::getDistancesToModel: 1 - f_meas . normalize(R^T (P - t))
inline double bearing_score(const Pose& T_w_c, const Vec3& f, const Vec3& P) {
  const Vec3 q = transpose(T_w_c.R) * (P - T_w_c.t);
  return 1.0 - dot(f, normalized(q));
}

// 1 - f . q / |q| < threshold  <=>  f . q > (1 - threshold) |q|: compared in squared form (no square root or division
// per point; RANSAC calls this for every correspondence on every iteration).  threshold < 1 (a few pixels).
inline void select_within(const Pose& T, const std::vector<Vec3>& f, const std::vector<Vec3>& P, double threshold,
                          std::vector<int>& inliers) {
  inliers.clear();
  const Mat3 Rt = transpose(T.R);
  const double c = 1.0 - threshold, c2 = c * c;
  for (size_t i = 0; i < f.size(); i++) {
    const Vec3 q = Rt * (P[i] - T.t);
    const double d = dot(f[i], q);
    if (d > 0.0 && d * d > c2 * dot(q, q)) inliers.push_back((int)i);
  }
}

struct RansacResult {
  bool ok = false;
  Pose T_w_c;
  std::vector<int> inliers;
  int iterations = 0;
};

// Ransac<AbsolutePoseSacProblem(KNEIP)>::computeModel: 4-point samples (3 for P3P + 1 to pick the root).
inline RansacResult ransac_p3p(const std::vector<Vec3>& f, const std::vector<Vec3>& P, double threshold, XorShift& rng,
                               int max_iterations = 1000, double probability = 0.99) {
  RansacResult res;
  const int n = (int)f.size();
  if (n < 4) return res;
  int best = 0;
  double k = 1.0;
  const double log_p = std::log(1.0 - probability);
  int skipped = 0;
  const int max_skip = max_iterations * 10;
  std::vector<int> inl;
  while (res.iterations < k && skipped < max_skip && res.iterations < max_iterations) {
    int id[4];
    for (int i = 0; i < 4; i++) {
      bool dup;
      do {
        id[i] = rng.below(n);
        dup = false;
        for (int j = 0; j < i; j++) dup = dup || id[j] == id[i];
      } while (dup);
    }
    const Vec3 fs[3] = {f[id[0]], f[id[1]], f[id[2]]};
    const Vec3 Ps[3] = {P[id[0]], P[id[1]], P[id[2]]};
    Pose sol[4];
    const int ns = p3p(fs, Ps, sol);
    if (ns == 0) {
      skipped++;
      continue;
    }
    int pick = 0;
    double pick_score = 1e300;
    for (int s = 0; s < ns; s++) {
      const double sc = bearing_score(sol[s], f[id[3]], P[id[3]]);
      if (sc < pick_score) {
        pick_score = sc;
        pick = s;
      }
    }
    select_within(sol[pick], f, P, threshold, inl);
    if ((int)inl.size() > best) {
      best = (int)inl.size();
      res.T_w_c = sol[pick];
      res.inliers = inl;
      res.ok = true;
      const double w = (double)best / n;
      double p_no_outliers = 1.0 - std::pow(w, 4.0);
      p_no_outliers = std::min(std::max(p_no_outliers, 1e-12), 1.0 - 1e-12);
      k = log_p / std::log(p_no_outliers);
    }
    res.iterations++;
  }
  return res;
}

// optimize_nonlinear: Gauss-Newton over T <- T * exp(delta) on r_i = normalize(R^T (P_i - t)) - f_i.
// With q = R^T (P - t), u = q / |q|:  du/dq = (I - u u^T) / |q|,  dq/d(upsilon) = -I,  dq/d(omega) = [q]x, and because
// u^T [q]x = 0 the rotation block of the Jacobian is simply [u]x.  The normal equations then need only 21 sums per point
// instead of a dense 3 x 6 Jacobian product (the refinement was 2/3 of the host time of a tracked frame):
//   H_uu = sum (I - u u^T) / |q|^2,  H_ww = sum (I - u u^T),  H_uw = -[sum u / |q|]x,
//   g_u = -sum (r - u (u . r)) / |q|,  g_w = sum r x u.
inline Pose refine_pose(const Pose& T0, const std::vector<Vec3>& f, const std::vector<Vec3>& P, const std::vector<int>& idx,
                        int iterations = 10) {
  Pose T = T0;
  for (int it = 0; it < iterations; it++) {
    double huu[6] = {0}, hww[6] = {0}, w[3] = {0}, gu[3] = {0}, gw[3] = {0};  // symmetric 3 x 3 as xx xy xz yy yz zz
    const Mat3 Rt = transpose(T.R);
    for (int i : idx) {
      const Vec3 q = Rt * (P[i] - T.t);
      const double n2 = dot(q, q);
      if (n2 < 1e-24) continue;
      const double inv = 1.0 / std::sqrt(n2);
      const Vec3 u = inv * q;
      const Vec3 r = u - f[i];
      const double a[6] = {1.0 - u.x * u.x, -u.x * u.y, -u.x * u.z, 1.0 - u.y * u.y, -u.y * u.z, 1.0 - u.z * u.z};
      const double inv2 = inv * inv;
      for (int k = 0; k < 6; k++) {
        hww[k] += a[k];
        huu[k] += a[k] * inv2;
      }
      w[0] += u.x * inv;
      w[1] += u.y * inv;
      w[2] += u.z * inv;
      const double ur = dot(u, r);
      gu[0] -= (r.x - u.x * ur) * inv;
      gu[1] -= (r.y - u.y * ur) * inv;
      gu[2] -= (r.z - u.z * ur) * inv;
      gw[0] += r.y * u.z - r.z * u.y;
      gw[1] += r.z * u.x - r.x * u.z;
      gw[2] += r.x * u.y - r.y * u.x;
    }
    double H[6][6], g[6];
    const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        H[a][b] = huu[sym[a][b]];
        H[3 + a][3 + b] = hww[sym[a][b]];
      }
    // H_uw = -[w]x, H_wu = its transpose = [w]x
    const double wx[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        H[a][3 + b] = -wx[a][b];
        H[3 + a][b] = wx[a][b];
      }
    for (int a = 0; a < 3; a++) {
      g[a] = gu[a];
      g[3 + a] = gw[a];
    }
    // solve H d = -g (Gaussian elimination with partial pivoting; tiny damping for rank safety)
    double M[6][7];
    for (int a = 0; a < 6; a++) {
      for (int b = 0; b < 6; b++) M[a][b] = H[a][b] + (a == b ? 1e-12 : 0.0);
      M[a][6] = -g[a];
    }
    bool singular = false;
    for (int c = 0; c < 6; c++) {
      int piv = c;
      for (int r2 = c + 1; r2 < 6; r2++)
        if (std::fabs(M[r2][c]) > std::fabs(M[piv][c])) piv = r2;
      if (std::fabs(M[piv][c]) < 1e-300) {
        singular = true;
        break;
      }
      if (piv != c)
        for (int b = 0; b < 7; b++) std::swap(M[c][b], M[piv][b]);
      for (int r2 = 0; r2 < 6; r2++) {
        if (r2 == c) continue;
        const double fct = M[r2][c] / M[c][c];
        for (int b = c; b < 7; b++) M[r2][b] -= fct * M[c][b];
      }
    }
    if (singular) break;
    double d[6];
    for (int a = 0; a < 6; a++) d[a] = M[a][6] / M[a][a];
    const Vec3 ups(d[0], d[1], d[2]), om(d[3], d[4], d[5]);
    // first-order translation (V(omega) ~ I for the small steps taken here)
    T.t = T.t + T.R * ups;
    T.R = T.R * exp_so3(om);
    // Gauss-Newton converges quadratically here: a step below 1e-5 (10 um / 1e-5 rad) leaves an error of the order of its
    // square (measured on 600 correspondences: steps 1e-2, 7e-6, 4e-10) -- far below the measurement noise.  (The
    // earlier 1e-8 bought a third pass over the inliers that moved the pose by 4e-10.)
    if (norm(ups) + norm(om) < 1e-5) break;
  }
  return T;
}

// Stereo triangulation: f1 in frame 1, f2 in frame 2, p_1 = R_1_2 p_2 + t_1_2; returns the point in
// frame 1 (midpoint of the shortest segment between the two rays).
inline Vec3 triangulate_midpoint(const Vec3& f1, const Vec3& f2, const Mat3& R_1_2, const Vec3& t_1_2) {
  const Vec3 d1 = f1, d2 = R_1_2 * f2;
  const double a = dot(d1, d1), b = dot(d1, d2), c = dot(d2, d2);
  const double e = dot(d1, t_1_2), g = dot(d2, t_1_2);
  const double den = a * c - b * b;
  if (std::fabs(den) < 1e-18) return 1e6 * d1;  // parallel rays: a far point along the ray
  const double l1 = (e * c - b * g) / den, l2 = (b * e - a * g) / den;
  const Vec3 p1 = l1 * d1, p2 = t_1_2 + l2 * d2;
  return 0.5 * (p1 + p2);
}

}  // namespace harness
}  // namespace visnav
