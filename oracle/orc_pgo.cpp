// orc_pgo.cpp -- CPU restatement of pose_graph_optimization's numerical core
// (include/visnav/loop_closure_utils.h:446-587): residual blocks
//     r = log(T_w_c^-1 * T_w_n) - upsilon_omega          (reprojection.h:107-126)
// on SE3 parameter blocks with the tangent-space parameterisation T * exp(delta)
// (local_parameterization_se3.hpp:43-63), HuberLoss(1.0), Ceres LM (max 20 iterations).  TEST INFRASTRUCTURE ONLY.
//
// [upstream] ceres::Solve / Sophus::SE3::log are not in the tree: parity with those binaries is unpinned.  The
// Jacobians come from forward-mode dual numbers (what ceres::AutoDiffCostFunction does), the LM policy is the
// one restated in orc_ba.cpp (radius 1e4, Jacobi scaling, diagonal clamp, rho > 1e-3, Ceres' radius update
// and tolerances); the reduced system is dense here (SPARSE_SCHUR in the reference: same solution).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vslam_oracle.h"

namespace {

template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0) { std::memset(d, 0, sizeof(d)); }
  Dual(double x) : v(x) { std::memset(d, 0, sizeof(d)); }  // NOLINT
};
template <int N>
Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  r.v = a.v + b.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int N>
Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  r.v = a.v - b.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int N>
Dual<N> operator-(const Dual<N>& a) {
  Dual<N> r;
  r.v = -a.v;
  for (int i = 0; i < N; i++) r.d[i] = -a.d[i];
  return r;
}
template <int N>
Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  r.v = a.v * b.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int N>
Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
  for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <int N>
Dual<N> dsqrt(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::sqrt(a.v);
  const double k = 0.5 / r.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * k;
  return r;
}
template <int N>
Dual<N> datan2(const Dual<N>& y, const Dual<N>& x) {
  Dual<N> r;
  r.v = std::atan2(y.v, x.v);
  const double den = 1.0 / (x.v * x.v + y.v * y.v);
  for (int i = 0; i < N; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den;
  return r;
}
template <int N>
Dual<N> dsin(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::sin(a.v);
  const double c = std::cos(a.v);
  for (int i = 0; i < N; i++) r.d[i] = c * a.d[i];
  return r;
}
template <int N>
Dual<N> dcos(const Dual<N>& a) {
  Dual<N> r;
  r.v = std::cos(a.v);
  const double s = -std::sin(a.v);
  for (int i = 0; i < N; i++) r.d[i] = s * a.d[i];
  return r;
}
inline double dsqrt(double a) { return std::sqrt(a); }
inline double datan2(double y, double x) { return std::atan2(y, x); }
inline double dsin(double a) { return std::sin(a); }
inline double dcos(double a) { return std::cos(a); }
inline double val(double a) { return a; }
template <int N>
double val(const Dual<N>& a) {
  return a.v;
}

// quaternion (x y z w) helpers
template <class T>
void qmul(const T* a, const T* b, T* o) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
template <class T>
void qrot(const T* q, const T* p, T* o) {  // o = R(q) p
  T uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
  const T c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) o[i] = p[i] + q[3] * uv[i] + c[i];
}

// Sophus::SE3::log of (q, t): out = (upsilon, omega)
template <class T>
void se3_log(const T* q, const T* t, T* out) {
  const T sq_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const T w = q[3];
  // Sophus so3.hpp logAndTheta: omega = (2 atan(n / w) / n) q.vec, with the series for a vanishing vector part
  T two_atan_nbyw_by_n, c;
  if (val(sq_n) < 1e-20) {
    two_atan_nbyw_by_n = T(2.0) / w - T(2.0 / 3.0) * sq_n / (w * w * w);
    c = T(1.0 / 12.0);
  } else {
    const T n = dsqrt(sq_n);
    const T half = val(w) < 0 ? datan2(-n, -w) : datan2(n, w);  // atan(n / w), continuous through w = 0
    two_atan_nbyw_by_n = T(2.0) * half / n;
    const T theta = two_atan_nbyw_by_n * n;
    // V^-1 = I - 1/2 Om + c Om^2,  c = (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2   (se3.hpp log)
    if (std::fabs(val(theta)) < 1e-6) {
      c = T(1.0 / 12.0);
    } else {
      const T ht = T(0.5) * theta;
      c = (T(1.0) - theta * dcos(ht) / (T(2.0) * dsin(ht))) / (theta * theta);
    }
  }
  T om[3] = {two_atan_nbyw_by_n * q[0], two_atan_nbyw_by_n * q[1], two_atan_nbyw_by_n * q[2]};
  // Om t = om x t; Om^2 t = om x (om x t)
  const T a[3] = {om[1] * t[2] - om[2] * t[1], om[2] * t[0] - om[0] * t[2], om[0] * t[1] - om[1] * t[0]};
  const T b[3] = {om[1] * a[2] - om[2] * a[1], om[2] * a[0] - om[0] * a[2], om[0] * a[1] - om[1] * a[0]};
  for (int i = 0; i < 3; i++) {
    out[i] = t[i] - T(0.5) * a[i] + c * b[i];
    out[3 + i] = om[i];
  }
}

// T * exp(delta), delta = (upsilon, omega): local_parameterization_se3.hpp:43-50
void se3_plus(const double* p7, const double* d6, double* o7) {
  const double* ups = d6;
  const double* om = d6 + 3;
  const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2], th = std::sqrt(th2);
  double imag, real;
  if (th < 1e-10) {
    imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0;
    real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0;
  } else {
    imag = std::sin(0.5 * th) / th;
    real = std::cos(0.5 * th);
  }
  const double dq[4] = {imag * om[0], imag * om[1], imag * om[2], real};
  double A, B;  // V = I + A Om + B Om^2
  if (th < 1e-10) {
    A = 0.5;
    B = 1.0 / 6.0;
  } else {
    A = (1.0 - std::cos(th)) / th2;
    B = (th - std::sin(th)) / (th2 * th);
  }
  const double a[3] = {om[1] * ups[2] - om[2] * ups[1], om[2] * ups[0] - om[0] * ups[2], om[0] * ups[1] - om[1] * ups[0]};
  const double b[3] = {om[1] * a[2] - om[2] * a[1], om[2] * a[0] - om[0] * a[2], om[0] * a[1] - om[1] * a[0]};
  double dt[3], rt[3];
  for (int i = 0; i < 3; i++) dt[i] = ups[i] + A * a[i] + B * b[i];
  qrot(p7, dt, rt);
  double q[4];
  qmul(p7, dq, q);
  const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) o7[i] = q[i] / nq;
  for (int i = 0; i < 3; i++) o7[4 + i] = p7[4 + i] + rt[i];
}

// residual of one edge; with T = Dual<12> the first six partials belong to delta_c, the last six to delta_n
template <class T>
void edge_residual(const T* qc, const T* tc, const T* qn, const T* tn, const double* meas, T* r) {
  const T qci[4] = {-qc[0], -qc[1], -qc[2], qc[3]};
  T q[4], dt[3], t[3];
  qmul(qci, qn, q);
  for (int i = 0; i < 3; i++) dt[i] = tn[i] - tc[i];
  qrot(qci, dt, t);
  T lg[6];
  se3_log(q, t, lg);
  for (int i = 0; i < 6; i++) r[i] = lg[i] - T(meas[i]);
}

void seed_pose(const double* p7, int first, Dual<12>* q, Dual<12>* t) {
  // T exp(delta) to first order: q' = q (x) (omega / 2, 1), t' = t + R upsilon
  for (int i = 0; i < 4; i++) q[i] = Dual<12>(p7[i]);
  for (int i = 0; i < 3; i++) t[i] = Dual<12>(p7[4 + i]);
  for (int k = 0; k < 3; k++) {
    double e[3] = {0, 0, 0}, re[3];
    e[k] = 1.0;
    qrot(p7, e, re);
    for (int i = 0; i < 3; i++) t[i].d[first + k] = re[i];  // d t / d upsilon_k = R e_k
    const double w[4] = {0.5 * e[0], 0.5 * e[1], 0.5 * e[2], 0.0};
    double qq[4];
    qmul(p7, w, qq);
    for (int i = 0; i < 4; i++) q[i].d[first + 3 + k] = qq[i];  // d q / d omega_k = q (x) (e_k / 2, 0)
  }
}

struct Lin {
  std::vector<double> r, Ja, Jb;  // per edge: 6, 36, 36 (robustified)
  double cost = 0;
};

void linearize(const orc_pgo_problem* p, const double* poses, bool use_huber, double huber, Lin& L, bool jac) {
  const int E = p->n_edges;
  L.r.assign(6 * (size_t)E, 0);
  if (jac) {
    L.Ja.assign(36 * (size_t)E, 0);
    L.Jb.assign(36 * (size_t)E, 0);
  }
  L.cost = 0;
  for (int e = 0; e < E; e++) {
    const double* pa = poses + 7 * (size_t)p->edge_a[e];
    const double* pb = poses + 7 * (size_t)p->edge_b[e];
    double r[6], Ja[36], Jb[36];
    if (jac) {
      Dual<12> qc[4], tc[3], qn[4], tn[3], rd[6];
      seed_pose(pa, 0, qc, tc);
      seed_pose(pb, 6, qn, tn);
      edge_residual(qc, tc, qn, tn, p->edge_meas + 6 * (size_t)e, rd);
      for (int i = 0; i < 6; i++) {
        r[i] = rd[i].v;
        for (int k = 0; k < 6; k++) {
          Ja[6 * i + k] = rd[i].d[k];
          Jb[6 * i + k] = rd[i].d[6 + k];
        }
      }
    } else {
      edge_residual(pa, pa + 4, pb, pb + 4, p->edge_meas + 6 * (size_t)e, r);
    }
    double s = 0;
    for (int i = 0; i < 6; i++) s += r[i] * r[i];
    double rho0 = s, rho1 = 1.0;
    if (use_huber && s > huber * huber) {  // ceres::HuberLoss: rho = 2 a sqrt(s) - a^2, rho' = a / sqrt(s), rho'' < 0
      const double rt = std::sqrt(s);
      rho0 = 2.0 * huber * rt - huber * huber;
      rho1 = huber / rt;
    }
    L.cost += 0.5 * rho0;
    const double k = std::sqrt(rho1);  // Corrector with rho'' <= 0: residual and Jacobian scaled by sqrt(rho')
    for (int i = 0; i < 6; i++) L.r[6 * (size_t)e + i] = k * r[i];
    if (jac)
      for (int i = 0; i < 36; i++) {
        L.Ja[36 * (size_t)e + i] = k * Ja[i];
        L.Jb[36 * (size_t)e + i] = k * Jb[i];
      }
  }
}

// dense H = J^T J, g = J^T r over the free nodes (6 columns each)
void normal_equations(const orc_pgo_problem* p, const std::vector<int>& free_idx, int n, const Lin& L, const double* scale,
                      std::vector<double>& H, std::vector<double>& g) {
  H.assign((size_t)n * n, 0);
  g.assign(n, 0);
  for (int e = 0; e < p->n_edges; e++) {
    const int ia = free_idx[p->edge_a[e]], ib = free_idx[p->edge_b[e]];
    const double* J[2] = {&L.Ja[36 * (size_t)e], &L.Jb[36 * (size_t)e]};
    const int idx[2] = {ia, ib};
    const double* r = &L.r[6 * (size_t)e];
    for (int x = 0; x < 2; x++) {
      if (idx[x] < 0) continue;
      for (int c = 0; c < 6; c++) {
        const double sc = scale ? scale[6 * idx[x] + c] : 1.0;
        double gv = 0;
        for (int i = 0; i < 6; i++) gv += J[x][6 * i + c] * r[i];
        g[6 * idx[x] + c] += sc * gv;
        for (int y = 0; y < 2; y++) {
          if (idx[y] < 0) continue;
          for (int c2 = 0; c2 < 6; c2++) {
            const double sc2 = scale ? scale[6 * idx[y] + c2] : 1.0;
            double hv = 0;
            for (int i = 0; i < 6; i++) hv += J[x][6 * i + c] * J[y][6 * i + c2];
            H[(size_t)(6 * idx[x] + c) * n + 6 * idx[y] + c2] += sc * sc2 * hv;
          }
        }
      }
    }
  }
}

bool chol_solve(std::vector<double>& A, std::vector<double>& b, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

}  // namespace

extern "C" {

void orc_se3_log(const double* pose7, double* out6) { se3_log(pose7, pose7 + 4, out6); }

void orc_pgo_residual_jacobian(const double* pose_c7, const double* pose_n7, const double* meas6, double* r6, double* J_c,
                               double* J_n) {
  Dual<12> qc[4], tc[3], qn[4], tn[3], rd[6];
  seed_pose(pose_c7, 0, qc, tc);
  seed_pose(pose_n7, 6, qn, tn);
  edge_residual(qc, tc, qn, tn, meas6, rd);
  for (int i = 0; i < 6; i++) {
    r6[i] = rd[i].v;
    for (int k = 0; k < 6; k++) {
      J_c[6 * i + k] = rd[i].d[k];
      J_n[6 * i + k] = rd[i].d[6 + k];
    }
  }
}

// H (n x n, n = 6 x free nodes, node order), g, cost of the robustified problem at the given poses
int orc_pgo_linearize(const orc_pgo_problem* p, const orc_ba_options* opt, double* H, double* g, double* cost, int* n_free) {
  std::vector<int> free_idx(p->n_nodes, -1);
  int nf = 0;
  for (int i = 0; i < p->n_nodes; i++)
    if (!p->node_fixed[i]) free_idx[i] = nf++;
  Lin L;
  linearize(p, p->poses, opt->use_huber != 0, opt->huber_parameter, L, true);
  std::vector<double> Hv, gv;
  normal_equations(p, free_idx, 6 * nf, L, nullptr, Hv, gv);
  if (H) std::memcpy(H, Hv.data(), sizeof(double) * Hv.size());
  if (g) std::memcpy(g, gv.data(), sizeof(double) * gv.size());
  if (cost) *cost = L.cost;
  if (n_free) *n_free = nf;
  return 0;
}

int orc_pose_graph_optimize(const orc_pgo_problem* p, const orc_ba_options* opt, orc_ba_summary* sum) {
  const int N = p->n_nodes;
  std::vector<int> free_idx(N, -1);
  int nf = 0;
  for (int i = 0; i < N; i++)
    if (!p->node_fixed[i]) free_idx[i] = nf++;
  const int n = 6 * nf;
  std::vector<double> poses(p->poses, p->poses + 7 * (size_t)N), cand(7 * (size_t)N);
  const bool huber = opt->use_huber != 0;
  Lin L;
  linearize(p, poses.data(), huber, opt->huber_parameter, L, true);
  std::vector<double> H, g, scale(n, 1.0);
  normal_equations(p, free_idx, n, L, nullptr, H, g);
  for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + std::sqrt(H[(size_t)i * n + i]));
  normal_equations(p, free_idx, n, L, scale.data(), H, g);
  double cost = L.cost;
  orc_ba_summary s;
  std::memset(&s, 0, sizeof(s));
  s.initial_cost = cost;
  auto gmax_of = [&]() {
    double m = 0;
    for (int i = 0; i < n; i++) m = std::max(m, std::fabs(g[i] / scale[i]));
    return m;
  };
  double gmax = gmax_of();
  double radius = 1e4, decrease = 2.0;
  int it = 0, invalid = 0;
  std::vector<double> diag(n);
  auto refresh_diag = [&]() {
    for (int i = 0; i < n; i++) diag[i] = std::min(std::max(H[(size_t)i * n + i], 1e-6), 1e32);
  };
  refresh_diag();
  while (true) {
    if (it >= opt->max_num_iterations) { s.termination = 0; break; }
    if (gmax <= 1e-10) { s.termination = 2; break; }
    if (radius <= 1e-32) { s.termination = 4; break; }
    it++;
    std::vector<double> A = H, d(n);
    for (int i = 0; i < n; i++) {
      A[(size_t)i * n + i] += diag[i] / radius;
      d[i] = -g[i];
    }
    bool ok = n == 0 || chol_solve(A, d, n);
    double model = 0, step2 = 0, x2 = 0;
    if (ok) {
      for (int i = 0; i < n; i++) {
        double hd = 0;
        for (int j = 0; j < n; j++) hd += H[(size_t)i * n + j] * d[j];
        model -= d[i] * (g[i] + 0.5 * hd);
        ok = ok && std::isfinite(d[i]);
      }
      for (int i = 0; i < N; i++) {
        if (free_idx[i] < 0) {
          std::memcpy(&cand[7 * (size_t)i], &poses[7 * (size_t)i], 56);
          continue;
        }
        double dl[6];
        for (int c = 0; c < 6; c++) {
          dl[c] = scale[6 * free_idx[i] + c] * d[6 * free_idx[i] + c];
          step2 += dl[c] * dl[c];
        }
        for (int c = 0; c < 7; c++) x2 += poses[7 * (size_t)i + c] * poses[7 * (size_t)i + c];
        se3_plus(&poses[7 * (size_t)i], dl, &cand[7 * (size_t)i]);
      }
      ok = ok && model > 0.0;
    }
    if (!ok) {
      if (++invalid >= 5) { s.termination = 4; break; }
      radius *= 0.5;
      continue;
    }
    invalid = 0;
    if (std::sqrt(step2) <= 1e-8 * (std::sqrt(x2) + 1e-8)) { s.termination = 3; break; }
    Lin Lc;
    linearize(p, cand.data(), huber, opt->huber_parameter, Lc, false);
    const double change = cost - Lc.cost;
    if (std::fabs(change) <= 1e-6 * cost) { s.termination = 1; break; }
    const double rel = change / model;
    if (rel > 1e-3) {
      poses.swap(cand);
      linearize(p, poses.data(), huber, opt->huber_parameter, L, true);
      normal_equations(p, free_idx, n, L, scale.data(), H, g);
      cost = L.cost;
      gmax = gmax_of();
      refresh_diag();
      s.successful_steps++;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
      decrease = 2.0;
    } else {
      radius /= decrease;
      decrease *= 2.0;
    }
  }
  s.iterations = it;
  s.final_cost = cost;
  std::memcpy(p->poses, poses.data(), sizeof(double) * 7 * (size_t)N);
  if (sum) *sum = s;
  return 0;
}

}  // extern "C"
