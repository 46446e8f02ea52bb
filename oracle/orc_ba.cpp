// orc_ba.cpp -- CPU restatement of the bundle-adjustment path.  TEST INFRASTRUCTURE ONLY
// (see vslam_oracle.h).
//
// In-tree reference code restated here:
//   camera_models.h project() of the four models, reprojection.h:81-105 (the functor),
//   local_parameterization_se3.hpp:43-63 (Plus = T*exp(delta); Jacobian = Dx_this_mul_exp_x_at_0),
//   map_utils.h:337-421 / loop_closure_utils.h:672-748 (problem structure, Huber loss, options).
// [upstream], parity unpinned (not in the tree; restated from the libraries' published algorithms):
//   Sophus SE3/SO3 (quaternion x,y,z,w + translation; inverse; point action; exp),
//   Ceres 2.0/2.1: AutoDiffCostFunction (dual numbers), HuberLoss + Corrector, Jacobi scaling,
//   LevenbergMarquardtStrategy, TrustRegionMinimizer, Schur elimination of the point blocks.
//
// Derivatives are taken the way the reference takes them -- forward-mode dual numbers through the
// quaternion formulas, times the 7x6 plus-Jacobian -- NOT with the closed-form 2x6 / 2x3 blocks the
// HIP kernel uses, so the parity test compares two independent derivations.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "vslam_oracle.h"

namespace {

// ---------------------------------------------------------------------------------- dual numbers
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0) { for (int i = 0; i < N; i++) v[i] = 0; }
  Jet(double x) : a(x) { for (int i = 0; i < N; i++) v[i] = 0; }  // NOLINT
  Jet(double x, int k) : a(x) { for (int i = 0; i < N; i++) v[i] = 0; v[k] = 1; }
};
template <int N> Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; i++) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; i++) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> Jet<N> operator-(const Jet<N>& x) { Jet<N> r; r.a = -x.a; for (int i = 0; i < N; i++) r.v[i] = -x.v[i]; return r; }
template <int N> Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; i++) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) {
  Jet<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv;
  for (int i = 0; i < N; i++) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv;
  return r;
}
template <int N> Jet<N>& operator+=(Jet<N>& x, const Jet<N>& y) { x = x + y; return x; }
template <int N> bool operator==(const Jet<N>& x, const Jet<N>& y) { return x.a == y.a; }
template <int N> Jet<N> sqrt(const Jet<N>& x) { Jet<N> r; r.a = std::sqrt(x.a); const double t = 0.5 / r.a; for (int i = 0; i < N; i++) r.v[i] = x.v[i] * t; return r; }
template <int N> Jet<N> atan2(const Jet<N>& y, const Jet<N>& x) {
  Jet<N> r; r.a = std::atan2(y.a, x.a); const double t = 1.0 / (x.a * x.a + y.a * y.a);
  for (int i = 0; i < N; i++) r.v[i] = t * (-y.a * x.v[i] + x.a * y.v[i]);
  return r;
}
inline double sqrt(double x) { return std::sqrt(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

// --------------------------------------------------------------------------------- camera models
// camera_models.h:75-94 (pinhole), :158-178 (eucm), :246-270 (ds), :341-374 (kb4)
template <class T, class P = double>
void project(int model, const P* param, const T p[3], T res[2]) {
  const T fx(param[0]), fy(param[1]), cx(param[2]), cy(param[3]);
  const T& x = p[0];
  const T& y = p[1];
  const T& z = p[2];
  switch (model) {
    case 1: {  // pinhole
      res[0] = fx * x / z + cx;
      res[1] = fy * y / z + cy;
      break;
    }
    case 2: {  // eucm
      const T alpha(param[4]), beta(param[5]);
      T d = sqrt(beta * (x * x + y * y) + z * z);
      res[0] = fx * x / (alpha * d + (T(1) - alpha) * z) + cx;
      res[1] = fy * y / (alpha * d + (T(1) - alpha) * z) + cy;
      break;
    }
    case 3: {  // kb4
      const T k1(param[4]), k2(param[5]), k3(param[6]), k4(param[7]);
      T r = sqrt(x * x + y * y);
      T theta = atan2(r, z);
      T d = theta + k1 * theta * theta * theta + k2 * theta * theta * theta * theta * theta +
            k3 * theta * theta * theta * theta * theta * theta * theta +
            k4 * theta * theta * theta * theta * theta * theta * theta * theta * theta;
      if (r == T(0)) {
        res[0] = cx;
        res[1] = cy;
      } else {
        res[0] = fx * d * x / r + cx;
        res[1] = fy * d * y / r + cy;
      }
      break;
    }
    default: {  // ds
      const T xi(param[4]), alpha(param[5]);
      T d1 = sqrt(x * x + y * y + z * z);
      T d2 = sqrt(x * x + y * y + (xi * d1 + z) * (xi * d1 + z));
      res[0] = fx * x / (alpha * d2 + (T(1) - alpha) * (xi * d1 + z)) + cx;
      res[1] = fy * y / (alpha * d2 + (T(1) - alpha) * (xi * d1 + z)) + cy;
      break;
    }
  }
}

// [upstream] Sophus: SO3 * point with a unit quaternion (x, y, z, w): uv = 2 (qv x p); p + w uv + qv x uv
template <class T>
void quat_rotate(const T q[4], const T p[3], T out[3]) {
  T uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
  const T c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) out[i] = p[i] + q[3] * uv[i] + c[i];
}

// reprojection.h:91-99: residuals = p_2d - cam->project(T_w_c.inverse() * p_3d_w)
// [upstream] SE3::inverse() = (R^-1, R^-1 * (-t)); R^-1 = conjugate quaternion.
template <class T>
void functor(int model, const double* intr, const double uv[2], const T pose[7], const T pw[3], T res[2]) {
  const T qi[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
  const T nt[3] = {pose[4] * T(-1.0), pose[5] * T(-1.0), pose[6] * T(-1.0)};
  T ti[3], rp[3], pc[3];
  quat_rotate(qi, nt, ti);
  quat_rotate(qi, pw, rp);
  for (int i = 0; i < 3; i++) pc[i] = rp[i] + ti[i];
  T proj[2];
  project(model, intr, pc, proj);
  res[0] = T(uv[0]) - proj[0];
  res[1] = T(uv[1]) - proj[1];
}

void quat_to_R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

void residual_jacobian(int model, const double* pose, const double* point, const double* intr,
                       const double* uv, double* r, double* Jp /*2x6*/, double* Jl /*2x3*/) {
  typedef Jet<10> J;
  J jp[7], jl[3], res[2];
  for (int i = 0; i < 7; i++) jp[i] = J(pose[i], i);
  for (int i = 0; i < 3; i++) jl[i] = J(point[i], 7 + i);
  functor<J>(model, intr, uv, jp, jl, res);
  // [upstream] Sophus SE3::Dx_this_mul_exp_x_at_0: rows (qx qy qz qw tx ty tz), cols (upsilon, omega)
  double P[7][6];
  std::memset(P, 0, sizeof(P));
  const double qx = pose[0], qy = pose[1], qz = pose[2], qw = pose[3];
  const double c0 = 0.5 * qw, c1 = 0.5 * qz, c2 = -c1, c3 = 0.5 * qy, c4 = 0.5 * qx, c5 = -c4, c6 = -c3;
  P[0][3] = c0; P[0][4] = c2; P[0][5] = c3;
  P[1][3] = c1; P[1][4] = c0; P[1][5] = c5;
  P[2][3] = c6; P[2][4] = c4; P[2][5] = c0;
  P[3][3] = c5; P[3][4] = c6; P[3][5] = c2;
  double R[9];
  quat_to_R(pose, R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) P[4 + i][j] = R[3 * i + j];
  for (int a = 0; a < 2; a++) {
    r[a] = res[a].a;
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 7; k++) s += res[a].v[k] * P[k][j];
      Jp[6 * a + j] = s;
    }
    for (int j = 0; j < 3; j++) Jl[3 * a + j] = res[a].v[7 + j];
  }
}

// The third parameter block of AutoDiffCostFunction<., 2, 7, 3, 8> (map_utils.h:377-380): d residual / d intrinsics,
// 2 x 8 row-major, by dual numbers over the eight intrinsic parameters (pose and point constant).
void residual_jacobian_intr(int model, const double* pose, const double* point, const double* intr, const double* uv,
                            double* Ji /*2x8*/) {
  (void)uv;  // the detected corner is a constant of the residual
  typedef Jet<8> J;
  J ji[8], jp[7], jl[3];
  for (int i = 0; i < 8; i++) ji[i] = J(intr[i], i);
  for (int i = 0; i < 7; i++) jp[i] = J(pose[i]);
  for (int i = 0; i < 3; i++) jl[i] = J(point[i]);
  const J qi[4] = {-jp[0], -jp[1], -jp[2], jp[3]};
  const J nt[3] = {jp[4] * J(-1.0), jp[5] * J(-1.0), jp[6] * J(-1.0)};
  J ti[3], rp[3], pc[3], proj[2];
  quat_rotate(qi, nt, ti);
  quat_rotate(qi, jl, rp);
  for (int i = 0; i < 3; i++) pc[i] = rp[i] + ti[i];
  project<J, J>(model, ji, pc, proj);
  for (int a = 0; a < 2; a++)
    for (int j = 0; j < 8; j++) Ji[8 * a + j] = -proj[a].v[j];  // residual = p_2d - projection
}

// [upstream] Sophus SO3::exp / SE3::exp and the group product, then q re-normalised.
void se3_plus(const double* T, const double* d, double* out) {
  const double ux = d[0], uy = d[1], uz = d[2], wx = d[3], wy = d[4], wz = d[5];
  const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
  double imag, real;
  if (th < 1e-10) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
  } else {
    imag = std::sin(0.5 * th) / th;
    real = std::cos(0.5 * th);
  }
  const double dq[4] = {imag * wx, imag * wy, imag * wz, real};
  // V * upsilon
  double Vu[3];
  {
    double Rd[9];
    quat_to_R(dq, Rd);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += O[3 * i + k] * O[3 * k + j];
        O2[3 * i + j] = s;
      }
    double V[9];
    if (th < 1e-10) {
      for (int i = 0; i < 9; i++) V[i] = Rd[i];
    } else {
      const double a = (1 - std::cos(th)) / th2, b = (th - std::sin(th)) / (th2 * th);
      for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
    }
    const double u[3] = {ux, uy, uz};
    for (int i = 0; i < 3; i++) Vu[i] = V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2];
  }
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  double R[9];
  quat_to_R(T, R);
  for (int i = 0; i < 3; i++) out[4 + i] = T[4 + i] + R[3 * i] * Vu[0] + R[3 * i + 1] * Vu[1] + R[3 * i + 2] * Vu[2];
  // q * dq (Hamilton)
  double nq[4];
  nq[0] = qw * dq[0] + qx * dq[3] + qy * dq[2] - qz * dq[1];
  nq[1] = qw * dq[1] - qx * dq[2] + qy * dq[3] + qz * dq[0];
  nq[2] = qw * dq[2] + qx * dq[1] - qy * dq[0] + qz * dq[3];
  nq[3] = qw * dq[3] - qx * dq[0] - qy * dq[1] - qz * dq[2];
  const double nn = std::sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
  for (int i = 0; i < 4; i++) out[i] = nq[i] / nn;
}

// ------------------------------------------------------------------------------ problem plumbing
struct Lin {
  std::vector<double> r;   // 2 per obs (robustified)
  std::vector<double> F;   // 12 per obs (2x6), robustified, possibly Jacobi-scaled
  std::vector<double> E;   // 6 per obs (2x3)
  double cost = 0;
};

struct Work {
  const orc_ba_problem* p;
  const orc_ba_options* o;
  int n_free = 0;
  std::vector<int> free_idx;            // cam -> free index or -1
  std::vector<int> lm_start, lm_obs;    // CSR landmark -> observation ids (stable order)
  int threads = 1;
};

template <class Fn>
void parallel_for(int n, int threads, Fn fn) {
  if (threads <= 1 || n < 256) {
    fn(0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  const int chunk = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    const int a = t * chunk, b = std::min(n, a + chunk);
    if (a >= b) break;
    th.emplace_back([=] { fn(a, b, t); });
  }
  for (auto& t : th) t.join();
}

void setup(Work& w, const orc_ba_problem* p, const orc_ba_options* o) {
  w.p = p;
  w.o = o;
  w.free_idx.assign(p->n_cams, -1);
  w.n_free = 0;
  for (int c = 0; c < p->n_cams; c++)
    if (!p->cam_fixed[c]) w.free_idx[c] = w.n_free++;
  w.lm_start.assign(p->n_lms + 1, 0);
  for (int i = 0; i < p->n_obs; i++) w.lm_start[p->obs_lm[i] + 1]++;
  for (int l = 0; l < p->n_lms; l++) w.lm_start[l + 1] += w.lm_start[l];
  w.lm_obs.resize(p->n_obs);
  std::vector<int> fill(w.lm_start.begin(), w.lm_start.end() - 1);
  for (int i = 0; i < p->n_obs; i++) w.lm_obs[fill[p->obs_lm[i]]++] = i;
  w.threads = std::max(1, (int)o->num_threads);
}

// [upstream] ceres::HuberLoss(a): rho(s) = s (s <= a^2) | 2 a sqrt(s) - a^2 ; rho' = 1 | a / sqrt(s)
inline void huber(double s, double a, double& rho0, double& rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho0 = 2 * a * r - b;
    rho1 = std::max(std::numeric_limits<double>::min(), a / r);
  } else {
    rho0 = s;
    rho1 = 1.0;
  }
}

double eval_cost(const Work& w, const double* poses, const double* points) {
  const orc_ba_problem* p = w.p;
  std::vector<double> partial(w.threads, 0.0);
  parallel_for(p->n_obs, w.threads, [&](int a, int b, int t) {
    double c = 0;
    for (int i = a; i < b; i++) {
      const int cam = p->obs_cam[i], lm = p->obs_lm[i], k = p->cam_intr[cam];
      double r[2];
      functor<double>(p->cam_model[k], p->intr + 8 * k, p->obs_uv + 2 * i, poses + 7 * cam, points + 3 * lm, r);
      const double s = r[0] * r[0] + r[1] * r[1];
      double rho0 = s, rho1 = 1;
      if (w.o->use_huber) huber(s, w.o->huber_parameter, rho0, rho1);
      c += 0.5 * rho0;
    }
    partial[t] = c;
  });
  double c = 0;
  for (double v : partial) c += v;
  return c;
}

void linearize(const Work& w, const double* poses, const double* points, Lin& L) {
  const orc_ba_problem* p = w.p;
  L.r.resize(2 * (size_t)p->n_obs);
  L.F.resize(12 * (size_t)p->n_obs);
  L.E.resize(6 * (size_t)p->n_obs);
  std::vector<double> partial(w.threads, 0.0);
  parallel_for(p->n_obs, w.threads, [&](int a, int b, int t) {
    double c = 0;
    for (int i = a; i < b; i++) {
      const int cam = p->obs_cam[i], lm = p->obs_lm[i], k = p->cam_intr[cam];
      double* r = &L.r[2 * (size_t)i];
      double* F = &L.F[12 * (size_t)i];
      double* E = &L.E[6 * (size_t)i];
      residual_jacobian(p->cam_model[k], poses + 7 * cam, points + 3 * lm, p->intr + 8 * k, p->obs_uv + 2 * i, r, F, E);
      const double s = r[0] * r[0] + r[1] * r[1];
      double rho0 = s, rho1 = 1;
      if (w.o->use_huber) huber(s, w.o->huber_parameter, rho0, rho1);
      c += 0.5 * rho0;
      // [upstream] ceres Corrector with rho'' <= 0: residual and Jacobian scaled by sqrt(rho')
      const double sr = std::sqrt(rho1);
      r[0] *= sr; r[1] *= sr;
      for (int j = 0; j < 12; j++) F[j] *= sr;
      for (int j = 0; j < 6; j++) E[j] *= sr;
    }
    partial[t] = c;
  });
  L.cost = 0;
  for (double v : partial) L.cost += v;
}

// dense Cholesky solve (lower), in place; returns false if not positive definite
bool chol_solve(std::vector<double>& A, std::vector<double>& b, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      const double* ai = &A[(size_t)i * n];
      const double* aj = &A[(size_t)j * n];
      for (int k = 0; k < j; k++) s -= ai[k] * aj[k];
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

bool inv3_spd(const double* P, double* Pi) {
  const double a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(std::fabs(det) > 0) || !std::isfinite(det)) return false;
  const double id = 1.0 / det;
  Pi[0] = c00 * id; Pi[1] = c01 * id; Pi[2] = c02 * id;
  Pi[3] = Pi[1];    Pi[4] = (a * f - c * c) * id; Pi[5] = (b * c - a * e) * id;
  Pi[6] = Pi[2];    Pi[7] = Pi[5]; Pi[8] = (a * d - b * b) * id;
  return true;
}

// Schur-eliminate the landmark blocks of (J^T J + diag(Dc2, Dl2)) y = J^T r.
//   S  = sum_c (F^T F + Dc2) - sum_l W P^-1 W^T,   rhs = sum F^T r - sum_l W P^-1 (E^T r)
// Dc2 / Dl2 may be null (no damping).  lm range [l0, l1).  Outputs S (n x n row-major), rhs (n),
// and optionally Pinv (9 per landmark) and bl (3 per landmark) for back-substitution.
void schur(const Work& w, const Lin& L, const double* Dc2, const double* Dl2, int l0, int l1,
           std::vector<double>& S, std::vector<double>& rhs, std::vector<double>* Pinv,
           std::vector<double>* bl) {
  const orc_ba_problem* p = w.p;
  const int n = 6 * w.n_free;
  S.assign((size_t)n * n, 0.0);
  rhs.assign(n, 0.0);
  if (Pinv) Pinv->assign(9 * (size_t)p->n_lms, 0.0);
  if (bl) bl->assign(3 * (size_t)p->n_lms, 0.0);
  if (Dc2)
    for (int i = 0; i < n; i++) S[(size_t)i * n + i] += Dc2[i];
  std::vector<double> Wbuf, Ybuf;
  std::vector<int> cams;
  for (int l = l0; l < l1; l++) {
    const int a = w.lm_start[l], b = w.lm_start[l + 1];
    if (a == b) continue;
    double P[9] = {0}, bb[3] = {0};
    if (Dl2) { P[0] = Dl2[3 * l]; P[4] = Dl2[3 * l + 1]; P[8] = Dl2[3 * l + 2]; }
    cams.clear();
    Wbuf.assign(18 * (size_t)(b - a), 0.0);
    for (int q = a; q < b; q++) {
      const int i = w.lm_obs[q];
      const double* E = &L.E[6 * (size_t)i];
      const double* F = &L.F[12 * (size_t)i];
      const double* r = &L.r[2 * (size_t)i];
      for (int x = 0; x < 3; x++) {
        for (int y = 0; y < 3; y++) P[3 * x + y] += E[x] * E[y] + E[3 + x] * E[3 + y];
        bb[x] += E[x] * r[0] + E[3 + x] * r[1];
      }
      const int fc = w.free_idx[p->obs_cam[i]];
      cams.push_back(fc);
      if (fc < 0) continue;
      double* W = &Wbuf[18 * (size_t)(q - a)];
      for (int x = 0; x < 6; x++) {
        for (int y = 0; y < 6; y++) S[(size_t)(6 * fc + x) * n + 6 * fc + y] += F[x] * F[y] + F[6 + x] * F[6 + y];
        rhs[6 * fc + x] += F[x] * r[0] + F[6 + x] * r[1];
        for (int y = 0; y < 3; y++) W[3 * x + y] = F[x] * E[y] + F[6 + x] * E[3 + y];
      }
    }
    double Pi[9];
    if (!inv3_spd(P, Pi)) continue;
    if (Pinv) std::memcpy(&(*Pinv)[9 * (size_t)l], Pi, sizeof(Pi));
    if (bl) std::memcpy(&(*bl)[3 * (size_t)l], bb, sizeof(bb));
    // Y = W P^-1 (6x3 per obs)
    Ybuf.assign(18 * (size_t)(b - a), 0.0);
    for (int q = 0; q < b - a; q++) {
      if (cams[q] < 0) continue;
      const double* W = &Wbuf[18 * (size_t)q];
      double* Y = &Ybuf[18 * (size_t)q];
      for (int x = 0; x < 6; x++)
        for (int y = 0; y < 3; y++) Y[3 * x + y] = W[3 * x] * Pi[y] + W[3 * x + 1] * Pi[3 + y] + W[3 * x + 2] * Pi[6 + y];
    }
    for (int q1 = 0; q1 < b - a; q1++) {
      const int c1 = cams[q1];
      if (c1 < 0) continue;
      const double* Y = &Ybuf[18 * (size_t)q1];
      for (int x = 0; x < 6; x++) rhs[6 * c1 + x] -= Y[3 * x] * bb[0] + Y[3 * x + 1] * bb[1] + Y[3 * x + 2] * bb[2];
      for (int q2 = 0; q2 < b - a; q2++) {
        const int c2 = cams[q2];
        if (c2 < 0) continue;
        const double* W2 = &Wbuf[18 * (size_t)q2];
        for (int x = 0; x < 6; x++)
          for (int y = 0; y < 6; y++)
            S[(size_t)(6 * c1 + x) * n + 6 * c2 + y] -= Y[3 * x] * W2[3 * y] + Y[3 * x + 1] * W2[3 * y + 1] + Y[3 * x + 2] * W2[3 * y + 2];
      }
    }
  }
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

void orc_project(int model, const double* intr8, const double* p3, double* uv2) { project<double>(model, intr8, p3, uv2); }

void orc_ba_residual(int model, const double* pose7, const double* point3, const double* intr8,
                     const double* uv2, double* r2) {
  functor<double>(model, intr8, uv2, pose7, point3, r2);
}

void orc_ba_residual_jacobian(int model, const double* pose7, const double* point3, const double* intr8,
                              const double* uv2, double* r2, double* J_pose, double* J_point) {
  residual_jacobian(model, pose7, point3, intr8, uv2, r2, J_pose, J_point);
}

void orc_se3_plus(const double* pose7, const double* delta6, double* out7) { se3_plus(pose7, delta6, out7); }

int orc_ba_linearize(const orc_ba_problem* p, const orc_ba_options* o, int lm_first, int lm_count, double* S,
                     double* g, double* cost, int* n_free) {
  Work w;
  setup(w, p, o);
  Lin L;
  linearize(w, p->poses, p->points, L);
  int l0 = 0, l1 = p->n_lms;
  if (lm_count >= 0) { l0 = lm_first; l1 = std::min(p->n_lms, lm_first + lm_count); }
  if (lm_count >= 0) {
    // cost restricted to the landmark range
    double c = 0;
    for (int l = l0; l < l1; l++)
      for (int q = w.lm_start[l]; q < w.lm_start[l + 1]; q++) {
        const int i = w.lm_obs[q];
        const int cam = p->obs_cam[i], k = p->cam_intr[cam];
        double r[2];
        functor<double>(p->cam_model[k], p->intr + 8 * k, p->obs_uv + 2 * i, p->poses + 7 * cam, p->points + 3 * p->obs_lm[i], r);
        const double s = r[0] * r[0] + r[1] * r[1];
        double rho0 = s, rho1 = 1;
        if (o->use_huber) huber(s, o->huber_parameter, rho0, rho1);
        c += 0.5 * rho0;
      }
    L.cost = c;
  }
  std::vector<double> Sv, gv;
  schur(w, L, nullptr, nullptr, l0, l1, Sv, gv, nullptr, nullptr);
  std::memcpy(S, Sv.data(), sizeof(double) * Sv.size());
  std::memcpy(g, gv.data(), sizeof(double) * gv.size());
  *cost = L.cost;
  *n_free = w.n_free;
  return 0;
}

// [upstream] ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT / SPARSE_SCHUR and defaults:
// initial_trust_region_radius 1e4, max 1e16, min 1e-32, min/max_lm_diagonal 1e-6 / 1e32,
// jacobi_scaling, min_relative_decrease 1e-3, function_tolerance 1e-6, gradient_tolerance 1e-10,
// parameter_tolerance 1e-8, max_num_consecutive_invalid_steps 5, monotonic steps.
int orc_bundle_adjust(const orc_ba_problem* p, const orc_ba_options* o, orc_ba_summary* sum) {
  const double t_start = now_ms();
  Work w;
  setup(w, p, o);
  const int nc = 6 * w.n_free, nl = 3 * p->n_lms;
  std::vector<double> x_pose(p->poses, p->poses + 7 * (size_t)p->n_cams);
  std::vector<double> x_pt(p->points, p->points + 3 * (size_t)p->n_lms);
  std::vector<double> c_pose(x_pose), c_pt(x_pt);
  orc_ba_summary s;
  std::memset(&s, 0, sizeof(s));

  auto x_norm_of = [&](const std::vector<double>& ps, const std::vector<double>& pt) {
    double q = 0;
    for (int c = 0; c < p->n_cams; c++)
      if (w.free_idx[c] >= 0)
        for (int j = 0; j < 7; j++) q += ps[7 * c + j] * ps[7 * c + j];
    for (double v : pt) q += v * v;
    return std::sqrt(q);
  };

  Lin L;
  double t0 = now_ms();
  linearize(w, x_pose.data(), x_pt.data(), L);
  s.linearize_ms += now_ms() - t0;
  double cost = L.cost;
  s.initial_cost = cost;
  double x_norm = x_norm_of(x_pose, x_pt);

  std::vector<double> scale_c(nc, 1.0), scale_l(nl, 1.0);
  std::vector<double> grad_c(nc), grad_l(nl);
  auto gradient_and_scale = [&](bool compute_scale) {
    std::fill(grad_c.begin(), grad_c.end(), 0.0);
    std::fill(grad_l.begin(), grad_l.end(), 0.0);
    std::vector<double> n2c(nc, 0.0), n2l(nl, 0.0);
    for (int i = 0; i < p->n_obs; i++) {
      const int fc = w.free_idx[p->obs_cam[i]], lm = p->obs_lm[i];
      const double* F = &L.F[12 * (size_t)i];
      const double* E = &L.E[6 * (size_t)i];
      const double* r = &L.r[2 * (size_t)i];
      if (fc >= 0)
        for (int j = 0; j < 6; j++) {
          grad_c[6 * fc + j] += F[j] * r[0] + F[6 + j] * r[1];
          n2c[6 * fc + j] += F[j] * F[j] + F[6 + j] * F[6 + j];
        }
      for (int j = 0; j < 3; j++) {
        grad_l[3 * lm + j] += E[j] * r[0] + E[3 + j] * r[1];
        n2l[3 * lm + j] += E[j] * E[j] + E[3 + j] * E[3 + j];
      }
    }
    if (compute_scale) {
      for (int i = 0; i < nc; i++) scale_c[i] = 1.0 / (1.0 + std::sqrt(n2c[i]));
      for (int i = 0; i < nl; i++) scale_l[i] = 1.0 / (1.0 + std::sqrt(n2l[i]));
    }
  };
  auto apply_scale = [&]() {
    for (int i = 0; i < p->n_obs; i++) {
      const int fc = w.free_idx[p->obs_cam[i]], lm = p->obs_lm[i];
      double* F = &L.F[12 * (size_t)i];
      double* E = &L.E[6 * (size_t)i];
      if (fc >= 0)
        for (int j = 0; j < 6; j++) { F[j] *= scale_c[6 * fc + j]; F[6 + j] *= scale_c[6 * fc + j]; }
      for (int j = 0; j < 3; j++) { E[j] *= scale_l[3 * lm + j]; E[3 + j] *= scale_l[3 * lm + j]; }
    }
  };
  auto grad_max = [&]() {
    double m = 0;
    for (double v : grad_c) m = std::max(m, std::fabs(v));
    for (double v : grad_l) m = std::max(m, std::fabs(v));
    return m;
  };

  gradient_and_scale(true);
  apply_scale();
  double gmax = grad_max();

  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  std::vector<double> diag_c(nc), diag_l(nl), Dc2(nc), Dl2(nl);
  std::vector<double> S, rhs, Pinv, bl, dc(nc), dl(nl);
  int iteration = 0, invalid = 0;
  s.termination = 0;
  if (o->verbosity >= 2)
    std::fprintf(stderr, "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e\n", 0, cost);

  while (true) {
    if (iteration >= o->max_num_iterations) { s.termination = 0; break; }
    if (gmax <= 1e-10) { s.termination = 2; break; }
    if (radius <= 1e-32) { s.termination = 4; break; }
    iteration++;
    // ---- LevenbergMarquardtStrategy::ComputeStep
    if (!reuse_diagonal) {
      std::fill(diag_c.begin(), diag_c.end(), 0.0);
      std::fill(diag_l.begin(), diag_l.end(), 0.0);
      for (int i = 0; i < p->n_obs; i++) {
        const int fc = w.free_idx[p->obs_cam[i]], lm = p->obs_lm[i];
        const double* F = &L.F[12 * (size_t)i];
        const double* E = &L.E[6 * (size_t)i];
        if (fc >= 0)
          for (int j = 0; j < 6; j++) diag_c[6 * fc + j] += F[j] * F[j] + F[6 + j] * F[6 + j];
        for (int j = 0; j < 3; j++) diag_l[3 * lm + j] += E[j] * E[j] + E[3 + j] * E[3 + j];
      }
      for (auto& v : diag_c) v = std::min(std::max(v, 1e-6), 1e32);
      for (auto& v : diag_l) v = std::min(std::max(v, 1e-6), 1e32);
    }
    for (int i = 0; i < nc; i++) Dc2[i] = diag_c[i] / radius;
    for (int i = 0; i < nl; i++) Dl2[i] = diag_l[i] / radius;
    t0 = now_ms();
    schur(w, L, Dc2.data(), Dl2.data(), 0, p->n_lms, S, rhs, &Pinv, &bl);
    s.schur_ms += now_ms() - t0;
    t0 = now_ms();
    std::vector<double> y(rhs);
    bool ok = nc == 0 ? true : chol_solve(S, y, nc);
    // step = -(J^T J + D^2)^-1 J^T r
    if (ok) {
      for (int i = 0; i < nc; i++) dc[i] = -y[i];
      // delta_l = -P^-1 (b_l + sum_c W_cl^T delta_c)
      for (int l = 0; l < p->n_lms; l++) {
        double t[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
        for (int q = w.lm_start[l]; q < w.lm_start[l + 1]; q++) {
          const int i = w.lm_obs[q];
          const int fc = w.free_idx[p->obs_cam[i]];
          if (fc < 0) continue;
          const double* F = &L.F[12 * (size_t)i];
          const double* E = &L.E[6 * (size_t)i];
          double fd[2] = {0, 0};
          for (int j = 0; j < 6; j++) { fd[0] += F[j] * dc[6 * fc + j]; fd[1] += F[6 + j] * dc[6 * fc + j]; }
          for (int j = 0; j < 3; j++) t[j] += E[j] * fd[0] + E[3 + j] * fd[1];
        }
        const double* Pi = &Pinv[9 * (size_t)l];
        for (int j = 0; j < 3; j++) dl[3 * l + j] = -(Pi[3 * j] * t[0] + Pi[3 * j + 1] * t[1] + Pi[3 * j + 2] * t[2]);
      }
      for (double v : dc) ok = ok && std::isfinite(v);
      for (double v : dl) ok = ok && std::isfinite(v);
    }
    s.solve_ms += now_ms() - t0;
    double model_cost_change = 0;
    if (ok) {
      for (int i = 0; i < p->n_obs; i++) {
        const int fc = w.free_idx[p->obs_cam[i]], lm = p->obs_lm[i];
        const double* F = &L.F[12 * (size_t)i];
        const double* E = &L.E[6 * (size_t)i];
        const double* r = &L.r[2 * (size_t)i];
        double m[2] = {0, 0};
        if (fc >= 0)
          for (int j = 0; j < 6; j++) { m[0] += F[j] * dc[6 * fc + j]; m[1] += F[6 + j] * dc[6 * fc + j]; }
        for (int j = 0; j < 3; j++) { m[0] += E[j] * dl[3 * lm + j]; m[1] += E[3 + j] * dl[3 * lm + j]; }
        model_cost_change -= m[0] * (r[0] + m[0] / 2.0) + m[1] * (r[1] + m[1] / 2.0);
      }
      ok = model_cost_change > 0.0;
    }
    if (!ok) {
      // TrustRegionMinimizer::HandleInvalidStep + LM StepIsInvalid
      if (++invalid >= 5) { s.termination = 4; break; }
      radius *= 0.5;
      reuse_diagonal = true;
      if (o->verbosity >= 2) std::fprintf(stderr, "%4d  invalid step, radius %.3e\n", iteration, radius);
      continue;
    }
    invalid = 0;
    // delta = step .* jacobi scaling ; candidate = Plus(x, delta)
    double step_norm2 = 0;
    for (int c = 0; c < p->n_cams; c++) {
      const int fc = w.free_idx[c];
      if (fc < 0) continue;
      double d6[6];
      for (int j = 0; j < 6; j++) { d6[j] = dc[6 * fc + j] * scale_c[6 * fc + j]; step_norm2 += d6[j] * d6[j]; }
      se3_plus(&x_pose[7 * c], d6, &c_pose[7 * c]);
    }
    for (int i = 0; i < nl; i++) {
      const double d = dl[i] * scale_l[i];
      step_norm2 += d * d;
      c_pt[i] = x_pt[i] + d;
    }
    t0 = now_ms();
    const double cand_cost = eval_cost(w, c_pose.data(), c_pt.data());
    s.linearize_ms += now_ms() - t0;
    const double step_norm = std::sqrt(step_norm2);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { s.termination = 3; break; }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * cost) { s.termination = 1; break; }
    const double rel = cost_change / model_cost_change;
    if (o->verbosity >= 2)
      std::fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", iteration, cand_cost, cost_change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      x_pose = c_pose;
      x_pt = c_pt;
      x_norm = x_norm_of(x_pose, x_pt);
      cost = cand_cost;
      t0 = now_ms();
      linearize(w, x_pose.data(), x_pt.data(), L);
      s.linearize_ms += now_ms() - t0;
      gradient_and_scale(false);
      apply_scale();
      gmax = grad_max();
      s.successful_steps++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  s.iterations = iteration;
  s.final_cost = cost;
  std::memcpy(p->poses, x_pose.data(), sizeof(double) * x_pose.size());
  std::memcpy(p->points, x_pt.data(), sizeof(double) * x_pt.size());
  s.total_ms = now_ms() - t_start;
  if (o->verbosity >= 1)
    std::fprintf(stderr, "orc BA: iterations %d, initial cost %.6e, final cost %.6e, termination %d\n", s.iterations, s.initial_cost, s.final_cost, s.termination);
  if (sum) *sum = s;
  return 0;
}

// map_utils.h:337-421 with options.optimize_intrinsics = true (:397-403: the two intrinsics blocks are NOT set constant):
// the same Levenberg-Marquardt / Schur loop as orc_bundle_adjust with two more 8-wide parameter blocks on the camera
// side of the reduced system.  Every residual depends on its camera pose (6 tangent columns, absent for a fixed
// camera), on the intrinsics block cam_intr[cam] (8 columns, all of them like Ceres' size-8 block: the unused trailing
// parameters of a model have zero columns and keep their values) and on its landmark (3 columns, eliminated).
// intr_io [16]: in = initial intrinsics, out = optimised.  p->intr is not used.
int orc_bundle_adjust_intrinsics(const orc_ba_problem* p, const orc_ba_options* o, double* intr_io, orc_ba_summary* sum) {
  const double t_start = now_ms();
  Work w;
  setup(w, p, o);
  const int nc = 6 * w.n_free, ni = 16, nt = nc + ni, nl = 3 * p->n_lms, O = p->n_obs;
  std::vector<double> x_pose(p->poses, p->poses + 7 * (size_t)p->n_cams), x_pt(p->points, p->points + 3 * (size_t)p->n_lms);
  std::vector<double> x_in(intr_io, intr_io + 16);
  std::vector<double> c_pose(x_pose), c_pt(x_pt), c_in(x_in);
  orc_ba_summary s;
  std::memset(&s, 0, sizeof(s));
  // per observation: residual (2), camera-side Jacobian J (2 x 14: pose 6 | intrinsics 8), landmark Jacobian E (2 x 3)
  std::vector<double> R(2 * (size_t)O), J(28 * (size_t)O), E(6 * (size_t)O);
  auto col_of = [&](int i, int j) {  // column of the reduced system of camera-side column j of observation i, or -1
    const int cam = p->obs_cam[i];
    if (j < 6) return w.free_idx[cam] >= 0 ? 6 * w.free_idx[cam] + j : -1;
    return nc + 8 * p->cam_intr[cam] + (j - 6);
  };
  auto cost_at = [&](const double* poses, const double* pts, const double* in) {
    double c = 0;
    for (int i = 0; i < O; i++) {
      const int cam = p->obs_cam[i], k = p->cam_intr[cam];
      double r[2];
      functor<double>(p->cam_model[k], in + 8 * k, p->obs_uv + 2 * i, poses + 7 * cam, pts + 3 * p->obs_lm[i], r);
      const double sq = r[0] * r[0] + r[1] * r[1];
      double rho0 = sq, rho1 = 1;
      if (o->use_huber) huber(sq, o->huber_parameter, rho0, rho1);
      c += 0.5 * rho0;
    }
    return c;
  };
  auto linearize_all = [&]() {
    double c = 0;
    for (int i = 0; i < O; i++) {
      const int cam = p->obs_cam[i], k = p->cam_intr[cam], model = p->cam_model[k];
      double r[2], F[12], El[6], G[16];
      residual_jacobian(model, &x_pose[7 * cam], &x_pt[3 * p->obs_lm[i]], &x_in[8 * k], p->obs_uv + 2 * i, r, F, El);
      residual_jacobian_intr(model, &x_pose[7 * cam], &x_pt[3 * p->obs_lm[i]], &x_in[8 * k], p->obs_uv + 2 * i, G);
      const double sq = r[0] * r[0] + r[1] * r[1];
      double rho0 = sq, rho1 = 1;
      if (o->use_huber) huber(sq, o->huber_parameter, rho0, rho1);
      c += 0.5 * rho0;
      const double sr = std::sqrt(rho1);
      for (int a = 0; a < 2; a++) {
        R[2 * (size_t)i + a] = r[a] * sr;
        for (int j = 0; j < 6; j++) J[28 * (size_t)i + 14 * a + j] = F[6 * a + j] * sr;
        for (int j = 0; j < 8; j++) J[28 * (size_t)i + 14 * a + 6 + j] = G[8 * a + j] * sr;
        for (int j = 0; j < 3; j++) E[6 * (size_t)i + 3 * a + j] = El[3 * a + j] * sr;
      }
    }
    return c;
  };
  auto x_norm_of = [&](const std::vector<double>& ps, const std::vector<double>& pt, const std::vector<double>& in) {
    double q = 0;
    for (int c = 0; c < p->n_cams; c++)
      if (w.free_idx[c] >= 0)
        for (int j = 0; j < 7; j++) q += ps[7 * c + j] * ps[7 * c + j];
    for (double v : pt) q += v * v;
    for (double v : in) q += v * v;
    return std::sqrt(q);
  };
  double cost = linearize_all();
  s.initial_cost = cost;
  double x_norm = x_norm_of(x_pose, x_pt, x_in);
  std::vector<double> scale_c(nt, 1.0), scale_l(nl, 1.0), grad_c(nt), grad_l(nl), n2c(nt), n2l(nl);
  auto column_stats = [&]() {  // gradient and squared column norms of the current (possibly scaled) Jacobian
    std::fill(grad_c.begin(), grad_c.end(), 0.0);
    std::fill(grad_l.begin(), grad_l.end(), 0.0);
    std::fill(n2c.begin(), n2c.end(), 0.0);
    std::fill(n2l.begin(), n2l.end(), 0.0);
    for (int i = 0; i < O; i++) {
      const double* Ji = &J[28 * (size_t)i];
      const double* Ei = &E[6 * (size_t)i];
      const double* r = &R[2 * (size_t)i];
      for (int j = 0; j < 14; j++) {
        const int c = col_of(i, j);
        if (c < 0) continue;
        grad_c[c] += Ji[j] * r[0] + Ji[14 + j] * r[1];
        n2c[c] += Ji[j] * Ji[j] + Ji[14 + j] * Ji[14 + j];
      }
      const int lm = p->obs_lm[i];
      for (int j = 0; j < 3; j++) {
        grad_l[3 * lm + j] += Ei[j] * r[0] + Ei[3 + j] * r[1];
        n2l[3 * lm + j] += Ei[j] * Ei[j] + Ei[3 + j] * Ei[3 + j];
      }
    }
  };
  auto apply_scale = [&]() {
    for (int i = 0; i < O; i++) {
      double* Ji = &J[28 * (size_t)i];
      double* Ei = &E[6 * (size_t)i];
      for (int j = 0; j < 14; j++) {
        const int c = col_of(i, j);
        if (c < 0) continue;
        Ji[j] *= scale_c[c];
        Ji[14 + j] *= scale_c[c];
      }
      const int lm = p->obs_lm[i];
      for (int j = 0; j < 3; j++) { Ei[j] *= scale_l[3 * lm + j]; Ei[3 + j] *= scale_l[3 * lm + j]; }
    }
  };
  auto grad_max = [&]() {
    double m = 0;
    for (double v : grad_c) m = std::max(m, std::fabs(v));
    for (double v : grad_l) m = std::max(m, std::fabs(v));
    return m;
  };
  column_stats();
  for (int i = 0; i < nt; i++) scale_c[i] = 1.0 / (1.0 + std::sqrt(n2c[i]));
  for (int i = 0; i < nl; i++) scale_l[i] = 1.0 / (1.0 + std::sqrt(n2l[i]));
  apply_scale();
  column_stats();
  double gmax = grad_max();

  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  std::vector<double> diag_c(nt), diag_l(nl), S, rhs, Pinv(9 * (size_t)p->n_lms), bl(3 * (size_t)p->n_lms), dc(nt), dl(nl);
  int iteration = 0, invalid = 0;
  s.termination = 0;
  while (true) {
    if (iteration >= o->max_num_iterations) { s.termination = 0; break; }
    if (gmax <= 1e-10) { s.termination = 2; break; }
    if (radius <= 1e-32) { s.termination = 4; break; }
    iteration++;
    if (!reuse_diagonal) {
      for (int i = 0; i < nt; i++) diag_c[i] = std::min(std::max(n2c[i], 1e-6), 1e32);
      for (int i = 0; i < nl; i++) diag_l[i] = std::min(std::max(n2l[i], 1e-6), 1e32);
    }
    // Schur complement of the landmark blocks
    S.assign((size_t)nt * nt, 0.0);
    rhs.assign(nt, 0.0);
    for (int i = 0; i < nt; i++) S[(size_t)i * nt + i] += diag_c[i] / radius;
    std::fill(Pinv.begin(), Pinv.end(), 0.0);
    std::fill(bl.begin(), bl.end(), 0.0);
    std::vector<double> Wb, Yb;
    for (int l = 0; l < p->n_lms; l++) {
      const int a = w.lm_start[l], b = w.lm_start[l + 1];
      if (a == b) continue;
      double P[9] = {0}, bb[3] = {0};
      P[0] = diag_l[3 * l] / radius; P[4] = diag_l[3 * l + 1] / radius; P[8] = diag_l[3 * l + 2] / radius;
      Wb.assign(42 * (size_t)(b - a), 0.0);
      for (int q = a; q < b; q++) {
        const int i = w.lm_obs[q];
        const double* Ji = &J[28 * (size_t)i];
        const double* Ei = &E[6 * (size_t)i];
        const double* r = &R[2 * (size_t)i];
        for (int x = 0; x < 3; x++) {
          for (int y = 0; y < 3; y++) P[3 * x + y] += Ei[x] * Ei[y] + Ei[3 + x] * Ei[3 + y];
          bb[x] += Ei[x] * r[0] + Ei[3 + x] * r[1];
        }
        double* W = &Wb[42 * (size_t)(q - a)];
        for (int x = 0; x < 14; x++) {
          const int cx = col_of(i, x);
          if (cx < 0) continue;
          for (int y = 0; y < 14; y++) {
            const int cy = col_of(i, y);
            if (cy >= 0) S[(size_t)cx * nt + cy] += Ji[x] * Ji[y] + Ji[14 + x] * Ji[14 + y];
          }
          rhs[cx] += Ji[x] * r[0] + Ji[14 + x] * r[1];
          for (int y = 0; y < 3; y++) W[3 * x + y] = Ji[x] * Ei[y] + Ji[14 + x] * Ei[3 + y];
        }
      }
      double Pi[9];
      if (!inv3_spd(P, Pi)) continue;
      std::memcpy(&Pinv[9 * (size_t)l], Pi, sizeof(Pi));
      std::memcpy(&bl[3 * (size_t)l], bb, sizeof(bb));
      Yb.assign(42 * (size_t)(b - a), 0.0);
      for (int q = 0; q < b - a; q++)
        for (int x = 0; x < 14; x++)
          for (int y = 0; y < 3; y++)
            Yb[42 * (size_t)q + 3 * x + y] = Wb[42 * (size_t)q + 3 * x] * Pi[y] + Wb[42 * (size_t)q + 3 * x + 1] * Pi[3 + y] +
                                            Wb[42 * (size_t)q + 3 * x + 2] * Pi[6 + y];
      for (int q1 = 0; q1 < b - a; q1++) {
        const int i1 = w.lm_obs[a + q1];
        const double* Y = &Yb[42 * (size_t)q1];
        for (int x = 0; x < 14; x++) {
          const int cx = col_of(i1, x);
          if (cx < 0) continue;
          rhs[cx] -= Y[3 * x] * bb[0] + Y[3 * x + 1] * bb[1] + Y[3 * x + 2] * bb[2];
          for (int q2 = 0; q2 < b - a; q2++) {
            const int i2 = w.lm_obs[a + q2];
            const double* W2 = &Wb[42 * (size_t)q2];
            for (int y = 0; y < 14; y++) {
              const int cy = col_of(i2, y);
              if (cy >= 0) S[(size_t)cx * nt + cy] -= Y[3 * x] * W2[3 * y] + Y[3 * x + 1] * W2[3 * y + 1] + Y[3 * x + 2] * W2[3 * y + 2];
            }
          }
        }
      }
    }
    std::vector<double> y(rhs);
    bool ok = chol_solve(S, y, nt);
    if (ok) {
      for (int i = 0; i < nt; i++) dc[i] = -y[i];
      for (int l = 0; l < p->n_lms; l++) {
        double t[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
        for (int q = w.lm_start[l]; q < w.lm_start[l + 1]; q++) {
          const int i = w.lm_obs[q];
          const double* Ji = &J[28 * (size_t)i];
          const double* Ei = &E[6 * (size_t)i];
          double fd[2] = {0, 0};
          for (int j = 0; j < 14; j++) {
            const int c = col_of(i, j);
            if (c >= 0) { fd[0] += Ji[j] * dc[c]; fd[1] += Ji[14 + j] * dc[c]; }
          }
          for (int j = 0; j < 3; j++) t[j] += Ei[j] * fd[0] + Ei[3 + j] * fd[1];
        }
        const double* Pi = &Pinv[9 * (size_t)l];
        for (int j = 0; j < 3; j++) dl[3 * l + j] = -(Pi[3 * j] * t[0] + Pi[3 * j + 1] * t[1] + Pi[3 * j + 2] * t[2]);
      }
      for (double v : dc) ok = ok && std::isfinite(v);
      for (double v : dl) ok = ok && std::isfinite(v);
    }
    double model_cost_change = 0;
    if (ok) {
      for (int i = 0; i < O; i++) {
        const double* Ji = &J[28 * (size_t)i];
        const double* Ei = &E[6 * (size_t)i];
        const double* r = &R[2 * (size_t)i];
        const int lm = p->obs_lm[i];
        double m[2] = {0, 0};
        for (int j = 0; j < 14; j++) {
          const int c = col_of(i, j);
          if (c >= 0) { m[0] += Ji[j] * dc[c]; m[1] += Ji[14 + j] * dc[c]; }
        }
        for (int j = 0; j < 3; j++) { m[0] += Ei[j] * dl[3 * lm + j]; m[1] += Ei[3 + j] * dl[3 * lm + j]; }
        model_cost_change -= m[0] * (r[0] + m[0] / 2.0) + m[1] * (r[1] + m[1] / 2.0);
      }
      ok = model_cost_change > 0.0;
    }
    if (!ok) {
      if (++invalid >= 5) { s.termination = 4; break; }
      radius *= 0.5;
      reuse_diagonal = true;
      continue;
    }
    invalid = 0;
    double step_norm2 = 0;
    for (int c = 0; c < p->n_cams; c++) {
      const int fc = w.free_idx[c];
      if (fc < 0) continue;
      double d6[6];
      for (int j = 0; j < 6; j++) { d6[j] = dc[6 * fc + j] * scale_c[6 * fc + j]; step_norm2 += d6[j] * d6[j]; }
      se3_plus(&x_pose[7 * c], d6, &c_pose[7 * c]);
    }
    for (int j = 0; j < ni; j++) {
      const double d = dc[nc + j] * scale_c[nc + j];
      step_norm2 += d * d;
      c_in[j] = x_in[j] + d;
    }
    for (int i = 0; i < nl; i++) {
      const double d = dl[i] * scale_l[i];
      step_norm2 += d * d;
      c_pt[i] = x_pt[i] + d;
    }
    const double cand_cost = cost_at(c_pose.data(), c_pt.data(), c_in.data());
    const double step_norm = std::sqrt(step_norm2);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { s.termination = 3; break; }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= 1e-6 * cost) { s.termination = 1; break; }
    const double rel = cost_change / model_cost_change;
    if (o->verbosity >= 2)
      std::fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", iteration, cand_cost, cost_change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      x_pose = c_pose;
      x_pt = c_pt;
      x_in = c_in;
      x_norm = x_norm_of(x_pose, x_pt, x_in);
      cost = cand_cost;
      linearize_all();
      apply_scale();
      column_stats();
      gmax = grad_max();
      s.successful_steps++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
  s.iterations = iteration;
  s.final_cost = cost;
  std::memcpy(p->poses, x_pose.data(), sizeof(double) * x_pose.size());
  std::memcpy(p->points, x_pt.data(), sizeof(double) * x_pt.size());
  std::memcpy(intr_io, x_in.data(), sizeof(double) * 16);
  s.total_ms = now_ms() - t_start;
  if (o->verbosity >= 1)
    std::fprintf(stderr, "orc BA (intrinsics): iterations %d, initial cost %.6e, final cost %.6e, termination %d\n", s.iterations, s.initial_cost, s.final_cost, s.termination);
  if (sum) *sum = s;
  return 0;
}

// d residual / d intrinsics (2 x 8 row-major) for the parity tests of the device Jacobian
void orc_ba_residual_jacobian_intr(int model, const double* pose7, const double* point3, const double* intr8,
                                   const double* uv2, double* J_intr) {
  residual_jacobian_intr(model, pose7, point3, intr8, uv2, J_intr);
}

}  // extern "C"
