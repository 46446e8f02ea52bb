// pgo.hip -- pose graph optimisation: the numerical core of
// visnav::pose_graph_optimization (include/visnav/loop_closure_utils.h:446-587).  Residual blocks
//     r = log(T_w_c^-1 * T_w_n) - upsilon_omega                  (PoseGraphRelativePoseCostFunctor, reprojection.h:107-126)
// on SE3 blocks with the tangent parameterisation T * exp(delta) (local_parameterization_se3.hpp:43-63),
// HuberLoss, Levenberg-Marquardt with the [upstream] Ceres policy restated for bundle adjustment in ba.hip.
//
// One thread per edge evaluates the residual and both 6x6 Jacobian blocks with forward-mode dual numbers
// (12 partials: what ceres::AutoDiffCostFunction<., 6, 7, 7> followed by the SE3 plus-Jacobian computes), the
// normal equations of the (at most a few thousand) keyframes are accumulated densely -- one thread owns one ROW of H and
// walks the node's incident edges in edge order (lists built once per solve on the host): no atomics, results do not
// depend on timing -- and solved by the blocked Cholesky of chol.hip.  This runs once per loop closure.
#include <algorithm>
#include <cmath>
#include <vector>

#include "vsl_common.h"

namespace {

struct D12 {
  double v;
  double d[12];
};
__device__ __forceinline__ D12 mk(double x) {
  D12 r;
  r.v = x;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = 0.0;
  return r;
}
__device__ __forceinline__ D12 operator+(const D12& a, const D12& b) {
  D12 r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
__device__ __forceinline__ D12 operator-(const D12& a, const D12& b) {
  D12 r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
__device__ __forceinline__ D12 operator-(const D12& a) {
  D12 r;
  r.v = -a.v;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = -a.d[i];
  return r;
}
__device__ __forceinline__ D12 operator*(const D12& a, const D12& b) {
  D12 r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
__device__ __forceinline__ D12 operator/(const D12& a, const D12& b) {
  D12 r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
__device__ __forceinline__ D12 dsqrt(const D12& a) {
  D12 r;
  r.v = sqrt(a.v);
  const double k = 0.5 / r.v;
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = a.d[i] * k;
  return r;
}
__device__ __forceinline__ D12 datan2(const D12& y, const D12& x) {
  D12 r;
  r.v = atan2(y.v, x.v);
  const double den = 1.0 / (x.v * x.v + y.v * y.v);
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den;
  return r;
}
__device__ __forceinline__ D12 dsin(const D12& a) {
  D12 r;
  r.v = sin(a.v);
  const double c = cos(a.v);
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = c * a.d[i];
  return r;
}
__device__ __forceinline__ D12 dcos(const D12& a) {
  D12 r;
  r.v = cos(a.v);
  const double s = -sin(a.v);
#pragma unroll
  for (int i = 0; i < 12; i++) r.d[i] = s * a.d[i];
  return r;
}
__device__ __forceinline__ double dsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ double datan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ double dsin(double a) { return sin(a); }
__device__ __forceinline__ double dcos(double a) { return cos(a); }
__device__ __forceinline__ double val(double a) { return a; }
__device__ __forceinline__ double val(const D12& a) { return a.v; }
__device__ __forceinline__ double lift(double, double x) { return x; }
__device__ __forceinline__ D12 lift(const D12&, double x) { return mk(x); }

template <class T>
__device__ void qmul(const T* a, const T* b, T* o) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
template <class T>
__device__ void qrot(const T* q, const T* p, T* o) {
  T uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
  const T c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) o[i] = p[i] + q[3] * uv[i] + c[i];
}

// Sophus::SE3::log of (q, t): out = (upsilon, omega)   (so3.hpp logAndTheta, se3.hpp log)
template <class T>
__device__ void se3_log(const T* q, const T* t, T* out) {
  const T sq_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const T w = q[3];
  T two_atan, c;
  if (val(sq_n) < 1e-20) {
    two_atan = lift(w, 2.0) / w - lift(w, 2.0 / 3.0) * sq_n / (w * w * w);
    c = lift(w, 1.0 / 12.0);
  } else {
    const T n = dsqrt(sq_n);
    const T half = val(w) < 0 ? datan2(-n, -w) : datan2(n, w);
    two_atan = lift(w, 2.0) * half / n;
    const T theta = two_atan * n;
    if (fabs(val(theta)) < 1e-6) {
      c = lift(w, 1.0 / 12.0);
    } else {
      const T ht = lift(w, 0.5) * theta;
      c = (lift(w, 1.0) - theta * dcos(ht) / (lift(w, 2.0) * dsin(ht))) / (theta * theta);
    }
  }
  const T om[3] = {two_atan * q[0], two_atan * q[1], two_atan * q[2]};
  const T a[3] = {om[1] * t[2] - om[2] * t[1], om[2] * t[0] - om[0] * t[2], om[0] * t[1] - om[1] * t[0]};
  const T b[3] = {om[1] * a[2] - om[2] * a[1], om[2] * a[0] - om[0] * a[2], om[0] * a[1] - om[1] * a[0]};
  for (int i = 0; i < 3; i++) {
    out[i] = t[i] - lift(w, 0.5) * a[i] + c * b[i];
    out[3 + i] = om[i];
  }
}

template <class T>
__device__ void edge_residual(const T* qc, const T* tc, const T* qn, const T* tn, const double* meas, T* r) {
  const T qci[4] = {-qc[0], -qc[1], -qc[2], qc[3]};
  T q[4], dt[3], t[3];
  qmul(qci, qn, q);
  for (int i = 0; i < 3; i++) dt[i] = tn[i] - tc[i];
  qrot(qci, dt, t);
  T lg[6];
  se3_log(q, t, lg);
  for (int i = 0; i < 6; i++) r[i] = lg[i] - lift(q[3], meas[i]);
}

// T exp(delta) to first order in dual arithmetic: q' = q (x) (omega / 2, 1), t' = t + R upsilon
__device__ void seed_pose(const double* p7, int first, D12* q, D12* t) {
  for (int i = 0; i < 4; i++) q[i] = mk(p7[i]);
  for (int i = 0; i < 3; i++) t[i] = mk(p7[4 + i]);
  for (int k = 0; k < 3; k++) {
    double e[3] = {0, 0, 0}, re[3];
    e[k] = 1.0;
    qrot(p7, e, re);
    for (int i = 0; i < 3; i++) t[i].d[first + k] = re[i];
    const double w[4] = {0.5 * e[0], 0.5 * e[1], 0.5 * e[2], 0.0};
    double qq[4];
    qmul(p7, w, qq);
    for (int i = 0; i < 4; i++) q[i].d[first + 3 + k] = qq[i];
  }
}

// r (6), Ja / Jb (6x6 row-major) per edge, robustified (ceres Corrector with rho'' <= 0: sqrt(rho') on both),
// cost[e] = rho(|r|^2) / 2
template <bool JAC>
__global__ __launch_bounds__(64) void pgo_linearize_kernel(int n_edges, const double* __restrict__ poses, const int* __restrict__ edge_a,
                                                           const int* __restrict__ edge_b, const double* __restrict__ meas,
                                                           int use_huber, double huber, double* __restrict__ r_out,
                                                           double* __restrict__ Ja, double* __restrict__ Jb, double* __restrict__ cost) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= n_edges) return;
  const double* pa = poses + 7 * (size_t)edge_a[e];
  const double* pb = poses + 7 * (size_t)edge_b[e];
  double r[6], s = 0;
  if (JAC) {
    D12 qc[4], tc[3], qn[4], tn[3], rd[6];
    seed_pose(pa, 0, qc, tc);
    seed_pose(pb, 6, qn, tn);
    edge_residual(qc, tc, qn, tn, meas + 6 * (size_t)e, rd);
    for (int i = 0; i < 6; i++) {
      r[i] = rd[i].v;
      s += r[i] * r[i];
    }
    double k = 1.0;
    if (use_huber && s > huber * huber) k = sqrt(huber / sqrt(s));
    for (int i = 0; i < 6; i++)
      for (int c = 0; c < 6; c++) {
        Ja[36 * (size_t)e + 6 * i + c] = k * rd[i].d[c];
        Jb[36 * (size_t)e + 6 * i + c] = k * rd[i].d[6 + c];
      }
    for (int i = 0; i < 6; i++) r_out[6 * (size_t)e + i] = k * r[i];
  } else {
    edge_residual(pa, pa + 4, pb, pb + 4, meas + 6 * (size_t)e, r);
    for (int i = 0; i < 6; i++) s += r[i] * r[i];
  }
  double rho0 = s;
  if (use_huber && s > huber * huber) rho0 = 2.0 * huber * sqrt(s) - huber * huber;
  cost[e] = 0.5 * rho0;
}

// squared column norms of the (robustified) Jacobian, for the Jacobi scaling
// incidence lists: node i's entries inc[inc_off[i] .. inc_off[i + 1]) = 2 * edge + side (0: the node is edge_a, 1: edge_b),
// ascending.  One thread per (free node, column): squared column norm of the Jacobian, summed in list order.
__global__ void pgo_colsq_kernel(int n_nodes, const int* __restrict__ inc_off, const int* __restrict__ inc,
                                 const int* __restrict__ free_idx, const double* __restrict__ Ja, const double* __restrict__ Jb,
                                 double* __restrict__ colsq) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_nodes * 6) return;
  const int node = t / 6, c = t - node * 6;
  const int nx = free_idx[node];
  if (nx < 0) return;
  double s = 0;
  for (int k = inc_off[node]; k < inc_off[node + 1]; k++) {
    const int e = inc[k] >> 1, x = inc[k] & 1;
    const double* J = (x ? Jb : Ja) + 36 * (size_t)e;
    double q = 0;
    for (int i = 0; i < 6; i++) q += J[6 * i + c] * J[6 * i + c];
    s += q;
  }
  colsq[6 * nx + c] = s;
}

__global__ void pgo_scale_kernel(int n, const double* __restrict__ colsq, double* __restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale[i] = 1.0 / (1.0 + sqrt(colsq[i]));
}

// H (n x n, scaled) += J^T J, g += J^T r: one thread per (edge, block x, column c)
// One thread per (free node, row c of its 6 x 6 block row): row 6 nx + c of the scaled normal equations and entry
// 6 nx + c of the gradient, accumulated over the node's incident edges in list order.  The thread is the only writer
// of its row (H and g are zeroed before the launch).
// Band forms (round 4): ld > 0 -- row i of H keeps its columns [i - (ld), i] at H[i * (ld + 1) + (j - i + ld)] (the
// LAPACK-style lower band storage of chol.hip, ld = bandwidth + 32); the thread writes the entries at or left of the
// diagonal (the mirror entries belong to the other endpoint's thread), and in the CYCLIC form (`cyclic`) the entries of
// the wrap-around corner (j - i >= n - bw: nodes that are neighbours around the loop) as (i, j - n) in the leading slots
// of its row.  ld = 0: dense, every entry.
__global__ void pgo_build_kernel(int n_nodes, int n, const int* __restrict__ inc_off, const int* __restrict__ inc,
                                 const int* __restrict__ edge_a, const int* __restrict__ edge_b,
                                 const int* __restrict__ free_idx, const double* __restrict__ r, const double* __restrict__ Ja,
                                 const double* __restrict__ Jb, const double* __restrict__ scale, double* __restrict__ H,
                                 double* __restrict__ g, int ld, int bw, int cyclic) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_nodes * 6) return;
  const int node = t / 6, c = t - node * 6;
  const int nx = free_idx[node];
  if (nx < 0) return;
  const double sx = scale[6 * nx + c];
  const int i = 6 * nx + c;
  double* __restrict__ Hrow = ld > 0 ? H + (size_t)i * (ld + 1) + (ld - i) : H + (size_t)i * n;  // Hrow[j] = entry (i, j)
  double gacc = 0;
  for (int k = inc_off[node]; k < inc_off[node + 1]; k++) {
    const int e = inc[k] >> 1, x = inc[k] & 1;
    const double* Jx = (x ? Jb : Ja) + 36 * (size_t)e;
    const double* re = r + 6 * (size_t)e;
    double col[6], gv = 0;
    for (int q = 0; q < 6; q++) {
      col[q] = Jx[6 * q + c];
      gv += col[q] * re[q];
    }
    gacc += sx * gv;
    for (int y = 0; y < 2; y++) {
      const int ny = free_idx[y ? edge_b[e] : edge_a[e]];
      if (ny < 0) continue;
      const double* Jy = (y ? Jb : Ja) + 36 * (size_t)e;
      for (int c2 = 0; c2 < 6; c2++) {
        int j = 6 * ny + c2;
        if (ld > 0) {
          if (j > i) {
            if (!(cyclic && j - i >= n - bw)) continue;  // the mirror entry is the other thread's
            j -= n;                                        // wrap-around corner: column "below zero"
          } else if (i - j > bw) {
            continue;  // (cyclic: the far side of a wrap pair -- stored by the row of the smaller index)
          }
        }
        double hv = 0;
        for (int q = 0; q < 6; q++) hv += col[q] * Jy[6 * q + c2];
        Hrow[j] += sx * scale[6 * ny + c2] * hv;
      }
    }
  }
  g[6 * nx + c] = gacc;
}

// band forms: A = H (all s_elems slots) with the LM damping on the diagonal, b = -g, gabs
__global__ void pgo_damp_band_kernel(int n, size_t elems, int ld, const double* __restrict__ H, const double* __restrict__ g,
                                     const double* __restrict__ scale, double inv_radius, double* __restrict__ A,
                                     double* __restrict__ b, double* __restrict__ gabs) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= elems) return;
  double v = H[t];
  const size_t row = t / (size_t)(ld + 1);
  if (row < (size_t)n && t - row * (ld + 1) == (size_t)ld) {  // the diagonal slot of row `row`
    v += fmin(fmax(v, 1e-6), 1e32) * inv_radius;
    b[row] = -g[row];
    gabs[row] = fabs(g[row] / scale[row]);
  }
  A[t] = v;
}

// band forms: the model cost change without H: d^T H d = sum over the edges of |Ja S_a d_a + Jb S_b d_b|^2.
// part[e] = |J_e S d|^2 / 2 per edge; the gradient part -d_i g_i per unknown goes to part2 (flag cleared on a non-finite d_i)
__global__ void pgo_model_edges_kernel(int n_edges, const int* __restrict__ edge_a, const int* __restrict__ edge_b,
                                       const int* __restrict__ free_idx, const double* __restrict__ Ja,
                                       const double* __restrict__ Jb, const double* __restrict__ scale,
                                       const double* __restrict__ d, double* __restrict__ part) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int fa = free_idx[edge_a[e]], fb = free_idx[edge_b[e]];
  double v[6] = {0, 0, 0, 0, 0, 0};
  // (H is the SCALED system S J^T J S and d its solution: the kernels' Ja / Jb are unscaled, build applies S -- so here)
  for (int side = 0; side < 2; side++) {
    const int f = side ? fb : fa;
    if (f < 0) continue;
    const double* J = (side ? Jb : Ja) + 36 * (size_t)e;
    for (int q = 0; q < 6; q++)
      for (int c = 0; c < 6; c++) v[q] += J[6 * q + c] * (scale[6 * f + c] * d[6 * f + c]);
  }
  double s = 0;
  for (int q = 0; q < 6; q++) s += v[q] * v[q];
  part[e] = 0.5 * s;
}
__global__ void pgo_model_grad_kernel(int n, const double* __restrict__ g, const double* __restrict__ d, double* __restrict__ part,
                                      int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  part[i] = -d[i] * g[i];
  if (!isfinite(d[i])) *flag = 0;
}

// A = H + diag(clamp(H_ii)) / radius, b = -g; gabs[i] = |g_i / scale_i|
__global__ void pgo_damp_kernel(int n, const double* __restrict__ H, const double* __restrict__ g, const double* __restrict__ scale,
                                double inv_radius, double* __restrict__ A, double* __restrict__ b, double* __restrict__ gabs) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * n) return;
  const int i = (int)(t / n), j = (int)(t - (size_t)i * n);
  double v = H[t];
  if (i == j) {
    v += fmin(fmax(v, 1e-6), 1e32) * inv_radius;
    b[i] = -g[i];
    gabs[i] = fabs(g[i] / scale[i]);
  }
  A[t] = v;
}

// part[i] = -d_i (g_i + (H d)_i / 2); flag cleared if d_i is not finite
__global__ void pgo_model_kernel(int n, const double* __restrict__ H, const double* __restrict__ g, const double* __restrict__ d,
                                 double* __restrict__ part, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double hd = 0;
  for (int j = 0; j < n; j++) hd += H[(size_t)i * n + j] * d[j];
  part[i] = -d[i] * (g[i] + 0.5 * hd);
  if (!isfinite(d[i])) *flag = 0;
}

// candidate poses T * exp(scale .* d) for the free nodes; part_step[i] / part_x[i] = squared norms per node
__global__ void pgo_update_kernel(int n_nodes, const int* __restrict__ free_idx, const double* __restrict__ poses,
                                  const double* __restrict__ d, const double* __restrict__ scale, double* __restrict__ cand,
                                  double* __restrict__ part_step, double* __restrict__ part_x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const double* p7 = poses + 7 * (size_t)i;
  double* o7 = cand + 7 * (size_t)i;
  const int f = free_idx[i];
  if (f < 0) {
    for (int c = 0; c < 7; c++) o7[c] = p7[c];
    part_step[i] = 0;
    part_x[i] = 0;
    return;
  }
  double dl[6], s2 = 0, x2 = 0;
  for (int c = 0; c < 6; c++) {
    dl[c] = scale[6 * f + c] * d[6 * f + c];
    s2 += dl[c] * dl[c];
  }
  for (int c = 0; c < 7; c++) x2 += p7[c] * p7[c];
  part_step[i] = s2;
  part_x[i] = x2;
  // Sophus SE3 exp (local_parameterization_se3.hpp:43-50: T * exp(delta))
  const double* ups = dl;
  const double* om = dl + 3;
  const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2], th = sqrt(th2);
  double imag, real, A, B;
  if (th < 1e-10) {
    imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0;
    real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0;
    A = 0.5;
    B = 1.0 / 6.0;
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
    A = (1.0 - cos(th)) / th2;
    B = (th - sin(th)) / (th2 * th);
  }
  const double dq[4] = {imag * om[0], imag * om[1], imag * om[2], real};
  const double a[3] = {om[1] * ups[2] - om[2] * ups[1], om[2] * ups[0] - om[0] * ups[2], om[0] * ups[1] - om[1] * ups[0]};
  const double b[3] = {om[1] * a[2] - om[2] * a[1], om[2] * a[0] - om[0] * a[2], om[0] * a[1] - om[1] * a[0]};
  double dt[3], rt[3], q[4];
  for (int c = 0; c < 3; c++) dt[c] = ups[c] + A * a[c] + B * b[c];
  qrot(p7, dt, rt);
  qmul(p7, dq, q);
  const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int c = 0; c < 4; c++) o7[c] = q[c] / nq;
  for (int c = 0; c < 3; c++) o7[4 + c] = p7[4 + c] + rt[c];
}

__global__ __launch_bounds__(256) void pgo_reduce_kernel(const double* __restrict__ v, int n, double* __restrict__ out, int is_max) {
  __shared__ double sh[256];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 256) a = is_max ? fmax(a, v[i]) : a + v[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + o]) : sh[threadIdx.x] + sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

struct Buf {
  void* p = nullptr;
  bool owned = true;  // (PgoState carves its buffers out of ONE allocation: 23 hipMalloc / hipFree pairs were ~1 ms of a 3 ms solve)
  ~Buf() {
    if (p && owned) (void)hipFree(p);
  }
  template <class T>
  T* as() {
    return (T*)p;
  }
};

}  // namespace

// (buffers of a solve: sizes collected first, ONE device allocation, carved in 256-byte steps)
#define PGO_ALLOC(buf, bytes) want.push_back({&(buf), (size_t)(bytes)})

static int pgo_validate(vsl_ctx* ctx, const vsl_pgo_problem* p) {
  if (!ctx) return VSL_ERR_INVALID;
  if (!p || p->n_nodes < 0 || p->n_edges < 0 || (p->n_nodes > 0 && (!p->poses || !p->node_fixed)) ||
      (p->n_edges > 0 && (!p->edge_a || !p->edge_b || !p->edge_meas)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "pose graph: null arrays or negative sizes");
  for (int e = 0; e < p->n_edges; e++)
    if (p->edge_a[e] < 0 || p->edge_a[e] >= p->n_nodes || p->edge_b[e] < 0 || p->edge_b[e] >= p->n_nodes || p->edge_a[e] == p->edge_b[e])
      return vsl_fail(ctx, VSL_ERR_INVALID, "pose graph: edge %d connects %d and %d (%d nodes)", e, p->edge_a[e], p->edge_b[e], p->n_nodes);
  return VSL_OK;
}

namespace {
struct PgoState {
  int N = 0, E = 0, n = 0;
  // storage of the normal equations: dense (ld = 0) or, when the graph is narrow in the nodes' own order (keyframes in
  // time order: odometry + covisibility edges, and the loop edge that closes the ring), lower band storage -- linear or
  // CYCLIC (the band closes on itself; chol.hip's ring solver) -- with ld = bw + 32 slots left of the diagonal
  int ld = 0, bw = 0, cyclic = 0;
  size_t h_elems = 0;
  bool force_dense = false;
  Buf arena;
  Buf poses, cand, free_idx, edge_a, edge_b, inc_off, inc, meas, r, Ja, Jb, cost, colsq, scale, H, g, A, b, gabs, part, part2, scalars, flag;
};

int pgo_setup(vsl_ctx* ctx, const vsl_pgo_problem* p, PgoState& st) {
  st.N = p->n_nodes;
  st.E = p->n_edges;
  std::vector<int> free_idx(st.N, -1);
  int nf = 0;
  for (int i = 0; i < st.N; i++)
    if (!p->node_fixed[i]) free_idx[i] = nf++;
  st.n = 6 * nf;
  {
    int lin = 0, cyc = 0;
    for (int e = 0; e < st.E; e++) {
      const int fa = free_idx[p->edge_a[e]], fb = free_idx[p->edge_b[e]];
      if (fa < 0 || fb < 0) continue;
      const int d = std::abs(fa - fb);
      lin = std::max(lin, d);
      cyc = std::max(cyc, std::min(d, nf - d));
    }
    const int bw_lin = 6 * lin + 5, bw_cyc = 6 * cyc + 5;
    int Bc = 0, nc = 0;
    static const bool env_dense = getenv("VSL_PGO_DENSE") != nullptr;
    st.ld = 0;
    st.h_elems = (size_t)st.n * st.n;
    if (!st.force_dense && !env_dense && !ctx->ba_force_dense && st.n > 128) {
      if (!ctx->chol_no_bcr && !ctx->chol_no_fused && 2 * bw_cyc < bw_lin && vsl_chol_bcr_cyclic_layout(st.n, bw_cyc, &Bc, &nc)) {
        st.cyclic = 1;
        st.bw = bw_cyc;
      } else if ((size_t)(bw_lin + 33) * 2 < (size_t)st.n) {
        st.cyclic = 0;
        st.bw = bw_lin;
      } else {
        st.bw = 0;
      }
      if (st.bw > 0) {
        st.ld = st.bw + 32;
        st.h_elems = (size_t)st.n * (st.ld + 1) + 64;
      }
    }
  }
  const size_t N = st.N, E = st.E, n = st.n;
  struct Want { Buf* b; size_t bytes; };
  std::vector<Want> want;
  PGO_ALLOC(st.poses, 56 * N);
  PGO_ALLOC(st.cand, 56 * N);
  PGO_ALLOC(st.free_idx, 4 * N);
  PGO_ALLOC(st.edge_a, 4 * E);
  PGO_ALLOC(st.edge_b, 4 * E);
  PGO_ALLOC(st.inc_off, 4 * (N + 1));
  PGO_ALLOC(st.inc, 8 * E);
  PGO_ALLOC(st.meas, 48 * E);
  PGO_ALLOC(st.r, 48 * E);
  PGO_ALLOC(st.Ja, 288 * E);
  PGO_ALLOC(st.Jb, 288 * E);
  PGO_ALLOC(st.cost, 8 * E);
  PGO_ALLOC(st.colsq, 8 * n);
  PGO_ALLOC(st.scale, 8 * n);
  PGO_ALLOC(st.H, 8 * st.h_elems);
  PGO_ALLOC(st.g, 8 * n);
  PGO_ALLOC(st.A, 8 * st.h_elems);
  PGO_ALLOC(st.b, 8 * n);
  PGO_ALLOC(st.gabs, 8 * n);
  PGO_ALLOC(st.part, 8 * std::max(std::max(n, N), E));
  PGO_ALLOC(st.part2, 8 * std::max(std::max(n, N), E));
  PGO_ALLOC(st.scalars, 64);
  PGO_ALLOC(st.flag, 8);
  {
    size_t total = 0;
    for (auto& w : want) total += (std::max<size_t>(w.bytes, 8) + 255) & ~(size_t)255;
    if (hipMalloc(&st.arena.p, total) != hipSuccess)
      return vsl_fail(ctx, VSL_ERR_NOMEM, "vsl_pose_graph_optimize: device allocation of %zu bytes failed", total);
    size_t off = 0;
    for (auto& w : want) {
      w.b->p = (char*)st.arena.p + off;
      w.b->owned = false;
      off += (std::max<size_t>(w.bytes, 8) + 255) & ~(size_t)255;
    }
  }
  hipStream_t s = ctx->stream;
  if (N) VSL_HIP(ctx, hipMemcpyAsync(st.poses.p, p->poses, 56 * N, hipMemcpyHostToDevice, s));
  if (N) VSL_HIP(ctx, hipMemcpyAsync(st.free_idx.p, free_idx.data(), 4 * N, hipMemcpyHostToDevice, s));
  // incidence lists in edge order (a counting sort over the nodes keeps the edge indices ascending inside a list)
  std::vector<int> inc_off(N + 1, 0), inc(2 * E);
  for (size_t e = 0; e < E; e++) {
    inc_off[p->edge_a[e] + 1]++;
    inc_off[p->edge_b[e] + 1]++;
  }
  for (size_t i = 0; i < N; i++) inc_off[i + 1] += inc_off[i];
  {
    std::vector<int> cur(inc_off.begin(), inc_off.end() - 1);
    for (size_t e = 0; e < E; e++) {
      inc[cur[p->edge_a[e]]++] = 2 * (int)e;
      inc[cur[p->edge_b[e]]++] = 2 * (int)e + 1;
    }
  }
  VSL_HIP(ctx, hipMemcpyAsync(st.inc_off.p, inc_off.data(), 4 * (N + 1), hipMemcpyHostToDevice, s));
  if (E) VSL_HIP(ctx, hipMemcpyAsync(st.inc.p, inc.data(), 8 * E, hipMemcpyHostToDevice, s));
  if (E) {
    VSL_HIP(ctx, hipMemcpyAsync(st.edge_a.p, p->edge_a, 4 * E, hipMemcpyHostToDevice, s));
    VSL_HIP(ctx, hipMemcpyAsync(st.edge_b.p, p->edge_b, 4 * E, hipMemcpyHostToDevice, s));
    VSL_HIP(ctx, hipMemcpyAsync(st.meas.p, p->edge_meas, 48 * E, hipMemcpyHostToDevice, s));
  }
  VSL_HIP(ctx, hipStreamSynchronize(s));  // free_idx and the incidence lists are locals
  return VSL_OK;
}

// residuals + Jacobians at st.poses, optional (re)build of the scaled normal equations; *cost_out = total cost
int pgo_linearize(vsl_ctx* ctx, PgoState& st, const vsl_ba_options* opt, bool first, double* cost_out) {
  hipStream_t s = ctx->stream;
  const int E = st.E, n = st.n;
  if (E > 0)
    hipLaunchKernelGGL(pgo_linearize_kernel<true>, dim3((E + 63) / 64), dim3(64), 0, s, E, st.poses.as<double>(), st.edge_a.as<int>(),
                       st.edge_b.as<int>(), st.meas.as<double>(), opt->use_huber, opt->huber_parameter, st.r.as<double>(),
                       st.Ja.as<double>(), st.Jb.as<double>(), st.cost.as<double>());
  if (first && n > 0) {
    VSL_HIP(ctx, hipMemsetAsync(st.colsq.p, 0, 8 * (size_t)n, s));
    if (st.N > 0)
      hipLaunchKernelGGL(pgo_colsq_kernel, dim3((st.N * 6 + 255) / 256), dim3(256), 0, s, st.N, st.inc_off.as<int>(), st.inc.as<int>(),
                         st.free_idx.as<int>(), st.Ja.as<double>(), st.Jb.as<double>(), st.colsq.as<double>());
    hipLaunchKernelGGL(pgo_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, st.colsq.as<double>(), st.scale.as<double>());
  }
  if (n > 0) {
    VSL_HIP(ctx, hipMemsetAsync(st.H.p, 0, 8 * st.h_elems, s));
    VSL_HIP(ctx, hipMemsetAsync(st.g.p, 0, 8 * (size_t)n, s));
    if (st.N > 0)
      hipLaunchKernelGGL(pgo_build_kernel, dim3((st.N * 6 + 255) / 256), dim3(256), 0, s, st.N, n, st.inc_off.as<int>(), st.inc.as<int>(),
                         st.edge_a.as<int>(), st.edge_b.as<int>(), st.free_idx.as<int>(), st.r.as<double>(), st.Ja.as<double>(),
                         st.Jb.as<double>(), st.scale.as<double>(), st.H.as<double>(), st.g.as<double>(), st.ld, st.bw, st.cyclic);
  }
  hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.cost.as<double>(), E, st.scalars.as<double>(), 0);
  VSL_CHECK_LAUNCH(ctx);
  VSL_HIP(ctx, hipMemcpyAsync(cost_out, st.scalars.p, 8, hipMemcpyDeviceToHost, s));
  VSL_HIP(ctx, hipStreamSynchronize(s));
  return VSL_OK;
}
}  // namespace

// H (n x n row-major, n = 6 x free nodes in node order), g and cost of the robustified problem at the given
// poses, unscaled -- the test hook next to orc_pgo_linearize.
extern "C" int vsl_pgo_linearize(vsl_ctx* ctx, const vsl_pgo_problem* prob, const vsl_ba_options* opt, double* H, double* g,
                                 double* cost, int* n_free) {
  int rc = pgo_validate(ctx, prob);
  if (rc) return rc;
  if (!opt) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_pgo_linearize: options are null");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  PgoState st;
  st.force_dense = true;  // (the hook returns H as a dense matrix)
  if ((rc = pgo_setup(ctx, prob, st))) return rc;
  // unit scaling: fill scale with ones by pretending every column norm is zero
  if (st.n > 0) {
    VSL_HIP(ctx, hipMemsetAsync(st.colsq.p, 0, 8 * (size_t)st.n, ctx->stream));
    hipLaunchKernelGGL(pgo_scale_kernel, dim3((st.n + 255) / 256), dim3(256), 0, ctx->stream, st.n, st.colsq.as<double>(),
                       st.scale.as<double>());
  }
  double c = 0;
  if ((rc = pgo_linearize(ctx, st, opt, false, &c))) return rc;
  if (H && st.n) VSL_HIP(ctx, hipMemcpy(H, st.H.p, 8 * (size_t)st.n * st.n, hipMemcpyDeviceToHost));
  if (g && st.n) VSL_HIP(ctx, hipMemcpy(g, st.g.p, 8 * (size_t)st.n, hipMemcpyDeviceToHost));
  if (cost) *cost = c;
  if (n_free) *n_free = st.n / 6;
  return VSL_OK;
}

extern "C" int vsl_pose_graph_optimize(vsl_ctx* ctx, const vsl_pgo_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary) {
  int rc = pgo_validate(ctx, prob);
  if (rc) return rc;
  if (!opt) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_pose_graph_optimize: options are null");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  PgoState st;
  if ((rc = pgo_setup(ctx, prob, st))) return rc;
  hipStream_t s = ctx->stream;
  const int n = st.n, N = st.N, E = st.E;
  vsl_ba_summary sum;
  memset(&sum, 0, sizeof(sum));
  double cost = 0;
  if ((rc = pgo_linearize(ctx, st, opt, true, &cost))) return rc;
  sum.initial_cost = cost;
  double radius = 1e4, decrease = 2.0, gmax = 1e300;
  int it = 0, invalid = 0;
  bool need_gmax = true;
  while (true) {
    if (it >= opt->max_num_iterations) { sum.termination = 0; break; }
    if (!need_gmax && gmax <= 1e-10) { sum.termination = 2; break; }
    if (radius <= 1e-32) { sum.termination = 4; break; }
    if (n == 0) { sum.termination = 2; break; }
    // damped system + gradient norm
    if (st.ld > 0)
      hipLaunchKernelGGL(pgo_damp_band_kernel, dim3((unsigned)((st.h_elems + 255) / 256)), dim3(256), 0, s, n, st.h_elems, st.ld,
                         st.H.as<double>(), st.g.as<double>(), st.scale.as<double>(), 1.0 / radius, st.A.as<double>(), st.b.as<double>(),
                         st.gabs.as<double>());
    else
      hipLaunchKernelGGL(pgo_damp_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, n, st.H.as<double>(), st.g.as<double>(),
                         st.scale.as<double>(), 1.0 / radius, st.A.as<double>(), st.b.as<double>(), st.gabs.as<double>());
    const bool gmax_pending = need_gmax;
    if (need_gmax) {
      // max |gradient| of this linearisation: reduced here, READ with the step's scalars below (round 4: one host round
      // trip per iteration instead of three -- a gradient below tolerance is found one step late and that step is dropped)
      hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.gabs.as<double>(), n, st.scalars.as<double>() + 1, 1);
      need_gmax = false;
    }
    it++;
    if (st.ld > 0)
      rc = vsl_chol_solve_band_dev(ctx, st.A.as<double>() + st.ld, st.b.as<double>(), n, st.ld, st.bw, st.flag.as<int>(), st.cyclic);
    else
      rc = vsl_chol_solve_dev(ctx, st.A.as<double>(), st.b.as<double>(), n, st.flag.as<int>());
    if (rc) return rc;
    // (the factorisation leaves flag = 1 on success, 0 on a bad pivot; the kernels below only ever CLEAR it -- on a
    // non-finite step --, so one read at the end tells both; after a failed factorisation they run on numbers nobody uses)
    double sc[6] = {0, 0, 0, 0, 0, 0};  // |g| max | model (gradient part in band form) | step^2 | x^2 | candidate cost | band form: sum |J S d|^2 / 2
    int finite = 1;
    {
      if (st.ld > 0) {
        // model change = sum_i -d_i g_i - sum_e |J_e S d|^2 / 2 (no product with H: the band keeps one triangle only)
        hipLaunchKernelGGL(pgo_model_edges_kernel, dim3((E + 255) / 256 > 0 ? (E + 255) / 256 : 1), dim3(256), 0, s, E, st.edge_a.as<int>(),
                           st.edge_b.as<int>(), st.free_idx.as<int>(), st.Ja.as<double>(), st.Jb.as<double>(), st.scale.as<double>(),
                           st.b.as<double>(), st.part2.as<double>());
        hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.part2.as<double>(), E, st.scalars.as<double>() + 6, 0);
        hipLaunchKernelGGL(pgo_model_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, st.g.as<double>(), st.b.as<double>(),
                           st.part.as<double>(), st.flag.as<int>());
      } else {
        hipLaunchKernelGGL(pgo_model_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, st.H.as<double>(), st.g.as<double>(),
                           st.b.as<double>(), st.part.as<double>(), st.flag.as<int>());
      }
      hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.part.as<double>(), n, st.scalars.as<double>() + 2, 0);
      hipLaunchKernelGGL(pgo_update_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, st.free_idx.as<int>(), st.poses.as<double>(),
                         st.b.as<double>(), st.scale.as<double>(), st.cand.as<double>(), st.part.as<double>(), st.part2.as<double>());
      hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.part.as<double>(), N, st.scalars.as<double>() + 3, 0);
      hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.part2.as<double>(), N, st.scalars.as<double>() + 4, 0);
      if (E > 0)
        hipLaunchKernelGGL(pgo_linearize_kernel<false>, dim3((E + 63) / 64), dim3(64), 0, s, E, st.cand.as<double>(), st.edge_a.as<int>(),
                           st.edge_b.as<int>(), st.meas.as<double>(), opt->use_huber, opt->huber_parameter, (double*)nullptr,
                           (double*)nullptr, (double*)nullptr, st.cost.as<double>());
      hipLaunchKernelGGL(pgo_reduce_kernel, dim3(1), dim3(256), 0, s, st.cost.as<double>(), E, st.scalars.as<double>() + 5, 0);
      VSL_CHECK_LAUNCH(ctx);
      VSL_HIP(ctx, hipMemcpyAsync(sc, st.scalars.as<double>() + 1, 48, hipMemcpyDeviceToHost, s));
      VSL_HIP(ctx, hipMemcpyAsync(&finite, st.flag.p, 4, hipMemcpyDeviceToHost, s));
      VSL_HIP(ctx, hipStreamSynchronize(s));
    }
    if (gmax_pending) {
      gmax = sc[0];
      if (gmax <= 1e-10) {
        it--;
        sum.termination = 2;
        break;
      }
    }
    if (st.ld > 0) sc[1] -= sc[5];
    const bool ok = finite != 0 && sc[1] > 0.0;
    if (!ok) {
      if (++invalid >= 5) { sum.termination = 4; break; }
      radius *= 0.5;
      continue;
    }
    invalid = 0;
    const double model = sc[1], step_norm = sqrt(sc[2]), x_norm = sqrt(sc[3]), cand_cost = sc[4];
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 3; break; }
    const double change = cost - cand_cost;
    if (fabs(change) <= 1e-6 * cost) { sum.termination = 1; break; }
    const double rel = change / model;
    if (opt->verbosity >= 2) fprintf(stderr, "pgo %3d cost %.6e change %.3e |g| %.3e step %.3e rho %.3e radius %.3e\n", it, cand_cost, change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      std::swap(st.poses.p, st.cand.p);
      if ((rc = pgo_linearize(ctx, st, opt, false, &cost))) return rc;
      need_gmax = true;
      sum.successful_steps++;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
      decrease = 2.0;
    } else {
      radius /= decrease;
      decrease *= 2.0;
    }
  }
  sum.iterations = it;
  sum.final_cost = cost;
  if (N) VSL_HIP(ctx, hipMemcpy(prob->poses, st.poses.p, 56 * (size_t)N, hipMemcpyDeviceToHost));
  if (opt->verbosity >= 1)
    fprintf(stderr, "vsl PGO: iterations %d, initial cost %.6e, final cost %.6e, termination %d\n", sum.iterations, sum.initial_cost,
            sum.final_cost, sum.termination);
  if (summary) *summary = sum;
  return VSL_OK;
}
