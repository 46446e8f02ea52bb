// ba_device.h -- device helpers shared by the bundle-adjustment kernels (ba.hip: operator-by-operator and large-system
// paths; ba_fused.hip: the fused small-system LM iteration): the four camera models of include/visnav/camera_models.h
// with closed-form derivatives, the reprojection residual of include/visnav/reprojection.h:81-105 with its 2x6 / 2x3
// Jacobian blocks in the tangent space of T*exp(delta) (include/visnav/local_parameterization_se3.hpp:43-63), the
// Huber corrector, a 3x3 SPD inverse.
#pragma once
#include "vsl_common.h"

namespace {

// ------------------------------------------------------------------------------- device helpers
struct Proj {
  double u, v;
  double J[6];  // d(u,v)/d(x,y,z) row-major 2x3
};

// camera_models.h project() of the four models + closed-form derivative.
__device__ __forceinline__ void project_jac(int model, const double* __restrict__ ip, double x, double y, double z,
                                            Proj& o, bool want_jac) {
  const double fx = ip[0], fy = ip[1], cx = ip[2], cy = ip[3];
  if (model == VSL_CAM_PINHOLE) {
    o.u = fx * x / z + cx;
    o.v = fy * y / z + cy;
    if (want_jac) {
      const double iz = 1.0 / z;
      o.J[0] = fx * iz; o.J[1] = 0; o.J[2] = -fx * x * iz * iz;
      o.J[3] = 0; o.J[4] = fy * iz; o.J[5] = -fy * y * iz * iz;
    }
  } else if (model == VSL_CAM_EUCM) {
    const double alpha = ip[4], beta = ip[5];
    const double d = sqrt(beta * (x * x + y * y) + z * z);
    const double den = alpha * d + (1.0 - alpha) * z;
    o.u = fx * x / den + cx;
    o.v = fy * y / den + cy;
    if (want_jac) {
      const double dd[3] = {beta * x / d, beta * y / d, z / d};
      const double dn[3] = {alpha * dd[0], alpha * dd[1], alpha * dd[2] + (1.0 - alpha)};
      const double id = 1.0 / den, id2 = id * id;
      o.J[0] = fx * (id - x * dn[0] * id2); o.J[1] = -fx * x * dn[1] * id2; o.J[2] = -fx * x * dn[2] * id2;
      o.J[3] = -fy * y * dn[0] * id2; o.J[4] = fy * (id - y * dn[1] * id2); o.J[5] = -fy * y * dn[2] * id2;
    }
  } else if (model == VSL_CAM_KB4) {
    const double k1 = ip[4], k2 = ip[5], k3 = ip[6], k4 = ip[7];
    const double r = sqrt(x * x + y * y);
    const double th = atan2(r, z);
    const double t2 = th * th;
    const double d = th + k1 * th * th * th + k2 * th * th * th * th * th + k3 * th * th * th * th * th * th * th +
                     k4 * th * th * th * th * th * th * th * th * th;
    if (r == 0.0) {
      o.u = cx;
      o.v = cy;
      if (want_jac) {  // limit r -> 0: d/r -> 1/z
        o.J[0] = fx / z; o.J[1] = 0; o.J[2] = 0;
        o.J[3] = 0; o.J[4] = fy / z; o.J[5] = 0;
      }
    } else {
      o.u = fx * d * x / r + cx;
      o.v = fy * d * y / r + cy;
      if (want_jac) {
        const double dp = 1.0 + t2 * (3 * k1 + t2 * (5 * k2 + t2 * (7 * k3 + t2 * 9 * k4)));
        const double n2 = r * r + z * z;
        const double dth[3] = {z * x / (r * n2), z * y / (r * n2), -r / n2};
        const double ir = 1.0 / r, ir3 = ir * ir * ir;
        const double xr[3] = {y * y * ir3, -x * y * ir3, 0.0};  // d(x/r)
        const double yr[3] = {-x * y * ir3, x * x * ir3, 0.0};  // d(y/r)
        for (int k = 0; k < 3; k++) {
          o.J[k] = fx * (x * ir * dp * dth[k] + d * xr[k]);
          o.J[3 + k] = fy * (y * ir * dp * dth[k] + d * yr[k]);
        }
      }
    }
  } else {  // double sphere
    const double xi = ip[4], alpha = ip[5];
    const double d1 = sqrt(x * x + y * y + z * z);
    const double k = xi * d1 + z;
    const double d2 = sqrt(x * x + y * y + k * k);
    const double den = alpha * d2 + (1.0 - alpha) * k;
    o.u = fx * x / den + cx;
    o.v = fy * y / den + cy;
    if (want_jac) {
      const double id1 = 1.0 / d1;
      const double dk[3] = {xi * x * id1, xi * y * id1, xi * z * id1 + 1.0};
      const double id2 = 1.0 / d2;
      const double dd2[3] = {(x + k * dk[0]) * id2, (y + k * dk[1]) * id2, (k * dk[2]) * id2};
      const double dn[3] = {alpha * dd2[0] + (1.0 - alpha) * dk[0], alpha * dd2[1] + (1.0 - alpha) * dk[1],
                            alpha * dd2[2] + (1.0 - alpha) * dk[2]};
      const double id = 1.0 / den, idd = id * id;
      o.J[0] = fx * (id - x * dn[0] * idd); o.J[1] = -fx * x * dn[1] * idd; o.J[2] = -fx * x * dn[2] * idd;
      o.J[3] = -fy * y * dn[0] * idd; o.J[4] = fy * (id - y * dn[1] * idd); o.J[5] = -fy * y * dn[2] * idd;
    }
  }
}

__device__ __forceinline__ void quat_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// residual r = uv - project(R^T (p - t));  F = dr/d(upsilon, omega) (2x6), E = dr/dp_w (2x3).  Rt = the camera's
// rotation matrix (9, row-major, from quat_R) followed by its translation (3): the fused kernels keep that per camera
// in LDS, the per-observation kernels build it from the pose on the fly (residual_blocks below) -- same arithmetic.
__device__ __forceinline__ void residual_blocks_Rt(int model, const double* __restrict__ intr, const double* Rt,
                                                   const double* pw, const double* uv, double* r, double* F, double* E,
                                                   bool want_jac) {
  const double* R = Rt;
  const double d[3] = {pw[0] - Rt[9], pw[1] - Rt[10], pw[2] - Rt[11]};
  const double pc[3] = {R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                        R[2] * d[0] + R[5] * d[1] + R[8] * d[2]};
  Proj pj;
  project_jac(model, intr, pc[0], pc[1], pc[2], pj, want_jac);
  r[0] = uv[0] - pj.u;
  r[1] = uv[1] - pj.v;
  if (!want_jac) return;
  const double* J = pj.J;
  // d p_c / d(upsilon, omega) = [-I | [p_c]x]  =>  F = [J | -J [p_c]x]
  for (int a = 0; a < 2; a++) {
    const double j0 = J[3 * a], j1 = J[3 * a + 1], j2 = J[3 * a + 2];
    F[6 * a + 0] = j0;
    F[6 * a + 1] = j1;
    F[6 * a + 2] = j2;
    // -J [p]x, [p]x = [0 -pz py; pz 0 -px; -py px 0]
    F[6 * a + 3] = -(j1 * pc[2] - j2 * pc[1]);
    F[6 * a + 4] = -(-j0 * pc[2] + j2 * pc[0]);
    F[6 * a + 5] = -(j0 * pc[1] - j1 * pc[0]);
    // E = -J R^T
    for (int k = 0; k < 3; k++) E[3 * a + k] = -(j0 * R[3 * k] + j1 * R[3 * k + 1] + j2 * R[3 * k + 2]);
  }
}

__device__ __forceinline__ void residual_blocks(int model, const double* __restrict__ intr, const double* pose,
                                                const double* pw, const double* uv, double* r, double* F, double* E,
                                                bool want_jac) {
  double Rt[12];
  quat_R(pose, Rt);
  Rt[9] = pose[4];
  Rt[10] = pose[5];
  Rt[11] = pose[6];
  residual_blocks_Rt(model, intr, Rt, pw, uv, r, F, E, want_jac);
}

// candidate pose = T * exp(d) (include/visnav/local_parameterization_se3.hpp:43-50; [upstream] Sophus SE3::exp):
// T = qx qy qz qw tx ty tz, d = (upsilon, omega); o receives the 7 values of the candidate
__device__ __forceinline__ void se3_plus(const double* T, const double* d, double* o) {
  const double wx = d[3], wy = d[4], wz = d[5];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  double imag, real;
  if (th < 1e-10) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
  } else {
    imag = sin(0.5 * th) / th;
    real = cos(0.5 * th);
  }
  const double dq[4] = {imag * wx, imag * wy, imag * wz, real};
  double V[9];
  if (th < 1e-10) {
    quat_R(dq, V);
  } else {
    const double a = (1 - cos(th)) / th2, b = (th - sin(th)) / (th2 * th);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += O[3 * i + k] * O[3 * k + j];
        V[3 * i + j] = (i == j ? 1.0 : 0.0) + a * O[3 * i + j] + b * s;
      }
  }
  double Vu[3];
  for (int i = 0; i < 3; i++) Vu[i] = V[3 * i] * d[0] + V[3 * i + 1] * d[1] + V[3 * i + 2] * d[2];
  double R[9];
  quat_R(T, R);
  for (int i = 0; i < 3; i++) o[4 + i] = T[4 + i] + R[3 * i] * Vu[0] + R[3 * i + 1] * Vu[1] + R[3 * i + 2] * Vu[2];
  const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
  double nq[4];
  nq[0] = qw * dq[0] + qx * dq[3] + qy * dq[2] - qz * dq[1];
  nq[1] = qw * dq[1] - qx * dq[2] + qy * dq[3] + qz * dq[0];
  nq[2] = qw * dq[2] + qx * dq[1] - qy * dq[0] + qz * dq[3];
  nq[3] = qw * dq[3] - qx * dq[0] - qy * dq[1] - qz * dq[2];
  const double nn = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
  for (int i = 0; i < 4; i++) o[i] = nq[i] / nn;
}

__device__ __forceinline__ void huber(double s, double a, double& rho0, double& rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho0 = 2 * a * r - b;
    rho1 = fmax(2.2250738585072014e-308, a / r);
  } else {
    rho0 = s;
    rho1 = 1.0;
  }
}

// deterministic workgroup sum (256 threads): value returned in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  return sh[0];
}

struct BaDims {
  int C, L, O, nfree, n;
  int model0, model1;
  int use_huber;
  double huber;
};

inline __device__ bool inv3(const double* P, double* Pi) {
  const double a = P[0], b = P[1], c = P[2], d = P[4], e = P[5], f = P[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (!(fabs(det) > 0.0) || !isfinite(det)) return false;
  const double id = 1.0 / det;
  Pi[0] = c00 * id; Pi[1] = c01 * id; Pi[2] = c02 * id;
  Pi[3] = Pi[1]; Pi[4] = (a * f - c * c) * id; Pi[5] = (b * c - a * e) * id;
  Pi[6] = Pi[2]; Pi[7] = Pi[5]; Pi[8] = (a * d - b * b) * id;
  return true;
}

}  // namespace
