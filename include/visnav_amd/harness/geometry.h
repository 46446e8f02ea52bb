// harness/geometry.h -- the little linear algebra the headless pipeline needs on the host (the
// reference uses Eigen + Sophus, which are not available offline): 3-vectors, 3x3 matrices, rigid
// poses with the Sophus::SE3d storage order (qx qy qz qw tx ty tz, include/visnav/serialization.h:153-162),
// and a 3x3 SVD for the trajectory alignment (src/slam.cpp:1664-1675).
#pragma once
#include <cmath>
#include <cstring>

namespace visnav {
namespace harness {

struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(const Vec3& a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(double s, const Vec3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(const Vec3& a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(const Vec3& a, const Vec3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(const Vec3& a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalized(const Vec3& a) { return (1.0 / norm(a)) * a; }

struct Mat3 {
  double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  static Mat3 zero() {
    Mat3 r;
    std::memset(r.m, 0, sizeof(r.m));
    return r;
  }
  static Mat3 from_cols(const Vec3& a, const Vec3& b, const Vec3& c) {
    Mat3 r;
    for (int i = 0; i < 3; i++) {
      r.m[i][0] = a[i];
      r.m[i][1] = b[i];
      r.m[i][2] = c[i];
    }
    return r;
  }
  Vec3 col(int j) const { return {m[0][j], m[1][j], m[2][j]}; }
};
inline Mat3 operator*(const Mat3& a, const Mat3& b) {
  Mat3 r = Mat3::zero();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) r.m[i][j] += a.m[i][k] * b.m[k][j];
  return r;
}
inline Vec3 operator*(const Mat3& a, const Vec3& v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline Mat3 transpose(const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
inline double det(const Mat3& a) {
  return a.m[0][0] * (a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1]) - a.m[0][1] * (a.m[1][0] * a.m[2][2] - a.m[1][2] * a.m[2][0]) +
         a.m[0][2] * (a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0]);
}
inline Mat3 skew(const Vec3& w) {  // include/visnav/matching_utils.h:51-55
  Mat3 r = Mat3::zero();
  r.m[0][1] = -w.z;
  r.m[0][2] = w.y;
  r.m[1][0] = w.z;
  r.m[1][2] = -w.x;
  r.m[2][0] = -w.y;
  r.m[2][1] = w.x;
  return r;
}
// Rodrigues
inline Mat3 exp_so3(const Vec3& w) {
  const double th2 = dot(w, w), th = std::sqrt(th2);
  const Mat3 K = skew(w);
  const Mat3 K2 = K * K;
  const double a = th < 1e-8 ? 1.0 - th2 / 6.0 : std::sin(th) / th;
  const double b = th < 1e-8 ? 0.5 - th2 / 24.0 : (1.0 - std::cos(th)) / th2;
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = (i == j ? 1.0 : 0.0) + a * K.m[i][j] + b * K2.m[i][j];
  return r;
}

// Rigid transform p_a = R * p_b + t
struct Pose {
  Mat3 R;
  Vec3 t;
};
inline Vec3 operator*(const Pose& T, const Vec3& p) { return T.R * p + T.t; }
inline Pose operator*(const Pose& a, const Pose& b) { return {a.R * b.R, a.R * b.t + a.t}; }
inline Pose inverse(const Pose& T) {
  const Mat3 Rt = transpose(T.R);
  return {Rt, -(Rt * T.t)};
}

inline Mat3 quat_to_rot(double qx, double qy, double qz, double qw) {
  const double n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  qx /= n;
  qy /= n;
  qz /= n;
  qw /= n;
  Mat3 r;
  r.m[0][0] = 1 - 2 * (qy * qy + qz * qz);
  r.m[0][1] = 2 * (qx * qy - qz * qw);
  r.m[0][2] = 2 * (qx * qz + qy * qw);
  r.m[1][0] = 2 * (qx * qy + qz * qw);
  r.m[1][1] = 1 - 2 * (qx * qx + qz * qz);
  r.m[1][2] = 2 * (qy * qz - qx * qw);
  r.m[2][0] = 2 * (qx * qz - qy * qw);
  r.m[2][1] = 2 * (qy * qz + qx * qw);
  r.m[2][2] = 1 - 2 * (qx * qx + qy * qy);
  return r;
}
inline void rot_to_quat(const Mat3& R, double* q /* x y z w */) {
  const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
  double qw, qx, qy, qz;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    qw = 0.25 * s;
    qx = (R.m[2][1] - R.m[1][2]) / s;
    qy = (R.m[0][2] - R.m[2][0]) / s;
    qz = (R.m[1][0] - R.m[0][1]) / s;
  } else if (R.m[0][0] > R.m[1][1] && R.m[0][0] > R.m[2][2]) {
    const double s = std::sqrt(1.0 + R.m[0][0] - R.m[1][1] - R.m[2][2]) * 2;
    qw = (R.m[2][1] - R.m[1][2]) / s;
    qx = 0.25 * s;
    qy = (R.m[0][1] + R.m[1][0]) / s;
    qz = (R.m[0][2] + R.m[2][0]) / s;
  } else if (R.m[1][1] > R.m[2][2]) {
    const double s = std::sqrt(1.0 + R.m[1][1] - R.m[0][0] - R.m[2][2]) * 2;
    qw = (R.m[0][2] - R.m[2][0]) / s;
    qx = (R.m[0][1] + R.m[1][0]) / s;
    qy = 0.25 * s;
    qz = (R.m[1][2] + R.m[2][1]) / s;
  } else {
    const double s = std::sqrt(1.0 + R.m[2][2] - R.m[0][0] - R.m[1][1]) * 2;
    qw = (R.m[1][0] - R.m[0][1]) / s;
    qx = (R.m[0][2] + R.m[2][0]) / s;
    qy = (R.m[1][2] + R.m[2][1]) / s;
    qz = 0.25 * s;
  }
  if (qw < 0) {
    qw = -qw;
    qx = -qx;
    qy = -qy;
    qz = -qz;
  }
  q[0] = qx;
  q[1] = qy;
  q[2] = qz;
  q[3] = qw;
}
// Sophus::SE3d::data() layout: qx qy qz qw tx ty tz
inline Pose pose_from7(const double* d) { return {quat_to_rot(d[0], d[1], d[2], d[3]), Vec3(d[4], d[5], d[6])}; }
inline void pose_to7(const Pose& T, double* d) {
  rot_to_quat(T.R, d);
  d[4] = T.t.x;
  d[5] = T.t.y;
  d[6] = T.t.z;
}

// A = U diag(s) V^T for a 3x3 matrix (one-sided Jacobi on A^T A; s sorted descending, U completed to a
// full orthonormal basis when A is rank deficient).
inline void svd3(const Mat3& A, Mat3& U, double s[3], Mat3& V) {
  Mat3 B = transpose(A) * A;  // symmetric
  V = Mat3();
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 3; q++) off += B.m[p][q] * B.m[p][q];
    if (off < 1e-30) break;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 3; q++) {
        if (std::fabs(B.m[p][q]) < 1e-300) continue;
        const double theta = (B.m[q][q] - B.m[p][p]) / (2.0 * B.m[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        Mat3 J;
        J.m[p][p] = c;
        J.m[q][q] = c;
        J.m[p][q] = sn;
        J.m[q][p] = -sn;
        B = transpose(J) * B * J;
        V = V * J;
      }
  }
  int idx[3] = {0, 1, 2};
  double ev[3] = {B.m[0][0], B.m[1][1], B.m[2][2]};
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 3; j++)
      if (ev[idx[j]] > ev[idx[i]]) {
        const int t = idx[i];
        idx[i] = idx[j];
        idx[j] = t;
      }
  V = Mat3::from_cols(V.col(idx[0]), V.col(idx[1]), V.col(idx[2]));
  Vec3 u[3];
  for (int k = 0; k < 3; k++) {
    u[k] = A * V.col(k);
    s[k] = norm(u[k]);  // more accurate than sqrt(eigenvalue) for small singular values
  }
  const double tol = 1e-12 * (s[0] > 0 ? s[0] : 1.0);
  if (s[0] > tol) u[0] = normalized(u[0]);
  else u[0] = Vec3(1, 0, 0);
  if (s[1] > tol) {
    u[1] = normalized(u[1] - dot(u[1], u[0]) * u[0]);
  } else {
    const Vec3 e = std::fabs(u[0].x) < 0.9 ? Vec3(1, 0, 0) : Vec3(0, 1, 0);
    u[1] = normalized(cross(u[0], e));
  }
  if (s[2] > tol) {
    u[2] = normalized(u[2] - dot(u[2], u[0]) * u[0] - dot(u[2], u[1]) * u[1]);
  } else {
    u[2] = cross(u[0], u[1]);
  }
  U = Mat3::from_cols(u[0], u[1], u[2]);
}

}  // namespace harness
}  // namespace visnav
