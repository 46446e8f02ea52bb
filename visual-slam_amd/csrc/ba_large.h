// ba_large.h -- kernels of the LARGE reduced camera system (global bundle adjustment: > 21 free cameras), recompute form.
//
// Replaces, per Levenberg-Marquardt iteration of visnav::global_bundle_adjustment
// (include/visnav/loop_closure_utils.h:672-748 -> ceres::Solve, SPARSE_SCHUR), the chain
//   ba_linearize (r / F / E arrays: 160 B per observation written, then read by five kernels) -> ba_lm_cols ->
//   ba_cam_block -> ba_schur_prep -> ba_schur_rhs ... ba_backsub -> ba_model -> ba_update -> ba_cost
// with kernels that EVALUATE an observation's blocks from its 20 bytes (camera, landmark, pixel) each time they are
// needed -- the residual and its two Jacobian blocks cost ~200 fp64 operations, the arrays they replace cost 160 B of
// HBM traffic per observation and pass:
//   bal_prep_kernel    workgroup = a run of landmarks (<= BL_THREADS observations), thread = observation:
//                      P = sum E^T E, b = sum E^T r per landmark, damped P^-1 = L L^T, and per free observation
//                      Z = F^T E L in camera-major order (ba_schur_gather_kernel reads it on both sides of a pair:
//                      Y_i W_j^T = Z_i Z_j^T); cost and max |gradient|.
//   bal_cam_kernel     workgroup = (free camera, segment) over the camera's observation list: H = sum F^T F,
//                      g = sum F^T r and the landmark part of the reduced right-hand side, -sum F^T E (P^-1 b).
//   bal_pose_kernel    candidate poses T exp(d), camera parts of the step / x norms.
//   bal_step_kernel    same partition as bal_prep: back-substitution dl = -P^-1 (b + sum W^T dc), model cost change,
//                      candidate points, cost at the candidate.
// The INIT variants run once per solve for the Jacobi scaling (unscaled column norms).  Every sum runs in a fixed
// order (observation order inside a landmark, xor tree inside a wavefront, wavefront / segment / workgroup order
// afterwards): a solve is bit-reproducible from run to run.
#pragma once
#include "ba_device.h"

namespace {

#define BL_THREADS 512  // observations of a workgroup (one per thread)
#define BL_LMW 256      // landmarks of a workgroup
#define BL_WAVES (BL_THREADS / 64)

__device__ __forceinline__ double bl_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double bl_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

struct BlArgs {
  BaDims D;
  const double* poses;
  const double* points;
  const double* intr;
  const int* cam_intr;
  const int* cam_free;
  const int* obs_cam;
  const int* obs_lm;
  const double* obs_uv;
  const int* lm_start;
  const int* wg_lm;      // workgroup g covers landmarks [wg_lm[g], wg_lm[g + 1])
  const double* scale_c;
  const double* scale_l;
};

// an observation evaluated at (poses, points): robustified, Jacobi-scaled blocks (as ba_linearize_kernel stores them)
struct BlObs {
  double r[2], F[12], E[6];
  double cost;
  int fc;
};

template <bool SCALED, bool WANT_F>
__device__ __forceinline__ void bl_eval(const BlArgs& a, int cam, const double* pw, const double* scl, const double* uv,
                                        BlObs& o) {
  const BaDims& D = a.D;
  const int k = a.cam_intr[cam];
  o.fc = a.cam_free[cam];
  residual_blocks(k ? D.model1 : D.model0, a.intr + 8 * k, a.poses + 7 * (size_t)cam, pw, uv, o.r, o.F, o.E, true);
  const double s = o.r[0] * o.r[0] + o.r[1] * o.r[1];
  double rho0 = s, rho1 = 1.0;
  if (D.use_huber) huber(s, D.huber, rho0, rho1);
  o.cost = 0.5 * rho0;
  const double sr = sqrt(rho1);
  o.r[0] *= sr;
  o.r[1] *= sr;
  if (WANT_F) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double sc = (SCALED && o.fc >= 0) ? a.scale_c[6 * o.fc + j] : 1.0;
      o.F[j] *= sr * sc;
      o.F[6 + j] *= sr * sc;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double sc = SCALED ? scl[j] : 1.0;
    o.E[j] *= sr * sc;
    o.E[3 + j] *= sr * sc;
  }
}

// INIT: n2l_out[3 l + x] = squared norm of landmark column x (unscaled), cost partial.  Otherwise: the damped inverse
// P^-1 and b per landmark (kept for the back-substitution), Z = F^T E chol(P^-1) per free observation (the gather's
// block: Y_i W_j^T = Z_i Z_j^T), cost and gradient partials.
// part[0 * G + g] = cost, part[1 * G + g] = max |gradient| over the workgroup's landmark columns (unscaled problem).
template <bool INIT>
__global__ __launch_bounds__(BL_THREADS) void bal_prep_kernel(BlArgs a, const int* __restrict__ cam_pos, double inv_radius,
                                                              double* __restrict__ Zg, double* __restrict__ Pinv,
                                                              double* __restrict__ bl,
                                                              double* __restrict__ pbs, double* __restrict__ n2l_out,
                                                              double* __restrict__ part) {
  __shared__ double stage_s[BL_THREADS * 9];  // E^T E (6) | E^T r (3) per observation
  __shared__ double pi_s[BL_LMW * 9];
  __shared__ double pts_s[BL_LMW * 3];
  __shared__ double scl_s[BL_LMW * 3];
  __shared__ int lmo_s[BL_LMW + 1];
  __shared__ double red_s[2][BL_WAVES];
  const int tid = threadIdx.x, bid = blockIdx.x, lane = tid & 63, wave = tid >> 6, G = gridDim.x;
  const int lm0 = a.wg_lm[bid], n_lm = a.wg_lm[bid + 1] - lm0;
  const int obs0 = a.lm_start[lm0], n_obs = a.lm_start[lm0 + n_lm] - obs0;
  if (tid < n_lm) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      pts_s[3 * tid + j] = a.points[3 * (size_t)(lm0 + tid) + j];
      scl_s[3 * tid + j] = INIT ? 1.0 : a.scale_l[3 * (size_t)(lm0 + tid) + j];
    }
  }
  if (tid <= n_lm) lmo_s[tid] = a.lm_start[lm0 + tid] - obs0;
  const bool have = tid < n_obs;
  const size_t q = (size_t)obs0 + tid;
  int cam = 0, lml = 0;
  double uv[2] = {0.0, 0.0};
  if (have) {
    cam = a.obs_cam[q];
    lml = a.obs_lm[q] - lm0;
    uv[0] = a.obs_uv[2 * q];
    uv[1] = a.obs_uv[2 * q + 1];
  }
  __syncthreads();
  BlObs o;
  o.cost = 0.0;
  o.fc = -1;
  if (have) {
    bl_eval<!INIT, !INIT>(a, cam, pts_s + 3 * lml, scl_s + 3 * lml, uv, o);
    double* st = stage_s + 9 * tid;
    const double* e = o.E;
    st[0] = e[0] * e[0] + e[3] * e[3];
    st[1] = e[0] * e[1] + e[3] * e[4];
    st[2] = e[0] * e[2] + e[3] * e[5];
    st[3] = e[1] * e[1] + e[4] * e[4];
    st[4] = e[1] * e[2] + e[4] * e[5];
    st[5] = e[2] * e[2] + e[5] * e[5];
#pragma unroll
    for (int x = 0; x < 3; x++) st[6 + x] = e[x] * o.r[0] + e[3 + x] * o.r[1];
  }
  {
    const double cs = bl_wave_sum(o.cost);
    if (lane == 0) red_s[0][wave] = cs;
  }
  __syncthreads();
  double gl = 0.0;
  if (tid < n_lm) {
    double P6[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    const int i0 = lmo_s[tid], i1 = lmo_s[tid + 1];
    for (int i = i0; i < i1; i++) {
      const double* st = stage_s + 9 * i;
#pragma unroll
      for (int x = 0; x < 6; x++) P6[x] += st[x];
#pragma unroll
      for (int x = 0; x < 3; x++) bb[x] += st[6 + x];
    }
    const size_t l = (size_t)(lm0 + tid);
    if (INIT) {
      n2l_out[3 * l] = P6[0];
      n2l_out[3 * l + 1] = P6[3];
      n2l_out[3 * l + 2] = P6[5];
    } else {
      double P[9] = {P6[0], P6[1], P6[2], P6[1], P6[3], P6[4], P6[2], P6[4], P6[5]};
      // LM damping: clamp(column norm^2, 1e-6, 1e32) / radius (recomputing the clamp after a rejected step IS keeping
      // it: the linearisation point has not moved)
      P[0] += fmin(fmax(P6[0], 1e-6), 1e32) * inv_radius;
      P[4] += fmin(fmax(P6[3], 1e-6), 1e32) * inv_radius;
      P[8] += fmin(fmax(P6[5], 1e-6), 1e32) * inv_radius;
      double Pi[9];
      bool ok = i1 > i0 && inv3(P, Pi);
      // P^-1 = L L^T (lower Cholesky factor, 3 x 3): the gather forms Y_i W_j^T = (W_i P^-1) W_j^T as Z_i Z_j^T with
      // Z = W L -- ONE 144-byte block per observation instead of two
      double L6[6] = {0, 0, 0, 0, 0, 0};  // l00 l10 l11 l20 l21 l22
      if (ok) {
        const double l00 = sqrt(Pi[0]), l10 = Pi[3] / l00, l20 = Pi[6] / l00;
        const double l11 = sqrt(Pi[4] - l10 * l10), l21 = (Pi[7] - l20 * l10) / l11;
        const double l22 = sqrt(Pi[8] - l20 * l20 - l21 * l21);
        ok = l00 > 0.0 && l11 > 0.0 && l22 > 0.0 && isfinite(l00) && isfinite(l11) && isfinite(l22) && isfinite(l10) &&
             isfinite(l20) && isfinite(l21);  // (a P^-1 that is not positive definite in fp64: treated like a singular P)
        if (ok) {
          L6[0] = l00; L6[1] = l10; L6[2] = l11; L6[3] = l20; L6[4] = l21; L6[5] = l22;
        }
      }
#pragma unroll
      for (int x = 0; x < 9; x++) {
        Pi[x] = ok ? Pi[x] : 0.0;  // a singular block contributes nothing (Z = 0) and its landmark does not move
        Pinv[9 * l + x] = Pi[x];
      }
#pragma unroll
      for (int x = 0; x < 6; x++) pi_s[9 * tid + x] = L6[x];
#pragma unroll
      for (int x = 0; x < 3; x++) {
        bl[3 * l + x] = ok ? bb[x] : 0.0;
        // (P^-1 b) in the scale of the UNSCALED landmark block: the camera kernel multiplies it with E sqrt(rho')
        pbs[3 * l + x] = (Pi[3 * x] * bb[0] + Pi[3 * x + 1] * bb[1] + Pi[3 * x + 2] * bb[2]) * scl_s[3 * tid + x];
        gl = fmax(gl, fabs(bb[x] / scl_s[3 * tid + x]));
      }
    }
  }
  {
    const double gm = bl_wave_max(gl);
    if (lane == 0) red_s[1][wave] = gm;
  }
  __syncthreads();
  if (!INIT) {
    // Z = F^T E L of the free observations, stored COOPERATIVELY: a lane's own 144-byte block as nine 16-byte stores is
    // 64 different cache lines per store instruction (the same lines-not-bytes limit that bounded the gather's loads);
    // instead the blocks of 32 lanes go through the wavefront's 4.6 KB of the (now dead) staging area as 288 pieces of
    // 16 bytes, piece m of block m / 9, and nine consecutive lanes store one contiguous block.
    const bool is_free = have && o.fc >= 0;
    double z[18];
    if (is_free) {
      const double* Lm = pi_s + 9 * lml;  // l00 l10 l11 l20 l21 l22
#pragma unroll
      for (int x = 0; x < 6; x++) {
        double w[3];
#pragma unroll
        for (int k = 0; k < 3; k++) w[k] = o.F[x] * o.E[k] + o.F[6 + x] * o.E[3 + k];
        z[3 * x] = w[0] * Lm[0] + w[1] * Lm[1] + w[2] * Lm[3];
        z[3 * x + 1] = w[1] * Lm[2] + w[2] * Lm[4];
        z[3 * x + 2] = w[2] * Lm[5];
      }
    }
    const int pos = is_free ? cam_pos[q] : -1;
    double2* region = (double2*)stage_s + (size_t)wave * (BL_THREADS * 9 / 2 / BL_WAVES);  // 288 double2 per wavefront
    static_assert(BL_THREADS * 9 / 2 / BL_WAVES == 288, "32 blocks of 9 pieces per wavefront and round");
#pragma unroll
    for (int half = 0; half < 2; half++) {
      __syncthreads();  // (the staging area: dead since the landmark sums / the previous round's pieces are out)
      if (is_free && (lane >> 5) == half) {
#pragma unroll
        for (int x = 0; x < 9; x++) region[9 * (lane & 31) + x] = make_double2(z[2 * x], z[2 * x + 1]);
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 5; t++) {
        const int m = 64 * t + lane, b = m < 288 ? m / 9 : 0, part = m - 9 * b;
        const int bpos = __shfl(pos, 32 * half + b);
        if (m < 288 && bpos >= 0) *(double2*)(Zg + 18 * (size_t)bpos + 2 * part) = region[m];
      }
    }
  }
  if (tid == 0) {
    double c = 0.0, g = 0.0;
    for (int wv = 0; wv < BL_WAVES; wv++) {
      c += red_s[0][wv];
      g = fmax(g, red_s[1][wv]);
    }
    part[bid] = c;
    part[G + bid] = g;
  }
}

// camera-major copies of (landmark, pixel) for bal_cam_kernel: once per session
__global__ void bal_cam_major_kernel(int O, const int* __restrict__ cam_obs, const int* __restrict__ obs_lm,
                                     const double* __restrict__ obs_uv, int* __restrict__ cam_lm, double* __restrict__ cam_uv) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= O) return;
  const size_t q = (size_t)cam_obs[k];
  cam_lm[k] = obs_lm[q];
  cam_uv[2 * (size_t)k] = obs_uv[2 * q];
  cam_uv[2 * (size_t)k + 1] = obs_uv[2 * q + 1];
}

// per free camera (grid = (free cameras, segments), like ba_cam_block_kernel): 21 entries of the upper triangle of
// H = sum F^T F, g = sum F^T r and the landmark part of the reduced right-hand side, sum over the camera's observations
// of Y b = F^T (E (P^-1 b)) -- from the landmark's 24 bytes pbs, not from the observation's 144-byte Y block (whose 16-byte
// loads touch 64 cache lines per instruction: 80 -> 57 us) --, evaluated from the observations of the camera's list;
// part[(fc * nseg + seg) * 33 + e].
template <bool INIT>
__global__ __launch_bounds__(256) void bal_cam_kernel(BlArgs a, const int* __restrict__ free_cams,
                                                      const int* __restrict__ cam_start, const int* __restrict__ cam_lm,
                                                      const double* __restrict__ cam_uv, const double* __restrict__ pbs,
                                                      double* __restrict__ part) {
  constexpr int NE = INIT ? 27 : 33;
  __shared__ double sh[4][NE];
  const int fc = blockIdx.x, seg = blockIdx.y, nseg = gridDim.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cam = free_cams[fc];
  double acc[NE];
#pragma unroll
  for (int k = 0; k < NE; k++) acc[k] = 0.0;
  for (int k = cam_start[cam] + seg * 256 + threadIdx.x; k < cam_start[cam + 1]; k += 256 * nseg) {
    // (landmark and pixel of the camera's k-th observation from camera-major copies: through cam_obs -> obs_lm / obs_uv
    // they were two more dependent, scattered loads per observation)
    const int lm = cam_lm[k];
    const double2 uv2 = *(const double2*)(cam_uv + 2 * (size_t)k);
    const double uv[2] = {uv2.x, uv2.y};
    const double pw[3] = {a.points[3 * (size_t)lm], a.points[3 * (size_t)lm + 1], a.points[3 * (size_t)lm + 2]};
    const double one[3] = {1.0, 1.0, 1.0};
    BlObs o;
    bl_eval<!INIT, true>(a, cam, pw, one, uv, o);  // (E robustified, NOT Jacobi-scaled: pbs carries the landmark's scale)
    const double* f = o.F;
    int e = 0;
#pragma unroll
    for (int x = 0; x < 6; x++)
#pragma unroll
      for (int y = x; y < 6; y++) acc[e++] += f[x] * f[y] + f[6 + x] * f[6 + y];
#pragma unroll
    for (int x = 0; x < 6; x++) acc[21 + x] += f[x] * o.r[0] + f[6 + x] * o.r[1];
    if (!INIT) {
      const double b0 = pbs[3 * (size_t)lm], b1 = pbs[3 * (size_t)lm + 1], b2 = pbs[3 * (size_t)lm + 2];
      const double u0 = o.E[0] * b0 + o.E[1] * b1 + o.E[2] * b2, u1 = o.E[3] * b0 + o.E[4] * b1 + o.E[5] * b2;
#pragma unroll
      for (int x = 0; x < 6; x++) acc[27 + x] += f[x] * u0 + f[6 + x] * u1;
    }
  }
#pragma unroll
  for (int k = 0; k < NE; k++) {
    const double v = bl_wave_sum(acc[k]);
    if (lane == 0) sh[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NE)
    part[((size_t)fc * nseg + seg) * 33 + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// segments summed in order: H (6 x 6, both triangles), g_c, and rhs = -(sum Y b) (the camera blocks are added to S
// afterwards).  The LAST workgroup also folds bal_prep_kernel's per-workgroup (cost, max |landmark gradient|) into
// scalars[0] / gl_out[0] (lpart given; it was a one-workgroup launch of its own).
__global__ __launch_bounds__(256) void bal_cam_finish_kernel(int nfree, int nseg, int with_rhs, const double* __restrict__ part,
                                                             double* __restrict__ H, double* __restrict__ g,
                                                             double* __restrict__ rhs, int G, const double* __restrict__ lpart,
                                                             double* __restrict__ scalars, double* __restrict__ gl_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nfree * 33) {
    const int fc = t / 33, k = t - fc * 33;
    if (k < 27 || with_rhs) {
      double v = 0;
      for (int sgm = 0; sgm < nseg; sgm++) v += part[((size_t)fc * nseg + sgm) * 33 + k];
      if (k >= 27) {
        rhs[6 * (size_t)fc + (k - 27)] = -v;
      } else if (k >= 21) {
        g[6 * (size_t)fc + (k - 21)] = v;
      } else {
        int x = 0, rem = k;  // k-th entry of the upper triangle, row-major
        while (rem >= 6 - x) {
          rem -= 6 - x;
          x++;
        }
        const int y = x + rem;
        H[36 * (size_t)fc + 6 * x + y] = v;
        H[36 * (size_t)fc + 6 * y + x] = v;
      }
    }
  }
  if (!lpart || blockIdx.x != gridDim.x - 1) return;
  __shared__ double sh[2][256];
  double c = 0.0, gm = 0.0;
  for (int i = threadIdx.x; i < G; i += 256) {
    c += lpart[i];
    gm = fmax(gm, lpart[G + i]);
  }
  sh[0][threadIdx.x] = c;
  sh[1][threadIdx.x] = gm;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] = fmax(sh[1][threadIdx.x], sh[1][threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    scalars[0] = sh[0][0];
    if (gl_out) gl_out[0] = sh[1][0];
  }
}

// candidate poses = T exp(dc .* scale_c) (fixed cameras copied); scalars[6] / [7] = squared step / x norm of the free
// cameras; flag[0] cleared when the camera step is not finite.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void bal_pose_kernel(BaDims D, const int* __restrict__ cam_free,
                                                        const double* __restrict__ poses, const double* __restrict__ dc,
                                                        const double* __restrict__ scale_c, double* __restrict__ cand_poses,
                                                        double* __restrict__ scalars, int* __restrict__ flag) {
  __shared__ double sh[2][16];
  double step2 = 0, x2 = 0;
  bool bad = false;
  for (int c = threadIdx.x; c < D.C; c += 1024) {
    const int fc = cam_free[c];
    const double* T = poses + 7 * (size_t)c;
    double* o = cand_poses + 7 * (size_t)c;
    if (fc < 0) {
      for (int j = 0; j < 7; j++) o[j] = T[j];
      continue;
    }
    double d[6];
    for (int j = 0; j < 6; j++) {
      const double v = dc[6 * fc + j];
      bad = bad || !isfinite(v);
      d[j] = v * scale_c[6 * fc + j];
      step2 += d[j] * d[j];
    }
    for (int j = 0; j < 7; j++) x2 += T[j] * T[j];
    se3_plus(T, d, o);
  }
  if (bad) flag[0] = 0;
  const double s2 = bl_wave_sum(step2), xx = bl_wave_sum(x2);  // xor tree, then the 16 wavefronts in order
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s2;
    sh[1][threadIdx.x >> 6] = xx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; w++) {
      a += sh[0][w];
      b += sh[1][w];
    }
    scalars[6] = a;
    scalars[7] = b;
  }
}

// back-substitution, model cost change, candidate points and the cost at the candidate, for the landmarks of a
// workgroup (partition of bal_prep_kernel).  part[k * G + g], k = model change | squared step norm | squared x norm
// (landmarks) | candidate cost.
__global__ __launch_bounds__(BL_THREADS) void bal_step_kernel(BlArgs a, const double* __restrict__ Pinv,
                                                              const double* __restrict__ bl, const double* __restrict__ dc,
                                                              const double* __restrict__ cand_poses,
                                                              double* __restrict__ cand_points, double* __restrict__ part,
                                                              int* __restrict__ flag) {
  __shared__ double stage_s[BL_THREADS * 3];  // E^T (F dc) per observation
  __shared__ double pts_s[BL_LMW * 3];
  __shared__ double scl_s[BL_LMW * 3];
  __shared__ double dl_s[BL_LMW * 3];
  __shared__ double cpt_s[BL_LMW * 3];
  __shared__ int lmo_s[BL_LMW + 1];
  __shared__ double red_s[4][BL_WAVES];
  const BaDims& D = a.D;
  const int tid = threadIdx.x, bid = blockIdx.x, lane = tid & 63, wave = tid >> 6, G = gridDim.x;
  const int lm0 = a.wg_lm[bid], n_lm = a.wg_lm[bid + 1] - lm0;
  const int obs0 = a.lm_start[lm0], n_obs = a.lm_start[lm0 + n_lm] - obs0;
  if (tid < n_lm) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      pts_s[3 * tid + j] = a.points[3 * (size_t)(lm0 + tid) + j];
      scl_s[3 * tid + j] = a.scale_l[3 * (size_t)(lm0 + tid) + j];
    }
  }
  if (tid <= n_lm) lmo_s[tid] = a.lm_start[lm0 + tid] - obs0;
  const bool have = tid < n_obs;
  const size_t q = (size_t)obs0 + tid;
  int cam = 0, lml = 0;
  double uv[2] = {0.0, 0.0};
  if (have) {
    cam = a.obs_cam[q];
    lml = a.obs_lm[q] - lm0;
    uv[0] = a.obs_uv[2 * q];
    uv[1] = a.obs_uv[2 * q + 1];
  }
  __syncthreads();
  BlObs o;
  double u0 = 0.0, u1 = 0.0;
  if (have) {
    bl_eval<true, true>(a, cam, pts_s + 3 * lml, scl_s + 3 * lml, uv, o);
    if (o.fc >= 0) {
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const double d = dc[6 * o.fc + j];
        u0 += o.F[j] * d;
        u1 += o.F[6 + j] * d;
      }
    }
#pragma unroll
    for (int z = 0; z < 3; z++) stage_s[3 * tid + z] = o.E[z] * u0 + o.E[3 + z] * u1;
  }
  __syncthreads();
  double step2 = 0.0, x2 = 0.0;
  if (tid < n_lm) {
    const size_t l = (size_t)(lm0 + tid);
    double t[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
    for (int i = lmo_s[tid]; i < lmo_s[tid + 1]; i++) {
#pragma unroll
      for (int z = 0; z < 3; z++) t[z] += stage_s[3 * i + z];
    }
    const double* Pi = Pinv + 9 * l;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double v = -(Pi[3 * j] * t[0] + Pi[3 * j + 1] * t[1] + Pi[3 * j + 2] * t[2]);
      bad = bad || !isfinite(v);
      dl_s[3 * tid + j] = v;
      const double dd = v * scl_s[3 * tid + j];
      step2 += dd * dd;
      const double xv = pts_s[3 * tid + j];
      x2 += xv * xv;
      cpt_s[3 * tid + j] = xv + dd;
      cand_points[3 * l + j] = xv + dd;
    }
    if (bad) flag[0] = 0;
  }
  __syncthreads();
  double model = 0.0, ccost = 0.0;
  if (have) {
    const double* d = dl_s + 3 * lml;
    const double m0 = u0 + o.E[0] * d[0] + o.E[1] * d[1] + o.E[2] * d[2];
    const double m1 = u1 + o.E[3] * d[0] + o.E[4] * d[1] + o.E[5] * d[2];
    model = -(m0 * (o.r[0] + m0 / 2.0) + m1 * (o.r[1] + m1 / 2.0));
    const int k = a.cam_intr[cam];
    double rc[2];
    residual_blocks(k ? D.model1 : D.model0, a.intr + 8 * k, cand_poses + 7 * (size_t)cam, cpt_s + 3 * lml, uv, rc, nullptr,
                    nullptr, false);
    const double s = rc[0] * rc[0] + rc[1] * rc[1];
    double rho0 = s, rho1 = 1.0;
    if (D.use_huber) huber(s, D.huber, rho0, rho1);
    ccost = 0.5 * rho0;
  }
  const double v4[4] = {bl_wave_sum(model), bl_wave_sum(step2), bl_wave_sum(x2), bl_wave_sum(ccost)};
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) red_s[k][wave] = v4[k];
  }
  __syncthreads();
  if (tid < 4) {
    double v = 0.0;
    for (int wv = 0; wv < BL_WAVES; wv++) v += red_s[tid][wv];
    part[(size_t)tid * G + bid] = v;
  }
}

// scalars[2] = model change, [3] / [4] = squared step / x norms (landmarks + cameras), [5] = candidate cost.  One
// workgroup; scalars[6] / [7] were written by bal_pose_kernel.  packC (nullable): the session's step record as well.
__global__ __launch_bounds__(256) void bal_step_finish_kernel(int G, const double* __restrict__ part,
                                                              double* __restrict__ scalars, const int* __restrict__ flag,
                                                              double* __restrict__ packC) {
  __shared__ double sh[4][256];
  double v[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < G; i += 256)
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] += part[(size_t)k * G + i];
#pragma unroll
  for (int k = 0; k < 4; k++) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int k = 0; k < 4; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double s6 = scalars[6], s7 = scalars[7];
    scalars[2] = sh[0][0];
    scalars[3] = sh[1][0] + s6;
    scalars[4] = sh[2][0] + s7;
    scalars[5] = sh[3][0];
    if (packC) {  // the session's packC (sess_pack_c_kernel's layout; flag[0] = step finite, flag[1] = factorisation succeeded)
      packC[0] = (flag[0] && flag[1]) ? 0.0 : 1.0;
      packC[1] = sh[0][0];
      packC[2] = sh[1][0] + s6;
      packC[3] = sh[2][0] + s7;
      packC[4] = sh[3][0];
      packC[5] = s6;
      packC[6] = s7;
      packC[7] = 0.0;
    }
  }
}

}  // namespace
