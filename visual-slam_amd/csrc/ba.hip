// ba.hip -- K6 (reprojection residual + Jacobian blocks) and K7 (Schur-complement reduction) plus the
// Levenberg-Marquardt loop that drives them.
//
// Replaces visnav::bundle_adjustment (include/visnav/map_utils.h:337-421) and
// visnav::global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748), i.e. what
// ceres::Solve does with BundleAdjustmentReprojectionCostFunctor (include/visnav/reprojection.h:81-105),
// the four camera models (include/visnav/camera_models.h), LocalParameterizationSE3
// (include/visnav/local_parameterization_se3.hpp:43-63), HuberLoss and SPARSE_SCHUR.
//
// Differences in HOW (the WHAT is the same optimisation problem and the same LM policy, restated in
// oracle/orc_ba.cpp from [upstream] Ceres 2.0/2.1):
//   * derivatives are closed-form 2x6 / 2x3 blocks in the tangent space of T*exp(delta), not dual
//     numbers through the quaternion followed by the 7x6 plus-Jacobian;
//   * observations are stored sorted by landmark; every reduction runs in a fixed order (per-landmark
//     loops, per-camera tree reductions, thread-owned Schur entries), so a solve is bit-reproducible
//     from run to run -- no floating-point atomics on the small-system path;
//   * Schur complement, small systems (6*free cameras <= 128, the local-BA window): each of the
//     (6C)^2 entries of S is OWNED by one thread of a 1024-thread workgroup, which keeps it in a
//     register while the workgroup streams its share of the landmarks through LDS (per landmark:
//     W = F^T E and Y = W P^-1 per observation, plus a camera->observation slot table).  Per-workgroup
//     partial matrices are summed in a fixed order afterwards.
//   * large systems (global BA): one wavefront per landmark, fp64 hardware atomics into the dense S,
//     blocked dense Cholesky (chol.hip).
//
// Algorithmic bytes per LM iteration: n_obs*(16 + 8) + n_lms*24 + n_cams*56 + 128 in;
// (6C)^2*8 + 6C*8 + n_lms*96 out.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <numeric>
#include <thread>

#include "vsl_common.h"
#include "ba_device.h"
#include "ba_large.h"

namespace {

// ---------------------------------------------------------------------------------------- kernels
// K6.  One thread per observation (observations sorted by landmark).  Writes the robustified (and, when
// scale_c/scale_l are given, Jacobi-scaled) blocks and a per-workgroup cost partial.
__global__ __launch_bounds__(256) void ba_linearize_kernel(BaDims D, const double* __restrict__ poses,
                                                           const double* __restrict__ points,
                                                           const double* __restrict__ intr,
                                                           const int* __restrict__ cam_intr,
                                                           const int* __restrict__ cam_free,
                                                           const int* __restrict__ obs_cam,
                                                           const int* __restrict__ obs_lm,
                                                           const double* __restrict__ obs_uv,
                                                           const double* __restrict__ scale_c,
                                                           const double* __restrict__ scale_l, double* __restrict__ r_out,
                                                           double* __restrict__ F_out, double* __restrict__ E_out,
                                                           double* __restrict__ partials, int robust) {
  __shared__ double sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double c = 0;
  if (i < D.O) {
    const int cam = obs_cam[i], lm = obs_lm[i], k = cam_intr[cam];
    double r[2], F[12], E[6];
    residual_blocks(k ? D.model1 : D.model0, intr + 8 * k, poses + 7 * cam, points + 3 * lm, obs_uv + 2 * i, r, F, E,
                    true);
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho0 = s, rho1 = 1.0;
    if (D.use_huber) huber(s, D.huber, rho0, rho1);
    c = 0.5 * rho0;
    const double sr = robust ? sqrt(rho1) : 1.0;
    r_out[2 * (size_t)i] = r[0] * sr;
    r_out[2 * (size_t)i + 1] = r[1] * sr;
    const int fc = cam_free[cam];
    for (int j = 0; j < 6; j++) {
      const double sc = (scale_c && fc >= 0) ? scale_c[6 * fc + j] : 1.0;
      F_out[12 * (size_t)i + j] = F[j] * sr * sc;
      F_out[12 * (size_t)i + 6 + j] = F[6 + j] * sr * sc;
    }
    for (int j = 0; j < 3; j++) {
      const double sc = scale_l ? scale_l[3 * lm + j] : 1.0;
      E_out[6 * (size_t)i + j] = E[j] * sr * sc;
      E_out[6 * (size_t)i + 3 + j] = E[3 + j] * sr * sc;
    }
  }
  const double t = block_sum_256(c, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// cost only, at candidate parameters
__global__ __launch_bounds__(256) void ba_cost_kernel(BaDims D, const double* __restrict__ poses,
                                                      const double* __restrict__ points, const double* __restrict__ intr,
                                                      const int* __restrict__ cam_intr, const int* __restrict__ obs_cam,
                                                      const int* __restrict__ obs_lm, const double* __restrict__ obs_uv,
                                                      int o_first, int o_count, double* __restrict__ partials) {
  __shared__ double sh[256];
  const int t = blockIdx.x * 256 + threadIdx.x;
  double c = 0;
  if (t < o_count) {
    const int i = o_first + t;
    const int cam = obs_cam[i], lm = obs_lm[i], k = cam_intr[cam];
    double r[2];
    residual_blocks(k ? D.model1 : D.model0, intr + 8 * k, poses + 7 * cam, points + 3 * lm, obs_uv + 2 * i, r, nullptr,
                    nullptr, false);
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho0 = s, rho1 = 1.0;
    if (D.use_huber) huber(s, D.huber, rho0, rho1);
    c = 0.5 * rho0;
  }
  const double v = block_sum_256(c, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = v;
}

// scalars[slot] = sum / max of partials[0..n) in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void ba_reduce_kernel(const double* __restrict__ partials, int n,
                                                        double* __restrict__ scalars, int slot, int is_max) {
  __shared__ double sh[256];
  double v = 0;
  for (int i = threadIdx.x; i < n; i += 256) v = is_max ? fmax(v, partials[i]) : v + partials[i];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = is_max ? fmax(sh[threadIdx.x], sh[threadIdx.x + o]) : sh[threadIdx.x] + sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scalars[slot] = sh[0];
}

// per-landmark column statistics: squared column norms of E and E^T r
__global__ __launch_bounds__(256) void ba_lm_cols_kernel(BaDims D, const int* __restrict__ lm_start,
                                                         const double* __restrict__ r, const double* __restrict__ E,
                                                         double* __restrict__ n2l, double* __restrict__ grad_l) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= D.L) return;
  double n2[3] = {0, 0, 0}, g[3] = {0, 0, 0};
  for (int i = lm_start[l]; i < lm_start[l + 1]; i++) {
    const double* e = E + 6 * (size_t)i;
    const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
    for (int j = 0; j < 3; j++) {
      n2[j] += e[j] * e[j] + e[3 + j] * e[3 + j];
      g[j] += e[j] * r0 + e[3 + j] * r1;
    }
  }
  for (int j = 0; j < 3; j++) {
    n2l[3 * (size_t)l + j] = n2[j];
    grad_l[3 * (size_t)l + j] = g[j];
  }
}

// per-camera normal-equation block: H = sum F^T F (6x6) and g = sum F^T r over the camera's observations (camera
// CSR).  grid = (free cameras, segments): a workgroup sums every `segments`-th 256-observation slice of its camera
// (the local-BA window has ~11k observations per camera and only 12 free cameras: one workgroup per camera left
// the chip idle for 128 us behind a serial chain of dependent loads), all 27 accumulators go through ONE LDS tree
// (8 barriers, not 27 x 9), partials are summed per camera in segment order afterwards -- fixed order throughout.
__global__ __launch_bounds__(256) void ba_cam_block_kernel(const int* __restrict__ free_cams,
                                                           const int* __restrict__ cam_start,
                                                           const int* __restrict__ cam_obs, const double* __restrict__ r,
                                                           const double* __restrict__ F, double* __restrict__ part) {
  __shared__ double sh[27][256];
  const int fc = blockIdx.x, seg = blockIdx.y, nseg = gridDim.y;
  const int cam = free_cams[fc];
  double acc[27];
  for (int k = 0; k < 27; k++) acc[k] = 0;
  for (int q = cam_start[cam] + seg * 256 + threadIdx.x; q < cam_start[cam + 1]; q += 256 * nseg) {
    const int i = cam_obs[q];
    const double* f = F + 12 * (size_t)i;
    const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
    int k = 0;
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++) acc[k++] += f[a] * f[b] + f[6 + a] * f[6 + b];
    for (int a = 0; a < 6; a++) acc[21 + a] += f[a] * r0 + f[6 + a] * r1;
  }
#pragma unroll
  for (int k = 0; k < 27; k++) sh[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int k = 0; k < 27; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 27) part[((size_t)fc * nseg + seg) * 27 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void ba_cam_block_finish_kernel(int nfree, int nseg, const double* __restrict__ part, double* __restrict__ H,
                                           double* __restrict__ g) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nfree * 27) return;
  const int fc = t / 27, k = t - fc * 27;
  double v = 0;
  for (int sgm = 0; sgm < nseg; sgm++) v += part[((size_t)fc * nseg + sgm) * 27 + k];
  if (k >= 21) {
    g[6 * (size_t)fc + (k - 21)] = v;
  } else {
    int a = 0, rem = k;  // k-th entry of the upper triangle, row-major
    while (rem >= 6 - a) {
      rem -= 6 - a;
      a++;
    }
    const int bcol = a + rem;
    H[36 * (size_t)fc + 6 * a + bcol] = v;
    H[36 * (size_t)fc + 6 * bcol + a] = v;
  }
}

// Jacobi scaling (computed once): scale = 1 / (1 + sqrt(column norm^2)); for cameras from diag(H)
__global__ void ba_make_scale_kernel(int nfree, int L, const double* __restrict__ H, const double* __restrict__ n2l,
                                     double* __restrict__ scale_c, double* __restrict__ scale_l) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nfree) scale_c[i] = 1.0 / (1.0 + sqrt(H[36 * (size_t)(i / 6) + 7 * (i % 6)]));
  if (i < 3 * L) scale_l[i] = 1.0 / (1.0 + sqrt(n2l[i]));
}

__global__ void ba_apply_scale_kernel(int O, const int* __restrict__ cam_free, const int* __restrict__ obs_cam,
                                      const int* __restrict__ obs_lm, const double* __restrict__ scale_c,
                                      const double* __restrict__ scale_l, double* __restrict__ F, double* __restrict__ E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O) return;
  const int fc = cam_free[obs_cam[i]], lm = obs_lm[i];
  if (fc >= 0)
    for (int j = 0; j < 6; j++) {
      F[12 * (size_t)i + j] *= scale_c[6 * fc + j];
      F[12 * (size_t)i + 6 + j] *= scale_c[6 * fc + j];
    }
  for (int j = 0; j < 3; j++) {
    E[6 * (size_t)i + j] *= scale_l[3 * lm + j];
    E[6 * (size_t)i + 3 + j] *= scale_l[3 * lm + j];
  }
}

// gradient max-norm of the UNSCALED problem: |g_scaled / scale|, and LM diagonals
//   diag = clamp(col norm^2, 1e-6, 1e32) (kept when reuse != 0), D2 = diag / radius
__global__ void ba_diag_kernel(int nfree, int L, const double* __restrict__ H, const double* __restrict__ n2l,
                               const double* __restrict__ g_c, const double* __restrict__ grad_l,
                               const double* __restrict__ scale_c, const double* __restrict__ scale_l,
                               double* __restrict__ diag_c, double* __restrict__ diag_l, double* __restrict__ gabs) {
  // gabs[blockIdx.x] = the workgroup's maximum (one 45k-entry max by a single workgroup afterwards cost ~40 us)
  __shared__ double sh[256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nc = 6 * nfree, nl = 3 * L;
  double m = 0;
  if (i < nc) {
    diag_c[i] = fmin(fmax(H[36 * (size_t)(i / 6) + 7 * (i % 6)], 1e-6), 1e32);
    m = fabs(g_c[i] / scale_c[i]);
  }
  if (i < nl) {
    diag_l[i] = fmin(fmax(n2l[i], 1e-6), 1e32);
    m = fmax(m, fabs(grad_l[i] / scale_l[i]));
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) gabs[blockIdx.x] = sh[0];
}

// K7, small systems.  See the file header.  LB landmarks are staged per barrier pair.
#define SCH_THREADS 1024
#ifndef SCH_LB
#define SCH_LB 16
#endif
#ifndef SCH_GMAX
#define SCH_GMAX 256
#endif
#define SCH_KMAX 24   // observations of one landmark that hit FREE cameras (<= free cameras <= 21)
#define SCH_EPT 16    // owned entries per thread: n <= 128
#define SCH_CMAX 22   // free cameras

// BLOCK3 = false: every thread owns up to SCH_EPT single entries of the full n x n matrix (the first formulation;
//   6 LDS doubles per 3 multiply-adds, LDS-bandwidth-bound; kept as the cross-check of the other one).
// BLOCK3 = true (default): a thread owns ONE 3 x 3 sub-block (bi, bj), bi >= bj, of the LOWER block triangle --
//   n = 6 * cameras is a multiple of 3, nb = n / 3 <= 42, nb (nb + 1) / 2 <= 903 <= SCH_THREADS sub-blocks, so threads
//   beyond that count own nothing (own = false guards every use) -- 18 LDS doubles per 27 fused multiply-adds; the
//   finish kernel mirrors the strictly-upper sub-blocks (S is symmetric).  A first attempt at this variant in round 1
//   faulted on the GPU and was reverted uncommitted (its source is lost); this one was written against the extents
//   listed in DESIGN.md section 5 (the round-1 Schur fault) and is tested at n = 72, 108 and 126.
template <bool BLOCK3>
__global__ __launch_bounds__(SCH_THREADS) void ba_schur_small_kernel(
    BaDims D, const int* __restrict__ lm_start, const int* __restrict__ obs_cam, const int* __restrict__ cam_free,
    const double* __restrict__ r, const double* __restrict__ F, const double* __restrict__ E,
    const double* __restrict__ diag_l, double inv_radius, int l_first, int l_count, int lm_per_wg,
    double* __restrict__ S_part, double* __restrict__ rhs_part, double* __restrict__ Pinv_out,
    double* __restrict__ bl_out) {
  __shared__ double W[SCH_LB][SCH_KMAX][18];
  __shared__ double Y[SCH_LB][SCH_KMAX][18];
  __shared__ double Pi_s[SCH_LB][9];
  __shared__ double b_s[SCH_LB][3];
  __shared__ int slot[SCH_LB][SCH_CMAX];
  __shared__ int kcnt[SCH_LB];
  __shared__ int obs_of[SCH_LB][SCH_KMAX];  // observation index of the k-th free-camera observation
  __shared__ int ok_s[SCH_LB];
  const int n = D.n, tid = threadIdx.x;
  const int wl0 = l_first + blockIdx.x * lm_per_wg;
  const int wl1 = min(l_first + l_count, wl0 + lm_per_wg);
  double acc[BLOCK3 ? 9 : SCH_EPT];
  int ei[BLOCK3 ? 1 : SCH_EPT];  // packed (c1, x, c2, y) of the owned entries
  // BLOCK3: sub-block (bi, bj) of thread tid = bi (bi + 1) / 2 + bj, bj <= bi < nb
  int bi = 0, bj = 0;
  bool own = false;
  if (BLOCK3) {
    const int nb = n / 3;
    bi = (int)((sqrt(8.0 * tid + 1.0) - 1.0) * 0.5);
    while (bi * (bi + 1) / 2 > tid) bi--;
    while ((bi + 1) * (bi + 2) / 2 <= tid) bi++;
    bj = tid - bi * (bi + 1) / 2;
    own = bi < nb;  // bj <= bi by construction
#pragma unroll
    for (int e = 0; e < 9; e++) acc[e] = 0;
  } else {
#pragma unroll
    for (int e = 0; e < SCH_EPT; e++) {
      acc[e] = 0;
      const int idx = tid + e * SCH_THREADS;
      if (idx < n * n) {
        const int i = idx / n, j = idx - i * n;
        ei[e] = (i / 6) | ((i % 6) << 8) | ((j / 6) << 16) | ((j % 6) << 24);
      } else {
        ei[e] = -1;
      }
    }
  }
  double racc = 0;  // thread tid < n owns rhs[tid]
  for (int s0 = wl0; s0 < wl1; s0 += SCH_LB) {
    const int nl = min(SCH_LB, wl1 - s0);
    // phase A: per landmark P, b, P^-1 and the slot table -- one WAVE per staged landmark: lanes take
    // the observations (a landmark of the small-system path has at most 64), xor-shuffle tree sums
    // (fixed order), the free-camera rank of an observation from a ballot prefix
    {
      const int wv = tid >> 6, ln = tid & 63;
      if (wv < nl) {
        const int l = s0 + wv;
        const int o0 = lm_start[l], o1 = lm_start[l + 1];
        const int i = o0 + ln;
        const bool have = i < o1;
        double P6[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};  // P upper: 00 01 02 11 12 22
        int fc = -1;
        if (have) {
          const double* e = E + 6 * (size_t)i;
          const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
          P6[0] = e[0] * e[0] + e[3] * e[3];
          P6[1] = e[0] * e[1] + e[3] * e[4];
          P6[2] = e[0] * e[2] + e[3] * e[5];
          P6[3] = e[1] * e[1] + e[4] * e[4];
          P6[4] = e[1] * e[2] + e[4] * e[5];
          P6[5] = e[2] * e[2] + e[5] * e[5];
          for (int x = 0; x < 3; x++) bb[x] = e[x] * r0 + e[3 + x] * r1;
          fc = cam_free[obs_cam[i]];
        }
        // landmarks with more than 64 observations cannot occur here (one observation per camera,
        // at most SCH_CMAX free + the fixed cameras); extra observations are folded in serially anyway
        for (int j = o0 + 64 + ln; j < o1; j += 64) {
          const double* e = E + 6 * (size_t)j;
          const double r0 = r[2 * (size_t)j], r1 = r[2 * (size_t)j + 1];
          P6[0] += e[0] * e[0] + e[3] * e[3];
          P6[1] += e[0] * e[1] + e[3] * e[4];
          P6[2] += e[0] * e[2] + e[3] * e[5];
          P6[3] += e[1] * e[1] + e[4] * e[4];
          P6[4] += e[1] * e[2] + e[4] * e[5];
          P6[5] += e[2] * e[2] + e[5] * e[5];
          for (int x = 0; x < 3; x++) bb[x] += e[x] * r0 + e[3 + x] * r1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
          for (int q = 0; q < 6; q++) P6[q] += __shfl_xor(P6[q], o);
#pragma unroll
          for (int q = 0; q < 3; q++) bb[q] += __shfl_xor(bb[q], o);
        }
        double P[9] = {P6[0], P6[1], P6[2], P6[1], P6[3], P6[4], P6[2], P6[4], P6[5]};
        if (diag_l) {
          P[0] += diag_l[3 * (size_t)l] * inv_radius;
          P[4] += diag_l[3 * (size_t)l + 1] * inv_radius;
          P[8] += diag_l[3 * (size_t)l + 2] * inv_radius;
        }
        if (ln < SCH_CMAX) slot[wv][ln] = -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long fm = __ballot(have && fc >= 0);
        const int rank = __popcll(fm & ((1ull << ln) - 1ull));
        if (have && fc >= 0 && rank < SCH_KMAX) {
          slot[wv][fc] = rank;
          obs_of[wv][rank] = i;
        }
        const int k = min(__popcll(fm), SCH_KMAX);
        double Pi[9];
        const bool ok = o1 > o0 && inv3(P, Pi);
        if (ln == 0) {
          ok_s[wv] = ok;
          kcnt[wv] = ok ? k : 0;
          for (int q = 0; q < 9; q++) Pi_s[wv][q] = ok ? Pi[q] : 0.0;
          for (int q = 0; q < 3; q++) b_s[wv][q] = ok ? bb[q] : 0.0;
          if (Pinv_out)
            for (int q = 0; q < 9; q++) Pinv_out[9 * (size_t)l + q] = ok ? Pi[q] : 0.0;
          if (bl_out)
            for (int q = 0; q < 3; q++) bl_out[3 * (size_t)l + q] = ok ? bb[q] : 0.0;
        }
      }
    }
    __syncthreads();
    // phase B: W = F^T E (6x3) and Y = W P^-1 for every free-camera observation of the staged landmarks
    for (int item = tid; item < nl * SCH_KMAX * 6; item += SCH_THREADS) {
      const int li = item / (SCH_KMAX * 6), rem = item - li * (SCH_KMAX * 6);
      const int q = rem / 6, x = rem - q * 6;
      if (q >= kcnt[li]) continue;
      const int i = obs_of[li][q];
      const double* f = F + 12 * (size_t)i;
      const double* e = E + 6 * (size_t)i;
      double w[3];
      for (int y = 0; y < 3; y++) w[y] = f[x] * e[y] + f[6 + x] * e[3 + y];
      const double* Pi = Pi_s[li];
      for (int y = 0; y < 3; y++) {
        W[li][q][3 * x + y] = w[y];
        Y[li][q][3 * x + y] = w[0] * Pi[y] + w[1] * Pi[3 + y] + w[2] * Pi[6 + y];
      }
    }
    __syncthreads();
    // phase C: owned entries
    for (int li = 0; li < nl; li++) {
      if (!ok_s[li]) continue;
      if (BLOCK3) {
        if (own) {
          // cameras bi / 2 and bj / 2 (< nfree <= SCH_CMAX); rows 3 (bi % 2) .. +2 of Y = 9 contiguous doubles
          const int q1 = slot[li][bi >> 1], q2 = slot[li][bj >> 1];
          if (q1 >= 0 && q2 >= 0) {
            const double* y1 = &Y[li][q1][9 * (bi & 1)];
            const double* w2 = &W[li][q2][9 * (bj & 1)];
            double yv[9], wv[9];
#pragma unroll
            for (int t = 0; t < 9; t++) {
              yv[t] = y1[t];
              wv[t] = w2[t];
            }
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
              for (int b = 0; b < 3; b++)
                acc[3 * a + b] = fma(-yv[3 * a + 2], wv[3 * b + 2], fma(-yv[3 * a + 1], wv[3 * b + 1], fma(-yv[3 * a], wv[3 * b], acc[3 * a + b])));
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < SCH_EPT; e++) {
          const int pk = ei[e];
          if (pk < 0) continue;
          const int q1 = slot[li][pk & 0xFF], q2 = slot[li][(pk >> 16) & 0xFF];
          if (q1 < 0 || q2 < 0) continue;
          const double* y1 = &Y[li][q1][3 * ((pk >> 8) & 0xFF)];
          const double* w2 = &W[li][q2][3 * ((pk >> 24) & 0xFF)];
          acc[e] -= y1[0] * w2[0] + y1[1] * w2[1] + y1[2] * w2[2];
        }
      }
      if (tid < n) {
        const int q1 = slot[li][tid / 6];
        if (q1 >= 0) {
          const double* y1 = &Y[li][q1][3 * (tid % 6)];
          racc -= y1[0] * b_s[li][0] + y1[1] * b_s[li][1] + y1[2] * b_s[li][2];
        }
      }
    }
    __syncthreads();
  }
  if (BLOCK3) {
    if (own) {  // rows 3 bi .. 3 bi + 2 < n, columns 3 bj .. 3 bj + 2 < n
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) S_part[(size_t)blockIdx.x * n * n + (size_t)(3 * bi + a) * n + (3 * bj + b)] = acc[3 * a + b];
    }
  } else {
#pragma unroll
    for (int e = 0; e < SCH_EPT; e++) {
      const int idx = tid + e * SCH_THREADS;
      if (idx < n * n) S_part[(size_t)blockIdx.x * n * n + idx] = acc[e];
    }
  }
  if (tid < n) rhs_part[(size_t)blockIdx.x * n + tid] = racc;
}

// S = sum_g S_part[g] + blockdiag(H) + diag(D2_c);  rhs = sum_g rhs_part[g] + g_c.  A workgroup owns 16 consecutive
// entries; 16 threads per entry each sum a sixteenth of the partials, the sixteen sub-sums are added in order (fixed
// order, 16-deep dependent load chains instead of G-deep ones: 114 -> ~12 us for 256 partials of a 72 x 72 system).
__global__ __launch_bounds__(256) void ba_schur_finish_kernel(int n, int G, const double* __restrict__ S_part,
                                                              const double* __restrict__ rhs_part, const double* __restrict__ H,
                                                              const double* __restrict__ g_c, const double* __restrict__ diag_c,
                                                              double inv_radius, double* __restrict__ S, double* __restrict__ rhs,
                                                              int lower_blocks) {
  __shared__ double sh[16][17];
  const int e = threadIdx.x & 15, c = threadIdx.x >> 4;
  const int total = n * n + n;  // the n*n entries of S, then the n entries of rhs
  const int idx = blockIdx.x * 16 + e;
  const int per = (G + 15) / 16;
  double v = 0;
  if (idx < total) {
    const bool is_rhs = idx >= n * n;
    int sidx = idx;
    if (!is_rhs && lower_blocks) {  // the partials hold the lower triangle of 3 x 3 sub-blocks: mirror the rest
      const int i = idx / n, j = idx - i * n;
      if (i / 3 < j / 3) sidx = j * n + i;
    }
    const double* src = is_rhs ? rhs_part + (idx - n * n) : S_part + sidx;
    const size_t stride = is_rhs ? (size_t)n : (size_t)n * n;
    for (int g = c * per; g < min(G, (c + 1) * per); g++) v += src[(size_t)g * stride];
  }
  sh[c][e] = v;
  __syncthreads();
  if (c == 0 && idx < total) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) t += sh[k][e];
    if (idx < n * n) {
      const int i = idx / n, j = idx - i * n;
      if (i / 6 == j / 6) t += H[36 * (size_t)(i / 6) + 6 * (i % 6) + (j % 6)];
      if (i == j && diag_c) t += diag_c[i] * inv_radius;
      S[idx] = t;
    } else {
      rhs[idx - n * n] = t + g_c[idx - n * n];
    }
  }
}

// K7, large systems: one wavefront per landmark, fp64 hardware atomics into the dense S.
__global__ __launch_bounds__(256) void ba_schur_atomic_kernel(BaDims D, const int* __restrict__ lm_start,
                                                              const int* __restrict__ obs_cam,
                                                              const int* __restrict__ cam_free, const double* __restrict__ r,
                                                              const double* __restrict__ F, const double* __restrict__ E,
                                                              const double* __restrict__ diag_l, double inv_radius,
                                                              int l_first, int l_count, double* __restrict__ S,
                                                              double* __restrict__ rhs, double* __restrict__ Pinv_out,
                                                              double* __restrict__ bl_out, int lower_only, int ldS) {
  // S is addressed as S[row * ldS + col]: ldS = n for the dense matrix; band storage (chol.hip "BAND FORM") passes
  // storage + bws and ldS = bws together with lower_only = 2: only entries with col <= row exist there
  constexpr int KM = 64;
  __shared__ double Ws[4][KM][18];
  __shared__ double Ys[4][KM][18];
  __shared__ int cs[4][KM];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l = l_first + blockIdx.x * 4 + wave;
  if (l >= l_first + l_count) return;
  const int a = lm_start[l], b = lm_start[l + 1];
  if (a == b) return;
  // P and b: lanes stride the observations, xor-shuffle tree reduction (fixed order)
  double P[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};  // P upper: 00 01 02 11 12 22
  for (int i = a + lane; i < b; i += 64) {
    const double* e = E + 6 * (size_t)i;
    const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
    P[0] += e[0] * e[0] + e[3] * e[3];
    P[1] += e[0] * e[1] + e[3] * e[4];
    P[2] += e[0] * e[2] + e[3] * e[5];
    P[3] += e[1] * e[1] + e[4] * e[4];
    P[4] += e[1] * e[2] + e[4] * e[5];
    P[5] += e[2] * e[2] + e[5] * e[5];
    for (int x = 0; x < 3; x++) bb[x] += e[x] * r0 + e[3 + x] * r1;
  }
  for (int o = 32; o > 0; o >>= 1) {
    for (int q = 0; q < 6; q++) P[q] += __shfl_xor(P[q], o);
    for (int q = 0; q < 3; q++) bb[q] += __shfl_xor(bb[q], o);
  }
  double Pf[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
  if (diag_l) {
    Pf[0] += diag_l[3 * (size_t)l] * inv_radius;
    Pf[4] += diag_l[3 * (size_t)l + 1] * inv_radius;
    Pf[8] += diag_l[3 * (size_t)l + 2] * inv_radius;
  }
  double Pi[9];
  const bool ok = inv3(Pf, Pi);
  if (lane == 0) {
    if (Pinv_out)
      for (int q = 0; q < 9; q++) Pinv_out[9 * (size_t)l + q] = ok ? Pi[q] : 0.0;
    if (bl_out)
      for (int q = 0; q < 3; q++) bl_out[3 * (size_t)l + q] = ok ? bb[q] : 0.0;
  }
  if (!ok) return;
  // observations in chunks of KM free-camera hits (a landmark seen by more than KM free cameras is
  // processed chunk x chunk)
  for (int qa = a; qa < b; qa += KM) {
    const int na = min(KM, b - qa);
    for (int qb = a; qb < b; qb += KM) {
      const int nb = min(KM, b - qb);
      // stage Y of chunk a and W of chunk b
      for (int item = lane; item < na * 6; item += 64) {
        const int q = item / 6, x = item - q * 6, i = qa + q;
        const double* f = F + 12 * (size_t)i;
        const double* e = E + 6 * (size_t)i;
        double w[3];
        for (int y = 0; y < 3; y++) w[y] = f[x] * e[y] + f[6 + x] * e[3 + y];
        for (int y = 0; y < 3; y++) Ys[wave][q][3 * x + y] = w[0] * Pi[y] + w[1] * Pi[3 + y] + w[2] * Pi[6 + y];
        if (x == 0) cs[wave][q] = cam_free[obs_cam[i]];
      }
      for (int item = lane; item < nb * 6; item += 64) {
        const int q = item / 6, x = item - q * 6, i = qb + q;
        const double* f = F + 12 * (size_t)i;
        const double* e = E + 6 * (size_t)i;
        for (int y = 0; y < 3; y++) Ws[wave][q][3 * x + y] = f[x] * e[y] + f[6 + x] * e[3 + y];
      }
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      for (int item = lane; item < na * nb * 36; item += 64) {
        const int q1 = item / (nb * 36), rem = item - q1 * (nb * 36);
        const int q2 = rem / 36, xy = rem - q2 * 36, x = xy / 6, y = xy - x * 6;
        const int c1 = cs[wave][q1], c2 = cam_free[obs_cam[qb + q2]];
        // lower_only: the Cholesky solve reads only the lower triangle -- blocks above the block diagonal
        // are not accumulated (half the atomics); the diagonal blocks stay complete
        if (c1 < 0 || c2 < 0 || (lower_only && c1 < c2) || (lower_only == 2 && c1 == c2 && x < y)) continue;
        const double* y1 = &Ys[wave][q1][3 * x];
        const double* w2 = &Ws[wave][q2][3 * y];
        unsafeAtomicAdd(&S[(size_t)(6 * c1 + x) * ldS + 6 * c2 + y], -(y1[0] * w2[0] + y1[1] * w2[1] + y1[2] * w2[2]));
      }
      if (qb == a) {
        for (int item = lane; item < na * 6; item += 64) {
          const int q = item / 6, x = item - q * 6;
          const int c1 = cs[wave][q];
          if (c1 < 0) continue;
          const double* y1 = &Ys[wave][q][3 * x];
          unsafeAtomicAdd(&rhs[6 * c1 + x], -(y1[0] * bb[0] + y1[1] * bb[1] + y1[2] * bb[2]));
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- K7, large systems, gather form (the default).  The block structure of the reduced camera system is fixed over
// the LM iterations of a solve: block (c1, c2), c1 >= c2, receives one term -Y_i W_j^T per landmark seen by both
// cameras (i, j = that landmark's observations by c1 and c2).  The (i, j) pairs are listed per block ONCE per solve (count / scan / fill, below); every
// iteration a per-landmark kernel writes W_i = F_i^T E_i and Y_i = W_i P^-1 for every observation, and a per-block
// kernel sums its list -- every entry of S is written once, by one lane, no atomics (157 M fp64 atomics per iteration at
// 1000 cameras / 881k observations took 2.5 ms).
// Slot of block (c1, c2): band form c1 * (hb + 1) + (c1 - c2) with hb = the half bandwidth in cameras; otherwise the
// packed lower triangle c1 (c1 + 1) / 2 + c2.
__device__ __forceinline__ int ba_pair_slot(int c1, int c2, int hbp1) {
  return hbp1 > 0 ? c1 * hbp1 + (c1 - c2) : c1 * (c1 + 1) / 2 + c2;
}

// FILL = false: cnt[slot] += 1 per pair; FILL = true: pairs[start[slot] + cursor[slot]++] = (i, j)
template <bool FILL>
__global__ void ba_pair_list_kernel(int l_first, int l_count, const int* __restrict__ lm_start, const int* __restrict__ obs_cam,
                                    const int* __restrict__ cam_free, const int* __restrict__ cam_pos, int hbp1, int nfree_cyclic,
                                    int* __restrict__ cnt, const int* __restrict__ start, int* __restrict__ pairs) {
  const int l = l_first + blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= l_first + l_count) return;
  const int a = lm_start[l], b = lm_start[l + 1];
  for (int i = a; i < b; i++) {
    const int ci = cam_free[obs_cam[i]];
    if (ci < 0) continue;
    for (int j = a; j < b; j++) {
      const int cj = cam_free[obs_cam[j]];
      if (cj < 0 || cj > ci) continue;  // (two observations of one landmark by the SAME camera: both orders land in the diagonal block)
      int slot, pi = i, pj = j;
      if (nfree_cyclic > 0 && ci - cj >= hbp1) {
        // cyclic band: the two cameras are neighbours AROUND the loop -- the block is (cj, ci - nfree), stored in row cj
        // (the smaller index) at distance cj + nfree - ci, and the roles of the two observations swap with it
        slot = cj * hbp1 + (cj + nfree_cyclic - ci);
        pi = j;
        pj = i;
      } else {
        slot = ba_pair_slot(ci, cj, hbp1);
      }
      const int k = atomicAdd(&cnt[slot], 1);
      if (FILL) {
        const size_t pos = (size_t)start[slot] + k;
        pairs[2 * pos] = cam_pos[pi];  // where the block of the observation lives (camera-major order)
        pairs[2 * pos + 1] = cam_pos[pj];
      }
    }
  }
}

// exclusive prefix sum of cnt[0 .. n) into start[0 .. n], one workgroup
__global__ __launch_bounds__(1024) void ba_pair_scan_kernel(int n, const int* __restrict__ cnt, int* __restrict__ start) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int idx = base + tid;
    const int v = idx < n ? cnt[idx] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wave; w++) off += wsum[w];
    if (idx < n) start[idx] = off + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = off + x;
    __syncthreads();
  }
  if (tid == 0) start[n] = carry_s;
}

// the fill order of a slot's list depends on the scheduling of the atomics above; sorting every list by its
// (i, j) keys once makes the summation order of the gather -- and with it the whole large-system solve -- reproducible
// run to run.  One wavefront per slot, rank sort in LDS (lists hold a few hundred pairs; beyond PAIR_SORT_MAX a list
// stays in fill order).
#define PAIR_SORT_MAX 2048
__global__ __launch_bounds__(64) void ba_pair_sort_kernel(const int* __restrict__ start, int* __restrict__ pairs) {
  __shared__ unsigned long long key[PAIR_SORT_MAX];
  const int slot = blockIdx.x, lane = threadIdx.x;
  const int p0 = start[slot], n = start[slot + 1] - p0;
  if (n < 2 || n > PAIR_SORT_MAX) return;
  for (int t = lane; t < n; t += 64) {
    const int2 ij = *(const int2*)(pairs + 2 * (size_t)(p0 + t));
    key[t] = ((unsigned long long)(unsigned)ij.x << 32) | (unsigned)ij.y;
  }
  __syncthreads();
  for (int t = lane; t < n; t += 64) {
    const unsigned long long mine = key[t];
    int rank = 0;
    for (int u = 0; u < n; u++) rank += key[u] < mine;  // keys are distinct: (i, j) occurs once
    int2 ij;
    ij.x = (int)(mine >> 32);
    ij.y = (int)(mine & 0xffffffffu);
    *(int2*)(pairs + 2 * (size_t)(p0 + rank)) = ij;
  }
}

// rhs of the reduced system, gather form: rhs[6 fc + x] = -sum over the observations of free camera fc (contiguous in
// camera-major order) whose landmark lies in [l_first, l_first + l_count) of Y_i[x] . b_l -- one wavefront per camera,
// lanes stride the observations, xor-tree (fixed order).  ba_add_cam_blocks_kernel adds g_c afterwards.
__global__ __launch_bounds__(64) void ba_schur_rhs_kernel(int nfree, const int* __restrict__ free_cams,
                                                          const int* __restrict__ cam_start, const int* __restrict__ cam_obs,
                                                          const int* __restrict__ obs_lm, const double* __restrict__ Yg,
                                                          const double* __restrict__ bl, int l_first, int l_count,
                                                          double* __restrict__ rhs) {
  const int fc = blockIdx.x, lane = threadIdx.x;
  const int c = free_cams[fc];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int k = cam_start[c] + lane; k < cam_start[c + 1]; k += 64) {
    const int l = obs_lm[cam_obs[k]];
    if (l < l_first || l >= l_first + l_count) continue;
    const double* y = Yg + 18 * (size_t)k;
    const double b0 = bl[3 * (size_t)l], b1 = bl[3 * (size_t)l + 1], b2 = bl[3 * (size_t)l + 2];
#pragma unroll
    for (int x = 0; x < 6; x++) acc[x] += y[3 * x] * b0 + y[3 * x + 1] * b1 + y[3 * x + 2] * b2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int x = 0; x < 6; x++) acc[x] += __shfl_xor(acc[x], o);
  if (lane < 6) rhs[6 * fc + lane] = -acc[lane];
}

// per landmark (one wavefront): P, b, P^-1 (kept for the back-substitution), and for every observation W and Y = W P^-1
// (Y = 0 for a landmark whose P is singular: it contributes nothing)
__global__ __launch_bounds__(256) void ba_schur_prep_kernel(BaDims D, const int* __restrict__ lm_start,
                                                            const int* __restrict__ obs_cam, const int* __restrict__ cam_free,
                                                            const int* __restrict__ cam_pos, const double* __restrict__ r,
                                                            const double* __restrict__ F, const double* __restrict__ E,
                                                            const double* __restrict__ diag_l, double inv_radius,
                                                            int l_first, int l_count, double* __restrict__ Wg,
                                                            double* __restrict__ Yg,
                                                            double* __restrict__ rhs, double* __restrict__ Pinv_out,
                                                            double* __restrict__ bl_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l = l_first + blockIdx.x * 4 + wave;
  if (l >= l_first + l_count) return;
  const int a = lm_start[l], b = lm_start[l + 1];
  if (a == b) return;
  double P[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};  // P upper: 00 01 02 11 12 22
  for (int i = a + lane; i < b; i += 64) {
    const double* e = E + 6 * (size_t)i;
    const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
    P[0] += e[0] * e[0] + e[3] * e[3];
    P[1] += e[0] * e[1] + e[3] * e[4];
    P[2] += e[0] * e[2] + e[3] * e[5];
    P[3] += e[1] * e[1] + e[4] * e[4];
    P[4] += e[1] * e[2] + e[4] * e[5];
    P[5] += e[2] * e[2] + e[5] * e[5];
    for (int x = 0; x < 3; x++) bb[x] += e[x] * r0 + e[3 + x] * r1;
  }
  for (int o = 32; o > 0; o >>= 1) {
    for (int q = 0; q < 6; q++) P[q] += __shfl_xor(P[q], o);
    for (int q = 0; q < 3; q++) bb[q] += __shfl_xor(bb[q], o);
  }
  double Pf[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
  if (diag_l) {
    Pf[0] += diag_l[3 * (size_t)l] * inv_radius;
    Pf[4] += diag_l[3 * (size_t)l + 1] * inv_radius;
    Pf[8] += diag_l[3 * (size_t)l + 2] * inv_radius;
  }
  double Pi[9];
  const bool ok = inv3(Pf, Pi);
  if (lane == 0) {
    if (Pinv_out)
      for (int q = 0; q < 9; q++) Pinv_out[9 * (size_t)l + q] = ok ? Pi[q] : 0.0;
    if (bl_out)
      for (int q = 0; q < 3; q++) bl_out[3 * (size_t)l + q] = ok ? bb[q] : 0.0;
  }
  for (int item = lane; item < (b - a) * 6; item += 64) {
    const int q = item / 6, x = item - q * 6, i = a + q;
    const int c = cam_free[obs_cam[i]];
    if (c < 0) continue;
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    double w[3], y[3];
    for (int z = 0; z < 3; z++) w[z] = f[x] * e[z] + f[6 + x] * e[3 + z];
    for (int z = 0; z < 3; z++) y[z] = ok ? w[0] * Pi[z] + w[1] * Pi[3 + z] + w[2] * Pi[6 + z] : 0.0;
    const size_t at = 18 * (size_t)cam_pos[i] + 3 * x;
    for (int z = 0; z < 3; z++) {
      Wg[at + z] = w[z];
      Yg[at + z] = y[z];
    }
  }
}

// per block slot (one wavefront): S_block = -sum over the slot's pairs of Y_i W_j^T, written once.  A lane takes every
// 64th pair of the list -- both 144-byte blocks with 16-byte loads, all 36 products -- so 64 pairs' loads are in flight
// at once (one pair per iteration with the 36 entries over the lanes was bound by the latency of the dependent loads
// pair -> W / Y: 0.85 ms per iteration at 4.4 M pairs); the 36 x 64 partial sums are folded through LDS in lane order.
// (Sessions in the recompute form pass ONE array for both operands: the block Z = F^T E chol(P^-1) of ba_large.h, with
// Y_i W_j^T = Z_i Z_j^T.)
// The blocks of a round's 64 pairs are FETCHED COOPERATIVELY (round 4): the 128 blocks are 1152 pieces of 16 bytes,
// piece m = 64 t + lane belongs to block m / 9, so nine consecutive lanes read one contiguous 144-byte block and an
// instruction touches ~10 cache lines; they land in LDS at double2[m] (linear, conflict-free) and every lane reads its
// own two blocks from there.  With a lane loading its own blocks every 16-byte load instruction touched 64 different
// lines, and the L1's one-line-per-cycle tag rate -- 18 instructions x 64 lines per round -- bounded the kernel.
// Band form keeps only x >= y of a diagonal block (lower_mode 2); lower_mode 1 = dense lower triangle (diagonal blocks
// complete), 0 = full matrix (the mirror block is written as well).
__global__ __launch_bounds__(64) void ba_schur_gather_kernel(int n_slots, int hbp1, const int* __restrict__ start,
                                                             const int* __restrict__ pairs, const double* __restrict__ Wg,
                                                             const double* __restrict__ Yg, double* __restrict__ S, int ldS,
                                                             int lower_mode) {
  __shared__ double red[36][65];  // (halved with a shuffle step first -- 12 instead of 8 wavefronts per compute unit -- the kernel was 9% SLOWER: more rows in flight than the L2 holds)
  int slot = blockIdx.x;
  if (hbp1 > 0) {
    // Band form, XCD-aware order (round 4): workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD gets a
    // CONTIGUOUS range of camera rows, walked in order.  The hbp1 slots of a row read the same Y blocks (the row camera's
    // observations) and the rows c .. c + hbp1 - 1 the same W blocks (camera c's): with slot = blockIdx.x those readers
    // sat on all eight XCDs, each L2 (4 MB) fetched every block for itself and the kernel ran at the fabric's ~5 TB/s
    // (1.3 GB per launch for 254 MB of distinct blocks).
    const int nrows = n_slots / hbp1, rpx = (nrows + 7) >> 3;
    const int k = (int)(blockIdx.x >> 3), row = (int)(blockIdx.x & 7) * rpx + k / hbp1;
    if (row >= nrows) return;
    slot = row * hbp1 + k % hbp1;
  }
  const int p0 = start[slot], p1 = start[slot + 1];
  if (p0 == p1) return;
  const int lane = threadIdx.x;
  double acc[36];
#pragma unroll
  for (int e = 0; e < 36; e++) acc[e] = 0.0;
  double2* stage = (double2*)&red[0][0];  // 1152 pieces of a round; the same LDS folds the partial sums afterwards
  static_assert(sizeof(red) >= 1152 * sizeof(double2), "the staging area must fit");
  for (int base = p0; base < p1; base += 64) {
    const int p = base + lane;
    const bool have = p < p1;
    int2 ij = make_int2(0, 0);  // (idle lanes fetch block 0: valid memory, never accumulated)
    if (have) ij = *(const int2*)(pairs + 2 * (size_t)p);
    double2 piece[18];
#pragma unroll
    for (int t = 0; t < 18; t++) {
      const int m = 64 * t + lane, b = m / 9, part = m - 9 * b;
      const int iy = __shfl(ij.x, b & 63), iw = __shfl(ij.y, b & 63);
      const double* src = b < 64 ? Yg + 18 * (size_t)iy : Wg + 18 * (size_t)iw;
      piece[t] = *(const double2*)(src + 2 * part);
    }
#pragma unroll
    for (int t = 0; t < 18; t++) stage[64 * t + lane] = piece[t];
    __syncthreads();
    if (have) {
      double yv[18], wv[18];
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const double2 a2 = stage[9 * lane + q], b2 = stage[9 * (64 + lane) + q];
        yv[2 * q] = a2.x;
        yv[2 * q + 1] = a2.y;
        wv[2 * q] = b2.x;
        wv[2 * q + 1] = b2.y;
      }
#pragma unroll
      for (int x = 0; x < 6; x++)
#pragma unroll
        for (int y = 0; y < 6; y++)
          acc[6 * x + y] += yv[3 * x] * wv[3 * y] + yv[3 * x + 1] * wv[3 * y + 1] + yv[3 * x + 2] * wv[3 * y + 2];
    }
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 36; e++) red[e][lane] = acc[e];
  __syncthreads();
  if (lane >= 36) return;
  const int x = lane / 6, y = lane - 6 * x;
  double tot = 0.0;
  const int nl = min(64, p1 - p0);
  for (int l = 0; l < nl; l++) tot += red[lane][l];
  const double v = -tot;
  int c1, c2;
  if (hbp1 > 0) {
    c1 = slot / hbp1;
    c2 = c1 - (slot - c1 * hbp1);
  } else {
    c1 = (int)((sqrt(8.0 * (double)slot + 1.0) - 1.0) * 0.5);
    while (c1 * (c1 + 1) / 2 > slot) c1--;
    while ((c1 + 1) * (c1 + 2) / 2 <= slot) c1++;
    c2 = slot - c1 * (c1 + 1) / 2;
  }
  if (!(lower_mode == 2 && c1 == c2 && x < y)) S[(size_t)(6 * c1 + x) * ldS + 6 * c2 + y] = v;
  if (lower_mode == 0 && c1 != c2) S[(size_t)(6 * c2 + y) * ldS + 6 * c1 + x] = v;
}

// S += blockdiag(H) + diag(D2); rhs += g_c   (large-system path, S pre-zeroed before the atomics)
__global__ void ba_add_cam_blocks_kernel(int nfree, const double* __restrict__ H, const double* __restrict__ g_c,
                                         const double* __restrict__ diag_c, double inv_radius, double* __restrict__ S,
                                         double* __restrict__ rhs, int ldS, int lower_elems) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = 6 * nfree;
  if (t < nfree * 36) {
    const int fc = t / 36, x = (t % 36) / 6, y = t % 6;
    double v = H[t];
    if (x == y && diag_c) v += diag_c[6 * fc + x] * inv_radius;
    if (!(lower_elems && y > x)) S[(size_t)(6 * fc + x) * ldS + 6 * fc + y] += v;
  }
  if (t < n) rhs[t] += g_c[t];
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {  // `lane` wave-uniform (a constant after unrolling)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// dense Cholesky solve in LDS, one workgroup; n <= 128.  dc = -(S^-1 rhs); flag = 0 on failure.
//
// Blocked right-looking factorisation, 8-column panels, THREE workgroup barriers per panel (the first version ran one
// pivot -> sqrt -> divide -> barrier round and a serial per-row trailing update per COLUMN: 124 us for 72 x 72):
//   panel:    a thread owns a row of the panel (8 values in registers).  Each wavefront factors the 8 x 8 diagonal
//             block redundantly in its lanes 0..7 (row per lane, column broadcasts by v_readlane: no LDS, no barrier),
//             then every row is solved against that factor with readlane broadcasts of its entries;
//   trailing: 4 x 4 register tiles of the lower triangle, 8-deep products from the panel rows in LDS.
// Rows are padded to an odd stride (column walks spread over the banks).  The substitutions run in one wavefront, two
// rows per lane, multiplying by the stored reciprocal pivots.
#define CH_NB 8
__global__ __launch_bounds__(256) void ba_chol_small_kernel(int n, const double* __restrict__ S, const double* __restrict__ rhs,
                                                            double* __restrict__ dc, int* __restrict__ ok_flag,
                                                            int* __restrict__ arm_flag = nullptr) {
  __shared__ double A[128 * 129 + 256];  // static: dynamic LDS above 64 KiB is refused by the runtime
  __shared__ int fail_s;
  const int ld = n | 1;
  double* bvec = A + 128 * 129;
  double* invd = bvec + 128;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < n * n; i += 256) {
    const int r = i / n, c = i - r * n;
    A[r * ld + c] = S[i];
  }
  for (int i = tid; i < n; i += 256) bvec[i] = rhs[i];
  if (tid == 0) fail_s = 0;
  __syncthreads();
#ifdef CHS_TIMING
  long long tp[6] = {0, 0, 0, 0, 0, 0}, tt = __builtin_amdgcn_s_memtime();
#define CHS_STAMP(q) { const long long t2 = __builtin_amdgcn_s_memtime(); tp[q] += t2 - tt; tt = t2; }
#else
#define CHS_STAMP(q)
#endif
  CHS_STAMP(0)
  for (int j0 = 0; j0 < n; j0 += CH_NB) {
    const int w = min(CH_NB, n - j0);  // a ragged last panel is completed with identity columns
    const int row = j0 + tid;
    const bool valid = tid < 128 && row < n;
    double a[CH_NB], dr[CH_NB];
    if (tid < 128) {
#pragma unroll
      for (int c = 0; c < CH_NB; c++) {
        a[c] = (valid && c < w) ? A[row * ld + j0 + c] : 0.0;
        dr[c] = (lane < w && c < w) ? A[(j0 + lane) * ld + j0 + c] : (lane == c ? 1.0 : 0.0);
      }
    }
    __syncthreads();  // the second wavefront has read the diagonal block before the first one overwrites it with L
    CHS_STAMP(1)
    if (tid < 128) {
      double inv[CH_NB], x[CH_NB];
      bool good = true;
#pragma unroll
      for (int c = 0; c < CH_NB; c++) {
        const double d = readlane_f64(dr[c], c);
        if (!(d > 0.0) || !isfinite(d)) good = false;  // wave-uniform
        // 1 / sqrt(d) by the hardware estimate and two Newton steps, sqrt(d) = d / sqrt(d): the IEEE sqrt and divide
        // expansions are ~100 dependent fp64 instructions each, and the pivots are a serial chain (measured in the band
        // solver of chol.hip: ~1500 cycles per pivot, i.e. most of this kernel's 60 us at n = 72)
        double iv = __builtin_amdgcn_rsq(d);
        iv = iv * (1.5 - 0.5 * d * iv * iv);
        iv = iv * (1.5 - 0.5 * d * iv * iv);
        inv[c] = iv;
        dr[c] = (lane == c) ? d * iv : dr[c] * iv;     // lanes > c: l(lane, c)
#pragma unroll
        for (int k = c + 1; k < CH_NB; k++) dr[k] -= dr[c] * readlane_f64(dr[c], k);  // meaningful for lanes >= k
      }
      // x L_d^T = a: the row of L in this panel (for a diagonal-block row this reproduces that row of L_d)
#pragma unroll
      for (int c = 0; c < CH_NB; c++) {
        double t = a[c];
#pragma unroll
        for (int k = 0; k < c; k++) t -= x[k] * readlane_f64(dr[k], c);
        x[c] = t * inv[c];
      }
      if (valid) {
#pragma unroll
        for (int c = 0; c < CH_NB; c++)
          if (c < w && tid >= c) A[row * ld + j0 + c] = x[c];
      }
      if (tid == 0) {
#pragma unroll
        for (int c = 0; c < CH_NB; c++)
          if (c < w) invd[j0 + c] = inv[c];
        if (!good) fail_s = 1;
      }
    }
    __syncthreads();
    CHS_STAMP(2)
    if (fail_s) break;  // workgroup-uniform
    const int r0 = j0 + CH_NB, rem = n - r0;
    if (rem > 0) {
      const int T = (rem + 3) >> 2, ntiles = T * (T + 1) / 2;
      for (int tile = tid; tile < ntiles; tile += 256) {
        int ti = (int)((__fsqrt_rn(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);  // fp32 estimate, corrected below (an fp64 sqrt costs more than the tile)
        while (ti * (ti + 1) / 2 > tile) ti--;
        while ((ti + 1) * (ti + 2) / 2 <= tile) ti++;
        const int tj = tile - ti * (ti + 1) / 2;
        const int ib = r0 + 4 * ti, kb = r0 + 4 * tj;
        double Li[4][CH_NB], Lk[4][CH_NB];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int ir = min(ib + q, n - 1), kr = min(kb + q, n - 1);
#pragma unroll
          for (int c = 0; c < CH_NB; c++) {
            Li[q][c] = A[ir * ld + j0 + c];
            Lk[q][c] = A[kr * ld + j0 + c];
          }
        }
#pragma unroll
        for (int qa = 0; qa < 4; qa++)
#pragma unroll
          for (int qb = 0; qb < 4; qb++) {
            const int i = ib + qa, k = kb + qb;
            if (i < n && k <= i) {
              double acc = A[i * ld + k];
#pragma unroll
              for (int c = 0; c < CH_NB; c++) acc = fma(-Li[qa][c], Lk[qb][c], acc);
              A[i * ld + k] = acc;
            }
          }
      }
    }
    __syncthreads();
    CHS_STAMP(3)
  }
  const bool good = fail_s == 0;
  if (good && tid < 64) {
    // substitutions by one wavefront, two rows per lane, no workgroup barriers: L y = b, then L^T x = y.  Panel by
    // panel: the eight factor entries a lane needs for a panel are read from LDS BEFORE the panel's eight serial steps
    // (with the read inside every step the chain was 144 x (LDS latency + broadcast): 52k of the kernel's 125k cycles)
    const int r0 = lane, r1 = lane + 64;
    double b0 = r0 < n ? bvec[r0] : 0.0, b1 = r1 < n ? bvec[r1] : 0.0;
    // the reciprocal pivots ride in registers too (lane l: pivots l and l + 64) and are broadcast like the solution
    // entries: an LDS read of invd[j] inside a step put the LDS latency back into the chain
    const double inv0 = r0 < n ? invd[r0] : 0.0, inv1 = r1 < n ? invd[r1] : 0.0;
    for (int j0 = 0; j0 < n; j0 += CH_NB) {
      double a0[CH_NB], a1[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; c++) {
        const int j = min(j0 + c, n - 1);
        a0[c] = (r0 > j && r0 < n) ? A[r0 * ld + j] : 0.0;
        a1[c] = (r1 > j && r1 < n) ? A[r1 * ld + j] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < CH_NB; c++) {
        const int j = j0 + c;
        if (j < n) {  // wave-uniform
          // j is wave-uniform: scalar broadcasts, not LDS permutes
          const double yj = readlane_f64(j < 64 ? b0 : b1, j & 63) * readlane_f64(j < 64 ? inv0 : inv1, j & 63);
          if (lane == (j & 63)) {
            if (j < 64) b0 = yj; else b1 = yj;
          }
          b0 -= a0[c] * yj;  // a0 / a1 are zero for rows at or above j
          b1 -= a1[c] * yj;
        }
      }
    }
    for (int j0 = ((n - 1) / CH_NB) * CH_NB; j0 >= 0; j0 -= CH_NB) {
      double a0[CH_NB], a1[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; c++) {
        const int j = min(j0 + c, n - 1);
        a0[c] = (r0 < j && j0 + c < n) ? A[j * ld + r0] : 0.0;
        a1[c] = (r1 < j && j0 + c < n) ? A[j * ld + r1] : 0.0;
      }
#pragma unroll
      for (int c = CH_NB - 1; c >= 0; c--) {
        const int j = j0 + c;
        if (j < n) {
          const double xj = readlane_f64(j < 64 ? b0 : b1, j & 63) * readlane_f64(j < 64 ? inv0 : inv1, j & 63);
          if (lane == (j & 63)) {
            if (j < 64) b0 = xj; else b1 = xj;
          }
          b0 -= a0[c] * xj;  // zero for rows at or below j
          b1 -= a1[c] * xj;
        }
      }
    }
    if (r0 < n) dc[r0] = -b0;
    if (r1 < n) dc[r1] = -b1;
  }
  CHS_STAMP(4)
#ifdef CHS_TIMING
  if (tid == 0) printf("chol_small n %d cycles: load %lld panel-load+barrier %lld factor+solve+store %lld trailing %lld substitutions %lld\n", n, tp[0], tp[1], tp[2], tp[3], tp[4]);
#endif
  if (tid == 0) {
    *ok_flag = good ? 1 : 0;
    if (arm_flag) *arm_flag = 1;  // the 'all finite' flag of the step: armed here, cleared by the back-substitution kernel
  }
}

__global__ void ba_negate_kernel(int n, const double* __restrict__ y, double* __restrict__ dc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dc[i] = -y[i];
}

// delta_l = -P^-1 (b_l + sum_q W_q^T delta_c)
__global__ __launch_bounds__(256) void ba_backsub_kernel(BaDims D, const int* __restrict__ lm_start,
                                                         const int* __restrict__ obs_cam, const int* __restrict__ cam_free,
                                                         const double* __restrict__ F, const double* __restrict__ E,
                                                         const double* __restrict__ Pinv, const double* __restrict__ bl,
                                                         const double* __restrict__ dc, double* __restrict__ dl,
                                                         int* __restrict__ finite_flag = nullptr) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (finite_flag)  // the camera step is checked here as well (it saves the launch of ba_all_finite2_kernel)
    for (int j = l; j < D.n; j += gridDim.x * 256)
      if (!isfinite(dc[j])) *finite_flag = 0;
  if (l >= D.L) return;
  double t[3] = {bl[3 * (size_t)l], bl[3 * (size_t)l + 1], bl[3 * (size_t)l + 2]};
  for (int i = lm_start[l]; i < lm_start[l + 1]; i++) {
    const int fc = cam_free[obs_cam[i]];
    if (fc < 0) continue;
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    double fd0 = 0, fd1 = 0;
    for (int j = 0; j < 6; j++) {
      fd0 += f[j] * dc[6 * fc + j];
      fd1 += f[6 + j] * dc[6 * fc + j];
    }
    for (int j = 0; j < 3; j++) t[j] += e[j] * fd0 + e[3 + j] * fd1;
  }
  const double* Pi = Pinv + 9 * (size_t)l;
  for (int j = 0; j < 3; j++) {
    const double v = -(Pi[3 * j] * t[0] + Pi[3 * j + 1] * t[1] + Pi[3 * j + 2] * t[2]);
    dl[3 * (size_t)l + j] = v;
    if (finite_flag && !isfinite(v)) *finite_flag = 0;
  }
}

// model cost change partials: -(J d)^T (r + J d / 2)
__global__ __launch_bounds__(256) void ba_model_kernel(BaDims D, const int* __restrict__ obs_cam,
                                                       const int* __restrict__ obs_lm, const int* __restrict__ cam_free,
                                                       const double* __restrict__ r, const double* __restrict__ F,
                                                       const double* __restrict__ E, const double* __restrict__ dc,
                                                       const double* __restrict__ dl, double* __restrict__ partials) {
  __shared__ double sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v = 0;
  if (i < D.O) {
    const int fc = cam_free[obs_cam[i]], lm = obs_lm[i];
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    double m0 = 0, m1 = 0;
    if (fc >= 0)
      for (int j = 0; j < 6; j++) {
        m0 += f[j] * dc[6 * fc + j];
        m1 += f[6 + j] * dc[6 * fc + j];
      }
    for (int j = 0; j < 3; j++) {
      m0 += e[j] * dl[3 * (size_t)lm + j];
      m1 += e[3 + j] * dl[3 * (size_t)lm + j];
    }
    v = -(m0 * (r[2 * (size_t)i] + m0 / 2.0) + m1 * (r[2 * (size_t)i + 1] + m1 / 2.0));
  }
  const double t = block_sum_256(v, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// candidate = Plus(x, step .* scale): T * exp(delta) for poses (local_parameterization_se3.hpp:43-50),
// x + delta for points.  Per-workgroup partials: [0..nb) squared step norm, [nb..2nb) squared x norm
// (of the CURRENT x, non-constant blocks only).
__global__ __launch_bounds__(256) void ba_update_kernel(BaDims D, const int* __restrict__ cam_free,
                                                        const double* __restrict__ poses, const double* __restrict__ points,
                                                        const double* __restrict__ dc, const double* __restrict__ dl,
                                                        const double* __restrict__ scale_c, const double* __restrict__ scale_l,
                                                        double* __restrict__ cand_poses, double* __restrict__ cand_points,
                                                        double* __restrict__ partials, int nb) {
  __shared__ double sh[256];
  const int t = blockIdx.x * 256 + threadIdx.x;
  double step2 = 0, x2 = 0;
  if (t < D.C) {
    const int fc = cam_free[t];
    const double* T = poses + 7 * (size_t)t;
    double* o = cand_poses + 7 * (size_t)t;
    if (fc < 0) {
      for (int j = 0; j < 7; j++) o[j] = T[j];
    } else {
      double d[6];
      for (int j = 0; j < 6; j++) {
        d[j] = dc[6 * fc + j] * scale_c[6 * fc + j];
        step2 += d[j] * d[j];
      }
      for (int j = 0; j < 7; j++) x2 += T[j] * T[j];
      se3_plus(T, d, o);  // [upstream] Sophus SE3::exp (ba_device.h)
    }
  }
  if (t < D.L) {
    for (int j = 0; j < 3; j++) {
      const double dd = dl[3 * (size_t)t + j] * scale_l[3 * (size_t)t + j];
      step2 += dd * dd;
      const double xv = points[3 * (size_t)t + j];
      x2 += xv * xv;
      cand_points[3 * (size_t)t + j] = xv + dd;
    }
  }
  const double a = block_sum_256(step2, sh);
  __syncthreads();
  const double b = block_sum_256(x2, sh);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = a;
    partials[nb + blockIdx.x] = b;
  }
}

__global__ void ba_set_flags_kernel(int* __restrict__ flag) {
  if (threadIdx.x < 2) flag[threadIdx.x] = 1;
}

// both step vectors in one launch
__global__ void ba_all_finite2_kernel(int na, const double* __restrict__ a, int nb, const double* __restrict__ b,
                                      int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((i < na && !isfinite(a[i])) || (i < nb && !isfinite(b[i]))) *flag = 0;
}

// scalars[slot] and scalars[slot + 1] = sums of partials[0..n) and partials[n..2n): one launch, a workgroup each
__global__ __launch_bounds__(256) void ba_reduce2_kernel(const double* __restrict__ partials, int n, double* __restrict__ scalars,
                                                         int slot) {
  __shared__ double sh[256];
  const double* p = partials + (size_t)blockIdx.x * n;
  double v = 0;
  for (int i = threadIdx.x; i < n; i += 256) v += p[i];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) scalars[slot + blockIdx.x] = sh[0];
}

__global__ void ba_all_finite_kernel(int n, const double* __restrict__ v, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !isfinite(v[i])) *flag = 0;
}

// raw per-observation residual / Jacobian blocks in the caller's observation order (parity hook)
__global__ void ba_raw_blocks_kernel(BaDims D, const double* __restrict__ poses, const double* __restrict__ points,
                                     const double* __restrict__ intr, const int* __restrict__ cam_intr,
                                     const int* __restrict__ obs_cam, const int* __restrict__ obs_lm,
                                     const double* __restrict__ obs_uv, double* __restrict__ r, double* __restrict__ F,
                                     double* __restrict__ E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.O) return;
  const int cam = obs_cam[i], lm = obs_lm[i], k = cam_intr[cam];
  double rr[2], FF[12], EE[6];
  residual_blocks(k ? D.model1 : D.model0, intr + 8 * k, poses + 7 * cam, points + 3 * lm, obs_uv + 2 * i, rr, FF, EE, true);
  r[2 * (size_t)i] = rr[0];
  r[2 * (size_t)i + 1] = rr[1];
  for (int j = 0; j < 12; j++) F[12 * (size_t)i + j] = FF[j];
  for (int j = 0; j < 6; j++) E[6 * (size_t)i + j] = EE[j];
}

// ------------------------------------------------------------------------------- host-side state
struct DevBuf {
  void* p = nullptr;
  bool owned = false;  // arena-backed buffers (BaState) are not freed one by one
  ~DevBuf() {
    if (p && owned) (void)hipFree(p);
  }
  template <class T>
  T* as() {
    return (T*)p;
  }
  hipError_t alloc(size_t bytes) {
    owned = true;
    return hipMalloc(&p, bytes > 0 ? bytes : 8);
  }
};

struct BaState {
  BaDims D;
  int G = 1, lm_per_wg = 1, nb_obs = 1, nb_upd = 1;
  int cb_seg = 1;  // workgroups per free camera in ba_cam_block_kernel
  int bl_seg = 1;  // ... in bal_cam_kernel (recompute form: an observation is a chain of dependent gathers, one per thread)
  bool small = true;
  std::vector<int> perm;  // sorted position -> caller observation index
  DevBuf poses, cand_poses, points, cand_points, intr, cam_intr, cam_free, free_cams, obs_cam, obs_lm, obs_uv, lm_start,
      cam_start, cam_obs, r, F, E, scale_c, scale_l, n2l, grad_l, H, g_c, diag_c, diag_l, gabs, S, rhs, S_part, rhs_part,
      Pinv, bl, dc, dl, partials, scalars, flag, cam_part;
  // second linearisation set (vsl_bundle_adjust linearises the CANDIDATE point speculatively, before the host has
  // read the step's verdict; an accepted step swaps the sets, a rejected one leaves the current set untouched)
  DevBuf r2, F2, E2, n2l2, grad_l2, H2, g_c2, diag_c2, diag_l2;
  bool want_alt_set = false;
  // large systems, gather form of the Schur complement (ba_schur_gather_kernel): per-block pair lists, built on the
  // first use for the landmark range they cover, and the per-observation W / Y blocks of the current linearisation
  DevBuf pair_cnt, pair_start, pairs, Wg, Yg, cam_pos;
  // recompute form of a session's iteration (ba_large.h): landmark runs of the workgroups, their partial sums
  DevBuf wg_lm, lpart, pbs, cam_lm, cam_uv;  // pbs[3 l + x] = scale_l (P^-1 b)_l: what the reduced right-hand side needs of a landmark
  int n_wg = 0;
  bool large_fused = false;
  int n_slots = 0, hbp1 = 0;
  size_t n_pairs_cap = 0;
  int pair_l0 = -1, pair_lc = -1;
  // Layout of the reduced camera system S: dense (ldS = n, offset 0) or, for large systems whose cameras can be
  // ordered into a narrow band (reverse Cuthill-McKee on the covisibility graph, ba_setup), LAPACK-style lower band
  // storage -- row i keeps columns [i - bws, i], bws = bw + VSL_CHOL_NB, entry (i, j) at S[i * ldS + j + offS] with
  // ldS = offS = bws (chol.hip "BAND FORM").  The free-camera numbering IS the band order.
  bool banded = false;
  bool cyclic = false;  // band form whose band closes on itself (camera loop in trajectory order): wrap blocks in the leading slots of the first rows
  int ldS = 0, offS = 0, bw = 0;
  size_t s_elems = 0;  // doubles to allocate / clear / exchange for S
  double* S_eff() { return (double*)S.p + offS; }
  // ONE device allocation per solve, carved into the buffers above (40 hipMalloc calls cost more than 3 LM iterations);
  // vsl_bundle_adjust lends the context's cached arena, a session owns its own
  void* arena = nullptr;
  size_t arena_cap = 0;
  bool arena_owned = false;
  vsl_ctx* arena_lender = nullptr;
  ~BaState() {
    if (arena && arena_owned) (void)hipFree(arena);
    if (arena_lender) arena_lender->ba_arena_busy = false;
  }
  void swap_sets() {
    std::swap(r.p, r2.p); std::swap(F.p, F2.p); std::swap(E.p, E2.p); std::swap(n2l.p, n2l2.p);
    std::swap(grad_l.p, grad_l2.p); std::swap(H.p, H2.p); std::swap(g_c.p, g_c2.p); std::swap(diag_c.p, diag_c2.p);
    std::swap(diag_l.p, diag_l2.p);
    std::swap(poses.p, cand_poses.p); std::swap(points.p, cand_points.p);
  }
};

#define BA_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return vsl_fail(ctx, VSL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

int ba_validate(vsl_ctx* ctx, const vsl_ba_problem* p) {
  if (!ctx) return VSL_ERR_INVALID;
  if (!p || p->n_cams <= 0 || p->n_lms <= 0 || p->n_obs <= 0 || !p->poses || !p->cam_fixed || !p->cam_intr || !p->intr ||
      !p->points || !p->obs_cam || !p->obs_lm || !p->obs_uv)
    return vsl_fail(ctx, VSL_ERR_INVALID, "bundle adjustment: null pointer or empty problem");
  for (int k = 0; k < 2; k++)
    if (p->cam_model[k] < 0 || p->cam_model[k] > 3) return vsl_fail(ctx, VSL_ERR_INVALID, "unknown camera model %d", p->cam_model[k]);
  for (int i = 0; i < p->n_obs; i++)
    if (p->obs_cam[i] < 0 || p->obs_cam[i] >= p->n_cams || p->obs_lm[i] < 0 || p->obs_lm[i] >= p->n_lms)
      return vsl_fail(ctx, VSL_ERR_INVALID, "observation %d references camera %d / landmark %d out of range", i, p->obs_cam[i], p->obs_lm[i]);
  for (int c = 0; c < p->n_cams; c++)
    if (p->cam_intr[c] < 0 || p->cam_intr[c] > 1) return vsl_fail(ctx, VSL_ERR_INVALID, "cam_intr[%d] = %d not in {0,1}", c, p->cam_intr[c]);
  return VSL_OK;
}

template <class T>
int upload(vsl_ctx* ctx, DevBuf& b, const T* src, size_t n) {
  BA_HIP(b.alloc(sizeof(T) * n));
  if (n) BA_HIP(hipMemcpyAsync(b.p, src, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
  return VSL_OK;
}

// host loops of the set-up over ranges of [0, n) on a few threads (the set-up of a 1000-camera solve was ~15 ms of
// single-threaded loops over 881k observations next to LM iterations of 2.5 ms)
template <class Fn>
void host_parallel(int n, Fn fn, int min_parallel = 1 << 19) {
  const int hw = (int)std::thread::hardware_concurrency();
  // (starting the threads costs ~0.4 ms: worth it for the global problems only -- a local window of 157k observations
  // went from 6.0 to 8.1 ms per solve with them)
  const int nt = n < min_parallel ? 1 : std::max(1, std::min(8, hw > 0 ? hw : 1));
  if (nt == 1) {
    fn(0, n, 0);
    return;
  }
  std::vector<std::thread> th;
  const int chunk = (n + nt - 1) / nt;
  for (int t = 0; t < nt; t++) {
    const int a = t * chunk, b = std::min(n, a + chunk);
    if (a < b) th.emplace_back([=] { fn(a, b, t); });
  }
  for (auto& t : th) t.join();
}

// Band order of the free cameras: reverse Cuthill-McKee on the covisibility graph (two free cameras are adjacent iff
// some landmark is observed by both: exactly the non-zero 6 x 6 blocks of the reduced camera system).  gp = the
// problem whose observations define the graph (a session passes the FULL problem so that every rank derives the same
// order).  order[position] = free index in ascending-camera numbering; returns the block half-bandwidth (max
// |position difference| over the edges).  A 500-keyframe loop comes out as a band of a few dozen cameras with no
// corner blocks (the breadth-first levels run both ways round the loop).
int camera_band_order(const vsl_ba_problem* gp, const std::vector<int>& cam_free0, int nfree, std::vector<int>& order,
                      int* half_cyclic_natural = nullptr) {
  const size_t words = ((size_t)nfree + 63) / 64;
  std::vector<uint64_t> adj((size_t)nfree * words, 0);
  {
    std::vector<int> start(gp->n_lms + 1, 0);
    bool sorted_in = true;
    for (int i = 0; i < gp->n_obs; i++) {
      start[gp->obs_lm[i] + 1]++;
      if (i > 0 && gp->obs_lm[i] < gp->obs_lm[i - 1]) sorted_in = false;
    }
    for (int l = 0; l < gp->n_lms; l++) start[l + 1] += start[l];
    // free-camera index of every observation in landmark order; the reference's own order is landmark order already:
    // then the threads below look the cameras up themselves (no gathered copy)
    std::vector<int> cams_v;
    if (!sorted_in) {
      cams_v.resize(gp->n_obs);
      std::vector<int> fill(start.begin(), start.end() - 1);
      for (int i = 0; i < gp->n_obs; i++) cams_v[fill[gp->obs_lm[i]]++] = cam_free0[gp->obs_cam[i]];
    }
    const int* cams_p = sorted_in ? nullptr : cams_v.data();
    const int32_t* ocam = gp->obs_cam;
    auto cam_at = [&](int a) { return cams_p ? cams_p[a] : cam_free0[ocam[a]]; };
    // a private bit matrix per thread, OR-ed together afterwards (shared atomics made the threads fight over its lines)
    std::vector<std::vector<uint64_t>> priv(8);
    host_parallel(gp->n_lms, [&](int l0, int l1, int t) {
      std::vector<uint64_t>& my = priv[t];
      my.assign((size_t)nfree * words, 0);
      for (int l = l0; l < l1; l++)
        for (int a = start[l]; a < start[l + 1]; a++) {
          const int ca = cam_at(a);
          if (ca < 0) continue;
          for (int b = a + 1; b < start[l + 1]; b++) {
            const int cb = cam_at(b);
            if (cb < 0 || cb == ca) continue;
            my[(size_t)ca * words + (cb >> 6)] |= 1ull << (cb & 63);
            my[(size_t)cb * words + (ca >> 6)] |= 1ull << (ca & 63);
          }
        }
    }, 1 << 15);  // ~k^2 = 80 bit operations per landmark: parallel from 32k landmarks
    for (auto& my : priv)
      if (!my.empty())
        for (size_t i = 0; i < adj.size(); i++) adj[i] |= my[i];
  }
  std::vector<std::vector<int>> nb(nfree);
  std::vector<int> deg(nfree, 0);
  for (int c = 0; c < nfree; c++)
    for (size_t w = 0; w < words; w++) {
      uint64_t m = adj[(size_t)c * words + w];
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        nb[c].push_back((int)(w * 64) + b);
      }
    }
  for (int c = 0; c < nfree; c++) deg[c] = (int)nb[c].size();
  for (int c = 0; c < nfree; c++) std::sort(nb[c].begin(), nb[c].end(), [&](int x, int y) { return deg[x] != deg[y] ? deg[x] < deg[y] : x < y; });
  std::vector<char> seen(nfree, 0);
  order.clear();
  order.reserve(nfree);
  auto bfs = [&](int root, std::vector<int>& out) {  // Cuthill-McKee from root over the unseen part; returns the last level's first node
    const size_t first = out.size();
    out.push_back(root);
    seen[root] = 1;
    for (size_t h = first; h < out.size(); h++)
      for (int v : nb[out[h]])
        if (!seen[v]) {
          seen[v] = 1;
          out.push_back(v);
        }
    return out.back();
  };
  for (;;) {
    int root = -1;
    for (int c = 0; c < nfree; c++)
      if (!seen[c] && (root < 0 || deg[c] < deg[root])) root = c;
    if (root < 0) break;
    // pseudo-peripheral start: two sweeps (the far end of a sweep from a minimum-degree node)
    std::vector<int> probe;
    const int far_end = bfs(root, probe);
    for (int v : probe) seen[v] = 0;
    bfs(far_end, order);
  }
  std::reverse(order.begin(), order.end());
  std::vector<int> pos(nfree);
  for (int k = 0; k < nfree; k++) pos[order[k]] = k;
  int half = 0;
  for (int c = 0; c < nfree; c++)
    for (int v : nb[c]) half = std::max(half, std::abs(pos[c] - pos[v]));
  if (half_cyclic_natural) {
    // the cameras as they come (ascending index = the reference's keyframe order, i.e. along the trajectory), distances
    // taken AROUND the ring: a closed loop has half the bandwidth of its best linear order this way
    int hc = 0;
    for (int c = 0; c < nfree; c++)
      for (int v : nb[c]) {
        const int d = std::abs(c - v);
        hc = std::max(hc, std::min(d, nfree - d));
      }
    *half_cyclic_natural = hc;
  }
  return half;
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// set-up phase times on stderr when VSL_BA_TRACE is set (developer aid)
struct BaTrace {
  bool on;
  double t0;
  BaTrace() : on(getenv("VSL_BA_TRACE") != nullptr), t0(now_ms()) {}
  void lap(const char* what) {
    if (!on) return;
    const double t = now_ms();
    fprintf(stderr, "  [ba set-up] %-28s %8.3f ms\n", what, t - t0);
    t0 = t;
  }
};

int ba_setup(vsl_ctx* ctx, const vsl_ba_problem* p, const vsl_ba_options* o, BaState& st, bool allow_band = false,
             const vsl_ba_problem* graph_prob = nullptr) {
  BaTrace tr;
  BaDims& D = st.D;
  D.C = p->n_cams;
  D.L = p->n_lms;
  D.O = p->n_obs;
  D.model0 = p->cam_model[0];
  D.model1 = p->cam_model[1];
  D.use_huber = o ? o->use_huber : 1;
  D.huber = o ? o->huber_parameter : 1.0;
  std::vector<int> cam_free(D.C, -1), free_cams;
  for (int c = 0; c < D.C; c++)
    if (!p->cam_fixed[c]) {
      cam_free[c] = (int)free_cams.size();
      free_cams.push_back(c);
    }
  D.nfree = (int)free_cams.size();
  D.n = 6 * D.nfree;
  // layout of the reduced camera system (see BaState): dense unless the cameras order into a narrow band
  st.banded = false;
  st.ldS = D.n;
  st.offS = 0;
  st.bw = D.n;
  st.s_elems = (size_t)D.n * D.n;
  if (allow_band && D.n > 128 && !ctx->ba_force_dense) {
    std::vector<int> order;
    int half_cyc = 0;
    const int half = camera_band_order(graph_prob ? graph_prob : p, cam_free, D.nfree, order, &half_cyc);
    const int bw = 6 * half + 5, bws = bw + VSL_CHOL_NB;
    // CYCLIC band (round 4): in the cameras' own order with distances around the ring.  Taken when the ring solver has a
    // block layout for it and its blocks are at most 3/4 of the linear form's (the solve costs ~ block size squared per
    // level): the 500-keyframe loop of configs[4] has half bandwidth 18 cameras around the ring, 36 in its best line
    const int bwc = 6 * half_cyc + 5, B_lin = (bw + 1 + 31) / 32 * 32;
    int B_cyc = 0, nblk_cyc = 0;
    static const bool env_no_cyclic = getenv("VSL_BA_NO_CYCLIC") != nullptr;
    if (!ctx->ba_no_cyclic && !env_no_cyclic && !ctx->ba_schur_atomics && !ctx->chol_no_bcr && !ctx->chol_no_fused &&
        vsl_chol_bcr_cyclic_layout(D.n, bwc, &B_cyc, &nblk_cyc) && 4 * B_cyc <= 3 * B_lin) {
      st.banded = true;
      st.cyclic = true;
      st.bw = bwc;
      st.ldS = st.offS = bwc + VSL_CHOL_NB;
      st.s_elems = (size_t)D.n * (bwc + VSL_CHOL_NB + 1) + 64;
    } else if ((size_t)(bws + 1) * 2 < (size_t)D.n) {  // worth it: the band holds less than half of the matrix
      std::vector<int> renum(D.nfree);
      for (int k = 0; k < D.nfree; k++) renum[k] = free_cams[order[k]];
      free_cams = renum;
      for (int k = 0; k < D.nfree; k++) cam_free[free_cams[k]] = k;
      st.banded = true;
      st.bw = bw;
      st.ldS = st.offS = bws;
      st.s_elems = (size_t)D.n * (bws + 1) + 64;  // + slack: the diagonal kernels read (never use) a few entries past a row
    }
  }
  // (diagnostic read-out for benchmarks: vsl_ctx_last_ba_layout)
  ctx->last_ba_s_elems = (int64_t)st.s_elems;
  ctx->last_ba_banded = st.cyclic ? 2 : (st.banded ? 1 : 0);
  ctx->last_ba_bw = st.bw;
  tr.lap("free cameras + band order");
  // sort observations by landmark (stable: keeps the caller's order inside a landmark).  The reference's own order
  // (map_utils.h:373 / loop_closure_utils.h:700: landmarks, then their observations) -- what
  // include/visnav_amd/bundle_adjustment.h hands over -- is sorted already: then the caller's arrays ARE the sorted ones
  // (no permutation, no 24 MB of gathered copies at 881 k observations; st.perm stays empty = identity)
  std::vector<int> lm_start(D.L + 1, 0), cam_start(D.C + 1, 0);
  bool sorted_in = true;
  for (int i = 0; i < D.O; i++) {
    lm_start[p->obs_lm[i] + 1]++;
    cam_start[p->obs_cam[i] + 1]++;
    if (i > 0 && p->obs_lm[i] < p->obs_lm[i - 1]) sorted_in = false;
  }
  for (int l = 0; l < D.L; l++) lm_start[l + 1] += lm_start[l];
  for (int c = 0; c < D.C; c++) cam_start[c + 1] += cam_start[c];
  std::vector<int> s_cam_v, s_lm_v;
  std::vector<double> s_uv_v;
  const int32_t *s_cam = p->obs_cam, *s_lm = p->obs_lm;
  const double* s_uv = p->obs_uv;
  st.perm.clear();
  if (!sorted_in) {
    st.perm.resize(D.O);
    {
      std::vector<int> fill(lm_start.begin(), lm_start.end() - 1);
      for (int i = 0; i < D.O; i++) st.perm[fill[p->obs_lm[i]]++] = i;
    }
    s_cam_v.resize(D.O);
    s_lm_v.resize(D.O);
    s_uv_v.resize(2 * (size_t)D.O);
    host_parallel(D.O, [&](int q0, int q1, int) {
      for (int q = q0; q < q1; q++) {
        const int i = st.perm[q];
        s_cam_v[q] = p->obs_cam[i];
        s_lm_v[q] = p->obs_lm[i];
        s_uv_v[2 * (size_t)q] = p->obs_uv[2 * (size_t)i];
        s_uv_v[2 * (size_t)q + 1] = p->obs_uv[2 * (size_t)i + 1];
      }
    });
    s_cam = s_cam_v.data();
    s_lm = s_lm_v.data();
    s_uv = s_uv_v.data();
  }
  int kmax_free = 0;
  size_t n_pairs = 0;  // (observation, observation) pairs of the block lists of the gather-form Schur complement
  // camera CSR over the sorted observation positions (counts gathered with the landmark counts above) and its inverse:
  // position of an observation in camera-major order (the gather-form Schur kernels keep their blocks in that order,
  // so that the blocks of one camera row read one contiguous segment).  ONE parallel region (round 4; three regions and
  // a serial scatter were 3.9 ms at 881 k observations, of which 1.2 ms starting threads): every thread counts the pairs
  // of its landmark range and the cameras of its observation chunk; after a barrier thread 0 turns the chunk histograms
  // into cursors; after another every thread scatters its chunk -- a stable counting sort, the caller's order inside a camera
  std::vector<int> cam_obs(D.O), cam_pos(D.O);
  {
    const int hw = (int)std::thread::hardware_concurrency();
    const int nt = D.O < (1 << 19) ? 1 : std::max(1, std::min(8, hw > 0 ? hw : 1));
    std::vector<std::vector<int>> hist(nt, std::vector<int>((size_t)D.C, 0));
    size_t np_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int km_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::atomic<int> arrived{0};
    auto barrier = [&](int phase) {  // (phase = how many barriers this thread has passed before)
      arrived.fetch_add(1, std::memory_order_acq_rel);
      while (arrived.load(std::memory_order_acquire) < nt * (phase + 1)) std::this_thread::yield();
    };
    auto work = [&](int t) {
      const int l0 = (int)((long long)D.L * t / nt), l1 = (int)((long long)D.L * (t + 1) / nt);
      const int q0 = (int)((long long)D.O * t / nt), q1 = (int)((long long)D.O * (t + 1) / nt);
      size_t np = 0;
      int km = 0;
      for (int l = l0; l < l1; l++) {
        int k = 0;
        for (int q = lm_start[l]; q < lm_start[l + 1]; q++) k += cam_free[s_cam[q]] >= 0;
        km = std::max(km, k);
        np += (size_t)k * k;  // upper bound (k (k + 1) / 2 when no camera observes a landmark twice)
      }
      np_t[t] = np;
      km_t[t] = km;
      std::vector<int>& h = hist[t];
      for (int q = q0; q < q1; q++) h[s_cam[q]]++;
      barrier(0);
      if (t == 0)
        for (int c = 0; c < D.C; c++) {
          int at = cam_start[c];
          for (int u = 0; u < nt; u++) {
            const int cnt = hist[u][c];
            hist[u][c] = at;  // becomes thread u's cursor for camera c
            at += cnt;
          }
        }
      barrier(1);
      for (int q = q0; q < q1; q++) {
        const int k = h[s_cam[q]]++;
        cam_obs[k] = q;
        cam_pos[q] = k;
      }
    };
    if (nt == 1) {
      work(0);
    } else {
      std::vector<std::thread> th;
      for (int t = 1; t < nt; t++) th.emplace_back(work, t);
      work(0);
      for (auto& x : th) x.join();
    }
    for (int t = 0; t < nt; t++) {
      n_pairs += np_t[t];
      kmax_free = std::max(kmax_free, km_t[t]);
    }
  }
  tr.lap("sort + CSRs");
  st.small = D.n <= 128 && D.nfree <= SCH_CMAX && kmax_free <= SCH_KMAX;
  // landmark runs of the recompute-form kernels: <= BL_THREADS observations and <= BL_LMW landmarks per workgroup
  std::vector<int> wg_lm;
  if (!st.small) {
    wg_lm.push_back(0);
    for (int l = 0, a = 0; l < D.L; l++) {
      if (lm_start[l + 1] - lm_start[l] > BL_THREADS) {  // a landmark seen by more cameras than a workgroup has threads
        wg_lm.clear();
        break;
      }
      if (lm_start[l + 1] - lm_start[a] > BL_THREADS || l - a == BL_LMW) {
        wg_lm.push_back(l);
        a = l;
      }
      if (l == D.L - 1) wg_lm.push_back(D.L);
    }
  }
  st.n_wg = wg_lm.empty() ? 0 : (int)wg_lm.size() - 1;
  st.nb_obs = (D.O + 255) / 256;
  st.nb_upd = (std::max(D.C, D.L) + 255) / 256;
  st.G = std::min(SCH_GMAX, (D.L + SCH_LB - 1) / SCH_LB);
  st.lm_per_wg = ((D.L + st.G - 1) / st.G + SCH_LB - 1) / SCH_LB * SCH_LB;
  st.G = (D.L + st.lm_per_wg - 1) / st.lm_per_wg;

  const size_t n = (size_t)D.n, L = (size_t)D.L, O = (size_t)D.O, C = (size_t)D.C;
  // enough workgroups per camera that a camera's observations are ~2 slices of 256 per workgroup (1 when cameras are many)
  st.cb_seg = D.nfree > 0 ? std::max(1, std::min(32, (int)(D.O / std::max(1, D.nfree) / 512))) : 1;
  st.bl_seg = st.cb_seg;  // (one workgroup per 256 observations of a camera measured 121 us against 80: more gathers in flight than the L2 holds)
  struct Want { DevBuf* b; size_t bytes; };
  std::vector<Want> want = {
      {&st.poses, 8 * 7 * C}, {&st.points, 8 * 3 * L}, {&st.intr, 8 * 16}, {&st.cam_intr, 4 * C}, {&st.cam_free, 4 * C},
      {&st.free_cams, 4 * free_cams.size()}, {&st.obs_cam, 4 * O}, {&st.obs_lm, 4 * O}, {&st.obs_uv, 16 * O},
      {&st.lm_start, 4 * (L + 1)}, {&st.cam_start, 4 * (C + 1)}, {&st.cam_obs, 4 * O},
      {&st.cand_poses, 8 * 7 * C}, {&st.cand_points, 8 * 3 * L}, {&st.r, 16 * O}, {&st.F, 96 * O}, {&st.E, 48 * O},
      {&st.scale_c, 8 * n}, {&st.scale_l, 24 * L}, {&st.n2l, 24 * L}, {&st.grad_l, 24 * L},
      {&st.cam_part, 8 * 33 * (size_t)std::max(1, D.nfree) * std::max(st.cb_seg, st.bl_seg)}, {&st.H, 8 * 36 * (size_t)D.nfree}, {&st.g_c, 8 * n},
      {&st.diag_c, 8 * n}, {&st.diag_l, 24 * L}, {&st.gabs, 8 * (n + 3 * L)}, {&st.S, 8 * st.s_elems}, {&st.rhs, 8 * n},
      {&st.Pinv, 72 * L}, {&st.bl, 24 * L}, {&st.dc, 8 * n}, {&st.dl, 24 * L},
      {&st.partials, 8 * (size_t)(2 * std::max(st.nb_obs, st.nb_upd) + 16)}, {&st.scalars, 8 * 16 + sizeof(int) * 4}};
  if (st.small) {
    want.push_back({&st.S_part, 8 * n * n * st.G});
    want.push_back({&st.rhs_part, 8 * n * st.G});
  } else {
    st.hbp1 = st.banded ? (st.bw - 5) / 6 + 1 : 0;
    st.n_slots = st.banded ? D.nfree * st.hbp1 : D.nfree * (D.nfree + 1) / 2;
    st.n_pairs_cap = n_pairs;
    st.pair_l0 = st.pair_lc = -1;
    want.push_back({&st.pair_cnt, 4 * ((size_t)st.n_slots + 1)});
    want.push_back({&st.pair_start, 4 * ((size_t)st.n_slots + 1)});
    want.push_back({&st.pairs, 8 * std::max<size_t>(n_pairs < ((size_t)1 << 31) ? n_pairs : 1, 1)});
    want.push_back({&st.cam_pos, 4 * O});
    want.push_back({&st.Wg, 8 * 18 * O});
    want.push_back({&st.Yg, 8 * 18 * O});
    want.push_back({&st.wg_lm, 4 * ((size_t)st.n_wg + 1)});
    want.push_back({&st.lpart, 8 * 4 * (size_t)std::max(1, st.n_wg)});
    want.push_back({&st.pbs, 24 * L});
    want.push_back({&st.cam_lm, 4 * O});
    want.push_back({&st.cam_uv, 16 * O});
  }
  if (st.want_alt_set) {
    const Want alt[] = {{&st.r2, 16 * O}, {&st.F2, 96 * O}, {&st.E2, 48 * O}, {&st.n2l2, 24 * L}, {&st.grad_l2, 24 * L},
                        {&st.H2, 8 * 36 * (size_t)D.nfree}, {&st.g_c2, 8 * n}, {&st.diag_c2, 8 * n}, {&st.diag_l2, 24 * L}};
    want.insert(want.end(), std::begin(alt), std::end(alt));
  }
  size_t total = 0;
  for (auto& wnt : want) total += (std::max<size_t>(wnt.bytes, 8) + 255) & ~(size_t)255;
  if (ctx->ba_arena_busy || !st.want_alt_set) {   // sessions (and nested use) own their arena
    BA_HIP(hipMalloc(&st.arena, total));
    st.arena_cap = total;
    st.arena_owned = true;
  } else {
    if (ctx->ba_arena_cap < total) {
      BA_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->ba_arena) (void)hipFree(ctx->ba_arena);
      ctx->ba_arena = nullptr;
      ctx->ba_arena_cap = 0;
      const size_t cap = total + total / 4;
      BA_HIP(hipMalloc(&ctx->ba_arena, cap));
      ctx->ba_arena_cap = cap;
    }
    st.arena = ctx->ba_arena;
    st.arena_cap = ctx->ba_arena_cap;
    ctx->ba_arena_busy = true;
    st.arena_lender = ctx;
  }
  {
    size_t off = 0;
    for (auto& wnt : want) {
      wnt.b->p = (char*)st.arena + off;
      off += (std::max<size_t>(wnt.bytes, 8) + 255) & ~(size_t)255;
    }
  }
  st.flag.p = (char*)st.scalars.p + 8 * 16;  // behind the 16 scalars: one copy brings both back
  tr.lap("arena");
  auto up = [&](DevBuf& bf, const void* src, size_t bytes) -> hipError_t {
    return bytes ? hipMemcpyAsync(bf.p, src, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
  };
  BA_HIP(up(st.poses, p->poses, 8 * 7 * C));
  BA_HIP(up(st.points, p->points, 8 * 3 * L));
  BA_HIP(up(st.intr, p->intr, 8 * 16));
  BA_HIP(up(st.cam_intr, p->cam_intr, 4 * C));
  BA_HIP(up(st.cam_free, cam_free.data(), 4 * C));
  BA_HIP(up(st.free_cams, free_cams.data(), 4 * free_cams.size()));
  BA_HIP(up(st.obs_cam, s_cam, 4 * O));
  BA_HIP(up(st.obs_lm, s_lm, 4 * O));
  BA_HIP(up(st.obs_uv, s_uv, 16 * O));
  BA_HIP(up(st.lm_start, lm_start.data(), 4 * (L + 1)));
  BA_HIP(up(st.cam_start, cam_start.data(), 4 * (C + 1)));
  BA_HIP(up(st.cam_obs, cam_obs.data(), 4 * O));
  if (!st.small) BA_HIP(up(st.cam_pos, cam_pos.data(), 4 * O));
  if (!st.small && st.n_wg > 0) BA_HIP(up(st.wg_lm, wg_lm.data(), 4 * wg_lm.size()));
  BA_HIP(hipStreamSynchronize(ctx->stream));  // the uploads above read host vectors that die here
  tr.lap("uploads");
  return VSL_OK;
}

// linearize at (poses, points): r, F, E (scaled when `scaled`), cost -> scalars[0]; per-landmark and
// per-camera column statistics.
int ba_linearize(vsl_ctx* ctx, BaState& st, bool scaled, int cost_slot = 0) {
  const BaDims& D = st.D;
  {
    VslStage s(ctx, VSL_STAGE_BA_LIN);
    hipLaunchKernelGGL(ba_linearize_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D, st.poses.as<double>(),
                       st.points.as<double>(), st.intr.as<double>(), st.cam_intr.as<int>(), st.cam_free.as<int>(),
                       st.obs_cam.as<int>(), st.obs_lm.as<int>(), st.obs_uv.as<double>(),
                       scaled ? st.scale_c.as<double>() : nullptr, scaled ? st.scale_l.as<double>() : nullptr,
                       st.r.as<double>(), st.F.as<double>(), st.E.as<double>(), st.partials.as<double>(), 1);
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>(), st.nb_obs,
                       st.scalars.as<double>(), cost_slot, 0);
    VSL_CHECK_LAUNCH(ctx);
  }
  return VSL_OK;
}

int ba_columns(vsl_ctx* ctx, BaState& st) {
  const BaDims& D = st.D;
  VslStage s(ctx, VSL_STAGE_BA_LIN);
  hipLaunchKernelGGL(ba_lm_cols_kernel, dim3((D.L + 255) / 256), dim3(256), 0, ctx->stream, D, st.lm_start.as<int>(),
                     st.r.as<double>(), st.E.as<double>(), st.n2l.as<double>(), st.grad_l.as<double>());
  if (D.nfree > 0) {
    hipLaunchKernelGGL(ba_cam_block_kernel, dim3(D.nfree, st.cb_seg), dim3(256), 0, ctx->stream, st.free_cams.as<int>(),
                       st.cam_start.as<int>(), st.cam_obs.as<int>(), st.r.as<double>(), st.F.as<double>(),
                       st.cam_part.as<double>());
    hipLaunchKernelGGL(ba_cam_block_finish_kernel, dim3((D.nfree * 27 + 255) / 256), dim3(256), 0, ctx->stream, D.nfree,
                       st.cb_seg, st.cam_part.as<double>(), st.H.as<double>(), st.g_c.as<double>());
  }
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// workgroups of ba_schur_gather_kernel: one per slot; in band form eight equal row ranges (see the kernel)
unsigned gather_grid(const BaState& st) {
  if (st.hbp1 <= 0) return (unsigned)st.n_slots;
  const int nrows = st.n_slots / st.hbp1;
  return 8u * (unsigned)((nrows + 7) / 8) * (unsigned)st.hbp1;
}

// block pair lists of the gather-form Schur complement for landmarks [l0, l0 + lc): built once per solve / session
int ba_pair_lists(vsl_ctx* ctx, BaState& st, int l0, int lc) {
  if (st.pair_l0 == l0 && st.pair_lc == lc) return VSL_OK;
  VSL_HIP(ctx, hipMemsetAsync(st.pair_cnt.p, 0, sizeof(int) * ((size_t)st.n_slots + 1), ctx->stream));
  hipLaunchKernelGGL(ba_pair_list_kernel<false>, dim3((lc + 255) / 256), dim3(256), 0, ctx->stream, l0, lc,
                     st.lm_start.as<int>(), st.obs_cam.as<int>(), st.cam_free.as<int>(), st.cam_pos.as<int>(), st.hbp1,
                     st.cyclic ? st.D.nfree : 0, st.pair_cnt.as<int>(), (const int*)nullptr, (int*)nullptr);
  hipLaunchKernelGGL(ba_pair_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, st.n_slots, st.pair_cnt.as<int>(),
                     st.pair_start.as<int>());
  VSL_HIP(ctx, hipMemsetAsync(st.pair_cnt.p, 0, sizeof(int) * ((size_t)st.n_slots + 1), ctx->stream));
  hipLaunchKernelGGL(ba_pair_list_kernel<true>, dim3((lc + 255) / 256), dim3(256), 0, ctx->stream, l0, lc,
                     st.lm_start.as<int>(), st.obs_cam.as<int>(), st.cam_free.as<int>(), st.cam_pos.as<int>(), st.hbp1,
                     st.cyclic ? st.D.nfree : 0, st.pair_cnt.as<int>(), st.pair_start.as<int>(), st.pairs.as<int>());
  hipLaunchKernelGGL(ba_pair_sort_kernel, dim3(st.n_slots), dim3(64), 0, ctx->stream, st.pair_start.as<int>(),
                     st.pairs.as<int>());
  VSL_CHECK_LAUNCH(ctx);
  st.pair_l0 = l0;
  st.pair_lc = lc;
  return VSL_OK;
}

// Schur complement of the landmark blocks over landmarks [l0, l0+lc); damping when diag != null.
int ba_schur(vsl_ctx* ctx, BaState& st, bool damp, double radius, int l0, int lc, bool keep_backsub, bool lower_only) {
  const BaDims& D = st.D;
  const int n = D.n;
  if (n == 0) return VSL_OK;
  VslStage s(ctx, VSL_STAGE_BA_SCHUR);
  const double inv_radius = damp ? 1.0 / radius : 0.0;
  const double* dgl = damp ? st.diag_l.as<double>() : nullptr;
  const double* dgc = damp ? st.diag_c.as<double>() : nullptr;
  double* Pinv = keep_backsub ? st.Pinv.as<double>() : nullptr;
  double* bl = keep_backsub ? st.bl.as<double>() : nullptr;
  if (st.small) {
    const int lpw = ((lc + st.G - 1) / st.G + SCH_LB - 1) / SCH_LB * SCH_LB;
    const int G = lpw > 0 ? (lc + lpw - 1) / lpw : 0;
    const bool block3 = !ctx->ba_schur_entries;
    if (G > 0) {
      if (block3)
        hipLaunchKernelGGL(ba_schur_small_kernel<true>, dim3(G), dim3(SCH_THREADS), 0, ctx->stream, D, st.lm_start.as<int>(),
                           st.obs_cam.as<int>(), st.cam_free.as<int>(), st.r.as<double>(), st.F.as<double>(),
                           st.E.as<double>(), dgl, inv_radius, l0, lc, lpw, st.S_part.as<double>(),
                           st.rhs_part.as<double>(), Pinv, bl);
      else
        hipLaunchKernelGGL(ba_schur_small_kernel<false>, dim3(G), dim3(SCH_THREADS), 0, ctx->stream, D, st.lm_start.as<int>(),
                           st.obs_cam.as<int>(), st.cam_free.as<int>(), st.r.as<double>(), st.F.as<double>(),
                           st.E.as<double>(), dgl, inv_radius, l0, lc, lpw, st.S_part.as<double>(),
                           st.rhs_part.as<double>(), Pinv, bl);
    }
    hipLaunchKernelGGL(ba_schur_finish_kernel, dim3((n * n + n + 15) / 16), dim3(256), 0, ctx->stream, n, G,
                       st.S_part.as<double>(), st.rhs_part.as<double>(), st.H.as<double>(), st.g_c.as<double>(), dgc,
                       inv_radius, st.S.as<double>(), st.rhs.as<double>(), block3 ? 1 : 0);
  } else {
    VSL_HIP(ctx, hipMemsetAsync(st.S.p, 0, sizeof(double) * st.s_elems, ctx->stream));
    VSL_HIP(ctx, hipMemsetAsync(st.rhs.p, 0, sizeof(double) * n, ctx->stream));
    const int lower_mode = st.banded ? 2 : ((lower_only && n > 128) ? 1 : 0);  // n <= 128 is solved by ba_chol_small_kernel (full matrix)
    // (pair lists index with 32-bit positions: a problem with 2^31 pairs or more keeps the atomic form)
    if (lc > 0 && (ctx->ba_schur_atomics || st.n_pairs_cap >= ((size_t)1 << 31)))
      hipLaunchKernelGGL(ba_schur_atomic_kernel, dim3((lc + 3) / 4), dim3(256), 0, ctx->stream, D, st.lm_start.as<int>(),
                         st.obs_cam.as<int>(), st.cam_free.as<int>(), st.r.as<double>(), st.F.as<double>(),
                         st.E.as<double>(), dgl, inv_radius, l0, lc, st.S_eff(), st.rhs.as<double>(), Pinv, bl, lower_mode,
                         st.ldS);
    else if (lc > 0) {
      int rc = ba_pair_lists(ctx, st, l0, lc);
      if (rc) return rc;
      hipLaunchKernelGGL(ba_schur_prep_kernel, dim3((lc + 3) / 4), dim3(256), 0, ctx->stream, D, st.lm_start.as<int>(),
                         st.obs_cam.as<int>(), st.cam_free.as<int>(), st.cam_pos.as<int>(), st.r.as<double>(), st.F.as<double>(),
                         st.E.as<double>(), dgl, inv_radius, l0, lc, st.Wg.as<double>(), st.Yg.as<double>(), st.rhs.as<double>(),
                         Pinv, st.bl.as<double>());
      if (D.nfree > 0)
        hipLaunchKernelGGL(ba_schur_rhs_kernel, dim3(D.nfree), dim3(64), 0, ctx->stream, D.nfree, st.free_cams.as<int>(),
                           st.cam_start.as<int>(), st.cam_obs.as<int>(), st.obs_lm.as<int>(), st.Yg.as<double>(),
                           st.bl.as<double>(), l0, lc, st.rhs.as<double>());
      hipLaunchKernelGGL(ba_schur_gather_kernel, dim3(gather_grid(st)), dim3(64), 0, ctx->stream, st.n_slots, st.hbp1,
                         st.pair_start.as<int>(), st.pairs.as<int>(), st.Wg.as<double>(), st.Yg.as<double>(), st.S_eff(),
                         st.ldS, lower_mode);
    }
    hipLaunchKernelGGL(ba_add_cam_blocks_kernel, dim3((D.nfree * 36 + 255) / 256), dim3(256), 0, ctx->stream, D.nfree,
                       st.H.as<double>(), st.g_c.as<double>(), dgc, inv_radius, st.S_eff(), st.rhs.as<double>(), st.ldS,
                       st.banded ? 1 : 0);
  }
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// dc = -(S^-1 rhs), enqueued only: flag[0] = 1 (the finite check clears it), flag[1] = Cholesky succeeded.
// flags_set: the caller's previous kernel has set both flags (saves the launch)
int ba_solve_enqueue(vsl_ctx* ctx, BaState& st, bool flags_set = false) {
  const int n = st.D.n;
  VslStage s(ctx, VSL_STAGE_BA_SOLVE);
  if ((n == 0 || n > 128) && !flags_set) hipLaunchKernelGGL(ba_set_flags_kernel, dim3(1), dim3(64), 0, ctx->stream, st.flag.as<int>());
  if (n == 0) return VSL_OK;
  if (n <= 128) {
    hipLaunchKernelGGL(ba_chol_small_kernel, dim3(1), dim3(256), 0, ctx->stream, n, st.S.as<double>(), st.rhs.as<double>(),
                       st.dc.as<double>(), st.flag.as<int>() + 1, st.flag.as<int>());
  } else {
    // dc = -(S^-1 rhs): the solver writes the negated solution as well
    int rc = vsl_chol_solve_band_dev(ctx, st.S_eff(), st.rhs.as<double>(), n, st.ldS, st.bw, st.flag.as<int>() + 1, st.cyclic ? 1 : 0,
                                     st.dc.as<double>());
    if (rc) return rc;
  }
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// dc = -(S^-1 rhs).  ok = false if S is not positive definite.
int ba_solve(vsl_ctx* ctx, BaState& st, bool& ok) {
  const int n = st.D.n;
  ok = true;
  if (n == 0) return VSL_OK;
  VslStage s(ctx, VSL_STAGE_BA_SOLVE);
  int flag = 1;
  if (n <= 128) {
    hipLaunchKernelGGL(ba_chol_small_kernel, dim3(1), dim3(256), 0, ctx->stream, n,
                       st.S.as<double>(), st.rhs.as<double>(), st.dc.as<double>(), st.flag.as<int>());
    VSL_CHECK_LAUNCH(ctx);
    VSL_HIP(ctx, hipMemcpyAsync(&flag, st.flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    // blocked right-looking Cholesky + substitutions (chol.hip); rhs <- S^-1 rhs
    int rc = vsl_chol_solve_band_dev(ctx, st.S_eff(), st.rhs.as<double>(), n, st.ldS, st.bw, st.flag.as<int>(), st.cyclic ? 1 : 0);
    if (rc) return rc;
    hipLaunchKernelGGL(ba_negate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, st.rhs.as<double>(), st.dc.as<double>());
    VSL_CHECK_LAUNCH(ctx);
    VSL_HIP(ctx, hipMemcpyAsync(&flag, st.flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  ok = flag != 0;
  return VSL_OK;
}

int read_scalars(vsl_ctx* ctx, BaState& st, double* out, int n) {
  VSL_HIP(ctx, hipMemcpyAsync(out, st.scalars.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}

BlArgs bal_args(BaState& st) {
  BlArgs a;
  a.D = st.D;
  a.poses = st.poses.as<double>();
  a.points = st.points.as<double>();
  a.intr = st.intr.as<double>();
  a.cam_intr = st.cam_intr.as<int>();
  a.cam_free = st.cam_free.as<int>();
  a.obs_cam = st.obs_cam.as<int>();
  a.obs_lm = st.obs_lm.as<int>();
  a.obs_uv = st.obs_uv.as<double>();
  a.lm_start = st.lm_start.as<int>();
  a.wg_lm = st.wg_lm.as<int>();
  a.scale_c = st.scale_c.as<double>();
  a.scale_l = st.scale_l.as<double>();
  return a;
}

// recompute form (ba_large.h), the Jacobi-scaling pass: unscaled column norms (st.n2l, diag of st.H) and the cost
int bal_init_pass(vsl_ctx* ctx, BaState& st) {
  const BaDims& D = st.D;
  VslStage s(ctx, VSL_STAGE_BA_LIN);
  const BlArgs a = bal_args(st);
  hipLaunchKernelGGL(bal_prep_kernel<true>, dim3(st.n_wg), dim3(BL_THREADS), 0, ctx->stream, a, (const int*)nullptr, 0.0,
                     (double*)nullptr, (double*)nullptr, (double*)nullptr, (double*)nullptr, st.n2l.as<double>(),
                     st.lpart.as<double>());
  hipLaunchKernelGGL(bal_cam_kernel<true>, dim3(D.nfree, st.bl_seg), dim3(256), 0, ctx->stream, a, st.free_cams.as<int>(),
                     st.cam_start.as<int>(), st.cam_lm.as<int>(), st.cam_uv.as<double>(), (const double*)nullptr, st.cam_part.as<double>());
  hipLaunchKernelGGL(bal_cam_finish_kernel, dim3((D.nfree * 33 + 255) / 256), dim3(256), 0, ctx->stream, D.nfree, st.bl_seg, 0,
                     st.cam_part.as<double>(), st.H.as<double>(), st.g_c.as<double>(), st.rhs.as<double>(), st.n_wg,
                     st.lpart.as<double>(), st.scalars.as<double>(), (double*)nullptr);
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// recompute form: S (landmark damping only, no camera damping), rhs, H, g_c, P^-1, b at the current point;
// scalars[0] = cost, gl_out[0] = max |gradient| over the landmark columns (unscaled problem).
int bal_reduce(vsl_ctx* ctx, BaState& st, double radius, double* gl_out) {
  const BaDims& D = st.D;
  VslStage s(ctx, VSL_STAGE_BA_SCHUR);
  const BlArgs a = bal_args(st);
  VSL_HIP(ctx, hipMemsetAsync(st.S.p, 0, sizeof(double) * st.s_elems, ctx->stream));
  int rc = ba_pair_lists(ctx, st, 0, D.L);
  if (rc) return rc;
  hipLaunchKernelGGL(bal_prep_kernel<false>, dim3(st.n_wg), dim3(BL_THREADS), 0, ctx->stream, a, st.cam_pos.as<int>(),
                     1.0 / radius, st.Yg.as<double>(), st.Pinv.as<double>(), st.bl.as<double>(), st.pbs.as<double>(),
                     (double*)nullptr, st.lpart.as<double>());
  hipLaunchKernelGGL(bal_cam_kernel<false>, dim3(D.nfree, st.bl_seg), dim3(256), 0, ctx->stream, a, st.free_cams.as<int>(),
                     st.cam_start.as<int>(), st.cam_lm.as<int>(), st.cam_uv.as<double>(), st.pbs.as<double>(), st.cam_part.as<double>());
  hipLaunchKernelGGL(bal_cam_finish_kernel, dim3((D.nfree * 33 + 255) / 256), dim3(256), 0, ctx->stream, D.nfree, st.bl_seg, 1,
                     st.cam_part.as<double>(), st.H.as<double>(), st.g_c.as<double>(), st.rhs.as<double>(), st.n_wg,
                     st.lpart.as<double>(), st.scalars.as<double>(), gl_out);
  hipLaunchKernelGGL(ba_schur_gather_kernel, dim3(gather_grid(st)), dim3(64), 0, ctx->stream, st.n_slots, st.hbp1,
                     st.pair_start.as<int>(), st.pairs.as<int>(), st.Yg.as<double>(), st.Yg.as<double>(), st.S_eff(), st.ldS,
                     st.banded ? 2 : (D.n > 128 ? 1 : 0));  // n <= 128 is solved by ba_chol_small_kernel (full matrix)
  // (the camera blocks are added by the caller together with the packing: sess_add_pack_kernel)
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// recompute form: candidate (cand_poses, cand_points) from dc, scalars[2..7] as the operator-by-operator chain leaves them
int bal_step(vsl_ctx* ctx, BaState& st, double* packC_dev) {
  const BaDims& D = st.D;
  VslStage s(ctx, VSL_STAGE_BA_STEP);
  const BlArgs a = bal_args(st);
  hipLaunchKernelGGL(bal_pose_kernel, dim3(1), dim3(1024), 0, ctx->stream, D, st.cam_free.as<int>(), st.poses.as<double>(),
                     st.dc.as<double>(), st.scale_c.as<double>(), st.cand_poses.as<double>(), st.scalars.as<double>(),
                     st.flag.as<int>());
  hipLaunchKernelGGL(bal_step_kernel, dim3(st.n_wg), dim3(BL_THREADS), 0, ctx->stream, a, st.Pinv.as<double>(),
                     st.bl.as<double>(), st.dc.as<double>(), st.cand_poses.as<double>(), st.cand_points.as<double>(),
                     st.lpart.as<double>(), st.flag.as<int>());
  hipLaunchKernelGGL(bal_step_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, st.n_wg, st.lpart.as<double>(),
                     st.scalars.as<double>(), st.flag.as<int>(), packC_dev);
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

}  // namespace

int vsl_ba_chol_small_launch(vsl_ctx* ctx, int n, const double* S, const double* rhs, double* dc, int* ok_flag) {
  hipLaunchKernelGGL(ba_chol_small_kernel, dim3(1), dim3(256), 0, ctx->stream, n, S, rhs, dc, ok_flag, (int*)nullptr);
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

extern "C" int vsl_ba_residuals_jacobians(vsl_ctx* ctx, const vsl_ba_problem* prob, double* r, double* J_pose,
                                          double* J_point) {
  int rc = ba_validate(ctx, prob);
  if (rc) return rc;
  if (!r || !J_pose || !J_point) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ba_residuals_jacobians: null output");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  BaState st;
  if ((rc = ba_setup(ctx, prob, nullptr, st))) return rc;
  const BaDims& D = st.D;
  hipLaunchKernelGGL(ba_raw_blocks_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D, st.poses.as<double>(),
                     st.points.as<double>(), st.intr.as<double>(), st.cam_intr.as<int>(), st.obs_cam.as<int>(),
                     st.obs_lm.as<int>(), st.obs_uv.as<double>(), st.r.as<double>(), st.F.as<double>(), st.E.as<double>());
  VSL_CHECK_LAUNCH(ctx);
  std::vector<double> hr(2 * (size_t)D.O), hF(12 * (size_t)D.O), hE(6 * (size_t)D.O);
  VSL_HIP(ctx, hipMemcpyAsync(hr.data(), st.r.p, 8 * hr.size(), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(hF.data(), st.F.p, 8 * hF.size(), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(hE.data(), st.E.p, 8 * hE.size(), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int q = 0; q < D.O; q++) {  // back to the caller's observation order
    const size_t i = st.perm.empty() ? (size_t)q : (size_t)st.perm[q];
    memcpy(r + 2 * i, &hr[2 * (size_t)q], 16);
    memcpy(J_pose + 12 * i, &hF[12 * (size_t)q], 96);
    memcpy(J_point + 6 * i, &hE[6 * (size_t)q], 48);
  }
  return VSL_OK;
}

extern "C" int vsl_ba_linearize(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, int lm_first,
                                int lm_count, double* S, double* g, double* cost, int* n_free) {
  int rc = ba_validate(ctx, prob);
  if (rc) return rc;
  if (!opt || !S || !g || !cost || !n_free) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ba_linearize: null argument");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  BaState st;
  if ((rc = ba_setup(ctx, prob, opt, st))) return rc;
  const BaDims& D = st.D;
  int l0 = 0, lc = D.L;
  if (lm_count >= 0) {
    if (lm_first < 0 || lm_first > D.L) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ba_linearize: lm_first out of range");
    l0 = lm_first;
    lc = std::min(lm_count, D.L - lm_first);
  }
  if ((rc = ba_linearize(ctx, st, false))) return rc;
  if (lm_count >= 0) {
    // restrict to the observations of the landmark range: zero the others' blocks so that the camera
    // sums and the cost only see the range (observations are sorted by landmark => one contiguous run)
    std::vector<int> lm_start(D.L + 1);
    VSL_HIP(ctx, hipMemcpyAsync(lm_start.data(), st.lm_start.p, sizeof(int) * lm_start.size(), hipMemcpyDeviceToHost, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t o0 = (size_t)lm_start[l0], o1 = (size_t)lm_start[l0 + lc];
    if (o0 > 0) {
      VSL_HIP(ctx, hipMemsetAsync(st.r.p, 0, 16 * o0, ctx->stream));
      VSL_HIP(ctx, hipMemsetAsync(st.F.p, 0, 96 * o0, ctx->stream));
      VSL_HIP(ctx, hipMemsetAsync(st.E.p, 0, 48 * o0, ctx->stream));
    }
    if (o1 < (size_t)D.O) {
      VSL_HIP(ctx, hipMemsetAsync(st.r.as<double>() + 2 * o1, 0, 16 * ((size_t)D.O - o1), ctx->stream));
      VSL_HIP(ctx, hipMemsetAsync(st.F.as<double>() + 12 * o1, 0, 96 * ((size_t)D.O - o1), ctx->stream));
      VSL_HIP(ctx, hipMemsetAsync(st.E.as<double>() + 6 * o1, 0, 48 * ((size_t)D.O - o1), ctx->stream));
    }
    const int oc = (int)(o1 - o0);
    const int nb = (oc + 255) / 256;
    if (nb > 0)
      hipLaunchKernelGGL(ba_cost_kernel, dim3(nb), dim3(256), 0, ctx->stream, D, st.poses.as<double>(), st.points.as<double>(),
                         st.intr.as<double>(), st.cam_intr.as<int>(), st.obs_cam.as<int>(), st.obs_lm.as<int>(),
                         st.obs_uv.as<double>(), (int)o0, oc, st.partials.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>(), nb, st.scalars.as<double>(), 0, 0);
    VSL_CHECK_LAUNCH(ctx);
  }
  if ((rc = ba_columns(ctx, st))) return rc;
  if ((rc = ba_schur(ctx, st, false, 1.0, l0, lc, false, false))) return rc;
  double sc[1];
  if ((rc = read_scalars(ctx, st, sc, 1))) return rc;
  *cost = sc[0];
  *n_free = D.nfree;
  if (D.n > 0) {
    VSL_HIP(ctx, hipMemcpy(S, st.S.p, sizeof(double) * (size_t)D.n * D.n, hipMemcpyDeviceToHost));
    VSL_HIP(ctx, hipMemcpy(g, st.rhs.p, sizeof(double) * D.n, hipMemcpyDeviceToHost));
  }
  return VSL_OK;
}

// [upstream] ceres::Solve, TRUST_REGION / LEVENBERG_MARQUARDT / Schur, defaults (see oracle/orc_ba.cpp
// for the option values this loop restates).
extern "C" int vsl_bundle_adjust(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt,
                                 vsl_ba_summary* summary) {
  int rc = ba_validate(ctx, prob);
  if (rc) return rc;
  if (!opt) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bundle_adjust: options are null");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  static const bool env_no_fused = getenv("VSL_BA_NO_FUSED") != nullptr;  // same switch for applications without a diagnostics hook
  if (!ctx->ba_no_fused && !env_no_fused) {  // local windows: the fused iteration (ba_fused.hip); everything else continues below
    int handled = 0;
    rc = vsl_ba_fused_solve(ctx, prob, opt, summary, &handled);
    if (rc || handled) return rc;
    // Large maps (more free cameras than the fused window takes: the reduced system goes through the pair-list gather
    // and the band solver): the session solver at world size 1 -- the recompute-form iteration of ba_large.h, the cyclic
    // band form, one host round trip per iteration -- the same LM policy on the same numbers
    // (tests/test_ba_dist_gpu.py::test_session_world1_matches_single_call_and_oracle).  What follows below remains for
    // mid-size windows the fused kernels decline (more than 64 cameras, a landmark seen more than 64 times), for
    // "ba_no_fused" and as the cross-check of both.
    int nfree = 0;
    for (int c = 0; c < prob->n_cams; c++) nfree += prob->cam_fixed[c] ? 0 : 1;
    if (6 * nfree > 128) return vsl_global_bundle_adjust(ctx, prob, opt, nullptr, nullptr, 0, 1, summary);
  }
  const double t_start = now_ms();
  BaState st;
  st.want_alt_set = true;
  if ((rc = ba_setup(ctx, prob, opt, st, true))) return rc;
  const BaDims& D = st.D;
  const int nc = D.n, nl = 3 * D.L;
  vsl_ba_summary sum;
  memset(&sum, 0, sizeof(sum));
  // per-stage device times (summary.linearize_ms / schur_ms / solve_ms) only when the context has profiling switched
  // on (vsl_ctx_set_profiling): the HIP events around every stage cost ~60 us per LM iteration of a local window
  // (0.36 -> 0.30 ms per iteration), so a plain call goes without them
  const bool prof_was = ctx->profiling;
  double base_ms[3];
  for (int k = 0; k < 3; k++) base_ms[k] = ctx->stage_ms[VSL_STAGE_BA_LIN + k];

  double sc[8];
  // iteration 0: evaluate, Jacobi scaling from the unscaled Jacobian, then scale it
  if ((rc = ba_linearize(ctx, st, false))) return rc;
  if ((rc = ba_columns(ctx, st))) return rc;
  const int nmax = std::max(nc, nl);
  hipLaunchKernelGGL(ba_make_scale_kernel, dim3((nmax + 255) / 256), dim3(256), 0, ctx->stream, D.nfree, D.L,
                     st.H.as<double>(), st.n2l.as<double>(), st.scale_c.as<double>(), st.scale_l.as<double>());
  hipLaunchKernelGGL(ba_apply_scale_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D.O, st.cam_free.as<int>(),
                     st.obs_cam.as<int>(), st.obs_lm.as<int>(), st.scale_c.as<double>(), st.scale_l.as<double>(),
                     st.F.as<double>(), st.E.as<double>());
  VSL_CHECK_LAUNCH(ctx);
  if ((rc = ba_columns(ctx, st))) return rc;

  auto diag_and_gmax = [&](int slot) -> int {
    hipLaunchKernelGGL(ba_diag_kernel, dim3((nmax + 255) / 256), dim3(256), 0, ctx->stream, D.nfree, D.L, st.H.as<double>(),
                       st.n2l.as<double>(), st.g_c.as<double>(), st.grad_l.as<double>(), st.scale_c.as<double>(),
                       st.scale_l.as<double>(), st.diag_c.as<double>(), st.diag_l.as<double>(), st.gabs.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.gabs.as<double>(), (nmax + 255) / 256,
                       st.scalars.as<double>(), slot, 1);
    VSL_CHECK_LAUNCH(ctx);
    return VSL_OK;
  };
  if ((rc = diag_and_gmax(1))) return rc;
  if ((rc = read_scalars(ctx, st, sc, 2))) return rc;
  double cost = sc[0], gmax = sc[1];
  sum.initial_cost = cost;

  double radius = 1e4, decrease_factor = 2.0;
  int iteration = 0, invalid = 0;
  sum.termination = 0;
  if (opt->verbosity >= 2)
    fprintf(stderr, "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e\n", 0, cost);
  // One host round trip per LM iteration: the whole iteration is enqueued without waiting -- Schur, Cholesky,
  // back-substitution, model / step norms, candidate parameters AND the linearisation at the candidate (into the
  // second set of blocks: its cost is the candidate's cost, and if the step is accepted it is the next iteration's
  // linearisation already) -- then ONE copy brings back [Cholesky ok, finite] + 7 scalars and the host applies the
  // [upstream] Ceres step policy.  A rejected or invalid step swaps the sets back; its speculative work (~60 us of
  // device time) is the price.  (The first version synchronised three times per iteration: after the Cholesky, after
  // the candidate cost, after the re-linearisation.)
  double* hsc = nullptr;
  {
    void* hp = nullptr;
    if ((rc = vsl_ctx_hpinned(ctx, 256, &hp))) return rc;
    hsc = (double*)hp;
  }
  const int* hflag = (const int*)(hsc + 16);
  while (true) {
    if (iteration >= opt->max_num_iterations) { sum.termination = 0; break; }
    if (gmax <= 1e-10) { sum.termination = 2; break; }
    if (radius <= 1e-32) { sum.termination = 4; break; }
    iteration++;
    if ((rc = ba_schur(ctx, st, true, radius, 0, D.L, true, true))) return rc;
    if ((rc = ba_solve_enqueue(ctx, st))) return rc;  // flag[1] = Cholesky ok
    {
      VslStage s(ctx, VSL_STAGE_BA_SOLVE);
      hipLaunchKernelGGL(ba_backsub_kernel, dim3((D.L + 255) / 256), dim3(256), 0, ctx->stream, D, st.lm_start.as<int>(),
                         st.obs_cam.as<int>(), st.cam_free.as<int>(), st.F.as<double>(), st.E.as<double>(),
                         st.Pinv.as<double>(), st.bl.as<double>(), st.dc.as<double>(), st.dl.as<double>(), st.flag.as<int>());
      hipLaunchKernelGGL(ba_model_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D, st.obs_cam.as<int>(),
                         st.obs_lm.as<int>(), st.cam_free.as<int>(), st.r.as<double>(), st.F.as<double>(), st.E.as<double>(),
                         st.dc.as<double>(), st.dl.as<double>(), st.partials.as<double>());
      hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>(), st.nb_obs,
                         st.scalars.as<double>(), 2, 0);
      hipLaunchKernelGGL(ba_update_kernel, dim3(st.nb_upd), dim3(256), 0, ctx->stream, D, st.cam_free.as<int>(),
                         st.poses.as<double>(), st.points.as<double>(), st.dc.as<double>(), st.dl.as<double>(),
                         st.scale_c.as<double>(), st.scale_l.as<double>(), st.cand_poses.as<double>(),
                         st.cand_points.as<double>(), st.partials.as<double>(), st.nb_upd);
      hipLaunchKernelGGL(ba_reduce2_kernel, dim3(2), dim3(256), 0, ctx->stream, st.partials.as<double>(), st.nb_upd,
                         st.scalars.as<double>(), 3);
      VSL_CHECK_LAUNCH(ctx);
    }
    // speculative: the candidate becomes the current point, its linearisation goes to the other set;
    // scalars[5] = cost there, scalars[6] = max |gradient| there (slots 0/1 keep the current point's values)
    st.swap_sets();
    if ((rc = ba_linearize(ctx, st, true, 5))) return rc;
    if ((rc = ba_columns(ctx, st))) return rc;
    if ((rc = diag_and_gmax(6))) return rc;
    VSL_HIP(ctx, hipMemcpyAsync(hsc, st.scalars.p, 16 * sizeof(double) + 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double model_change = hsc[2], step_norm = sqrt(hsc[3]), x_norm = sqrt(hsc[4]), cand_cost = hsc[5];
    const bool ok = hflag[0] != 0 && hflag[1] != 0 && model_change > 0.0;
    if (!ok) {
      st.swap_sets();
      if (++invalid >= 5) { sum.termination = 4; break; }
      radius *= 0.5;
      if (opt->verbosity >= 2) fprintf(stderr, "%4d  invalid step, radius %.3e\n", iteration, radius);
      continue;  // the LM diagonal is reused (the Jacobian is unchanged)
    }
    invalid = 0;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { st.swap_sets(); sum.termination = 3; break; }
    const double cost_change = cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * cost) { st.swap_sets(); sum.termination = 1; break; }
    const double rel = cost_change / model_change;
    if (opt->verbosity >= 2)
      fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", iteration, cand_cost, cost_change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      cost = cand_cost;   // the sets stay swapped: the speculative linearisation is the current one
      gmax = hsc[6];
      sum.successful_steps++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0;
    } else {
      st.swap_sets();
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
  }
  sum.iterations = iteration;
  sum.final_cost = cost;
  VSL_HIP(ctx, hipMemcpyAsync(prob->poses, st.poses.p, sizeof(double) * 7 * (size_t)D.C, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(prob->points, st.points.p, sizeof(double) * 3 * (size_t)D.L, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double ms;
  int64_t cnt;
  vsl_ctx_stage_ms(ctx, VSL_STAGE_BA_LIN, &ms, &cnt);
  sum.linearize_ms = ctx->stage_ms[VSL_STAGE_BA_LIN] - base_ms[0];
  sum.schur_ms = ctx->stage_ms[VSL_STAGE_BA_SCHUR] - base_ms[1];
  sum.solve_ms = ctx->stage_ms[VSL_STAGE_BA_SOLVE] - base_ms[2];
  vsl_ctx_set_profiling(ctx, prof_was ? 1 : 0);
  sum.total_ms = now_ms() - t_start;
  if (opt->verbosity >= 1)
    fprintf(stderr, "vsl BA: iterations %d, initial cost %.6e, final cost %.6e, termination %d, %.3f ms\n", sum.iterations,
            sum.initial_cost, sum.final_cost, sum.termination, sum.total_ms);
  if (summary) *summary = sum;
  return VSL_OK;
}

// =================================================================================================
// Step-wise session: the multi-GPU global-BA path.
//
// One process per GPU.  Every rank holds all camera poses and OWNS a contiguous landmark range with
// its observations (the session is built on that sub-problem).  Per LM iteration the ranks exchange
//   packB = [ S_part (n*n) | rhs_part (n) | diag(H_part) (n) | g_c part (n) | cost_part | 0 ]  SUM all-reduce
//   packC = [ bad, model_part, step2_lm, x2_lm, cand_cost_part ]                  SUM all-reduce
// (plus one MAX all-reduce of the landmark gradient norm after an accepted step, and one SUM of
// [diag(H_part) | cost_part] for the Jacobi scaling at iteration 0).  Every rank then factorises the
// same reduced camera system redundantly -- RCCL all-reduce leaves bit-identical buffers on all ranks,
// so the accept / reject decisions agree without a broadcast.  Landmark damping and back-substitution
// are local.  The host-side loop lives in visual-slam_amd/ba_dist.py (torch.distributed = RCCL).
struct vsl_ba_session {
  vsl_ctx* ctx = nullptr;
  BaState st;
  vsl_ba_options opt;
  int lm_first = 0, lm_count = 0, n_lms_total = 0;
  DevBuf diagc_keep;  // clamp(diag H_full): reused across rejected steps
  bool solo = false;  // vsl_ba_session_solve without a collective: S stays where it is (no copy into packB and back)
};

namespace {
__global__ void sess_pack_hdiag_kernel(int nfree, const double* __restrict__ H, const double* __restrict__ scalars,
                                       double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = 6 * nfree;
  if (i < n) out[i] = H[36 * (size_t)(i / 6) + 7 * (i % 6)];
  if (i == 0) out[n] = scalars[0];
}

__global__ void sess_scale_kernel(int nfree, int L, const double* __restrict__ hdiag_full, const double* __restrict__ n2l,
                                  double* __restrict__ scale_c, double* __restrict__ scale_l) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 6 * nfree) scale_c[i] = 1.0 / (1.0 + sqrt(hdiag_full[i]));
  if (i < 3 * L) scale_l[i] = 1.0 / (1.0 + sqrt(n2l[i]));
}

// landmark LM diagonal (own landmarks) and |gradient| of the unscaled problem for the landmark columns
__global__ void sess_diag_l_kernel(int L, const double* __restrict__ n2l, const double* __restrict__ grad_l,
                                   const double* __restrict__ scale_l, double* __restrict__ diag_l, double* __restrict__ gabs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * L) {
    diag_l[i] = fmin(fmax(n2l[i], 1e-6), 1e32);
    gabs[i] = fabs(grad_l[i] / scale_l[i]);
  }
}

// packB tail after the n*n block: [rhs_part | diag(H_part) | g_c part (raw sum F^T r) | cost_part | 0]
__global__ void sess_pack_b_kernel(int nfree, const double* __restrict__ rhs, const double* __restrict__ H,
                                   const double* __restrict__ g_c, const double* __restrict__ scalars,
                                   double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = 6 * nfree;
  if (i < n) {
    out[i] = rhs[i];
    out[n + i] = H[36 * (size_t)(i / 6) + 7 * (i % 6)];
    out[2 * n + i] = g_c[i];
  }
  if (i == 0) {
    out[3 * n] = scalars[0];
    out[3 * n + 1] = 0.0;
  }
}

// ba_add_cam_blocks_kernel (no camera damping) and sess_pack_b_kernel in one launch: S += blockdiag(H), rhs += g_c, and
// the tail of packB from the sums
__global__ void sess_add_pack_kernel(int nfree, const double* __restrict__ H, const double* __restrict__ g_c,
                                     const double* __restrict__ scalars, double* __restrict__ S, double* __restrict__ rhs, int ldS,
                                     int lower_elems, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = 6 * nfree;
  if (t < nfree * 36) {
    const int fc = t / 36, x = (t % 36) / 6, y = t % 6;
    if (!(lower_elems && y > x)) S[(size_t)(6 * fc + x) * ldS + 6 * fc + y] += H[t];
  }
  if (t < n) {
    const double r = rhs[t] + g_c[t];
    rhs[t] = r;
    out[t] = r;
    out[n + t] = H[36 * (size_t)(t / 6) + 7 * (t % 6)];
    out[2 * n + t] = g_c[t];
  }
  if (t == 0) {
    out[3 * n] = scalars[0];
    out[3 * n + 1] = 0.0;
  }
}

// S = S_full + diag(diag_c / radius); diag_c = clamp(diag H_full) when refresh, else kept.  S_full (the first
// `elems` doubles of packB, dense or band layout) has been copied into S already; this adds the damping to the
// diagonal (entry (i, i) at S_eff[i * ldS + i]) and unpacks rhs.
__global__ void sess_damp_kernel(int n, size_t elems, const double* __restrict__ packB, double inv_radius, int refresh,
                                 double* __restrict__ diag_keep, double* __restrict__ S_eff, int ldS, double* __restrict__ rhs,
                                 int* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (flags && i < 2) flags[i] = 1;  // (step finite / factorisation succeeded: what ba_set_flags_kernel would set)
  if (i < n) {
    double d = diag_keep[i];
    if (refresh) {
      d = fmin(fmax(packB[elems + n + i], 1e-6), 1e32);
      diag_keep[i] = d;
    }
    S_eff[(size_t)i * ldS + i] += d * inv_radius;
    rhs[i] = packB[elems + i];
  }
}

__global__ void sess_pack_c_kernel(const double* __restrict__ scalars, const int* __restrict__ flag, int both,
                                   double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    // number of ranks whose step is unusable (both: flag[0] = step finite, flag[1] = factorisation succeeded)
    out[0] = (flag[0] && (!both || flag[1])) ? 0.0 : 1.0;
    out[1] = scalars[2];           // model cost change, own observations
    out[2] = scalars[3];           // squared step norm (own landmarks + cameras, see ba_dist.py)
    out[3] = scalars[4];           // squared x norm   (own landmarks + cameras)
    out[4] = scalars[5];           // candidate cost, own observations
    out[5] = scalars[6];           // squared step norm of the cameras alone (replicated on every rank)
    out[6] = scalars[7];           // squared x norm of the cameras alone
    out[7] = 0.0;
  }
}

// camera-only parts of the step / x norms (identical on every rank)
__global__ __launch_bounds__(256) void sess_cam_norms_kernel(BaDims D, const int* __restrict__ cam_free,
                                                             const double* __restrict__ poses, const double* __restrict__ dc,
                                                             const double* __restrict__ scale_c, double* __restrict__ scalars) {
  __shared__ double sh[256];
  double step2 = 0, x2 = 0;
  for (int c = threadIdx.x; c < D.C; c += 256) {
    const int fc = cam_free[c];
    if (fc < 0) continue;
    for (int j = 0; j < 6; j++) {
      const double d = dc[6 * fc + j] * scale_c[6 * fc + j];
      step2 += d * d;
    }
    for (int j = 0; j < 7; j++) x2 += poses[7 * (size_t)c + j] * poses[7 * (size_t)c + j];
  }
  const double a = block_sum_256(step2, sh);
  __syncthreads();
  const double b = block_sum_256(x2, sh);
  if (threadIdx.x == 0) {
    scalars[6] = a;
    scalars[7] = b;
  }
}
}  // namespace

extern "C" int vsl_ba_session_create(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, int lm_first,
                                     int lm_count, vsl_ba_session** out) {
  int rc = ba_validate(ctx, prob);
  if (rc) return rc;
  if (!opt || !out) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ba_session_create: null argument");
  *out = nullptr;
  if (lm_first < 0 || lm_count < 1 || lm_first + lm_count > prob->n_lms)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ba_session_create: landmark range [%d, %d) must be non-empty and inside [0, %d)", lm_first, lm_first + lm_count, prob->n_lms);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  BaTrace tr;
  // sub-problem of the owned landmarks (all cameras)
  std::vector<int32_t> ocam, olm;
  std::vector<double> ouv;
  const bool whole = lm_first == 0 && lm_count == prob->n_lms;  // one rank: the problem itself, no copy
  if (!whole) {
    for (int i = 0; i < prob->n_obs; i++) {
      const int l = prob->obs_lm[i];
      if (l >= lm_first && l < lm_first + lm_count) {
        ocam.push_back(prob->obs_cam[i]);
        olm.push_back(l - lm_first);
        ouv.push_back(prob->obs_uv[2 * (size_t)i]);
        ouv.push_back(prob->obs_uv[2 * (size_t)i + 1]);
      }
    }
    if (ocam.empty()) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ba_session_create: landmark range has no observations");
  }
  tr.lap("session: sub-problem");
  vsl_ba_problem sub = *prob;
  if (!whole) {
    sub.n_lms = lm_count;
    sub.n_obs = (int32_t)ocam.size();
    sub.points = prob->points + 3 * (size_t)lm_first;
    sub.obs_cam = ocam.data();
    sub.obs_lm = olm.data();
    sub.obs_uv = ouv.data();
  }
  vsl_ba_session* s = new (std::nothrow) vsl_ba_session;
  if (!s) return vsl_fail(ctx, VSL_ERR_NOMEM, "out of host memory");
  s->ctx = ctx;
  s->opt = *opt;
  s->lm_first = lm_first;
  s->lm_count = lm_count;
  s->n_lms_total = prob->n_lms;
  if ((rc = ba_setup(ctx, &sub, opt, s->st, true, prob))) {  // band order from the FULL problem: identical on every rank
    delete s;
    return rc;
  }
  {
    // the recompute-form iteration (ba_large.h) for large systems in gather form; "ba_no_fused" / VSL_BA_NO_FUSED keep
    // the operator-by-operator chain over stored r / F / E blocks (A/B runs, tests)
    static const bool env_no_fused = getenv("VSL_BA_NO_FUSED") != nullptr;
    BaState& st = s->st;
    st.large_fused = !st.small && st.n_wg > 0 && st.D.nfree > 0 && !ctx->ba_no_fused && !env_no_fused &&
                     !ctx->ba_schur_atomics && st.n_pairs_cap < ((size_t)1 << 31);
    if (st.large_fused) {
      hipLaunchKernelGGL(bal_cam_major_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, st.D.O, st.cam_obs.as<int>(),
                         st.obs_lm.as<int>(), st.obs_uv.as<double>(), st.cam_lm.as<int>(), st.cam_uv.as<double>());
    }
  }
  if (s->diagc_keep.alloc(8 * (size_t)(s->st.D.n > 0 ? s->st.D.n : 1)) != hipSuccess) {
    delete s;
    return vsl_fail(ctx, VSL_ERR_NOMEM, "device allocation failed");
  }
  *out = s;
  return VSL_OK;
}

extern "C" int vsl_ba_session_destroy(vsl_ba_session* s) {
  if (!s) return VSL_OK;
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  delete s;
  return VSL_OK;
}

extern "C" int vsl_ba_session_dims(const vsl_ba_session* s, int* n, int* n_lms_own, int* n_obs_own, int* n_cams) {
  if (!s) return VSL_ERR_INVALID;
  if (n) *n = s->st.D.n;
  if (n_lms_own) *n_lms_own = s->st.D.L;
  if (n_obs_own) *n_obs_own = s->st.D.O;
  if (n_cams) *n_cams = s->st.D.C;
  return VSL_OK;
}

// Layout of the reduced camera system inside packB: *s_elems doubles (n * n dense; n * (ld + 1) + 64 in band form,
// where the cameras were renumbered into band order -- identical on every rank), then rhs / diag H / g_c / cost.
extern "C" int vsl_ba_session_layout(const vsl_ba_session* s, int64_t* s_elems, int* banded, int* bandwidth) {
  if (!s) return VSL_ERR_INVALID;
  if (s_elems) *s_elems = (int64_t)s->st.s_elems;
  if (banded) *banded = s->st.cyclic ? 2 : (s->st.banded ? 1 : 0);
  if (bandwidth) *bandwidth = s->st.bw;
  return VSL_OK;
}

// Linearise the owned observations at the current parameters (Jacobi-scaled when use_scale) and
// compute the per-landmark / per-camera column statistics.
extern "C" int vsl_ba_session_linearize(vsl_ba_session* s, int use_scale) {
  if (!s) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  if (s->st.large_fused)  // nothing is stored per observation: vsl_ba_session_reduce_dev evaluates at the current point
    return use_scale ? VSL_OK : bal_init_pass(ctx, s->st);
  int rc = ba_linearize(ctx, s->st, use_scale != 0);
  if (rc) return rc;
  return ba_columns(ctx, s->st);
}

// out_dev[n + 1] = [diag(H_part) | cost_part]   (device pointer; asynchronous on the context's stream)
extern "C" int vsl_ba_session_hdiag_cost_dev(vsl_ba_session* s, double* out_dev) {
  if (!s || !out_dev) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  const BaDims& D = s->st.D;
  hipLaunchKernelGGL(sess_pack_hdiag_kernel, dim3((D.n + 256) / 256), dim3(256), 0, ctx->stream, D.nfree, s->st.H.as<double>(),
                     s->st.scalars.as<double>(), out_dev);
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// Jacobi scaling from the all-reduced diag(H) (cameras) and the owned landmark column norms; scales the
// stored Jacobian blocks and refreshes the column statistics.
extern "C" int vsl_ba_session_set_scale_dev(vsl_ba_session* s, const double* hdiag_full_dev) {
  if (!s || !hdiag_full_dev) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  BaState& st = s->st;
  const BaDims& D = st.D;
  const int nmax = std::max(D.n, 3 * D.L);
  hipLaunchKernelGGL(sess_scale_kernel, dim3((nmax + 255) / 256), dim3(256), 0, ctx->stream, D.nfree, D.L, hdiag_full_dev,
                     st.n2l.as<double>(), st.scale_c.as<double>(), st.scale_l.as<double>());
  VSL_CHECK_LAUNCH(ctx);
  if (st.large_fused) return VSL_OK;
  hipLaunchKernelGGL(ba_apply_scale_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D.O, st.cam_free.as<int>(),
                     st.obs_cam.as<int>(), st.obs_lm.as<int>(), st.scale_c.as<double>(), st.scale_l.as<double>(),
                     st.F.as<double>(), st.E.as<double>());
  VSL_CHECK_LAUNCH(ctx);
  return ba_columns(ctx, st);
}

// packB_dev[n*n + 3n + 2] = [S_part | rhs_part | diag(H_part) | g_c part | cost_part | 0]: Schur complement of the
// owned landmarks with THEIR damping (diag_l / radius) plus this rank's camera blocks, no camera damping.
// gmax_l_dev[1] = max |gradient| over the owned landmark columns (unscaled problem).
extern "C" int vsl_ba_session_reduce_dev(vsl_ba_session* s, double radius, double* packB_dev, double* gmax_l_dev) {
  if (!s || !packB_dev || !(radius > 0)) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  BaState& st = s->st;
  const BaDims& D = st.D;
  const int n = D.n;
  int rc;
  if (st.large_fused) {
    if ((rc = bal_reduce(ctx, st, radius, gmax_l_dev))) return rc;
  } else {
    hipLaunchKernelGGL(sess_diag_l_kernel, dim3((3 * D.L + 255) / 256), dim3(256), 0, ctx->stream, D.L, st.n2l.as<double>(),
                       st.grad_l.as<double>(), st.scale_l.as<double>(), st.diag_l.as<double>(), st.gabs.as<double>());
    if (gmax_l_dev) {
      hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.gabs.as<double>(), 3 * D.L, gmax_l_dev, 0, 1);
    }
    VSL_CHECK_LAUNCH(ctx);
    // Schur with landmark damping only: reuse ba_schur with damping, but with a zero camera diagonal
    VSL_HIP(ctx, hipMemsetAsync(st.diag_c.p, 0, sizeof(double) * (size_t)(n > 0 ? n : 1), ctx->stream));
    if ((rc = ba_schur(ctx, st, true, radius, 0, D.L, true, true))) return rc;
  }
  if (n > 0 && st.large_fused) {
    hipLaunchKernelGGL(sess_add_pack_kernel, dim3((D.nfree * 36 + 255) / 256), dim3(256), 0, ctx->stream, D.nfree, st.H.as<double>(),
                       st.g_c.as<double>(), st.scalars.as<double>(), st.S_eff(), st.rhs.as<double>(), st.ldS, st.banded ? 1 : 0,
                       packB_dev + st.s_elems);
    VSL_CHECK_LAUNCH(ctx);
    if (!s->solo) VSL_HIP(ctx, hipMemcpyAsync(packB_dev, st.S.p, sizeof(double) * st.s_elems, hipMemcpyDeviceToDevice, ctx->stream));
  } else if (n > 0) {
    if (!s->solo) VSL_HIP(ctx, hipMemcpyAsync(packB_dev, st.S.p, sizeof(double) * st.s_elems, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(sess_pack_b_kernel, dim3((n + 256) / 256), dim3(256), 0, ctx->stream, D.nfree, st.rhs.as<double>(),
                       st.H.as<double>(), st.g_c.as<double>(), st.scalars.as<double>(), packB_dev + st.s_elems);
    VSL_CHECK_LAUNCH(ctx);
  }
  return VSL_OK;
}

// From the all-reduced packB: damp the cameras, solve, back-substitute the owned landmarks, build the
// candidate, and report packC_dev[8] (see sess_pack_c_kernel).  refresh_diag = 1 after an accepted step
// (or at the first iteration), 0 when the Jacobian is unchanged (LM reuses its diagonal).
extern "C" int vsl_ba_session_step_dev(vsl_ba_session* s, const double* packB_full_dev, double radius, int refresh_diag,
                                       double* packC_dev) {
  if (!s || !packB_full_dev || !packC_dev || !(radius > 0)) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  BaState& st = s->st;
  const BaDims& D = st.D;
  const int n = D.n, nl = 3 * D.L;
  if (n > 0) {
    if (!s->solo) VSL_HIP(ctx, hipMemcpyAsync(st.S.p, packB_full_dev, sizeof(double) * st.s_elems, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(sess_damp_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, st.s_elems, packB_full_dev, 1.0 / radius,
                       refresh_diag, s->diagc_keep.as<double>(), st.S_eff(), st.ldS, st.rhs.as<double>(),
                       st.large_fused ? st.flag.as<int>() : (int*)nullptr);
    VSL_CHECK_LAUNCH(ctx);
  }
  int rc;
  if (st.large_fused) {
    // everything is enqueued, nothing is read back here: a failed factorisation leaves flag[1] = 0 and numbers nobody
    // uses (the caller's one read of packC per iteration sees the step as unusable)
    if ((rc = ba_solve_enqueue(ctx, st, n > 0))) return rc;
    if ((rc = bal_step(ctx, st, packC_dev))) return rc;  // (packC written by the step's last kernel)
    return VSL_OK;
  }
  bool ok = true;
  if ((rc = ba_solve(ctx, st, ok))) return rc;
  const int okflag = ok ? 1 : 0;
  VSL_HIP(ctx, hipMemcpyAsync(st.flag.p, &okflag, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ok) {
    hipLaunchKernelGGL(ba_backsub_kernel, dim3((D.L + 255) / 256), dim3(256), 0, ctx->stream, D, st.lm_start.as<int>(),
                       st.obs_cam.as<int>(), st.cam_free.as<int>(), st.F.as<double>(), st.E.as<double>(), st.Pinv.as<double>(),
                       st.bl.as<double>(), st.dc.as<double>(), st.dl.as<double>());
    if (n > 0) hipLaunchKernelGGL(ba_all_finite_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, st.dc.as<double>(), st.flag.as<int>());
    hipLaunchKernelGGL(ba_all_finite_kernel, dim3((nl + 255) / 256), dim3(256), 0, ctx->stream, nl, st.dl.as<double>(), st.flag.as<int>());
    hipLaunchKernelGGL(ba_model_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D, st.obs_cam.as<int>(), st.obs_lm.as<int>(),
                       st.cam_free.as<int>(), st.r.as<double>(), st.F.as<double>(), st.E.as<double>(), st.dc.as<double>(),
                       st.dl.as<double>(), st.partials.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>(), st.nb_obs, st.scalars.as<double>(), 2, 0);
    hipLaunchKernelGGL(ba_update_kernel, dim3(st.nb_upd), dim3(256), 0, ctx->stream, D, st.cam_free.as<int>(), st.poses.as<double>(),
                       st.points.as<double>(), st.dc.as<double>(), st.dl.as<double>(), st.scale_c.as<double>(), st.scale_l.as<double>(),
                       st.cand_poses.as<double>(), st.cand_points.as<double>(), st.partials.as<double>(), st.nb_upd);
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>(), st.nb_upd, st.scalars.as<double>(), 3, 0);
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>() + st.nb_upd, st.nb_upd, st.scalars.as<double>(), 4, 0);
    hipLaunchKernelGGL(sess_cam_norms_kernel, dim3(1), dim3(256), 0, ctx->stream, D, st.cam_free.as<int>(), st.poses.as<double>(),
                       st.dc.as<double>(), st.scale_c.as<double>(), st.scalars.as<double>());
    hipLaunchKernelGGL(ba_cost_kernel, dim3(st.nb_obs), dim3(256), 0, ctx->stream, D, st.cand_poses.as<double>(), st.cand_points.as<double>(),
                       st.intr.as<double>(), st.cam_intr.as<int>(), st.obs_cam.as<int>(), st.obs_lm.as<int>(), st.obs_uv.as<double>(), 0,
                       D.O, st.partials.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, st.partials.as<double>(), st.nb_obs, st.scalars.as<double>(), 5, 0);
    VSL_CHECK_LAUNCH(ctx);
  }
  hipLaunchKernelGGL(sess_pack_c_kernel, dim3(1), dim3(64), 0, ctx->stream, st.scalars.as<double>(), st.flag.as<int>(), 0, packC_dev);
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

// The candidate becomes the current estimate.
extern "C" int vsl_ba_session_accept(vsl_ba_session* s) {
  if (!s) return VSL_ERR_INVALID;
  std::swap(s->st.poses.p, s->st.cand_poses.p);
  std::swap(s->st.points.p, s->st.cand_points.p);
  return VSL_OK;
}

// poses[7 * n_cams] (all cameras) and points_own[3 * lm_count] (the owned range), host pointers.
extern "C" int vsl_ba_session_download(vsl_ba_session* s, double* poses, double* points_own) {
  if (!s) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  if (poses) VSL_HIP(ctx, hipMemcpyAsync(poses, s->st.poses.p, sizeof(double) * 7 * (size_t)s->st.D.C, hipMemcpyDeviceToHost, ctx->stream));
  if (points_own) VSL_HIP(ctx, hipMemcpyAsync(points_own, s->st.points.p, sizeof(double) * 3 * (size_t)s->st.D.L, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}

// =================================================================================================
// The Levenberg-Marquardt loop over a session, in C++ (host code of the multi-GPU global bundle adjustment).
// Collectives go through ONE caller-supplied function -- ncclAllReduce on the context's stream for RCCL
// (include/visnav_amd/bundle_adjustment.h), a host hop for the gloo tests (visual-slam_amd/ba_dist.py) -- so the loop
// itself does not depend on a communication library.  Policy = the [upstream] Ceres policy of vsl_bundle_adjust.
// Per iteration: SUM of packB (the packed partial reduced camera system, band form when the cameras order into a band:
// ~6 MB in the cyclic band form, ~12 MB in the linear one, instead of 287 MB at 1000 cameras), MAX of one scalar after an accepted step, SUM of the 8 doubles of packC.
namespace {
__global__ __launch_bounds__(1024) void sess_gmax_c_kernel(int n, const double* __restrict__ g_c, const double* __restrict__ scale_c,
                                                           const double* __restrict__ cost_in, const double* __restrict__ gl,
                                                           double* __restrict__ out) {
  // out[0] = cost (copied), out[1] = max(max_i |g_c[i] / scale_c[i]|, gl[0])
  __shared__ double sh[1024];
  double m = 0;
  for (int i = threadIdx.x; i < n; i += 1024) m = fmax(m, fabs(g_c[i] / scale_c[i]));
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = cost_in[0];
    out[1] = fmax(sh[0], gl[0]);
  }
}
}  // namespace

extern "C" int vsl_ba_session_solve(vsl_ba_session* s, vsl_allreduce_fn allreduce, void* user, int world, int max_iters,
                                    int verbosity, double* poses_out, double* points_all_out, vsl_ba_summary* summary) {
  if (!s || world < 1 || (world > 1 && !allreduce)) return VSL_ERR_INVALID;
  vsl_ctx* ctx = s->ctx;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  BaState& st = s->st;
  const int n = st.D.n;
  const size_t elems = st.s_elems, nB = elems + 3 * (size_t)n + 2;
  const double t_start = now_ms();
  DevBuf bufA, packB, packC, gl, gather;
  if (bufA.alloc(8 * ((size_t)n + 1)) != hipSuccess || packB.alloc(8 * nB) != hipSuccess || packC.alloc(80) != hipSuccess ||
      gl.alloc(8) != hipSuccess)
    return vsl_fail(ctx, VSL_ERR_NOMEM, "vsl_ba_session_solve: device allocation failed");
  auto AR = [&](double* buf, size_t count, int op) -> int {
    if (!allreduce) return VSL_OK;  // a caller that passes a callback at world 1 gets its (trivial) collectives: tests
    const int rc = allreduce(user, buf, (int64_t)count, op, (void*)ctx->stream);
    return rc ? vsl_fail(ctx, VSL_ERR_HIP, "all-reduce callback failed (%d)", rc) : VSL_OK;
  };
  auto D2H = [&](void* dst, const void* src, size_t bytes) -> int {
    VSL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VSL_OK;
  };
  int rc;
  s->solo = !allreduce;
  vsl_ba_summary sum;
  memset(&sum, 0, sizeof(sum));
  VSL_HIP(ctx, hipMemsetAsync(gl.p, 0, 8, ctx->stream));
  // iteration 0: cost, Jacobi scaling from the global column norms
  if ((rc = vsl_ba_session_linearize(s, 0))) return rc;
  if ((rc = vsl_ba_session_hdiag_cost_dev(s, bufA.as<double>()))) return rc;
  if ((rc = AR(bufA.as<double>(), (size_t)n + 1, 0))) return rc;
  if ((rc = vsl_ba_session_set_scale_dev(s, bufA.as<double>()))) return rc;
  double h2[2];
  if ((rc = D2H(h2, bufA.as<double>() + n, 8))) return rc;
  sum.initial_cost = h2[0];
  double radius = 1e4, decrease = 2.0, cost = sum.initial_cost, gmax = INFINITY;
  int it = 0, invalid = 0, refresh = 1;
  bool have_h2 = false;
  double* const hostpack = packC.as<double>() + 8;  // [cost | max |gradient|] behind the 8 doubles of packC: one copy brings both
  sum.termination = 0;
  if (verbosity >= 2) fprintf(stderr, "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e\n", 0, cost);
  while (true) {
    if ((rc = vsl_ba_session_reduce_dev(s, radius, packB.as<double>(), gl.as<double>()))) return rc;
    if ((rc = AR(packB.as<double>(), nB, 0))) return rc;
    if (refresh) {
      if ((rc = AR(gl.as<double>(), 1, 1))) return rc;
      hipLaunchKernelGGL(sess_gmax_c_kernel, dim3(1), dim3(1024), 0, ctx->stream, n, packB.as<double>() + elems + 2 * (size_t)n,
                         st.scale_c.as<double>(), packB.as<double>() + elems + 3 * (size_t)n, gl.as<double>(), hostpack);
      VSL_CHECK_LAUNCH(ctx);
      have_h2 = false;  // (cost, |gradient|) of this linearisation: read together with the step's verdict below --
                        // ONE host round trip per iteration; a gradient below tolerance is found one step late, and
                        // that step is dropped
    }
    if (it >= max_iters) {
      if (!have_h2) {
        if ((rc = D2H(h2, hostpack, 16))) return rc;
        cost = h2[0];
        gmax = h2[1];
      }
      sum.termination = 0;
      break;
    }
    if (have_h2 && gmax <= 1e-10) { sum.termination = 2; break; }
    if (radius <= 1e-32) { sum.termination = 4; break; }
    it++;
    if ((rc = vsl_ba_session_step_dev(s, packB.as<double>(), radius, refresh, packC.as<double>()))) return rc;
    if ((rc = AR(packC.as<double>(), 8, 0))) return rc;
    double c[10];
    if ((rc = D2H(c, packC.p, 80))) return rc;
    if (!have_h2) {
      cost = c[8];
      gmax = c[9];
      have_h2 = true;
      if (gmax <= 1e-10) {
        it--;
        sum.termination = 2;
        break;
      }
    }
    const double cams_step2 = c[5] / world, cams_x2 = c[6] / world;
    const bool ok = c[0] == 0.0 && c[1] > 0.0;
    if (!ok) {
      if (++invalid >= 5) { sum.termination = 4; break; }
      radius *= 0.5;
      refresh = 0;
      continue;
    }
    invalid = 0;
    const double step_norm = sqrt(std::max(c[2] - (world - 1) * cams_step2, 0.0));
    const double x_norm = sqrt(std::max(c[3] - (world - 1) * cams_x2, 0.0));
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 3; break; }
    const double cost_change = cost - c[4];
    if (fabs(cost_change) <= 1e-6 * cost) { sum.termination = 1; break; }
    const double rel = cost_change / c[1];
    if (verbosity >= 2) fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", it, c[4], cost_change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      if ((rc = vsl_ba_session_accept(s))) return rc;
      if ((rc = vsl_ba_session_linearize(s, 1))) return rc;
      refresh = 1;
      sum.successful_steps++;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
      decrease = 2.0;
    } else {
      radius /= decrease;
      decrease *= 2.0;
      refresh = 0;
    }
  }
  sum.iterations = it;
  sum.final_cost = cost;
  if (poses_out) {
    VSL_HIP(ctx, hipMemcpyAsync(poses_out, st.poses.p, sizeof(double) * 7 * (size_t)st.D.C, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (points_all_out) {
    // every rank's landmarks: a zero buffer with the own range filled in, summed over the ranks
    const size_t total = 3 * (size_t)s->n_lms_total;
    if (gather.alloc(8 * total) != hipSuccess) return vsl_fail(ctx, VSL_ERR_NOMEM, "vsl_ba_session_solve: device allocation failed");
    VSL_HIP(ctx, hipMemsetAsync(gather.p, 0, 8 * total, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(gather.as<double>() + 3 * (size_t)s->lm_first, st.points.p, sizeof(double) * 3 * (size_t)st.D.L,
                                hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = AR(gather.as<double>(), total, 0))) return rc;
    VSL_HIP(ctx, hipMemcpyAsync(points_all_out, gather.p, 8 * total, hipMemcpyDeviceToHost, ctx->stream));
  }
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  sum.total_ms = now_ms() - t_start;
  if (verbosity >= 1)
    fprintf(stderr, "vsl global BA (%d rank%s, %s system, bandwidth %d of %d): iterations %d, initial cost %.6e, final cost %.6e, termination %d, %.3f ms\n",
            world, world > 1 ? "s" : "", st.cyclic ? "cyclic band" : (st.banded ? "band" : "dense"), st.bw, n, sum.iterations, sum.initial_cost, sum.final_cost,
            sum.termination, sum.total_ms);
  if (summary) *summary = sum;
  return VSL_OK;
}

// global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748) over `world` ranks: landmarks are split into
// contiguous ranges balanced by observation count, rank `rank` owns one; poses / points of `prob` are updated in place
// on every rank.  world = 1 (allreduce may be null) is the single-GPU session path.
extern "C" int vsl_global_bundle_adjust(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, vsl_allreduce_fn allreduce,
                                        void* user, int rank, int world, vsl_ba_summary* summary) {
  int rc = ba_validate(ctx, prob);
  if (rc) return rc;
  if (!opt || world < 1 || rank < 0 || rank >= world) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_global_bundle_adjust: bad arguments");
  // contiguous landmark ranges balanced by observation count (the same rule as visual-slam_amd/dist.py landmark_ranges)
  std::vector<int64_t> csum(prob->n_lms + 1, 0);
  for (int i = 0; i < prob->n_obs; i++) csum[prob->obs_lm[i] + 1]++;
  for (int l = 0; l < prob->n_lms; l++) csum[l + 1] += csum[l];
  std::vector<int> cuts(world + 1, 0);
  for (int r = 1; r < world; r++) {
    const double target = (double)csum[prob->n_lms] * r / world;
    cuts[r] = (int)(std::lower_bound(csum.begin(), csum.end(), target, [](int64_t v, double t) { return (double)v < t; }) - csum.begin());
    cuts[r] = std::min(std::max(cuts[r], cuts[r - 1]), prob->n_lms);
  }
  cuts[world] = prob->n_lms;
  // Every rank computes ALL ranges, so decisions about them are identical everywhere (a rank that returned alone
  // would leave the others waiting in the first all-reduce).  A range emptied by the balancing rule (few, heavy
  // landmarks) is widened to one landmark; fewer landmarks than ranks is an error on every rank alike.
  if (prob->n_lms < world)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_global_bundle_adjust: %d landmarks cannot be split over %d ranks (every rank fails alike)",
                    prob->n_lms, world);
  for (int r = 1; r < world; r++) cuts[r] = std::min(std::max(cuts[r], cuts[r - 1] + 1), prob->n_lms - (world - r));
  const int first = cuts[rank], count = cuts[rank + 1] - cuts[rank];
  vsl_ba_session* s = nullptr;
  rc = vsl_ba_session_create(ctx, prob, opt, first, count, &s);
  if (world > 1 && allreduce) {
    // rank-local failures (allocation, a bad range) are agreed on BEFORE the first data collective: MAX of a flag
    double* flag = ctx->status_word;  // allocated with the context: never null, so the collective is always entered
    int frc = 0;
    const double mine = rc ? 1.0 : 0.0;
    double any = mine;
    if (hipMemcpyAsync(flag, &mine, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) frc = VSL_ERR_HIP;
    // (a rank that cannot stage the flag still enters the collective with whatever the word holds: it is about to
    // fail anyway and must not leave the others hanging)
    const int arc = allreduce(user, flag, 1, 1, (void*)ctx->stream);
    if (!arc && !frc && hipMemcpyAsync(&any, flag, 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess)
      (void)hipStreamSynchronize(ctx->stream);
    if (rc || frc || arc || any != 0.0) {
      if (s) vsl_ba_session_destroy(s);
      if (rc) return rc;  // this rank's own message is already in place
      return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_global_bundle_adjust: set-up failed on %s (all ranks leave together)",
                      (frc || arc) ? "this rank's status exchange" : "another rank");
    }
  } else if (rc) {
    return rc;
  }
  rc = vsl_ba_session_solve(s, allreduce, user, world, opt->max_num_iterations, opt->verbosity, prob->poses, prob->points, summary);
  vsl_ba_session_destroy(s);
  return rc;
}

extern "C" int vsl_ctx_last_ba_layout(vsl_ctx* ctx, int64_t* s_elems, int* banded, int* bandwidth) {
  if (!ctx) return VSL_ERR_INVALID;
  if (s_elems) *s_elems = ctx->last_ba_s_elems;
  if (banded) *banded = ctx->last_ba_banded;
  if (bandwidth) *bandwidth = ctx->last_ba_bw;
  return VSL_OK;
}

// Plain copies for callers that hold device pointers of this library (the all-reduce callbacks of the tests):
// kind 0 host->device, 1 device->host, 2 device->device; synchronous.
extern "C" int vsl_ctx_memcpy(vsl_ctx* ctx, void* dst, const void* src, size_t bytes, int kind) {
  if (!ctx || !dst || !src || kind < 0 || kind > 2) return VSL_ERR_INVALID;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
  VSL_HIP(ctx, hipMemcpyAsync(dst, src, bytes, k, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}

// =================================================================================================
// bundle_adjustment with BundleAdjustmentOptions::optimize_intrinsics = true (include/visnav/map_utils.h:324, :397-403:
// the two 8-parameter intrinsics blocks are NOT set constant).  The reference never enables it (a hidden GUI variable,
// src/slam.cpp:304, :1545), so this path is written for correctness, not for the last microsecond: its own plain
// Levenberg-Marquardt loop (several host round trips per iteration) over the same restated Ceres policy, the reduced
// system [poses (6 per free camera) | intrinsics (2 x 8)] assembled densely with fp64 atomics -- the camera-camera part
// by the large-system kernel above, the intrinsics border by the kernels below -- and solved by the dense Cholesky.
// The unused trailing parameters of a model have zero Jacobian columns, exactly like Ceres' size-8 block: the LM
// diagonal floor (1e-6 / radius) keeps the system definite and they keep their values.
namespace {

// d(u, v) / d intrinsics at the camera-frame point (x, y, z): Gu[8], Gv[8] (camera_models.h project() of the four models)
__device__ __forceinline__ void project_intr_jac(int model, const double* __restrict__ ip, double x, double y, double z,
                                                 double* Gu, double* Gv) {
  const double fx = ip[0], fy = ip[1];
  for (int j = 0; j < 8; j++) Gu[j] = Gv[j] = 0.0;
  Gu[2] = 1.0;
  Gv[3] = 1.0;
  if (model == VSL_CAM_PINHOLE) {
    Gu[0] = x / z;
    Gv[1] = y / z;
  } else if (model == VSL_CAM_EUCM) {
    const double alpha = ip[4], beta = ip[5];
    const double rho2 = x * x + y * y;
    const double d = sqrt(beta * rho2 + z * z);
    const double den = alpha * d + (1.0 - alpha) * z;
    const double id = 1.0 / den, id2 = id * id;
    Gu[0] = x * id;
    Gv[1] = y * id;
    const double dden_da = d - z, dden_db = alpha * rho2 / (2.0 * d);
    Gu[4] = -fx * x * dden_da * id2; Gv[4] = -fy * y * dden_da * id2;
    Gu[5] = -fx * x * dden_db * id2; Gv[5] = -fy * y * dden_db * id2;
  } else if (model == VSL_CAM_KB4) {
    const double k1 = ip[4], k2 = ip[5], k3 = ip[6], k4 = ip[7];
    const double r = sqrt(x * x + y * y);
    if (r == 0.0) return;  // u = cx, v = cy
    const double th = atan2(r, z);
    const double t2 = th * th, t3 = t2 * th, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
    const double d = th + k1 * t3 + k2 * t5 + k3 * t7 + k4 * t9;
    const double xr = x / r, yr = y / r;
    Gu[0] = d * xr;
    Gv[1] = d * yr;
    Gu[4] = fx * xr * t3; Gv[4] = fy * yr * t3;
    Gu[5] = fx * xr * t5; Gv[5] = fy * yr * t5;
    Gu[6] = fx * xr * t7; Gv[6] = fy * yr * t7;
    Gu[7] = fx * xr * t9; Gv[7] = fy * yr * t9;
  } else {  // double sphere
    const double xi = ip[4], alpha = ip[5];
    const double d1 = sqrt(x * x + y * y + z * z);
    const double k = xi * d1 + z;
    const double d2 = sqrt(x * x + y * y + k * k);
    const double den = alpha * d2 + (1.0 - alpha) * k;
    const double id = 1.0 / den, id2 = id * id;
    Gu[0] = x * id;
    Gv[1] = y * id;
    const double dden_dxi = alpha * (k * d1 / d2) + (1.0 - alpha) * d1, dden_da = d2 - k;
    Gu[4] = -fx * x * dden_dxi * id2; Gv[4] = -fy * y * dden_dxi * id2;
    Gu[5] = -fx * x * dden_da * id2;  Gv[5] = -fy * y * dden_da * id2;
  }
}

// one thread per observation: robustified residual and the three Jacobian blocks F (2x6, pose tangent), G (2x8,
// intrinsics), E (2x3, landmark), Jacobi-scaled when `scale` (n camera columns followed by 16 intrinsics columns) is given
__global__ __launch_bounds__(256) void bai_linearize_kernel(BaDims D, const double* __restrict__ poses,
                                                            const double* __restrict__ points, const double* __restrict__ intr,
                                                            const int* __restrict__ cam_intr, const int* __restrict__ cam_free,
                                                            const int* __restrict__ obs_cam, const int* __restrict__ obs_lm,
                                                            const double* __restrict__ obs_uv, const double* __restrict__ scale,
                                                            const double* __restrict__ scale_l, double* __restrict__ r_out,
                                                            double* __restrict__ F_out, double* __restrict__ E_out,
                                                            double* __restrict__ G_out, double* __restrict__ partials) {
  __shared__ double sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double c = 0;
  if (i < D.O) {
    const int cam = obs_cam[i], lm = obs_lm[i], k = cam_intr[cam], model = k ? D.model1 : D.model0;
    const double* pose = poses + 7 * (size_t)cam;
    const double* pw = points + 3 * (size_t)lm;
    double r[2], F[12], E[6], Gu[8], Gv[8];
    residual_blocks(model, intr + 8 * k, pose, pw, obs_uv + 2 * (size_t)i, r, F, E, true);
    {
      double R[9];
      quat_R(pose, R);
      const double d[3] = {pw[0] - pose[4], pw[1] - pose[5], pw[2] - pose[6]};
      project_intr_jac(model, intr + 8 * k, R[0] * d[0] + R[3] * d[1] + R[6] * d[2], R[1] * d[0] + R[4] * d[1] + R[7] * d[2],
                       R[2] * d[0] + R[5] * d[1] + R[8] * d[2], Gu, Gv);
    }
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho0 = s, rho1 = 1.0;
    if (D.use_huber) huber(s, D.huber, rho0, rho1);
    c = 0.5 * rho0;
    const double sr = sqrt(rho1);
    r_out[2 * (size_t)i] = r[0] * sr;
    r_out[2 * (size_t)i + 1] = r[1] * sr;
    const int fc = cam_free[cam];
    for (int j = 0; j < 6; j++) {
      const double sc = (scale && fc >= 0) ? scale[6 * fc + j] : 1.0;
      F_out[12 * (size_t)i + j] = F[j] * sr * sc;
      F_out[12 * (size_t)i + 6 + j] = F[6 + j] * sr * sc;
    }
    for (int j = 0; j < 8; j++) {  // residual = p_2d - projection
      const double sc = scale ? scale[D.n + 8 * k + j] : 1.0;
      G_out[16 * (size_t)i + j] = -Gu[j] * sr * sc;
      G_out[16 * (size_t)i + 8 + j] = -Gv[j] * sr * sc;
    }
    for (int j = 0; j < 3; j++) {
      const double sc = scale_l ? scale_l[3 * (size_t)lm + j] : 1.0;
      E_out[6 * (size_t)i + j] = E[j] * sr * sc;
      E_out[6 * (size_t)i + 3 + j] = E[3 + j] * sr * sc;
    }
  }
  const double t = block_sum_256(c, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// squared column norms and gradient J^T r of all columns: cameras + intrinsics in n2 / grad (n + 16), landmarks in
// n2l / gradl (3 L).  Outputs pre-zeroed.  The 32 intrinsics sums go through LDS first.
__global__ __launch_bounds__(256) void bai_stats_kernel(BaDims D, const int* __restrict__ cam_free, const int* __restrict__ cam_intr,
                                                        const int* __restrict__ obs_cam, const int* __restrict__ obs_lm,
                                                        const double* __restrict__ r, const double* __restrict__ F,
                                                        const double* __restrict__ E, const double* __restrict__ G,
                                                        double* __restrict__ n2, double* __restrict__ grad,
                                                        double* __restrict__ n2l, double* __restrict__ gradl) {
  __shared__ double acc[32];
  if (threadIdx.x < 32) acc[threadIdx.x] = 0.0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D.O) {
    const int cam = obs_cam[i], fc = cam_free[cam], k = cam_intr[cam], lm = obs_lm[i];
    const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    const double* g = G + 16 * (size_t)i;
    if (fc >= 0)
      for (int j = 0; j < 6; j++) {
        unsafeAtomicAdd(&n2[6 * fc + j], f[j] * f[j] + f[6 + j] * f[6 + j]);
        unsafeAtomicAdd(&grad[6 * fc + j], f[j] * r0 + f[6 + j] * r1);
      }
    for (int j = 0; j < 8; j++) {
      unsafeAtomicAdd(&acc[8 * k + j], g[j] * g[j] + g[8 + j] * g[8 + j]);
      unsafeAtomicAdd(&acc[16 + 8 * k + j], g[j] * r0 + g[8 + j] * r1);
    }
    for (int j = 0; j < 3; j++) {
      unsafeAtomicAdd(&n2l[3 * (size_t)lm + j], e[j] * e[j] + e[3 + j] * e[3 + j]);
      unsafeAtomicAdd(&gradl[3 * (size_t)lm + j], e[j] * r0 + e[3 + j] * r1);
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) unsafeAtomicAdd(&n2[D.n + threadIdx.x], acc[threadIdx.x]);
  else if (threadIdx.x < 32) unsafeAtomicAdd(&grad[D.n + threadIdx.x - 16], acc[threadIdx.x]);
}

__global__ void bai_make_scale_kernel(int nt, int L3, const double* __restrict__ n2, const double* __restrict__ n2l,
                                      double* __restrict__ scale, double* __restrict__ scale_l) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nt) scale[i] = 1.0 / (1.0 + sqrt(n2[i]));
  if (i < L3) scale_l[i] = 1.0 / (1.0 + sqrt(n2l[i]));
}

__global__ void bai_apply_scale_kernel(BaDims D, const int* __restrict__ cam_free, const int* __restrict__ cam_intr,
                                       const int* __restrict__ obs_cam, const int* __restrict__ obs_lm,
                                       const double* __restrict__ scale, const double* __restrict__ scale_l,
                                       double* __restrict__ F, double* __restrict__ E, double* __restrict__ G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.O) return;
  const int cam = obs_cam[i], fc = cam_free[cam], k = cam_intr[cam], lm = obs_lm[i];
  if (fc >= 0)
    for (int j = 0; j < 6; j++) {
      F[12 * (size_t)i + j] *= scale[6 * fc + j];
      F[12 * (size_t)i + 6 + j] *= scale[6 * fc + j];
    }
  for (int j = 0; j < 8; j++) {
    G[16 * (size_t)i + j] *= scale[D.n + 8 * k + j];
    G[16 * (size_t)i + 8 + j] *= scale[D.n + 8 * k + j];
  }
  for (int j = 0; j < 3; j++) {
    E[6 * (size_t)i + j] *= scale_l[3 * (size_t)lm + j];
    E[6 * (size_t)i + 3 + j] *= scale_l[3 * (size_t)lm + j];
  }
}

// LM diagonal clamp(||column||^2) of all columns and max |gradient| (one workgroup; scalars[slot] = the maximum)
__global__ __launch_bounds__(256) void bai_diag_gmax_kernel(int nt, int L3, const double* __restrict__ n2, const double* __restrict__ n2l,
                                                            const double* __restrict__ grad, const double* __restrict__ gradl,
                                                            int write_diag, double* __restrict__ diag, double* __restrict__ diag_l,
                                                            double* __restrict__ scalars, int slot) {
  __shared__ double sh[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < nt; i += 256) {
    if (write_diag) diag[i] = fmin(fmax(n2[i], 1e-6), 1e32);
    m = fmax(m, fabs(grad[i]));
  }
  for (int i = threadIdx.x; i < L3; i += 256) {
    if (write_diag) diag_l[i] = fmin(fmax(n2l[i], 1e-6), 1e32);
    m = fmax(m, fabs(gradl[i]));
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) scalars[slot] = sh[0];
}

// J^T J of the camera side before the Schur correction, and J^T r: pose-pose diagonal blocks, pose-intrinsics blocks
// (both triangles of the full matrix), intrinsics-intrinsics blocks (through LDS), into S (nt x nt) and rhs (nt)
__global__ __launch_bounds__(256) void bai_hess_kernel(BaDims D, int nt, const int* __restrict__ cam_free,
                                                       const int* __restrict__ cam_intr, const int* __restrict__ obs_cam,
                                                       const double* __restrict__ r, const double* __restrict__ F,
                                                       const double* __restrict__ G, double* __restrict__ S,
                                                       double* __restrict__ rhs) {
  __shared__ double hii[2][64];
  __shared__ double gi[16];
  if (threadIdx.x < 128) hii[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
  if (threadIdx.x < 16) gi[threadIdx.x] = 0.0;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D.O) {
    const int cam = obs_cam[i], fc = cam_free[cam], k = cam_intr[cam];
    const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
    const double* f = F + 12 * (size_t)i;
    const double* g = G + 16 * (size_t)i;
    const int ci = D.n + 8 * k;
    if (fc >= 0) {
      for (int x = 0; x < 6; x++) {
        unsafeAtomicAdd(&rhs[6 * fc + x], f[x] * r0 + f[6 + x] * r1);
        for (int y = 0; y < 6; y++)
          unsafeAtomicAdd(&S[(size_t)(6 * fc + x) * nt + 6 * fc + y], f[x] * f[y] + f[6 + x] * f[6 + y]);
        for (int j = 0; j < 8; j++) {
          const double v = f[x] * g[j] + f[6 + x] * g[8 + j];
          if (v != 0.0) {
            unsafeAtomicAdd(&S[(size_t)(6 * fc + x) * nt + ci + j], v);
            unsafeAtomicAdd(&S[(size_t)(ci + j) * nt + 6 * fc + x], v);
          }
        }
      }
    }
    for (int a = 0; a < 8; a++) {
      unsafeAtomicAdd(&gi[8 * k + a], g[a] * r0 + g[8 + a] * r1);
      for (int b = 0; b < 8; b++) {
        const double v = g[a] * g[b] + g[8 + a] * g[8 + b];
        if (v != 0.0) unsafeAtomicAdd(&hii[k][8 * a + b], v);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x >> 6, e = threadIdx.x & 63, a = e >> 3, b = e & 7;
    const double v = hii[k][e];
    if (v != 0.0) unsafeAtomicAdd(&S[(size_t)(D.n + 8 * k + a) * nt + D.n + 8 * k + b], v);
  } else if (threadIdx.x < 144) {
    unsafeAtomicAdd(&rhs[D.n + threadIdx.x - 128], gi[threadIdx.x - 128]);
  }
}

// Schur correction of the intrinsics border, one wavefront per landmark (P^-1 and b of the landmark come from the
// camera-camera kernel): W_i = sum_o G_o^T E_o (16 x 3), T = P^-1 W_i^T (3 x 16, kept for the back-substitution);
//   S_ii -= W_i T,  rhs_i -= T^T b,  S_ci(camera of o) -= (F_o^T E_o) T  (and its mirror)
__global__ __launch_bounds__(256) void bai_border_kernel(BaDims D, int nt, const int* __restrict__ lm_start,
                                                         const int* __restrict__ obs_cam, const int* __restrict__ cam_free,
                                                         const int* __restrict__ cam_intr, const double* __restrict__ F,
                                                         const double* __restrict__ E, const double* __restrict__ G,
                                                         const double* __restrict__ Pinv, const double* __restrict__ bl,
                                                         double* __restrict__ T_out, double* __restrict__ S,
                                                         double* __restrict__ rhs) {
  __shared__ double Wi[4][16][3];
  __shared__ double Ts[4][3][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l = blockIdx.x * 4 + wave;
  if (l >= D.L) return;
  const int a = lm_start[l], b = lm_start[l + 1];
  const double* Pi = Pinv + 9 * (size_t)l;
  if (lane < 48) {
    const int col = lane / 3, y = lane - 3 * col, k = col >> 3, j = col & 7;
    double s = 0.0;
    for (int i = a; i < b; i++) {
      if (cam_intr[obs_cam[i]] != k) continue;
      const double* g = G + 16 * (size_t)i;
      const double* e = E + 6 * (size_t)i;
      s += g[j] * e[y] + g[8 + j] * e[3 + y];
    }
    Wi[wave][col][y] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane < 48) {
    const int y = lane >> 4, col = lane & 15;
    const double t = Pi[3 * y] * Wi[wave][col][0] + Pi[3 * y + 1] * Wi[wave][col][1] + Pi[3 * y + 2] * Wi[wave][col][2];
    Ts[wave][y][col] = t;
    T_out[48 * (size_t)l + 16 * y + col] = t;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < 256; e += 64) {
    const int c1 = e >> 4, c2 = e & 15;
    const double v = Wi[wave][c1][0] * Ts[wave][0][c2] + Wi[wave][c1][1] * Ts[wave][1][c2] + Wi[wave][c1][2] * Ts[wave][2][c2];
    if (v != 0.0) unsafeAtomicAdd(&S[(size_t)(D.n + c1) * nt + D.n + c2], -v);
  }
  if (lane < 16) {
    const double v = Ts[wave][0][lane] * bl[3 * (size_t)l] + Ts[wave][1][lane] * bl[3 * (size_t)l + 1] + Ts[wave][2][lane] * bl[3 * (size_t)l + 2];
    if (v != 0.0) unsafeAtomicAdd(&rhs[D.n + lane], -v);
  }
  for (int item = lane; item < (b - a) * 96; item += 64) {
    const int q = item / 96, rem = item - 96 * q, x = rem >> 4, c = rem & 15, i = a + q;
    const int fc = cam_free[obs_cam[i]];
    if (fc < 0) continue;
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    double v = 0.0;
    for (int z = 0; z < 3; z++) v += (f[x] * e[z] + f[6 + x] * e[3 + z]) * Ts[wave][z][c];
    if (v != 0.0) {
      unsafeAtomicAdd(&S[(size_t)(6 * fc + x) * nt + D.n + c], -v);
      unsafeAtomicAdd(&S[(size_t)(D.n + c) * nt + 6 * fc + x], -v);
    }
  }
}

__global__ void bai_damp_kernel(int nt, const double* __restrict__ diag, double inv_radius, double* __restrict__ S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nt) S[(size_t)i * nt + i] += diag[i] * inv_radius;
}

// delta_l = -P^-1 (b_l + sum_o E_o^T (F_o dc + G_o di)); d = [dc (n) | di (16)]
__global__ __launch_bounds__(256) void bai_backsub_kernel(BaDims D, const int* __restrict__ lm_start, const int* __restrict__ obs_cam,
                                                          const int* __restrict__ cam_free, const int* __restrict__ cam_intr,
                                                          const double* __restrict__ F, const double* __restrict__ E,
                                                          const double* __restrict__ G, const double* __restrict__ Pinv,
                                                          const double* __restrict__ bl, const double* __restrict__ d,
                                                          double* __restrict__ dl) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= D.L) return;
  double t[3] = {bl[3 * (size_t)l], bl[3 * (size_t)l + 1], bl[3 * (size_t)l + 2]};
  for (int i = lm_start[l]; i < lm_start[l + 1]; i++) {
    const int cam = obs_cam[i], fc = cam_free[cam], k = cam_intr[cam];
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    const double* g = G + 16 * (size_t)i;
    double m0 = 0, m1 = 0;
    if (fc >= 0)
      for (int j = 0; j < 6; j++) {
        m0 += f[j] * d[6 * fc + j];
        m1 += f[6 + j] * d[6 * fc + j];
      }
    for (int j = 0; j < 8; j++) {
      m0 += g[j] * d[D.n + 8 * k + j];
      m1 += g[8 + j] * d[D.n + 8 * k + j];
    }
    for (int j = 0; j < 3; j++) t[j] += e[j] * m0 + e[3 + j] * m1;
  }
  const double* Pi = Pinv + 9 * (size_t)l;
  for (int j = 0; j < 3; j++) dl[3 * (size_t)l + j] = -(Pi[3 * j] * t[0] + Pi[3 * j + 1] * t[1] + Pi[3 * j + 2] * t[2]);
}

__global__ __launch_bounds__(256) void bai_model_kernel(BaDims D, const int* __restrict__ obs_cam, const int* __restrict__ obs_lm,
                                                        const int* __restrict__ cam_free, const int* __restrict__ cam_intr,
                                                        const double* __restrict__ r, const double* __restrict__ F,
                                                        const double* __restrict__ E, const double* __restrict__ G,
                                                        const double* __restrict__ d, const double* __restrict__ dl,
                                                        double* __restrict__ partials) {
  __shared__ double sh[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v = 0;
  if (i < D.O) {
    const int cam = obs_cam[i], fc = cam_free[cam], k = cam_intr[cam], lm = obs_lm[i];
    const double* f = F + 12 * (size_t)i;
    const double* e = E + 6 * (size_t)i;
    const double* g = G + 16 * (size_t)i;
    double m0 = 0, m1 = 0;
    if (fc >= 0)
      for (int j = 0; j < 6; j++) {
        m0 += f[j] * d[6 * fc + j];
        m1 += f[6 + j] * d[6 * fc + j];
      }
    for (int j = 0; j < 8; j++) {
      m0 += g[j] * d[D.n + 8 * k + j];
      m1 += g[8 + j] * d[D.n + 8 * k + j];
    }
    for (int j = 0; j < 3; j++) {
      m0 += e[j] * dl[3 * (size_t)lm + j];
      m1 += e[3 + j] * dl[3 * (size_t)lm + j];
    }
    v = -(m0 * (r[2 * (size_t)i] + m0 / 2.0) + m1 * (r[2 * (size_t)i + 1] + m1 / 2.0));
  }
  const double t = block_sum_256(v, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// candidate intrinsics = intrinsics + step .* scale; scalars[slot] = squared step norm, scalars[slot + 1] = squared norm
// of the current intrinsics (both blocks are non-constant parameter blocks)
__global__ void bai_intr_update_kernel(int n, const double* __restrict__ intr, const double* __restrict__ d,
                                       const double* __restrict__ scale, double* __restrict__ cand_intr,
                                       double* __restrict__ scalars, int slot) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s2 = 0, x2 = 0;
  for (int j = 0; j < 16; j++) {
    const double dd = d[n + j] * scale[n + j];
    s2 += dd * dd;
    x2 += intr[j] * intr[j];
    cand_intr[j] = intr[j] + dd;
  }
  scalars[slot] = s2;
  scalars[slot + 1] = x2;
}

}  // namespace

extern "C" int vsl_bundle_adjust_intrinsics(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, double* intr_io,
                                            vsl_ba_summary* summary) {
  int rc = ba_validate(ctx, prob);
  if (rc) return rc;
  if (!opt || !intr_io) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bundle_adjust_intrinsics: null argument");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const double t_start = now_ms();
  vsl_ba_problem p2 = *prob;
  p2.intr = intr_io;
  BaState st;
  if ((rc = ba_setup(ctx, &p2, opt, st))) return rc;
  const BaDims& D = st.D;
  const int n = D.n, nt = n + 16, L3 = 3 * D.L;
  DevBuf G, scale, n2, grad, diag, Sf, rhsf, df, cand_intr, Tl;
  BA_HIP(G.alloc(8 * 16 * (size_t)D.O));
  BA_HIP(scale.alloc(8 * (size_t)nt));
  BA_HIP(n2.alloc(8 * (size_t)nt));
  BA_HIP(grad.alloc(8 * (size_t)nt));
  BA_HIP(diag.alloc(8 * (size_t)nt));
  BA_HIP(Sf.alloc(8 * (size_t)nt * nt));
  BA_HIP(rhsf.alloc(8 * (size_t)nt));
  BA_HIP(df.alloc(8 * (size_t)nt));
  BA_HIP(cand_intr.alloc(8 * 16));
  BA_HIP(Tl.alloc(8 * 48 * (size_t)D.L));
  vsl_ba_summary sum;
  memset(&sum, 0, sizeof(sum));
  hipStream_t q = ctx->stream;
  const int nbo = st.nb_obs, nbu = st.nb_upd;
  double* scal = st.scalars.as<double>();

  auto linearize = [&](bool scaled) -> int {  // at the CURRENT point; scalars[0] = cost
    hipLaunchKernelGGL(bai_linearize_kernel, dim3(nbo), dim3(256), 0, q, D, st.poses.as<double>(), st.points.as<double>(),
                       st.intr.as<double>(), st.cam_intr.as<int>(), st.cam_free.as<int>(), st.obs_cam.as<int>(),
                       st.obs_lm.as<int>(), st.obs_uv.as<double>(), scaled ? scale.as<double>() : (const double*)nullptr,
                       scaled ? st.scale_l.as<double>() : (const double*)nullptr, st.r.as<double>(), st.F.as<double>(),
                       st.E.as<double>(), G.as<double>(), st.partials.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, q, st.partials.as<double>(), nbo, scal, 0, 0);
    VSL_CHECK_LAUNCH(ctx);
    return VSL_OK;
  };
  auto stats = [&](bool write_diag) -> int {  // column norms / gradient of the current blocks; scalars[1] = max |gradient|
    VSL_HIP(ctx, hipMemsetAsync(n2.p, 0, 8 * (size_t)nt, q));
    VSL_HIP(ctx, hipMemsetAsync(grad.p, 0, 8 * (size_t)nt, q));
    VSL_HIP(ctx, hipMemsetAsync(st.n2l.p, 0, 8 * (size_t)L3, q));
    VSL_HIP(ctx, hipMemsetAsync(st.grad_l.p, 0, 8 * (size_t)L3, q));
    hipLaunchKernelGGL(bai_stats_kernel, dim3(nbo), dim3(256), 0, q, D, st.cam_free.as<int>(), st.cam_intr.as<int>(),
                       st.obs_cam.as<int>(), st.obs_lm.as<int>(), st.r.as<double>(), st.F.as<double>(), st.E.as<double>(),
                       G.as<double>(), n2.as<double>(), grad.as<double>(), st.n2l.as<double>(), st.grad_l.as<double>());
    hipLaunchKernelGGL(bai_diag_gmax_kernel, dim3(1), dim3(256), 0, q, nt, L3, n2.as<double>(), st.n2l.as<double>(),
                       grad.as<double>(), st.grad_l.as<double>(), write_diag ? 1 : 0, diag.as<double>(), st.diag_l.as<double>(),
                       scal, 1);
    VSL_CHECK_LAUNCH(ctx);
    return VSL_OK;
  };
  double h[16];
  // initial linearisation, Jacobi scaling from the unscaled column norms (once), statistics of the scaled blocks
  if ((rc = linearize(false))) return rc;
  if ((rc = stats(false))) return rc;
  hipLaunchKernelGGL(bai_make_scale_kernel, dim3((std::max(nt, L3) + 255) / 256), dim3(256), 0, q, nt, L3, n2.as<double>(),
                     st.n2l.as<double>(), scale.as<double>(), st.scale_l.as<double>());
  hipLaunchKernelGGL(bai_apply_scale_kernel, dim3(nbo), dim3(256), 0, q, D, st.cam_free.as<int>(), st.cam_intr.as<int>(),
                     st.obs_cam.as<int>(), st.obs_lm.as<int>(), scale.as<double>(), st.scale_l.as<double>(), st.F.as<double>(),
                     st.E.as<double>(), G.as<double>());
  if ((rc = stats(true))) return rc;
  if ((rc = read_scalars(ctx, st, h, 2))) return rc;
  double cost = h[0], gmax = h[1];
  sum.initial_cost = cost;
  double radius = 1e4, decrease_factor = 2.0;
  bool have_diag = true;  // stats(true) above wrote the LM diagonal of the current Jacobian
  int iteration = 0, invalid = 0;
  sum.termination = 0;
  if (opt->verbosity >= 2) fprintf(stderr, "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e\n", 0, cost);
  while (true) {
    if (iteration >= opt->max_num_iterations) { sum.termination = 0; break; }
    if (gmax <= 1e-10) { sum.termination = 2; break; }
    if (radius <= 1e-32) { sum.termination = 4; break; }
    iteration++;
    (void)have_diag;
    const double inv_radius = 1.0 / radius;
    // reduced system: J^T J of the camera side, damping, Schur corrections (camera-camera, then the border)
    VSL_HIP(ctx, hipMemsetAsync(Sf.p, 0, 8 * (size_t)nt * nt, q));
    VSL_HIP(ctx, hipMemsetAsync(rhsf.p, 0, 8 * (size_t)nt, q));
    hipLaunchKernelGGL(bai_hess_kernel, dim3(nbo), dim3(256), 0, q, D, nt, st.cam_free.as<int>(), st.cam_intr.as<int>(),
                       st.obs_cam.as<int>(), st.r.as<double>(), st.F.as<double>(), G.as<double>(), Sf.as<double>(),
                       rhsf.as<double>());
    hipLaunchKernelGGL(bai_damp_kernel, dim3((nt + 255) / 256), dim3(256), 0, q, nt, diag.as<double>(), inv_radius, Sf.as<double>());
    hipLaunchKernelGGL(ba_schur_atomic_kernel, dim3((D.L + 3) / 4), dim3(256), 0, q, D, st.lm_start.as<int>(), st.obs_cam.as<int>(),
                       st.cam_free.as<int>(), st.r.as<double>(), st.F.as<double>(), st.E.as<double>(), st.diag_l.as<double>(),
                       inv_radius, 0, D.L, Sf.as<double>(), rhsf.as<double>(), st.Pinv.as<double>(), st.bl.as<double>(), 0, nt);
    hipLaunchKernelGGL(bai_border_kernel, dim3((D.L + 3) / 4), dim3(256), 0, q, D, nt, st.lm_start.as<int>(), st.obs_cam.as<int>(),
                       st.cam_free.as<int>(), st.cam_intr.as<int>(), st.F.as<double>(), st.E.as<double>(), G.as<double>(),
                       st.Pinv.as<double>(), st.bl.as<double>(), Tl.as<double>(), Sf.as<double>(), rhsf.as<double>());
    hipLaunchKernelGGL(ba_set_flags_kernel, dim3(1), dim3(64), 0, q, st.flag.as<int>());
    if (nt <= 128) {
      hipLaunchKernelGGL(ba_chol_small_kernel, dim3(1), dim3(256), 0, q, nt, Sf.as<double>(), rhsf.as<double>(), df.as<double>(),
                         st.flag.as<int>() + 1);
    } else {
      if ((rc = vsl_chol_solve_band_dev(ctx, Sf.as<double>(), rhsf.as<double>(), nt, nt, nt, st.flag.as<int>() + 1))) return rc;
      hipLaunchKernelGGL(ba_negate_kernel, dim3((nt + 255) / 256), dim3(256), 0, q, nt, rhsf.as<double>(), df.as<double>());
    }
    hipLaunchKernelGGL(bai_backsub_kernel, dim3((D.L + 255) / 256), dim3(256), 0, q, D, st.lm_start.as<int>(), st.obs_cam.as<int>(),
                       st.cam_free.as<int>(), st.cam_intr.as<int>(), st.F.as<double>(), st.E.as<double>(), G.as<double>(),
                       st.Pinv.as<double>(), st.bl.as<double>(), df.as<double>(), st.dl.as<double>());
    hipLaunchKernelGGL(ba_all_finite2_kernel, dim3((std::max(nt, L3) + 255) / 256), dim3(256), 0, q, nt, df.as<double>(), L3,
                       st.dl.as<double>(), st.flag.as<int>());
    hipLaunchKernelGGL(bai_model_kernel, dim3(nbo), dim3(256), 0, q, D, st.obs_cam.as<int>(), st.obs_lm.as<int>(),
                       st.cam_free.as<int>(), st.cam_intr.as<int>(), st.r.as<double>(), st.F.as<double>(), st.E.as<double>(),
                       G.as<double>(), df.as<double>(), st.dl.as<double>(), st.partials.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, q, st.partials.as<double>(), nbo, scal, 2, 0);
    // candidate point and its cost
    hipLaunchKernelGGL(ba_update_kernel, dim3(nbu), dim3(256), 0, q, D, st.cam_free.as<int>(), st.poses.as<double>(),
                       st.points.as<double>(), df.as<double>(), st.dl.as<double>(), scale.as<double>(), st.scale_l.as<double>(),
                       st.cand_poses.as<double>(), st.cand_points.as<double>(), st.partials.as<double>(), nbu);
    hipLaunchKernelGGL(ba_reduce2_kernel, dim3(2), dim3(256), 0, q, st.partials.as<double>(), nbu, scal, 3);
    hipLaunchKernelGGL(bai_intr_update_kernel, dim3(1), dim3(64), 0, q, n, st.intr.as<double>(), df.as<double>(), scale.as<double>(),
                       cand_intr.as<double>(), scal, 6);
    hipLaunchKernelGGL(ba_cost_kernel, dim3(nbo), dim3(256), 0, q, D, st.cand_poses.as<double>(), st.cand_points.as<double>(),
                       cand_intr.as<double>(), st.cam_intr.as<int>(), st.obs_cam.as<int>(), st.obs_lm.as<int>(),
                       st.obs_uv.as<double>(), 0, D.O, st.partials.as<double>());
    hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, q, st.partials.as<double>(), nbo, scal, 5, 0);
    VSL_CHECK_LAUNCH(ctx);
    int hflag[2];
    VSL_HIP(ctx, hipMemcpyAsync(h, scal, sizeof(double) * 8, hipMemcpyDeviceToHost, q));
    VSL_HIP(ctx, hipMemcpyAsync(hflag, st.flag.p, sizeof(int) * 2, hipMemcpyDeviceToHost, q));
    VSL_HIP(ctx, hipStreamSynchronize(q));
    const double model_change = h[2], step_norm = sqrt(h[3] + h[6]), x_norm = sqrt(h[4] + h[7]), cand_cost = h[5];
    const bool ok = hflag[0] != 0 && hflag[1] != 0 && model_change > 0.0;
    if (!ok) {
      if (++invalid >= 5) { sum.termination = 4; break; }
      radius *= 0.5;
      if (opt->verbosity >= 2) fprintf(stderr, "%4d  invalid step, radius %.3e\n", iteration, radius);
      continue;  // the LM diagonal is reused (the Jacobian is unchanged)
    }
    invalid = 0;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 3; break; }
    const double cost_change = cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * cost) { sum.termination = 1; break; }
    const double rel = cost_change / model_change;
    if (opt->verbosity >= 2)
      fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", iteration, cand_cost, cost_change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      std::swap(st.poses.p, st.cand_poses.p);
      std::swap(st.points.p, st.cand_points.p);
      VSL_HIP(ctx, hipMemcpyAsync(st.intr.p, cand_intr.p, 8 * 16, hipMemcpyDeviceToDevice, q));
      cost = cand_cost;
      if ((rc = linearize(true))) return rc;
      if ((rc = stats(true))) return rc;   // new LM diagonal
      if ((rc = read_scalars(ctx, st, h, 2))) return rc;
      gmax = h[1];
      sum.successful_steps++;
      radius = radius / std::max(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
      radius = std::min(1e16, radius);
      decrease_factor = 2.0;
    } else {
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
  }
  sum.iterations = iteration;
  sum.final_cost = cost;
  VSL_HIP(ctx, hipMemcpyAsync(prob->poses, st.poses.p, sizeof(double) * 7 * (size_t)D.C, hipMemcpyDeviceToHost, q));
  VSL_HIP(ctx, hipMemcpyAsync(prob->points, st.points.p, sizeof(double) * 3 * (size_t)D.L, hipMemcpyDeviceToHost, q));
  VSL_HIP(ctx, hipMemcpyAsync(intr_io, st.intr.p, sizeof(double) * 16, hipMemcpyDeviceToHost, q));
  VSL_HIP(ctx, hipStreamSynchronize(q));
  sum.total_ms = now_ms() - t_start;
  if (opt->verbosity >= 1)
    fprintf(stderr, "vsl BA (intrinsics): iterations %d, initial cost %.6e, final cost %.6e, termination %d, %.3f ms\n", sum.iterations,
            sum.initial_cost, sum.final_cost, sum.termination, sum.total_ms);
  if (summary) *summary = sum;
  return VSL_OK;
}
