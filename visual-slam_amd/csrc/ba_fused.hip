// ba_fused.hip -- the local-window bundle adjustment (6 * free cameras <= 126 unknowns after the Schur complement) as
// FOUR launches and ONE host synchronisation per Levenberg-Marquardt iteration, with no per-observation array in
// global memory at all.
//
// Replaces, for the sliding window of visnav::bundle_adjustment (include/visnav/map_utils.h:337-421, called from
// src/slam.cpp optimize()), what ceres::Solve does per iteration with BundleAdjustmentReprojectionCostFunctor
// (include/visnav/reprojection.h:81-105): evaluate residuals + Jacobian blocks, eliminate the landmarks (SPARSE_SCHUR),
// solve the reduced camera system, back-substitute, evaluate the candidate.  Same optimisation problem, same restated
// [upstream] Ceres step policy as vsl_bundle_adjust's operator-by-operator path in ba.hip (which stays the path of
// large systems, of vsl_ba_linearize and of the "ba_no_fused" diagnostic) -- different data flow:
//
//   round 3:  linearize -> r / F / E (160 B per observation) in global memory, read back by Schur, back-substitution,
//             model, per-camera blocks, column norms: 14 launches per iteration, >= 4.5 x the algorithmic bytes.
//   here:     a workgroup owns a contiguous range of landmarks with <= 1024 observations, ONE THREAD PER OBSERVATION.
//     1 baf_schur_kernel   loads the 20 B of an observation (detected corner, static layout word with the camera), evaluates the
//                          residual and the Jacobian blocks into REGISTERS, reduces E^T E / E^T r per landmark through
//                          LDS and inverts the damped 3 x 3 blocks; the per-camera blocks F^T F, F^T r come from a
//                          camera-major LDS copy of F and r; then, <= 29 landmarks at a time, W = F^T E and -Y = -W P^-1
//                          are laid out as dense K-major operands in LDS (K = 3 columns per landmark, zeros where a camera
//                          does not see a landmark, one extra row holding -P^-1 b) and the lower triangle of
//                          -sum Y W^T -- the local window's reduced camera system is dense: every landmark is seen by most
//                          cameras -- is accumulated as 16 x 16 tiles on v_mfma_f64_16x16x4_f64, tiles dealt to the 16
//                          wavefronts, accumulators in registers over all chunks; the extra row delivers the right-hand
//                          side.  Out: per-workgroup tile partials, camera-block partials, P^-1 and b per landmark (96 B).
//                          (The first version gave every thread a 3 x 3 sub-block and walked the landmarks with slot-table
//                          lookups: 75 % of the kernel in a latency-bound loop of 5 of the 16 wavefronts.)
//     2 baf_finish_kernel  sums the partials in a fixed order, adds camera blocks and LM damping; gradient max-norm, cost.
//     3 baf_chol_kernel    solves the reduced system: one workgroup, DPP-broadcast panel factorisation, MFMA trailing
//                          update, the forward substitution folded into the factorisation (comment at the kernel).
//     4 baf_step_kernel    re-evaluates the blocks (cheaper than storing them), back-substitutes the landmark steps,
//                          accumulates the model cost change and the step / parameter norms, writes the candidate
//                          parameters and evaluates the cost there; its per-workgroup partials land in pinned host
//                          memory, the host adds them in workgroup order and takes the decision.
//   Every structural quantity (workgroup ranges, chunks, camera-major ranks, visibility masks) is fixed over the LM
//   iterations of a solve and is laid out once on the host (bf_plan).  All reductions run in a fixed order: a solve is
//   bit-reproducible run to run.
//
// Algorithmic bytes per LM iteration: n_obs * 24 + n_lms * 24 + n_cams * 56 + 128 in;
// (6C)^2 * 8 + 6C * 8 + n_lms * 96 out.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "ba_device.h"
#include "dpp_chol.h"
#include "vsl_common.h"

namespace {

#define BF_THREADS 1024
#define BF_WAVES (BF_THREADS / 64)
#define BF_OBS_CAP 1024  // observations of a workgroup (one per thread)
#define BF_LMW 128       // landmarks of a workgroup
#define BF_R 14336       // doubles of the shared work region: staging [1024][9] / camera-major [1024][14] / the two dense operands
#define BF_CAMS 64       // cameras of the problem
#define BF_MAXCH 16      // chunks per workgroup
#define BF_TILES 3       // 16 x 16 tiles per wavefront: 36 lower tiles at 128 padded unknowns / 16 wavefronts
#define BF_INFO 48       // ints per workgroup record: lm0, n_lm, obs0, n_obs, n_chunks, chunk boundaries [17], camera offsets [23]
#define BF_INFO_CB 5
#define BF_INFO_CAM 22

typedef double bf_v4d __attribute__((ext_vector_type(4)));

// static word of an observation: camera-major rank in the workgroup (10) | landmark in chunk (5) << 10 | chunk (4) << 15 |
// landmark in workgroup (7) << 19 | camera (6) << 26
__host__ __device__ inline unsigned bf_pack(unsigned rank, unsigned li, unsigned chunk, unsigned lml, unsigned cam) {
  return rank | (li << 10) | (chunk << 15) | (lml << 19) | (cam << 26);
}

// Levenberg-Marquardt state of a solve ON THE DEVICE (round 4, "device-decided" loop): baf_decide_kernel applies the
// [upstream] Ceres policy after every iteration and the kernels of the next one read radius / current buffer / done from
// here, so the host enqueues iterations without waiting for the previous one's verdict.
struct BfLm {
  double cost, radius, decrease;
  int iteration, invalid, successful, termination;
  int cur;    // which (poses, points) buffer pair holds the current estimate: 0 = (poses, points), 1 = the _alt pair
  int done;   // set with the termination reason: every later kernel of the stream returns at once
  int bodies; // loop bodies decided so far
  int pad;
};

struct BfArgs {
  BaDims D;
  const BfLm* lm;          // null: the host decides (radius as a kernel argument, buffers swapped by the host)
  double* poses_alt;       // device-decided loop: the second buffer pair (candidate of cur = 0, current of cur = 1)
  double* points_alt;
  int NP;     // unknowns + the right-hand-side row, padded to whole tiles: 16 * ((n + 16) / 16)
  int NPs;    // row stride of the K-major operands (doubles): = 16 mod 32, so the four k-groups of an operand read spread over the banks
  int NT;     // NP / 16
  int T;      // lower tiles: NT (NT + 1) / 2
  const double* poses;
  const double* points;
  const double* intr;
  const int* cam_intr;
  const int* cam_free;
  const double* obs_uv;
  const unsigned* obs_meta;
  const int* lm_start;
  const unsigned* lm_pres;  // bit c: free camera c observes the landmark
  const int* wg_info;
  const double* scale_c;
  double* scale_l;   // written by the INIT pass, read afterwards
  double* Pinv;
  double* bl;
  double* S_part;    // [G][T][256]: tiles in accumulator layout (lane * 4 + q)
  double* hc_part;   // [G][nfree][27]: upper triangle of F^T F (21) | F^T r (6)
  double* sc_part;   // [G][2]: cost | max |landmark gradient|
};

// device-decided loop: the kernel's view of the argument block -- current buffers by the state's `cur`, the radius
// from the state; returns false when the solve is over (the kernel leaves)
__device__ __forceinline__ bool bf_resolve(BfArgs& a, double& inv_radius, double*& cand_poses, double*& cand_points) {
  if (!a.lm) return true;
  const BfLm st = *a.lm;
  if (st.done) return false;
  inv_radius = 1.0 / st.radius;
  double* pA = const_cast<double*>(a.poses);
  double* qA = const_cast<double*>(a.points);
  if (st.cur) {
    a.poses = a.poses_alt;
    a.points = a.points_alt;
    cand_poses = pA;
    cand_points = qA;
  } else {
    cand_poses = a.poses_alt;
    cand_points = a.points_alt;
  }
  return true;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// What both per-iteration kernels start with: the workgroup's cameras / landmarks into LDS, this thread's observation
// evaluated at (poses, points) into registers -- robustified and Jacobi-scaled like ba_linearize_kernel does it.
struct BfObs {
  double r[2], F[12], E[6];
  double cost;
  int fc, cam, lml;
  unsigned meta;
  bool have;
};

struct BfShared {
  double* cam_s;   // [BF_CAMS][12] R | t
  double* intr_s;  // [16]
  double* scc_s;   // [128]
  double* pts_s;   // [BF_LMW][3]
  double* scl_s;   // [BF_LMW][3]
  int* lmo_s;      // [BF_LMW + 1]
  int* camk_s;     // [BF_CAMS]
  int* camf_s;     // [BF_CAMS]
};

template <bool INIT>
__device__ __forceinline__ void bf_load(const BfArgs& a, const BfShared& sh, int lm0, int n_lm, int obs0, int n_obs,
                                        BfObs& o, double* uv) {
  const BaDims& D = a.D;
  const int tid = threadIdx.x;
  if (tid < D.C) {
    const double* T = a.poses + 7 * (size_t)tid;
    double Rt[12];
    quat_R(T, Rt);
    Rt[9] = T[4];
    Rt[10] = T[5];
    Rt[11] = T[6];
#pragma unroll
    for (int j = 0; j < 12; j++) sh.cam_s[12 * tid + j] = Rt[j];
    sh.camk_s[tid] = a.cam_intr[tid];
    sh.camf_s[tid] = a.cam_free[tid];
  }
  if (tid < 16) sh.intr_s[tid] = a.intr[tid];
  if (tid < D.n) sh.scc_s[tid] = INIT ? 1.0 : a.scale_c[tid];
  if (tid < n_lm) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      sh.pts_s[3 * tid + j] = a.points[3 * (size_t)(lm0 + tid) + j];
      sh.scl_s[3 * tid + j] = INIT ? 1.0 : a.scale_l[3 * (size_t)(lm0 + tid) + j];
    }
  }
  if (tid <= n_lm) sh.lmo_s[tid] = a.lm_start[lm0 + tid] - obs0;
  o.have = tid < n_obs;
  o.cam = 0;
  o.meta = bf_pack(0, 0, 15, 0, 0);
  uv[0] = uv[1] = 0.0;
  if (o.have) {
    const size_t i = (size_t)obs0 + tid;
    o.meta = a.obs_meta[i];
    uv[0] = a.obs_uv[2 * i];
    uv[1] = a.obs_uv[2 * i + 1];
  }
  o.cam = (int)(o.meta >> 26);
  o.lml = (int)((o.meta >> 19) & 0x7Fu);
}

__device__ __forceinline__ void bf_eval(const BfArgs& a, const BfShared& sh, BfObs& o, const double* uv) {
  const BaDims& D = a.D;
  o.r[0] = o.r[1] = 0.0;
  o.cost = 0.0;
  o.fc = -1;
#pragma unroll
  for (int j = 0; j < 12; j++) o.F[j] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; j++) o.E[j] = 0.0;
  if (!o.have) return;
  const int k = sh.camk_s[o.cam];
  o.fc = sh.camf_s[o.cam];
  residual_blocks_Rt(k ? D.model1 : D.model0, sh.intr_s + 8 * k, sh.cam_s + 12 * o.cam, sh.pts_s + 3 * o.lml, uv, o.r, o.F,
                     o.E, true);
  const double s = o.r[0] * o.r[0] + o.r[1] * o.r[1];
  double rho0 = s, rho1 = 1.0;
  if (D.use_huber) huber(s, D.huber, rho0, rho1);
  o.cost = 0.5 * rho0;
  const double sr = sqrt(rho1);
  o.r[0] = o.r[0] * sr;
  o.r[1] = o.r[1] * sr;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const double sc = o.fc >= 0 ? sh.scc_s[6 * o.fc + j] : 1.0;
    o.F[j] = o.F[j] * sr * sc;
    o.F[6 + j] = o.F[6 + j] * sr * sc;
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double sc = sh.scl_s[3 * o.lml + j];
    o.E[j] = o.E[j] * sr * sc;
    o.E[3 + j] = o.E[3 + j] * sr * sc;
  }
}

// lower tile t = ti (ti + 1) / 2 + tj, tj <= ti
__device__ __forceinline__ void bf_tile_coords(int t, int& ti, int& tj) {
  ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (ti * (ti + 1) / 2 > t) ti--;
  while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
  tj = t - ti * (ti + 1) / 2;
}

#ifdef BF_TIMING
#define BF_STAMP(q) { const long long t2 = __builtin_amdgcn_s_memtime(); tp[q] += t2 - tt; tt = t2; }
#else
#define BF_STAMP(q)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Launch 1 of an iteration (INIT = false), and the Jacobi-scaling pass of a solve (INIT = true: unscaled blocks, only
// the column norms -- scale_l per landmark written directly, the camera blocks' diagonal for scale_c -- and the cost).
// TPW = 16 x 16 tiles per wavefront (1 up to 80 padded unknowns, 2 up to 112, 3 at 128): a template parameter because the
// operand look-ahead of the tile loop is paid in registers per tile.
template <bool INIT, int TPW>
__global__ __launch_bounds__(BF_THREADS) void baf_schur_kernel(BfArgs a, double inv_radius) {
  __shared__ double R[BF_R];  // 112 KB: (1) staging [obs][9] = E^T E | E^T r, (2) camera-major [rank][14] = F | r, (3) Yd | Wd
  __shared__ double cam_s[BF_CAMS * 12];
  __shared__ double intr_s[16];
  __shared__ double scc_s[128];
  __shared__ double pts_s[BF_LMW * 3];
  __shared__ double scl_s[BF_LMW * 3];
  __shared__ double pib_s[BF_LMW * 12];  // P^-1 (9) | -P^-1 b (3) per landmark
  __shared__ int lmo_s[BF_LMW + 1];
  __shared__ int camk_s[BF_CAMS];
  __shared__ int camf_s[BF_CAMS];
  __shared__ int cb_s[BF_MAXCH + 1];
  __shared__ int camoff_s[24];
  __shared__ unsigned pres_s[BF_LMW];
  __shared__ double red_s[2][BF_WAVES];
  static_assert(BF_OBS_CAP * 14 <= BF_R, "the camera-major view must fit the work region");
  {
    double *cp_unused, *cq_unused;
    if (!INIT && !bf_resolve(a, inv_radius, cp_unused, cq_unused)) return;
  }
  const BaDims& D = a.D;
  const int tid = threadIdx.x, bid = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = D.n;
  const int* info = a.wg_info + BF_INFO * (size_t)bid;
  const int lm0 = info[0], n_lm = info[1], obs0 = info[2], n_obs = info[3], n_ch = info[4];
  const BfShared sh = {cam_s, intr_s, scc_s, pts_s, scl_s, lmo_s, camk_s, camf_s};
#ifdef BF_TIMING
  long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tt = __builtin_amdgcn_s_memtime();
#endif
  BfObs o;
  double uv[2];
  bf_load<INIT>(a, sh, lm0, n_lm, obs0, n_obs, o, uv);
  if (tid <= BF_MAXCH) cb_s[tid] = info[BF_INFO_CB + tid];
  if (tid <= D.nfree) camoff_s[tid] = info[BF_INFO_CAM + tid];
  if (tid < n_lm) pres_s[tid] = a.lm_pres[lm0 + tid];
  __syncthreads();
  BF_STAMP(0)
  bf_eval(a, sh, o, uv);
  {
    double* st = R + 9 * tid;  // (idle threads write zeros nobody reads: a landmark's range covers real observations only)
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
    const double cs = wave_sum(o.cost);
    if (lane == 0) red_s[0][wave] = cs;
  }
  __syncthreads();
  BF_STAMP(1)
  // per landmark: P = sum E^T E, b = sum E^T r in observation order; damped inverse
  double gl = 0.0;
  if (tid < n_lm) {
    double P6[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    const int i0 = lmo_s[tid], i1 = lmo_s[tid + 1];
    for (int i = i0; i < i1; i++) {
      const double* st = R + 9 * i;
#pragma unroll
      for (int q = 0; q < 6; q++) P6[q] += st[q];
#pragma unroll
      for (int q = 0; q < 3; q++) bb[q] += st[6 + q];
    }
    if (INIT) {
      // Jacobi scaling of the landmark columns ([upstream] Ceres jacobi_scaling): 1 / (1 + sqrt(column norm^2))
      a.scale_l[3 * (size_t)(lm0 + tid)] = 1.0 / (1.0 + sqrt(P6[0]));
      a.scale_l[3 * (size_t)(lm0 + tid) + 1] = 1.0 / (1.0 + sqrt(P6[3]));
      a.scale_l[3 * (size_t)(lm0 + tid) + 2] = 1.0 / (1.0 + sqrt(P6[5]));
    } else {
      double P[9] = {P6[0], P6[1], P6[2], P6[1], P6[3], P6[4], P6[2], P6[4], P6[5]};
      // LM damping: diag = clamp(column norm^2, 1e-6, 1e32) / radius (the Jacobian of a rejected step is unchanged, so
      // recomputing the clamp here IS keeping it)
      P[0] += fmin(fmax(P6[0], 1e-6), 1e32) * inv_radius;
      P[4] += fmin(fmax(P6[3], 1e-6), 1e32) * inv_radius;
      P[8] += fmin(fmax(P6[5], 1e-6), 1e32) * inv_radius;
      double Pi[9];
      const bool ok = i1 > i0 && inv3(P, Pi);
#pragma unroll
      for (int q = 0; q < 9; q++) {
        Pi[q] = ok ? Pi[q] : 0.0;  // a singular block contributes nothing (Y = 0) and its landmark does not move
        pib_s[12 * tid + q] = Pi[q];
        a.Pinv[9 * (size_t)(lm0 + tid) + q] = Pi[q];
      }
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const double v = ok ? bb[q] : 0.0;
        a.bl[3 * (size_t)(lm0 + tid) + q] = v;
        pib_s[12 * tid + 9 + q] = -(Pi[3 * q] * bb[0] + Pi[3 * q + 1] * bb[1] + Pi[3 * q + 2] * bb[2]);
        gl = fmax(gl, fabs(bb[q] / scl_s[3 * tid + q]));  // gradient of the UNSCALED problem
      }
    }
  }
  {
    const double gm = wave_max(gl);
    if (lane == 0) red_s[1][wave] = gm;
  }
  __syncthreads();  // staging is dead from here on
  BF_STAMP(2)
  // per-camera blocks: F | r of the free-camera observations in camera-major order, then thread (camera, entry) walks
  // its camera's run -- addresses are affine in the loop counter, the loads pipeline
  const bool is_free = o.fc >= 0;  // (idle threads: -1)
  if (is_free) {
    double* f = R + 14 * (size_t)(o.meta & 0x3FFu);
#pragma unroll
    for (int j = 0; j < 12; j++) f[j] = o.F[j];
    f[12] = o.r[0];
    f[13] = o.r[1];
  }
  __syncthreads();
  if (tid < 27 * D.nfree) {
    const int c = tid / 27, e = tid - 27 * c;
    int ia = 0, ib = 12;  // entry (ia, ib) of F^T F, or F^T r (ib = 12)
    if (e < 21) {
      int rem = e;
      while (rem >= 6 - ia) {
        rem -= 6 - ia;
        ia++;
      }
      ib = ia + rem;
    } else {
      ia = e - 21;
    }
    const int ib1 = e < 21 ? ib + 6 : 13;
    double acc = 0.0;
    const int k1 = camoff_s[c + 1];
    int k = camoff_s[c];
    for (; k + 4 <= k1; k += 4) {  // the loads of four observations in flight, the sum in observation order
      double t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double* f = R + 14 * (k + u);
        t[u] = f[ia] * f[ib] + f[6 + ia] * f[ib1];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += t[u];
    }
    for (; k < k1; k++) {
      const double* f = R + 14 * k;
      acc += f[ia] * f[ib] + f[6 + ia] * f[ib1];
    }
    a.hc_part[((size_t)bid * D.nfree + c) * 27 + e] = acc;
  }
  if (tid == 0) {
    double cs = 0.0, gm = 0.0;
    for (int w = 0; w < BF_WAVES; w++) {
      cs += red_s[0][w];
      gm = fmax(gm, red_s[1][w]);
    }
    a.sc_part[2 * bid] = cs;
    a.sc_part[2 * bid + 1] = gm;
  }
  if (INIT) return;
  __syncthreads();  // the camera-major view is dead: the region becomes the two dense operands
  BF_STAMP(3)
  const int NPs = a.NPs, NP = a.NP, T = a.T;
  const int kq = lane >> 4, l16 = lane & 15;
  bf_v4d acc[TPW];
  int aoff[TPW], boff[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++) {
    acc[s] = (bf_v4d){0.0, 0.0, 0.0, 0.0};
    int ti = 0, tj = 0;
    if (wave + BF_WAVES * s < T) bf_tile_coords(wave + BF_WAVES * s, ti, tj);
    aoff[s] = kq * NPs + 16 * ti + l16;  // Yd rows of the tile's row block
    boff[s] = kq * NPs + 16 * tj + l16;  // Wd rows of its column block
  }
  const int my_ch = (int)((o.meta >> 15) & 0xFu), my_li = (int)((o.meta >> 10) & 0x1Fu);
  const int rowcam[2] = {lane / 6, (lane + 64) / 6};  // free camera of the two operand rows this lane zero-fills
  for (int c = 0; c < n_ch; c++) {
    const int l_a = cb_s[c], lc = cb_s[c + 1] - l_a;
    const int KP = (3 * lc + 3) & ~3;
    double* Yd = R;
    double* Wd = R + (size_t)KP * NPs;
    // zeros where a free camera does not see a landmark (static mask), in the padding rows / columns and in Wd's
    // right-hand-side row; everything else is written by the thread that owns the value -- disjoint, one barrier
    for (int k = wave; k < KP; k += BF_WAVES) {  // k is wave-uniform, the rows are the lanes (NP <= 128: two per lane)
      const int li = k / 3;
      const unsigned pm = li < lc ? pres_s[l_a + li] : 0u;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int row = lane + 64 * h;
        if (row < NP) {
          const bool vis = row < n && ((pm >> rowcam[h]) & 1u) != 0;
          const bool visy = vis || (row == n && li < lc);
          if (!visy) Yd[k * NPs + row] = 0.0;
          if (!vis) Wd[k * NPs + row] = 0.0;
        }
      }
    }
    BF_STAMP(7)
    if (is_free && my_ch == c) {
      const double* Pi = pib_s + 12 * o.lml;
      const int base = 3 * my_li * NPs + 6 * o.fc;
#pragma unroll
      for (int x = 0; x < 6; x++) {
        double w[3];
#pragma unroll
        for (int y = 0; y < 3; y++) w[y] = o.F[x] * o.E[y] + o.F[6 + x] * o.E[3 + y];
#pragma unroll
        for (int y = 0; y < 3; y++) {
          Wd[base + y * NPs + x] = w[y];
          Yd[base + y * NPs + x] = -(w[0] * Pi[y] + w[1] * Pi[3 + y] + w[2] * Pi[6 + y]);
        }
      }
    }
    BF_STAMP(3)
    if (tid >= l_a && tid < l_a + lc) {  // the landmark's own thread: row n of Yd = -P^-1 b
#pragma unroll
      for (int y = 0; y < 3; y++) Yd[(3 * (tid - l_a) + y) * NPs + n] = pib_s[12 * tid + 9 + y];
    }
    __syncthreads();
    BF_STAMP(4)
    {
      int k0 = 0;
      constexpr int AHEAD = TPW == 1 ? 4 : (TPW == 2 ? 2 : 1);  // steps whose operands are in flight before the first matrix instruction
      for (; k0 + 4 * AHEAD <= KP; k0 += 4 * AHEAD) {
        double ya[AHEAD][TPW], wb[AHEAD][TPW];
#pragma unroll
        for (int u = 0; u < AHEAD; u++)
#pragma unroll
          for (int s = 0; s < TPW; s++)
            if (wave + BF_WAVES * s < T) {  // wave-uniform
              ya[u][s] = Yd[(k0 + 4 * u) * NPs + aoff[s]];
              wb[u][s] = Wd[(k0 + 4 * u) * NPs + boff[s]];
            }
#pragma unroll
        for (int u = 0; u < AHEAD; u++)
#pragma unroll
          for (int s = 0; s < TPW; s++)
            if (wave + BF_WAVES * s < T) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[u][s], wb[u][s], acc[s], 0, 0, 0);
      }
      for (; k0 < KP; k0 += 4) {
#pragma unroll
        for (int s = 0; s < TPW; s++)
          if (wave + BF_WAVES * s < T)
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Yd[k0 * NPs + aoff[s]], Wd[k0 * NPs + boff[s]], acc[s], 0, 0, 0);
      }
    }
    __syncthreads();
    BF_STAMP(5)
  }
#pragma unroll
  for (int s = 0; s < TPW; s++)
    if (wave + BF_WAVES * s < T)
      *(bf_v4d*)(a.S_part + (((size_t)bid * T + wave + BF_WAVES * s) * 256 + 4 * lane)) = acc[s];
  BF_STAMP(6)
#ifdef BF_TIMING
  if (tid == 0 && (bid == 0 || bid == 100))
    printf("schur wg %d (lms %d obs %d chunks %d) ticks: load %lld eval+stage %lld landmarks %lld camera blocks+W/Y %lld zero fill %lld barrier after operands %lld mfma %lld epilogue %lld\n",
           bid, n_lm, n_obs, n_ch, tp[0], tp[1], tp[2], tp[3], tp[7], tp[4], tp[5], tp[6]);
#endif
}

// entry (a, b), a <= b, of the upper triangle of a 6 x 6 block, row-major
__device__ __forceinline__ int bf_tri6(int a, int b) { return 6 * a - a * (a - 1) / 2 + (b - a); }

// Launch 2: S = tile partials summed in workgroup order (the strictly upper tiles are mirror images) + camera blocks +
// LM damping of the camera diagonal; right-hand side and |g_c / scale| per unknown; cost and landmark gradient norm by
// the last workgroup.  Scalars go to the pinned mailbox: [0] cost, [1] max landmark gradient, gabs[0 .. n) camera part.
// Every sum is 16 slices of <= BF_PER partials whose loads are all in flight at once (a plain loop over 128 partials was
// a chain of 128 dependent round trips: 46 us), added in slice order -- fixed.
#define BF_PER 16  // partials per slice: G <= 256 -> 16 slices; larger G loops
template <int STRIDE_IS_TILE>
__device__ __forceinline__ double bf_slice_sum(const double* __restrict__ p, size_t stride, int g_begin, int g_end) {
  double v = 0;
  for (int g = g_begin; g < g_end; g += BF_PER) {
    double t[BF_PER];
#pragma unroll
    for (int u = 0; u < BF_PER; u++) t[u] = g + u < g_end ? p[(size_t)(g + u) * stride] : 0.0;
#pragma unroll
    for (int u = 0; u < BF_PER; u++) v += t[u];
  }
  return v;
}

__global__ __launch_bounds__(256) void baf_finish_kernel(int n, int nfree, int G, int T, const double* __restrict__ S_part,
                                                         const double* __restrict__ hc_part,
                                                         const double* __restrict__ sc_part,
                                                         const double* __restrict__ scale_c, double inv_radius,
                                                         double* __restrict__ S, double* __restrict__ rhs,
                                                         double* __restrict__ host_out, double* __restrict__ host_gabs,
                                                         const BfLm* __restrict__ lm) {
  if (lm) {  // device-decided loop: radius from the state; nothing to do once the solve is over
    if (lm->done) return;
    inv_radius = 1.0 / lm->radius;
  }
  // A workgroup sums 16 CONSECUTIVE doubles of the tile partials (128 contiguous bytes per partial: the first version
  // walked the partials entry by entry of S, 8 bytes out of every 32 -- FETCH_SIZE 45 MB for 8 MB of partials) and
  // scatters the sums to the entries of S they are: element q of lane l of tile (ti, tj) is
  // D[16 ti + (l >> 4) + 4 q][16 tj + (l & 15)], mirrored for the tiles below the diagonal; row n is the right-hand side.
  __shared__ double sh[16][17];
  __shared__ double sh2[16][17];
  const int nR = T * 16;  // workgroups over the raw tile elements
  const int per = (G + 15) / 16;
  const int e = threadIdx.x & 15, c = threadIdx.x >> 4;
  const int g_a = min(G, c * per), g_b = min(G, (c + 1) * per);
  if ((int)blockIdx.x < nR) {
    const int raw = blockIdx.x * 16 + e;
    const int tile = raw >> 8, l = (raw & 255) >> 2, q = raw & 3;
    int ti, tj;
    bf_tile_coords(tile, ti, tj);
    const int i = 16 * ti + (l >> 4) + 4 * q, j = 16 * tj + (l & 15);
    const bool in_S = i < n && j < n, is_rhs = i == n && j < n;
    double v = 0, hd = 0;
    if (in_S || is_rhs) v = bf_slice_sum<1>(S_part + raw, (size_t)T * 256, g_a, g_b);
    int hk = -1;
    if (in_S && i / 6 == j / 6) {
      const int a = i % 6, b = j % 6;
      hk = (i / 6) * 27 + bf_tri6(min(a, b), max(a, b));
    } else if (is_rhs) {
      hk = (j / 6) * 27 + 21 + j % 6;  // g_c
    }
    if (hk >= 0) hd = bf_slice_sum<0>(hc_part + hk, (size_t)nfree * 27, g_a, g_b);
    sh[c][e] = v;
    sh2[c][e] = hd;
    __syncthreads();
    if (c == 0 && (in_S || is_rhs)) {
      double t = 0, h = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        t += sh[k][e];
        h += sh2[k][e];
      }
      if (is_rhs) {
        rhs[j] = t + h;
        host_gabs[j] = fabs(h / scale_c[j]);  // gradient of the UNSCALED problem, camera part
      } else {
        t += h;
        if (i == j) t += fmin(fmax(h, 1e-6), 1e32) * inv_radius;
        S[(size_t)i * n + j] = t;
        if (ti != tj) S[(size_t)j * n + i] = t;
      }
    }
    return;
  }
  // scalars: one partial per thread and round, LDS tree in a fixed shape
  __shared__ double cs_s[256], gl_s[256];
  double cs = 0, gl = 0;
  for (int g = threadIdx.x; g < G; g += 256) {
    cs += sc_part[2 * g];
    gl = fmax(gl, sc_part[2 * g + 1]);
  }
  cs_s[threadIdx.x] = cs;
  gl_s[threadIdx.x] = gl;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      cs_s[threadIdx.x] += cs_s[threadIdx.x + o];
      gl_s[threadIdx.x] = fmax(gl_s[threadIdx.x], gl_s[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    host_out[0] = cs_s[0];
    host_out[1] = gl_s[0];
  }
}

// the Jacobi-scaling pass's second launch: scale_c = 1 / (1 + sqrt(diag H)) (workgroups 0 .. nV), initial cost (the last)
__global__ __launch_bounds__(256) void baf_init_finish_kernel(int n, int nfree, int G, const double* __restrict__ hc_part,
                                                              const double* __restrict__ sc_part,
                                                              double* __restrict__ scale_c, double* __restrict__ host_out) {
  __shared__ double sh[16][17];
  __shared__ double cs_s[256];
  const int nV = (n + 15) / 16, per = (G + 15) / 16;
  if ((int)blockIdx.x < nV) {
    const int e = threadIdx.x & 15, c = threadIdx.x >> 4;
    const int x = blockIdx.x * 16 + e;
    double h = 0;
    if (x < n) {
      const int a = x % 6;
      h = bf_slice_sum<0>(hc_part + (x / 6) * 27 + bf_tri6(a, a), (size_t)nfree * 27, min(G, c * per), min(G, (c + 1) * per));
    }
    sh[c][e] = h;
    __syncthreads();
    if (c == 0 && x < n) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) t += sh[k][e];
      scale_c[x] = 1.0 / (1.0 + sqrt(t));
    }
    return;
  }
  double cs = 0;
  for (int g = threadIdx.x; g < G; g += 256) cs += sc_part[2 * g];
  cs_s[threadIdx.x] = cs;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) cs_s[threadIdx.x] += cs_s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    host_out[0] = cs_s[0];
    host_out[1] = 0.0;
  }
}

// Launch 3: dc = -(S^-1 rhs) for n <= 127 by ONE workgroup, everything in LDS.
//
// The solve is a serial chain of n pivots whatever one does; what this kernel shortens is the work per pivot
// (ba_chol_small_kernel of ba.hip, the round-2 solver that the other paths keep: 50 us at n = 72, of which 41 % in the
// two substitutions):
//   * the right-hand side rides along as one more ROW of the matrix (row R below the last panel): the Cholesky
//     factorisation of [S b; b^T .] leaves y = L^-1 b in that row -- the forward substitution costs nothing;
//   * 16-column panels.  A wavefront keeps the 16 x 16 diagonal block one row per lane of each 16-lane DPP row (all four
//     rows of the wavefront hold a copy) and ONE PANEL ROW PER LANE next to it; a pivot step is: broadcast of the pivot
//     (v_mov_b64_dpp row_newbcast), rsq + two Newton steps, and per remaining column ONE v_fmac_f64_dpp for the
//     diagonal block and one for the panel row -- the neighbour's factor entry arrives as the DPP operand, no LDS, no
//     readlane, no barrier.  Every wavefront that has panel rows runs the same stream (factoring the diagonal block
//     redundantly), so factorisation and panel solve of a panel are one pass of ~400 instructions;
//   * the trailing update as 16 x 16 tiles on v_mfma_f64_16x16x4_f64 (four instructions per tile), tiles dealt to the
//     16 wavefronts: two barriers per panel;
//   * backward substitution by one wavefront without workgroup barriers: per panel the 16 x 16 transposed solve on DPP
//     broadcasts again, then one batched update of the remaining columns.
// flag = 0 if a pivot is not positive / finite.
#define CS_LD 129   // row stride (doubles): one matrix row per lane is conflict-free (2 dwords per lane and bank)
#define CS_ROWS 129 // 128 matrix rows + the right-hand-side row

__global__ __launch_bounds__(BF_THREADS) void baf_chol_kernel(int n, const double* __restrict__ S,
                                                              const double* __restrict__ rhs, double* __restrict__ dc,
                                                              int* __restrict__ ok_flag, const BfLm* __restrict__ lm) {
  if (lm && lm->done) return;
  __shared__ double A[CS_ROWS * CS_LD];  // 133 KB
  __shared__ double inv_s[128];
  __shared__ double t_s[128];
  __shared__ double xs_s[16];
  __shared__ int fail_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = 16 * ((n + 15) / 16);  // row of the right-hand side; rows n .. R - 1 are identity padding
  const int NPANEL = R / 16;
  // load: rows 0 .. R (wave-strided), columns by lane
  for (int i = wave; i <= R; i += BF_WAVES) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int j = lane + 64 * h;
      if (j < R) {
        double v;
        if (i < n)
          v = j < n ? S[(size_t)i * n + j] : 0.0;
        else if (i < R)
          v = i == j ? 1.0 : 0.0;
        else
          v = j < n ? rhs[j] : 0.0;
        A[i * CS_LD + j] = v;
      }
    }
  }
  if (tid == 0) fail_s = 0;
  __syncthreads();
  const int kq = lane >> 4, l16 = lane & 15;
  for (int pn = 0; pn < NPANEL; pn++) {
    const int c0 = 16 * pn;
    const int r0 = c0 + 16;               // first panel row
    const int nrow = R - r0 + 1;          // panel rows, the right-hand side included (>= 1)
    if (wave * 64 < nrow) {               // wave-uniform
      const int prow = r0 + wave * 64 + lane;
      const bool valid = prow <= R;
      double d[16], p[16];
      const double* drow = A + (c0 + l16) * CS_LD + c0;
      const double* pr = A + min(prow, R) * CS_LD + c0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        d[k] = drow[k];
        p[k] = valid ? pr[k] : 0.0;
      }
      bool good = true;
      CsCol<0>::run(d, p, (wave == 0 && lane == 0) ? inv_s + c0 : (double*)nullptr, good);
      if (wave == 0 && lane < 16) {
        double* dw = A + (c0 + lane) * CS_LD + c0;
#pragma unroll
        for (int k = 0; k < 16; k++) dw[k] = k <= lane ? d[k] : 0.0;  // L_d, zeros above the diagonal
      }
      if (valid) {
        double* pw = A + prow * CS_LD + c0;
#pragma unroll
        for (int k = 0; k < 16; k++) pw[k] = p[k];
      }
      if (!good) fail_s = 1;
    }
    __syncthreads();
    if (fail_s) break;  // workgroup-uniform
    // trailing update: rows / columns r0 .. R in 16 x 16 tiles (the tile column that holds column R is never needed)
    const int ntr = (R - r0) / 16 + 1;
    const int ntiles = ntr * (ntr + 1) / 2;
    for (int t = wave; t < ntiles; t += BF_WAVES) {
      int ti, tj;
      bf_tile_coords(t, ti, tj);
      if (tj == ntr - 1) continue;  // wave-uniform
      const int R0 = r0 + 16 * ti, C0 = r0 + 16 * tj;
      const int ra = min(R0 + l16, R), rb = C0 + l16;
      bf_v4d acc;
#pragma unroll
      for (int q = 0; q < 4; q++) acc[q] = A[min(R0 + kq + 4 * q, R) * CS_LD + C0 + l16];
#pragma unroll
      for (int m = 0; m < 4; m++)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-A[ra * CS_LD + c0 + 4 * m + kq], A[rb * CS_LD + c0 + 4 * m + kq], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = R0 + kq + 4 * q;
        if (row <= R) A[row * CS_LD + C0 + l16] = acc[q];
      }
    }
    __syncthreads();
  }
  const bool good = fail_s == 0;
  if (tid == 0) *ok_flag = good ? 1 : 0;
  if (!good || wave != 0) return;
  // L^T x = y (y = row R), one wavefront, panels in descending order
  t_s[lane] = lane < R ? A[R * CS_LD + lane] : 0.0;
  t_s[lane + 64] = lane + 64 < R ? A[R * CS_LD + lane + 64] : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (int pn = NPANEL - 1; pn >= 0; pn--) {
    const int c0 = 16 * pn;
    double c[16];
#pragma unroll
    for (int k = 0; k < 16; k++) c[k] = A[(c0 + k) * CS_LD + c0 + l16];  // l(k, lane); zero for k < lane (written so above)
    double tp = t_s[c0 + l16], x = 0.0;
    const double ivl = inv_s[c0 + l16];
    CsBack<15>::run(tp, x, c, ivl, l16);
    if (lane < 16) {
      xs_s[lane] = x;
      t_s[c0 + lane] = x;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // the columns in front of the panel lose l(c0 + k, col) x_k
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int col = lane + 64 * h;
      if (col < c0) {
        double t = t_s[col];
#pragma unroll
        for (int k = 0; k < 16; k++) t -= A[(c0 + k) * CS_LD + col] * xs_s[k];
        t_s[col] = t;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < n) dc[lane] = -t_s[lane];
  if (lane + 64 < n) dc[lane + 64] = -t_s[lane + 64];
}

// Launch 4: the step.  delta_l = -P^-1 (b_l + sum_obs E^T F delta_c); model cost change -(J d)^T (r + J d / 2); candidate
// = Plus(x, step .* scale); cost at the candidate.  step_part[5 * workgroup ..] = [model, candidate cost, |step|^2,
// |x|^2, non-finite count] in pinned host memory.
__global__ __launch_bounds__(BF_THREADS) void baf_step_kernel(BfArgs a, const double* __restrict__ dc,
                                                              double* cand_poses, double* cand_points,
                                                              double* __restrict__ step_part) {
  __shared__ double stage_v[BF_OBS_CAP * 3];
  __shared__ double cam_s[BF_CAMS * 12];
  __shared__ double camc_s[BF_CAMS * 12];  // candidate cameras
  __shared__ double intr_s[16];
  __shared__ double scc_s[128];
  __shared__ double dc_s[128];
  __shared__ double pts_s[BF_LMW * 3];
  __shared__ double scl_s[BF_LMW * 3];
  __shared__ double cpt_s[BF_LMW * 3];
  __shared__ double dl_s[BF_LMW * 3];
  __shared__ int lmo_s[BF_LMW + 1];
  __shared__ int camk_s[BF_CAMS];
  __shared__ int camf_s[BF_CAMS];
  __shared__ double red_s[BF_WAVES][5];
  {
    double ir_unused = 0.0;
    if (!bf_resolve(a, ir_unused, cand_poses, cand_points)) return;
  }
  const BaDims& D = a.D;
  const int tid = threadIdx.x, bid = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = D.n;
  const int* info = a.wg_info + BF_INFO * (size_t)bid;
  const int lm0 = info[0], n_lm = info[1], obs0 = info[2], n_obs = info[3];
  const BfShared sh = {cam_s, intr_s, scc_s, pts_s, scl_s, lmo_s, camk_s, camf_s};
  BfObs o;
  double uv[2];
  bf_load<false>(a, sh, lm0, n_lm, obs0, n_obs, o, uv);
  double step2 = 0.0, x2 = 0.0, bad = 0.0;
  if (tid < n) {
    const double v = dc[tid];
    dc_s[tid] = v;
    if (!isfinite(v)) bad = 1.0;
  }
  if (tid < D.C) {  // candidate cameras (every workgroup needs them; workgroup 0 publishes them and owns their norms)
    const int fc = a.cam_free[tid];
    const double* T = a.poses + 7 * (size_t)tid;
    double c7[7];
    double s2 = 0.0, xx = 0.0;
    if (fc < 0) {
#pragma unroll
      for (int j = 0; j < 7; j++) c7[j] = T[j];
    } else {
      double d[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        d[j] = dc[6 * fc + j] * a.scale_c[6 * fc + j];
        s2 += d[j] * d[j];
      }
#pragma unroll
      for (int j = 0; j < 7; j++) xx += T[j] * T[j];
      se3_plus(T, d, c7);
    }
    double Rt[12];
    quat_R(c7, Rt);
    Rt[9] = c7[4];
    Rt[10] = c7[5];
    Rt[11] = c7[6];
#pragma unroll
    for (int j = 0; j < 12; j++) camc_s[12 * tid + j] = Rt[j];
    if (bid == 0) {
#pragma unroll
      for (int j = 0; j < 7; j++) cand_poses[7 * (size_t)tid + j] = c7[j];
      step2 += s2;
      x2 += xx;
    }
  }
  double Pi[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bl[3] = {0, 0, 0};
  if (tid < n_lm) {
#pragma unroll
    for (int q = 0; q < 9; q++) Pi[q] = a.Pinv[9 * (size_t)(lm0 + tid) + q];
#pragma unroll
    for (int q = 0; q < 3; q++) bl[q] = a.bl[3 * (size_t)(lm0 + tid) + q];
  }
  __syncthreads();
  bf_eval(a, sh, o, uv);
  double u0 = 0.0, u1 = 0.0;
  if (o.fc >= 0) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      u0 += o.F[j] * dc_s[6 * o.fc + j];
      u1 += o.F[6 + j] * dc_s[6 * o.fc + j];
    }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) stage_v[3 * tid + j] = o.E[j] * u0 + o.E[3 + j] * u1;
  __syncthreads();
  if (tid < n_lm) {
    double t[3] = {bl[0], bl[1], bl[2]};
    for (int i = lmo_s[tid]; i < lmo_s[tid + 1]; i++) {
#pragma unroll
      for (int j = 0; j < 3; j++) t[j] += stage_v[3 * i + j];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double v = -(Pi[3 * j] * t[0] + Pi[3 * j + 1] * t[1] + Pi[3 * j + 2] * t[2]);
      if (!isfinite(v)) bad = 1.0;
      dl_s[3 * tid + j] = v;
      const double dd = v * scl_s[3 * tid + j];
      step2 += dd * dd;
      const double xv = pts_s[3 * tid + j];
      x2 += xv * xv;
      const double cp = xv + dd;
      cpt_s[3 * tid + j] = cp;
      cand_points[3 * (size_t)(lm0 + tid) + j] = cp;
    }
  }
  __syncthreads();
  double model = 0.0, ccost = 0.0;
  if (o.have) {
    double m0 = u0, m1 = u1;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      m0 += o.E[j] * dl_s[3 * o.lml + j];
      m1 += o.E[3 + j] * dl_s[3 * o.lml + j];
    }
    model = -(m0 * (o.r[0] + m0 / 2.0) + m1 * (o.r[1] + m1 / 2.0));
    const int k = camk_s[o.cam];
    double rc[2];
    residual_blocks_Rt(k ? D.model1 : D.model0, intr_s + 8 * k, camc_s + 12 * o.cam, cpt_s + 3 * o.lml, uv, rc, nullptr,
                       nullptr, false);
    const double s = rc[0] * rc[0] + rc[1] * rc[1];
    double rho0 = s, rho1 = 1.0;
    if (D.use_huber) huber(s, D.huber, rho0, rho1);
    ccost = 0.5 * rho0;
  }
  double v5[5] = {model, ccost, step2, x2, bad};
#pragma unroll
  for (int q = 0; q < 5; q++) {
    const double w = wave_sum(v5[q]);
    if (lane == 0) red_s[wave][q] = w;
  }
  __syncthreads();
  if (tid < 5) {
    double t = 0;
    for (int w = 0; w < BF_WAVES; w++) t += red_s[w][tid];
    step_part[5 * (size_t)bid + tid] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The Levenberg-Marquardt decision on the device (one wavefront): what the host loop of vsl_ba_fused_solve does with
// the mailbox after an iteration -- gradient tolerance, step validity, parameter / function tolerance, the
// [upstream] Ceres trust-region update, accept / reject -- on the same numbers in the same order (IEEE double
// throughout; the cube of the radius update is formed in double-double and rounded once, as the host's pow is), so the
// trajectory is the host-decided one.  box = [cost | max landmark gradient | (int) Cholesky ok | . | gabs[128] |
// 5 step partials per workgroup], the kernels' outputs in DEVICE memory; rec = one record per loop body in pinned host
// memory: [seq, done, termination, iteration, successful, cost, radius, gmax, step norm, rel, candidate cost, cost change,
// cur], seq written last behind a system-scope fence -- the host polls it.
#define BF_REC 16
__global__ __launch_bounds__(64) void baf_decide_kernel(BfLm* __restrict__ lm, const double* __restrict__ box, int n, int G,
                                                        int max_iters, double* __restrict__ rec_base) {
  const int lane = threadIdx.x;
  BfLm st = *lm;
  if (st.done) return;
  // step partials in workgroup order (one quantity per lane), camera gradient maximum over the lanes.  The 5 G partials
  // come in with all loads in flight (as a loop of load-and-add they were 256 dependent round trips: 35 us)
  __shared__ double part_s[5 * 1024];
  {
    const int total = 5 * G;
    for (int i0 = 0; i0 < total; i0 += 64 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + 64 * u + lane;
        v[u] = i < total ? box[132 + i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = i0 + 64 * u + lane;
        if (i < total) part_s[i] = v[u];
      }
    }
  }
  __syncthreads();
  double sum5 = 0.0;
  if (lane < 5) {
    int g = 0;
    for (; g + 8 <= G; g += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = part_s[5 * (g + u) + lane];
#pragma unroll
      for (int u = 0; u < 8; u++) sum5 += v[u];
    }
    for (; g < G; g++) sum5 += part_s[5 * g + lane];
  }
  double gm = 0.0;
  for (int x = lane; x < n; x += 64) gm = fmax(gm, box[4 + x]);
  gm = wave_max(gm);
  const double model_change = __shfl(sum5, 0), cand_cost = __shfl(sum5, 1), step2 = __shfl(sum5, 2), x2 = __shfl(sum5, 3),
               bad = __shfl(sum5, 4);
  if (lane != 0) return;
  const double gmax = fmax(box[1], gm);
  const int chol_ok = *(const int*)(box + 2);
  double step_norm = 0.0, rel = 0.0, cost_change = 0.0;
  const double radius_used = st.radius;
  int term = -1;
  if (gmax <= 1e-10) {
    term = 2;
  } else if (st.radius <= 1e-32) {
    term = 4;
  } else {
    st.iteration++;
    step_norm = sqrt(step2);
    const double x_norm = sqrt(x2);
    const bool ok = chol_ok != 0 && bad == 0.0 && model_change > 0.0;
    if (!ok) {
      if (++st.invalid >= 5)
        term = 4;
      else
        st.radius *= 0.5;
    } else {
      st.invalid = 0;
      cost_change = st.cost - cand_cost;
      if (step_norm <= 1e-8 * (x_norm + 1e-8)) {
        term = 3;
      } else if (fabs(cost_change) <= 1e-6 * st.cost) {
        term = 1;
      } else {
        rel = cost_change / model_change;
        if (rel > 1e-3) {
          st.cost = cand_cost;
          st.cur ^= 1;
          st.successful++;
          const double y = 2.0 * rel - 1.0;
          const double p = y * y, pe = fma(y, y, -p);   // y^2 = p + pe
          const double q = p * y, qe = fma(p, y, -q);   // p y = q + qe
          const double y3 = q + (qe + pe * y);          // y^3 rounded once
          st.radius = st.radius / fmax(1.0 / 3.0, 1.0 - y3);
          st.radius = fmin(1e16, st.radius);
          st.decrease = 2.0;
        } else {
          st.radius = st.radius / st.decrease;
          st.decrease *= 2.0;
        }
      }
    }
  }
  if (term < 0 && st.iteration >= max_iters) term = 0;
  if (term >= 0) {
    st.termination = term;
    st.done = 1;
  }
  const int body = st.bodies;
  st.bodies = body + 1;
  *lm = st;
  double* rec = rec_base + (size_t)BF_REC * (body + 1);
  rec[1] = (double)st.done;
  rec[2] = (double)st.termination;
  rec[3] = (double)st.iteration;
  rec[4] = (double)st.successful;
  rec[5] = st.cost;
  rec[6] = radius_used;
  rec[7] = gmax;
  rec[8] = step_norm;
  rec[9] = rel;
  rec[10] = cand_cost;
  rec[11] = cost_change;
  rec[12] = (double)st.cur;
  rec[13] = (double)(chol_ok != 0 && bad == 0.0 && model_change > 0.0);
  __threadfence_system();
  *(volatile double*)rec = (double)(body + 1);
}

// state of a fresh solve: the scaling pass left the initial cost in box[0]; record 0 carries it to the host
__global__ void baf_lm_init_kernel(BfLm* __restrict__ lm, const double* __restrict__ box, double* __restrict__ rec_base) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  BfLm st;
  st.cost = box[0];
  st.radius = 1e4;
  st.decrease = 2.0;
  st.iteration = st.invalid = st.successful = 0;
  st.termination = -1;
  st.cur = st.done = st.bodies = st.pad = 0;
  *lm = st;
  rec_base[5] = st.cost;
  __threadfence_system();
  *(volatile double*)rec_base = -1.0;  // (sequence numbers of the loop bodies start at 1; -1 marks record 0 as written)
}

// ------------------------------------------------------------------------------------------------ host side
double bf_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// The static layout of a solve, written straight into ONE pinned host block that goes to the device with ONE copy
// (eleven pageable hipMemcpyAsync calls were 0.15 ms, the std::vector passes of the first version 0.9 ms at 157 k
// observations):  [poses | points | intr | cam_intr | cam_free | obs_uv | obs_meta | lm_start | lm_pres | wg_info].
struct BfLayout {
  size_t poses, points, intr, cam_intr, cam_free, obs_uv, obs_meta, lm_start, lm_pres, wg_info, bytes;
  int g_cap;
};
BfLayout bf_layout(int C, int L, int O) {
  BfLayout y;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
  };
  y.poses = take(56 * (size_t)C);
  y.points = take(24 * (size_t)L);
  y.intr = take(128);
  y.cam_intr = take(4 * (size_t)C);
  y.cam_free = take(4 * (size_t)C);
  y.obs_uv = take(16 * (size_t)O);
  y.obs_meta = take(4 * (size_t)O);
  y.lm_start = take(4 * ((size_t)L + 1));
  y.lm_pres = take(4 * (size_t)L);
  // workgroups: <= 256 by observation count, + whatever the per-workgroup caps (128 landmarks, 16 chunks) force
  y.g_cap = 256 + O / 512 + L / 16 + 16;
  y.wg_info = take(4 * (size_t)BF_INFO * y.g_cap);
  y.bytes = off;
  return y;
}

struct BfPlan {
  int G = 0, nfree = 0, NP = 0, NPs = 0, lc_max = 0;
};

// Lays the problem out for the fused kernels into `blk`; false when it does not fit them (the caller takes the general path).
bool bf_plan(const vsl_ba_problem* p, const BfLayout& y, char* blk, BfPlan& pl) {
  const int C = p->n_cams, L = p->n_lms, O = p->n_obs;
  int* cam_free = (int*)(blk + y.cam_free);
  int nfree = 0;
  for (int c = 0; c < C; c++) cam_free[c] = p->cam_fixed[c] ? -1 : nfree++;
  pl.nfree = nfree;
  if (nfree < 1 || 6 * nfree > 126) return false;
  const int n = 6 * nfree;
  pl.NP = 16 * ((n + 16) / 16);  // the unknowns and the right-hand-side row, in whole tiles
  pl.NPs = (pl.NP % 32 == 16) ? pl.NP : pl.NP + 16;
  pl.lc_max = std::min(31, (((BF_R / 2) / pl.NPs) & ~3) / 3);  // K = 3 columns per landmark, padded to the matrix instruction's 4
  memcpy(blk + y.poses, p->poses, 56 * (size_t)C);
  memcpy(blk + y.points, p->points, 24 * (size_t)L);
  memcpy(blk + y.intr, p->intr, 128);
  memcpy(blk + y.cam_intr, p->cam_intr, 4 * (size_t)C);
  // observations by landmark (stable).  The reference's own order (map_utils.h:373: landmarks, then their observations)
  // -- what include/visnav_amd/bundle_adjustment.h hands over -- is sorted already: one copy, no scatter
  int* lm_start = (int*)(blk + y.lm_start);
  unsigned* meta = (unsigned*)(blk + y.obs_meta);
  double* s_uv = (double*)(blk + y.obs_uv);
  memset(lm_start, 0, 4 * ((size_t)L + 1));
  bool sorted = true;
  for (int i = 0; i < O; i++) {
    lm_start[p->obs_lm[i] + 1]++;
    if (i > 0 && p->obs_lm[i] < p->obs_lm[i - 1]) sorted = false;
  }
  for (int l = 0; l < L; l++) lm_start[l + 1] += lm_start[l];
  if (sorted) {
    memcpy(s_uv, p->obs_uv, 16 * (size_t)O);
    // (the camera field is written together with the rest of the word in the landmark loop below: one pass less)
  } else {
    unsigned* fill = (unsigned*)(blk + y.lm_pres);  // borrowed as the scatter cursor; rewritten below
    for (int l = 0; l < L; l++) fill[l] = (unsigned)lm_start[l];
    for (int i = 0; i < O; i++) {
      const unsigned q = fill[p->obs_lm[i]]++;
      meta[q] = (unsigned)p->obs_cam[i] << 26;
      s_uv[2 * (size_t)q] = p->obs_uv[2 * (size_t)i];
      s_uv[2 * (size_t)q + 1] = p->obs_uv[2 * (size_t)i + 1];
    }
  }
  // workgroups: contiguous landmark ranges balanced by observation count, one thread per observation
  unsigned* pres = (unsigned*)(blk + y.lm_pres);
  int* wg_info = (int*)(blk + y.wg_info);
  // (as many workgroups as compute units as soon as each gets ~100 observations: the per-workgroup cost is the tile
  // products on the fp64 matrix unit, proportional to its landmarks -- a 25 k-observation window on 78 workgroups
  // spent 49 us in this kernel, on 256 it is the fixed costs only)
  int G0 = std::max(1, std::min(256, (O + 95) / 96));
  if ((O + G0 - 1) / G0 > 960) G0 = (O + 959) / 960;
  int l = 0, g = 0;
  int cnt[BF_CAMS + 1];
  while (l < L) {
    if (g >= y.g_cap) return false;
    const long long target = (long long)O * (g + 1) / G0;  // cumulative observations this workgroup should reach
    int* rec = wg_info + (size_t)BF_INFO * g;
    memset(rec, 0, 4 * BF_INFO);
    for (int c = 0; c <= nfree; c++) cnt[c] = 0;
    const int lm0 = l, obs0 = lm_start[l];
    int n_ch = 0, ch_lm = 0;
    while (l < L) {
      const int a = lm_start[l], b = lm_start[l + 1];
      if (b - a > 64) return false;
      const bool wg_has = l > lm0;
      if (wg_has && (b - obs0 > BF_OBS_CAP || l - lm0 >= BF_LMW)) break;
      if (wg_has && a >= target && g + 1 < G0) break;
      if (ch_lm >= pl.lc_max) {  // the chunk is full
        if (n_ch + 1 >= BF_MAXCH) break;  // the next workgroup takes this landmark
        n_ch++;
        rec[BF_INFO_CB + n_ch] = l - lm0;
        ch_lm = 0;
      }
      // which free cameras see the landmark; a camera seeing it twice would write one operand entry twice
      unsigned m = 0;
      const unsigned word = bf_pack(0, (unsigned)ch_lm, (unsigned)n_ch, (unsigned)(l - lm0), 0);
      for (int q = a; q < b; q++) {
        const unsigned cam = sorted ? (unsigned)p->obs_cam[q] : meta[q] >> 26;
        const int fc = cam_free[cam];
        if (fc >= 0) {
          if (m & (1u << fc)) return false;
          m |= 1u << fc;
          cnt[fc + 1]++;  // (camera-major counts of the workgroup: gathered here, not in a pass of their own)
        }
        meta[q] = (cam << 26) | word;
      }
      pres[l] = m;
      ch_lm++;
      l++;
    }
    n_ch++;
    const int obs1 = lm_start[l];
    rec[0] = lm0;
    rec[1] = l - lm0;
    rec[2] = obs0;
    rec[3] = obs1 - obs0;
    rec[4] = n_ch;
    for (int c = n_ch; c <= BF_MAXCH; c++) rec[BF_INFO_CB + c] = l - lm0;
    if (rec[3] > BF_OBS_CAP) return false;  // (a single landmark has <= 64 observations: a logic guard)
    // camera-major ranks of the free-camera observations of this workgroup
    for (int c = 0; c < nfree; c++) cnt[c + 1] += cnt[c];
    for (int c = 0; c <= nfree; c++) rec[BF_INFO_CAM + c] = cnt[c];
    for (int q = obs0; q < obs1; q++) {
      const int fc = cam_free[meta[q] >> 26];
      if (fc >= 0) meta[q] |= (unsigned)cnt[fc]++;
    }
    g++;
  }
  pl.G = g;
  return g > 0;
}

#define BF_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return vsl_fail(ctx, VSL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)

struct ArenaLoan {  // the context's cached BA arena, or a private allocation when that one is lent out
  vsl_ctx* ctx = nullptr;
  void* p = nullptr;
  bool owned = false, lent = false;
  ~ArenaLoan() {
    if (owned && p) (void)hipFree(p);
    if (lent) ctx->ba_arena_busy = false;
  }
};

}  // namespace

// handled = 0: the problem does not fit the fused kernels (nothing was done); otherwise the solve ran (rc tells how).
int vsl_ba_fused_solve(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary,
                       int* handled) {
  *handled = 0;
  if (prob->n_cams > BF_CAMS) return VSL_OK;
  const double t_start = bf_now_ms();
  const bool trace = getenv("VSL_BA_TRACE") != nullptr;  // phase times on stderr (developer aid)
  double t_lap = t_start;
  auto lap = [&](const char* what) {
    if (!trace) return;
    const double t = bf_now_ms();
    fprintf(stderr, "  [fused ba] %-32s %8.3f ms\n", what, t - t_lap);
    t_lap = t;
  };
  BF_HIP(hipSetDevice(ctx->device));
  const size_t C = prob->n_cams, L = prob->n_lms, O = prob->n_obs;
  const BfLayout y = bf_layout((int)C, (int)L, (int)O);
  // the pinned block: the plan first, the kernels' mailbox behind it (one allocation, device-mapped)
  //   mailbox: [0] cost, [1] max |landmark gradient|, [2] (int) Cholesky ok, [4 .. 132) |camera gradient| per unknown,
  //            [132 .. 132 + 5 G) step partials
  static const bool env_host_lm = getenv("VSL_BA_HOST_LM") != nullptr;
  bool device_lm = !ctx->ba_host_lm && !env_host_lm;
  // (device-decided loop: the pinned area holds one BF_REC-double record per loop body instead of the kernels' outputs)
  const size_t mail_doubles = std::max<size_t>(132 + 5 * (size_t)y.g_cap, (size_t)BF_REC * ((size_t)std::max(opt->max_num_iterations, 0) + 4));
  const size_t pin_bytes = y.bytes + 8 * mail_doubles;
  if (ctx->ba_pin_cap < pin_bytes) {
    BF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->ba_pin) (void)hipHostFree(ctx->ba_pin);
    ctx->ba_pin = nullptr;
    ctx->ba_pin_cap = 0;
    const size_t cap = pin_bytes + pin_bytes / 4;
    BF_HIP(hipHostMalloc((void**)&ctx->ba_pin, cap, hipHostMallocMapped | hipHostMallocCoherent));
    ctx->ba_pin_cap = cap;
  }
  char* blk = (char*)ctx->ba_pin;
  double* mailbox = (double*)(blk + y.bytes);
  BfPlan pl;
  if (!bf_plan(prob, y, blk, pl)) return VSL_OK;
  *handled = 1;
  if (pl.G > 1024) device_lm = false;  // (baf_decide_kernel stages 5 partials of <= 1024 workgroups)
  lap("plan");
  BaDims D;
  D.C = prob->n_cams;
  D.L = prob->n_lms;
  D.O = prob->n_obs;
  D.nfree = pl.nfree;
  D.n = 6 * pl.nfree;
  D.model0 = prob->cam_model[0];
  D.model1 = prob->cam_model[1];
  D.use_huber = opt->use_huber;
  D.huber = opt->huber_parameter;
  const int n = D.n, G = pl.G, NT = pl.NP / 16, T = NT * (NT + 1) / 2;
  // one arena: the uploaded block, then what the kernels produce
  struct Want {
    void** p;
    size_t bytes;
  };
  char* dblk;
  double *cand_poses, *cand_points, *scale_c, *scale_l, *Pinv, *bl, *S_part, *hc_part, *sc_part, *S, *rhs, *dc, *box;
  BfLm* lm_dev;
  std::vector<Want> want = {{(void**)&dblk, y.bytes}, {(void**)&cand_poses, 56 * C}, {(void**)&cand_points, 24 * L},
                            {(void**)&box, 8 * (132 + 5 * (size_t)y.g_cap)}, {(void**)&lm_dev, sizeof(BfLm)},
                            {(void**)&scale_c, 8 * 128}, {(void**)&scale_l, 24 * L}, {(void**)&Pinv, 72 * L},
                            {(void**)&bl, 24 * L}, {(void**)&S_part, 2048 * (size_t)T * G},
                            {(void**)&hc_part, 216 * (size_t)pl.nfree * G}, {(void**)&sc_part, 16 * (size_t)G},
                            {(void**)&S, 8 * (size_t)n * n}, {(void**)&rhs, 8 * 128}, {(void**)&dc, 8 * 128}};
  size_t total = 0;
  for (auto& w : want) total += (std::max<size_t>(w.bytes, 8) + 255) & ~(size_t)255;
  ArenaLoan loan;
  loan.ctx = ctx;
  if (ctx->ba_arena_busy) {
    BF_HIP(hipMalloc(&loan.p, total));
    loan.owned = true;
  } else {
    if (ctx->ba_arena_cap < total) {
      BF_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->ba_arena) (void)hipFree(ctx->ba_arena);
      ctx->ba_arena = nullptr;
      ctx->ba_arena_cap = 0;
      const size_t cap = total + total / 4;
      BF_HIP(hipMalloc(&ctx->ba_arena, cap));
      ctx->ba_arena_cap = cap;
    }
    loan.p = ctx->ba_arena;
    ctx->ba_arena_busy = true;
    loan.lent = true;
  }
  {
    size_t off = 0;
    for (auto& w : want) {
      *w.p = (char*)loan.p + off;
      off += (std::max<size_t>(w.bytes, 8) + 255) & ~(size_t)255;
    }
  }
  volatile double* mail = mailbox;
  volatile int* chol_ok = (volatile int*)(mailbox + 2);
  volatile double* gabs = mailbox + 4;
  volatile double* step_part = mailbox + 132;
  // ONE copy: everything up to the used part of the workgroup records
  BF_HIP(hipMemcpyAsync(dblk, blk, y.wg_info + 4 * (size_t)BF_INFO * G, hipMemcpyHostToDevice, ctx->stream));
  double* poses = (double*)(dblk + y.poses);
  double* points = (double*)(dblk + y.points);

  BfArgs a;
  a.D = D;
  a.NP = pl.NP;
  a.NPs = pl.NPs;
  a.NT = NT;
  a.T = T;
  a.poses = poses;
  a.points = points;
  a.lm = device_lm ? lm_dev : nullptr;
  a.poses_alt = cand_poses;
  a.points_alt = cand_points;
  a.intr = (const double*)(dblk + y.intr);
  a.cam_intr = (const int*)(dblk + y.cam_intr);
  a.cam_free = (const int*)(dblk + y.cam_free);
  a.obs_uv = (const double*)(dblk + y.obs_uv);
  a.obs_meta = (const unsigned*)(dblk + y.obs_meta);
  a.lm_start = (const int*)(dblk + y.lm_start);
  a.lm_pres = (const unsigned*)(dblk + y.lm_pres);
  a.wg_info = (const int*)(dblk + y.wg_info);
  a.scale_c = scale_c;
  a.scale_l = scale_l;
  a.Pinv = Pinv;
  a.bl = bl;
  a.S_part = S_part;
  a.hc_part = hc_part;
  a.sc_part = sc_part;

  vsl_ba_summary sum;
  memset(&sum, 0, sizeof(sum));
  const bool prof_was = ctx->profiling;
  // summary: linearize_ms = scaling pass + step kernels, schur_ms = Schur + finish kernels, solve_ms = the Cholesky
  const int st_of[5] = {VSL_STAGE_BA_LIN, VSL_STAGE_BA_SCHUR, VSL_STAGE_BA_SOLVE, VSL_STAGE_BA_FINISH, VSL_STAGE_BA_STEP};
  double base_ms[5];
  for (int k = 0; k < 5; k++) base_ms[k] = ctx->stage_ms[st_of[k]];

  if (device_lm)  // (sequence numbers of an earlier solve must not be mistaken for this one's)
    for (int k = 0; k < std::max(opt->max_num_iterations, 0) + 3; k++) mailbox[(size_t)BF_REC * k] = 0.0;
  // Jacobi scaling from the unscaled Jacobian + the initial cost
  {
    VslStage s(ctx, VSL_STAGE_BA_LIN);
    hipLaunchKernelGGL((baf_schur_kernel<true, 1>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, 0.0);
    hipLaunchKernelGGL(baf_init_finish_kernel, dim3((n + 15) / 16 + 1), dim3(256), 0, ctx->stream, n, pl.nfree, G, hc_part,
                       sc_part, scale_c, device_lm ? box : mailbox);
    if (device_lm) hipLaunchKernelGGL(baf_lm_init_kernel, dim3(1), dim3(64), 0, ctx->stream, lm_dev, box, mailbox);
    VSL_CHECK_LAUNCH(ctx);
  }
  if (device_lm) {
    // ---- the device-decided loop: iterations are enqueued ONE AHEAD of the decision the host has seen; a body that
    // runs after the termination returns at its first instruction.  No synchronisation inside the loop: the host polls
    // the pinned records.
    volatile double* recs = mailbox;
    auto enqueue_body = [&]() -> int {
      {
        VslStage s(ctx, VSL_STAGE_BA_SCHUR);
        if (T <= BF_WAVES)
          hipLaunchKernelGGL((baf_schur_kernel<false, 1>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, 0.0);
        else if (T <= 2 * BF_WAVES)
          hipLaunchKernelGGL((baf_schur_kernel<false, 2>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, 0.0);
        else
          hipLaunchKernelGGL((baf_schur_kernel<false, 3>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, 0.0);
      }
      {
        VslStage s(ctx, VSL_STAGE_BA_FINISH);
        hipLaunchKernelGGL(baf_finish_kernel, dim3(16 * T + 1), dim3(256), 0, ctx->stream, n, pl.nfree, G, T, S_part, hc_part,
                           sc_part, scale_c, 0.0, S, rhs, box, box + 4, (const BfLm*)lm_dev);
      }
      {
        VslStage s(ctx, VSL_STAGE_BA_SOLVE);
        hipLaunchKernelGGL(baf_chol_kernel, dim3(1), dim3(BF_THREADS), 0, ctx->stream, n, S, rhs, dc, (int*)(box + 2),
                           (const BfLm*)lm_dev);
      }
      {
        VslStage s(ctx, VSL_STAGE_BA_STEP);
        hipLaunchKernelGGL(baf_step_kernel, dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, dc, cand_poses, cand_points, box + 132);
        hipLaunchKernelGGL(baf_decide_kernel, dim3(1), dim3(64), 0, ctx->stream, lm_dev, box, n, G, opt->max_num_iterations,
                           mailbox);
      }
      VSL_CHECK_LAUNCH(ctx);
      return VSL_OK;
    };
    const int max_bodies = std::max(opt->max_num_iterations, 0);
    int enq = 0, rc2;
    // (with the stage events on, a body is enqueued only once the previous decision is known: the speculative body that
    // finds `done` would count as a launch of every stage and pull the per-launch averages down)
    const int lead = ctx->profiling ? 1 : 2;
    for (; enq < std::min(lead, max_bodies); enq++)
      if ((rc2 = enqueue_body())) return rc2;
    double last[BF_REC];
    memset(last, 0, sizeof(last));
    int seen = 0;
    bool over = max_bodies == 0;
    if (opt->verbosity >= 2) fprintf(stderr, "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n");
    while (!over) {
      volatile double* r = recs + (size_t)BF_REC * (seen + 1);
      // wait for the decision of body `seen` (spin on pinned memory; a device error ends the wait through the query)
      for (long spins = 0; r[0] != (double)(seen + 1); spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xfffff) == 0xfffff) {
          const hipError_t qe = hipStreamQuery(ctx->stream);
          if (qe != hipSuccess && qe != hipErrorNotReady)
            return vsl_fail(ctx, VSL_ERR_HIP, "fused bundle adjustment: %s while waiting for the device's decision", hipGetErrorString(qe));
          if (qe == hipSuccess && r[0] != (double)(seen + 1))
            return vsl_fail(ctx, VSL_ERR_HIP, "fused bundle adjustment: the stream drained without the decision of body %d", seen);
        }
      }
      for (int k = 0; k < BF_REC; k++) last[k] = r[k];
      seen++;
      if (opt->verbosity >= 2) {
        if (last[13] != 0.0)
          fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", (int)last[3], last[10], last[11], last[7], last[8], last[9], last[6]);
        else
          fprintf(stderr, "%4d  invalid step or termination before a step, radius %.3e\n", (int)last[3], last[6]);
      }
      if (last[1] != 0.0) break;  // done
      if (enq < max_bodies) {
        if ((rc2 = enqueue_body())) return rc2;
        enq++;
      } else if (seen == enq) {
        over = true;  // (cannot happen: the last allowed body sets done)
      }
    }
    BF_HIP(hipStreamSynchronize(ctx->stream));
    lap("scaling pass + LM loop (device-decided)");
    sum.initial_cost = recs[5];
    if (seen > 0) {
      sum.iterations = (int)last[3];
      sum.successful_steps = (int)last[4];
      sum.final_cost = last[5];
      sum.termination = (int)last[2] < 0 ? 0 : (int)last[2];
      if ((int)last[12]) {
        std::swap(poses, cand_poses);
        std::swap(points, cand_points);
      }
    } else {
      sum.final_cost = sum.initial_cost;
    }
  } else {
  BF_HIP(hipStreamSynchronize(ctx->stream));
  lap("arena + upload + scaling pass");
  double cost = mail[0];
  sum.initial_cost = cost;

  double radius = 1e4, decrease_factor = 2.0;
  int iteration = 0, invalid = 0;
  sum.termination = 0;
  if (opt->verbosity >= 2)
    fprintf(stderr, "iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n%4d % .6e\n", 0, cost);
  while (true) {
    if (iteration >= opt->max_num_iterations) { sum.termination = 0; break; }
    // the whole iteration is enqueued without waiting; the gradient norm of the CURRENT point comes back with it
    // (it is a by-product of the Schur kernel), so the gradient-tolerance test is taken before the iteration counts
    const double inv_radius = 1.0 / radius;
    {
      VslStage s(ctx, VSL_STAGE_BA_SCHUR);
      if (T <= BF_WAVES)
        hipLaunchKernelGGL((baf_schur_kernel<false, 1>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, inv_radius);
      else if (T <= 2 * BF_WAVES)
        hipLaunchKernelGGL((baf_schur_kernel<false, 2>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, inv_radius);
      else
        hipLaunchKernelGGL((baf_schur_kernel<false, 3>), dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, inv_radius);
      VSL_CHECK_LAUNCH(ctx);
    }
    {
      VslStage s(ctx, VSL_STAGE_BA_FINISH);
      hipLaunchKernelGGL(baf_finish_kernel, dim3(16 * T + 1), dim3(256), 0, ctx->stream, n, pl.nfree,
                         G, T, S_part, hc_part, sc_part, scale_c, inv_radius, S, rhs, mailbox, mailbox + 4, (const BfLm*)nullptr);
      VSL_CHECK_LAUNCH(ctx);
    }
    {
      VslStage s(ctx, VSL_STAGE_BA_SOLVE);
      hipLaunchKernelGGL(baf_chol_kernel, dim3(1), dim3(BF_THREADS), 0, ctx->stream, n, S, rhs, dc, (int*)(mailbox + 2),
                         (const BfLm*)nullptr);
      VSL_CHECK_LAUNCH(ctx);
    }
    {
      VslStage s(ctx, VSL_STAGE_BA_STEP);
      hipLaunchKernelGGL(baf_step_kernel, dim3(G), dim3(BF_THREADS), 0, ctx->stream, a, dc, cand_poses, cand_points,
                         mailbox + 132);
      VSL_CHECK_LAUNCH(ctx);
    }
    BF_HIP(hipStreamSynchronize(ctx->stream));
    double gmax = mail[1];
    for (int x = 0; x < n; x++) gmax = std::max(gmax, (double)gabs[x]);
    if (gmax <= 1e-10) { sum.termination = 2; break; }
    if (radius <= 1e-32) { sum.termination = 4; break; }
    iteration++;
    double model_change = 0, cand_cost = 0, step2 = 0, x2 = 0, bad = 0;
    for (int g = 0; g < G; g++) {  // workgroup order: fixed
      model_change += step_part[5 * g];
      cand_cost += step_part[5 * g + 1];
      step2 += step_part[5 * g + 2];
      x2 += step_part[5 * g + 3];
      bad += step_part[5 * g + 4];
    }
    const double step_norm = sqrt(step2), x_norm = sqrt(x2);
    const bool ok = *chol_ok != 0 && bad == 0.0 && model_change > 0.0;
    if (!ok) {
      if (++invalid >= 5) { sum.termination = 4; break; }
      radius *= 0.5;
      if (opt->verbosity >= 2) fprintf(stderr, "%4d  invalid step, radius %.3e\n", iteration, radius);
      continue;
    }
    invalid = 0;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum.termination = 3; break; }
    const double cost_change = cost - cand_cost;
    if (fabs(cost_change) <= 1e-6 * cost) { sum.termination = 1; break; }
    const double rel = cost_change / model_change;
    if (opt->verbosity >= 2)
      fprintf(stderr, "%4d % .6e % .3e % .3e % .3e % .3e % .3e\n", iteration, cand_cost, cost_change, gmax, step_norm, rel, radius);
    if (rel > 1e-3) {
      cost = cand_cost;
      std::swap(poses, cand_poses);
      std::swap(points, cand_points);
      a.poses = poses;
      a.points = points;
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
  lap("LM loop");
  }  // host-decided loop
  BF_HIP(hipMemcpyAsync(prob->poses, poses, 56 * C, hipMemcpyDeviceToHost, ctx->stream));
  BF_HIP(hipMemcpyAsync(prob->points, points, 24 * L, hipMemcpyDeviceToHost, ctx->stream));
  BF_HIP(hipStreamSynchronize(ctx->stream));
  double ms;
  int64_t cnt;
  vsl_ctx_stage_ms(ctx, VSL_STAGE_BA_LIN, &ms, &cnt);  // drains the pending stage events
  double dms[5];
  for (int k = 0; k < 5; k++) dms[k] = ctx->stage_ms[st_of[k]] - base_ms[k];
  sum.linearize_ms = dms[0] + dms[4];
  sum.schur_ms = dms[1] + dms[3];
  sum.solve_ms = dms[2];
  vsl_ctx_set_profiling(ctx, prof_was ? 1 : 0);
  lap("download");
  sum.total_ms = bf_now_ms() - t_start;
  if (opt->verbosity >= 1)
    fprintf(stderr, "vsl BA: iterations %d, initial cost %.6e, final cost %.6e, termination %d, %.3f ms\n", sum.iterations,
            sum.initial_cost, sum.final_cost, sum.termination, sum.total_ms);
  if (summary) *summary = sum;
  return VSL_OK;
}
