// vo.hip -- per-frame landmark projection and guided descriptor matching.
//
// Replaces visnav::project_landmarks (include/visnav/vo_utils.h:48-81) and
// visnav::find_matches_landmarks (include/visnav/vo_utils.h:83-167), which src/slam.cpp runs on EVERY
// frame (:1099-1114, :1159, :1339).
//
//  * projection: one thread per landmark, fp64 in the oracle's operation order (no FMA), then an
//    order-preserving compaction -- the output order is the caller's landmark order (the reference
//    iterates its unordered_map);
//  * matching: one wavefront per keypoint.  Lanes test 64 projected points at a time against the 2-D
//    radius (double, sqrt(dx*dx + dy*dy) < r like Eigen's norm()); for every hit, in order, the lanes
//    stride the landmark's observation descriptors and a wave-wide minimum gives the landmark
//    distance.  The reference then calls std::partial_sort(first, first + 2, last) on the (landmark,
//    distance) list; which of two EQUALLY distant landmarks comes first is libstdc++'s heap-select
//    behaviour, reproduced here as the equivalent streaming state machine over the list:
//        first two:   top = (d1 < d0) ? e0 : e1,  other = the other one
//        each later e with d(e) < d(top):   (top, other) = d(other) < d(e) ? (e, other) : (other, e)
//        result[0] = other, result[1] = top
//    so ties are broken exactly like the reference (pinned against the oracle, which calls the real
//    std::partial_sort).
#include <cmath>

#include "vsl_common.h"

namespace {

__device__ __forceinline__ void quat_rotate_d(const double* q, const double* p, double* out) {
  double uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int i = 0; i < 3; i++) uv[i] = uv[i] + uv[i];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) out[i] = p[i] + q[3] * uv[i] + c[i];
}

// camera_models.h project(), expression for expression (include/visnav/camera_models.h:75-94, :158-178,
// :246-270, :341-374)
__device__ __forceinline__ void project_exact(int model, const double* ip, double x, double y, double z, double& u,
                                              double& v) {
  const double fx = ip[0], fy = ip[1], cx = ip[2], cy = ip[3];
  if (model == VSL_CAM_PINHOLE) {
    u = fx * x / z + cx;
    v = fy * y / z + cy;
  } else if (model == VSL_CAM_EUCM) {
    const double alpha = ip[4], beta = ip[5];
    const double d = sqrt(beta * (x * x + y * y) + z * z);
    u = fx * x / (alpha * d + (1.0 - alpha) * z) + cx;
    v = fy * y / (alpha * d + (1.0 - alpha) * z) + cy;
  } else if (model == VSL_CAM_KB4) {
    const double k1 = ip[4], k2 = ip[5], k3 = ip[6], k4 = ip[7];
    const double r = sqrt(x * x + y * y);
    const double theta = atan2(r, z);
    const double d = theta + k1 * theta * theta * theta + k2 * theta * theta * theta * theta * theta +
                     k3 * theta * theta * theta * theta * theta * theta * theta +
                     k4 * theta * theta * theta * theta * theta * theta * theta * theta * theta;
    if (r == 0.0) {
      u = cx;
      v = cy;
    } else {
      u = fx * d * x / r + cx;
      v = fy * d * y / r + cy;
    }
  } else {
    const double xi = ip[4], alpha = ip[5];
    const double d1 = sqrt(x * x + y * y + z * z);
    const double d2 = sqrt(x * x + y * y + (xi * d1 + z) * (xi * d1 + z));
    u = fx * x / (alpha * d2 + (1.0 - alpha) * (xi * d1 + z)) + cx;
    v = fy * y / (alpha * d2 + (1.0 - alpha) * (xi * d1 + z)) + cy;
  }
}

__global__ __launch_bounds__(256) void project_landmarks_kernel(const double* __restrict__ pose, int model,
                                                                const double* __restrict__ intr, int width, int height,
                                                                const double* __restrict__ points, int n, double z_thr,
                                                                double* __restrict__ uv, uint8_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double qi[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
  const double nt[3] = {pose[4] * -1.0, pose[5] * -1.0, pose[6] * -1.0};
  double ti[3], rp[3];
  quat_rotate_d(qi, nt, ti);
  const double p[3] = {points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]};
  quat_rotate_d(qi, p, rp);
  const double pc[3] = {rp[0] + ti[0], rp[1] + ti[1], rp[2] + ti[2]};
  bool ok = !(pc[2] < z_thr);
  double u = 0, v = 0;
  if (ok) {
    project_exact(model, intr, pc[0], pc[1], pc[2], u, v);
    ok = !(u > (double)width || v > (double)height || u < 0 || v < 0);
  }
  uv[2 * (size_t)i] = u;
  uv[2 * (size_t)i + 1] = v;
  keep[i] = ok ? 1 : 0;
}

// order-preserving compaction of (uv, index) by keep[]; one workgroup
__global__ __launch_bounds__(1024) void compact_projection_kernel(const double* __restrict__ uv, const uint8_t* __restrict__ keep,
                                                                  int n, double* __restrict__ out_uv, int32_t* __restrict__ out_idx,
                                                                  int32_t* __restrict__ n_out) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const bool ok = i < n && keep[i];
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    if (ok) {
      const int p = off + __popcll(m & ((1ull << lane) - 1ull));
      out_uv[2 * (size_t)p] = uv[2 * (size_t)i];
      out_uv[2 * (size_t)p + 1] = uv[2 * (size_t)i + 1];
      out_idx[p] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; w++) t += wave_tot[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = base_s;
}

// project_landmarks + the order-preserving compaction in ONE launch (vsl_map_track): a workgroup projects 1024 landmarks,
// compacts them in landmark order, and takes its base offset from the chain of its predecessors' totals ("stream
// scan": block b waits for block b - 1's running total, published as one 64-bit word (epoch << 32 | total) with
// release / acquire at device scope -- no flags to reset between calls, the epoch changes).  Forward progress of the
// wait: up to VO_CHAIN_RESIDENT_BLOCKS workgroups (one per compute unit: every one of them is resident from the start)
// the hardware's index-order dispatch makes blockIdx the chain position; larger maps (> 262 k landmarks) draw their
// chain position from an atomic ticket instead, so a workgroup only ever waits for workgroups that have STARTED --
// the decoupled look-back rule -- whatever the dispatch order (the ticket word is reset by the workgroup that draws
// the last one).
#define VO_CHAIN_RESIDENT_BLOCKS 256
struct PoseIntr {
  double v[16];  // pose (qx qy qz qw tx ty tz, pad) | intrinsics (8)
};
__global__ __launch_bounds__(1024) void project_compact_kernel(PoseIntr pi, int model, int width, int height,
                                                               const double* __restrict__ points, int n, double z_thr,
                                                               double* __restrict__ out_uv, int32_t* __restrict__ out_idx,
                                                               int32_t* __restrict__ n_out, unsigned long long* __restrict__ chain,
                                                               unsigned int epoch, unsigned int* __restrict__ ticket) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  __shared__ unsigned int bid_s;
  unsigned int bid = blockIdx.x;
  if (ticket) {  // workgroup-uniform
    if (threadIdx.x == 0) {
      const unsigned int t = atomicAdd(ticket, 1u);
      if (t == gridDim.x - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all drawn: ready for the next call
      bid_s = t;
    }
    __syncthreads();
    bid = bid_s;
  }
  const int i = (int)bid * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool ok = false;
  double u = 0, v = 0;
  if (i < n) {
    const double* pose = pi.v;
    const double qi[4] = {-pose[0], -pose[1], -pose[2], pose[3]};
    const double nt[3] = {pose[4] * -1.0, pose[5] * -1.0, pose[6] * -1.0};
    double ti[3], rp[3];
    quat_rotate_d(qi, nt, ti);
    const double p[3] = {points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]};
    quat_rotate_d(qi, p, rp);
    const double pc[3] = {rp[0] + ti[0], rp[1] + ti[1], rp[2] + ti[2]};
    ok = !(pc[2] < z_thr);
    if (ok) {
      project_exact(model, pi.v + 8, pc[0], pc[1], pc[2], u, v);
      ok = !(u > (double)width || v > (double)height || u < 0 || v < 0);
    }
  }
  const unsigned long long m = __ballot(ok);
  if (lane == 0) wave_tot[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < 16; w++) total += wave_tot[w];
    unsigned long long prev = 0;
    if (bid > 0) {
      do {
        prev = __hip_atomic_load(&chain[bid - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      } while ((unsigned int)(prev >> 32) != epoch);
    }
    const int base = (int)(unsigned int)prev;
    __hip_atomic_store(&chain[bid], ((unsigned long long)epoch << 32) | (unsigned int)(base + total), __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
    base_s = base;
    if (bid == gridDim.x - 1) *n_out = base + total;
  }
  __syncthreads();
  if (ok) {
    int off = base_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    const int p = off + __popcll(m & ((1ull << lane) - 1ull));
    out_uv[2 * (size_t)p] = u;
    out_uv[2 * (size_t)p + 1] = v;
    out_idx[p] = i;
  }
}

// The smallest double T with !(sqrt(T) < m): for x >= 0, sqrt(x) < m <=> x < T, because the correctly rounded square
// root is monotone (host libm and the device's fp64 sqrt are both correctly rounded).  Bisection over the bit patterns
// of the non-negative doubles (they order like the values).
static double sqrt_less_threshold(double m) {
  if (!(m > 0.0)) return 0.0;  // sqrt(x) < m never holds for x >= 0
  if (std::isinf(m)) return m;
  uint64_t lo = 0, hi;          // invariant: sqrt(value(lo)) < m, !(sqrt(value(hi)) < m)
  {
    const double inf = INFINITY;
    memcpy(&hi, &inf, 8);
  }
  while (hi - lo > 1) {
    const uint64_t mid = lo + (hi - lo) / 2;
    double v;
    memcpy(&v, &mid, 8);
    if (std::sqrt(v) < m) lo = mid; else hi = mid;
  }
  double T;
  memcpy(&T, &hi, 8);
  return T;
}

// one wavefront per keypoint; result[k] = matched landmark index or -1
__global__ __launch_bounds__(256) void find_matches_kernel(const double* __restrict__ kp_xy, const uint64_t* __restrict__ kp_desc,
                                                           int n_kp, const double* __restrict__ proj_uv,
                                                           const int32_t* __restrict__ proj_lm, int n_proj,
                                                           const int32_t* __restrict__ lm_obs_start,
                                                           const uint64_t* __restrict__ obs_desc, double max_dist_sq,
                                                           int threshold, double dist_2_best, int32_t* __restrict__ result,
                                                           const int32_t* __restrict__ kp_xy_i32,
                                                           const int32_t* __restrict__ obs_index,
                                                           const int32_t* __restrict__ n_kp_dev, int result_cap,
                                                           const int32_t* __restrict__ n_proj_dev,
                                                           int32_t* __restrict__ mail_hdr,
                                                           const int32_t* __restrict__ tie_count_dev,
                                                           int32_t* __restrict__ mail_xy) {
  // Device-resident callers (vsl_map_track) pass the keypoints of a frame store slot (int32 positions,
  // count on the device), the number of projected landmarks on the device, and observation descriptors
  // through an index into the map's descriptor pool; the host-buffer entry point passes none of them.
  if (n_kp_dev) n_kp = *n_kp_dev;
  if (n_proj_dev) n_proj = *n_proj_dev;
  if (mail_hdr && blockIdx.x == 0 && threadIdx.x == 0) {  // the header of the caller's mailbox (vsl_map_track)
    mail_hdr[0] = n_proj;
    mail_hdr[1] = tie_count_dev ? *tie_count_dev : 0;
    mail_hdr[2] = n_kp;
  }
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= n_kp) {  // wave-uniform; the slot still gets its "no match" (the callers no longer memset the array)
    if (lane == 0 && k < result_cap) result[k] = -1;
    return;
  }
  const double kx = kp_xy_i32 ? (double)kp_xy_i32[2 * (size_t)k] : kp_xy[2 * (size_t)k];
  const double ky = kp_xy_i32 ? (double)kp_xy_i32[2 * (size_t)k + 1] : kp_xy[2 * (size_t)k + 1];
  if (mail_xy && lane == 0 && kp_xy_i32) {  // the keypoint's position rides along (the host needs it for PnP)
    mail_xy[2 * (size_t)k] = kp_xy_i32[2 * (size_t)k];
    mail_xy[2 * (size_t)k + 1] = kp_xy_i32[2 * (size_t)k + 1];
  }
  uint32_t d[8];
  {
    const uint32_t* p = (const uint32_t*)(kp_desc + 4 * (size_t)k);
#pragma unroll
    for (int q = 0; q < 8; q++) d[q] = p[q];
  }
  int count = 0, top_d = 0, other_d = 0, other_id = 0;
  for (int base = 0; base < n_proj; base += 64) {
    const int j = base + lane;
    bool hit = false;
    if (j < n_proj) {
      const double dx = kx - proj_uv[2 * (size_t)j], dy = ky - proj_uv[2 * (size_t)j + 1];
      // the reference tests (p_2d - kp).norm() < match_max_dist_2d (vo_utils.h:108); max_dist_sq is the host-computed
      // double T with sqrt(x) < match_max_dist_2d <=> x < T for every x >= 0 (sqrt_less_threshold below): the same
      // decisions bit for bit without ~40 instructions of fp64 square root per lane and chunk
      hit = dx * dx + dy * dy < max_dist_sq;
    }
    unsigned long long mask = __ballot(hit);
    while (mask) {
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int l = proj_lm[base + b];
      const int o0 = lm_obs_start[l], o1 = lm_obs_start[l + 1];
      int best = 256;  // minimal_dist, vo_utils.h:116
      for (int o = o0 + lane; o < o1; o += 64) {
        const uint32_t* od = (const uint32_t*)(obs_desc + 4 * (size_t)(obs_index ? obs_index[o] : o));
        int dist = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) dist += __builtin_popcount(d[q] ^ od[q]);
        best = min(best, dist);
      }
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) best = min(best, __shfl_xor(best, s));
      // libstdc++ partial_sort(first, first + 2, last) as a streaming state machine (see the file header)
      if (count == 0) {
        other_d = best;
        other_id = l;
      } else if (count == 1) {
        // e0 = (other_id, other_d) so far, e1 = (l, best)
        if (best < other_d) {  // d1 < d0: top = e0, other = e1
          top_d = other_d;
          other_d = best;
          other_id = l;
        } else {  // top = e1, other = e0
          top_d = best;
        }
      } else if (best < top_d) {
        if (other_d < best) {
          top_d = best;
        } else {
          top_d = other_d;
          other_d = best;
          other_id = l;
        }
      }
      count++;
    }
  }
  if (lane == 0) {
    int res = -1;
    if (count > 0 && !(other_d >= threshold)) {
      const double second = count < 2 ? 256.0 : (double)top_d;  // vo_utils.h:146-160
      if (!(second < (double)other_d * dist_2_best)) res = other_id;
    }
    result[k] = res;
  }
}

__global__ __launch_bounds__(1024) void compact_matches_kernel(const int32_t* __restrict__ result, int n, int32_t* __restrict__ pairs,
                                                               int32_t* __restrict__ n_out) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const int r = i < n ? result[i] : -1;
    const bool ok = r >= 0;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    if (ok) {
      const int p = off + __popcll(m & ((1ull << lane) - 1ull));
      pairs[2 * (size_t)p] = i;
      pairs[2 * (size_t)p + 1] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; w++) t += wave_tot[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = base_s;
}

}  // namespace

extern "C" int vsl_project_landmarks(vsl_ctx* ctx, const double* pose7, int cam_model, const double* intr8, int width,
                                     int height, const double* points, int n, double cam_z_threshold, double* proj_uv,
                                     int32_t* proj_idx, int* n_out) {
  if (!ctx || !pose7 || !intr8 || !n_out || n < 0 || (n > 0 && (!points || !proj_uv || !proj_idx)) || cam_model < 0 ||
      cam_model > 3)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_project_landmarks: bad arguments");
  *n_out = 0;
  if (n == 0) return VSL_OK;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t N = (size_t)n;
  void* d = nullptr;
  // pose (8) | intr (8) | points (3N) | uv (2N) | out_uv (2N) | out_idx (N i32) | n_out | keep (N u8)
  int rc = vsl_ctx_dscratch(ctx, 8 * (16 + 7 * N) + 4 * (N + 4) + N + 64, &d);
  if (rc) return rc;
  double* dpose = (double*)d;
  double* dintr = dpose + 8;
  double* dpts = dintr + 8;
  double* duv = dpts + 3 * N;
  double* douv = duv + 2 * N;
  int32_t* didx = (int32_t*)(douv + 2 * N);
  int32_t* dn = didx + N;
  uint8_t* dkeep = (uint8_t*)(dn + 4);
  VSL_HIP(ctx, hipMemcpyAsync(dpose, pose7, 56, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(dintr, intr8, 64, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(dpts, points, 24 * N, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(project_landmarks_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dpose, cam_model, dintr, width,
                     height, dpts, n, cam_z_threshold, duv, dkeep);
  hipLaunchKernelGGL(compact_projection_kernel, dim3(1), dim3(1024), 0, ctx->stream, duv, dkeep, n, douv, didx, dn);
  VSL_CHECK_LAUNCH(ctx);
  // one round trip: count and full-capacity outputs into pinned memory together
  void* hp = nullptr;
  rc = vsl_ctx_hpinned(ctx, 64 + 20 * N, &hp);
  if (rc) return rc;
  int32_t* hn = (int32_t*)hp;
  double* huv = (double*)((char*)hp + 64);
  int32_t* hidx = (int32_t*)(huv + 2 * N);
  VSL_HIP(ctx, hipMemcpyAsync(hn, dn, 4, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(huv, douv, 16 * N, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(hidx, didx, 4 * N, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t m = hn[0];
  *n_out = m;
  if (m > 0) {
    std::memcpy(proj_uv, huv, 16 * (size_t)m);
    std::memcpy(proj_idx, hidx, 4 * (size_t)m);
  }
  return VSL_OK;
}

extern "C" int vsl_find_matches_landmarks(vsl_ctx* ctx, const double* kp_xy, const uint64_t* kp_desc, int n_kp,
                                          const double* proj_uv, const int32_t* proj_lm, int n_proj,
                                          const int32_t* lm_obs_start, int n_lms, const uint64_t* obs_desc,
                                          double match_max_dist_2d, int feature_match_threshold,
                                          double feature_match_dist_2_best, int32_t* pairs, int* n_out) {
  if (!ctx || !n_out || n_kp < 0 || n_proj < 0 || n_lms < 0 || (n_kp > 0 && (!kp_xy || !kp_desc || !pairs)) ||
      (n_proj > 0 && (!proj_uv || !proj_lm || !lm_obs_start)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_find_matches_landmarks: bad arguments");
  *n_out = 0;
  if (n_kp == 0 || n_proj == 0) return VSL_OK;
  for (int j = 0; j < n_proj; j++)
    if (proj_lm[j] < 0 || proj_lm[j] >= n_lms) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_find_matches_landmarks: landmark index %d out of range", proj_lm[j]);
  const int total = lm_obs_start[n_lms];
  if (total < 0 || (total > 0 && !obs_desc)) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_find_matches_landmarks: bad observation arrays");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t K = (size_t)n_kp, P = (size_t)n_proj, L = (size_t)n_lms, T = (size_t)total;
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, 8 * (2 * K + 4 * K + 2 * P + 4 * T) + 4 * (P + L + 1 + K + 2 * K + 4) + 64, &d);
  if (rc) return rc;
  double* dkxy = (double*)d;
  uint64_t* dkd = (uint64_t*)(dkxy + 2 * K);
  double* dpuv = (double*)(dkd + 4 * K);
  uint64_t* dod = (uint64_t*)(dpuv + 2 * P);
  int32_t* dplm = (int32_t*)(dod + 4 * T);
  int32_t* dstart = dplm + P;
  int32_t* dres = dstart + L + 1;
  int32_t* dpairs = dres + K;
  int32_t* dn = dpairs + 2 * K;
  VSL_HIP(ctx, hipMemcpyAsync(dkxy, kp_xy, 16 * K, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(dkd, kp_desc, 32 * K, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(dpuv, proj_uv, 16 * P, hipMemcpyHostToDevice, ctx->stream));
  if (T) VSL_HIP(ctx, hipMemcpyAsync(dod, obs_desc, 32 * T, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(dplm, proj_lm, 4 * P, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(dstart, lm_obs_start, 4 * (L + 1), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(find_matches_kernel, dim3((n_kp + 3) / 4), dim3(256), 0, ctx->stream, dkxy, dkd, n_kp, dpuv, dplm, n_proj,
                     dstart, dod, sqrt_less_threshold(match_max_dist_2d), feature_match_threshold, feature_match_dist_2_best, dres,
                     (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, n_kp, (const int32_t*)nullptr,
                     (int32_t*)nullptr, (const int32_t*)nullptr, (int32_t*)nullptr);
  hipLaunchKernelGGL(compact_matches_kernel, dim3(1), dim3(1024), 0, ctx->stream, dres, n_kp, dpairs, dn);
  VSL_CHECK_LAUNCH(ctx);
  void* hp = nullptr;
  rc = vsl_ctx_hpinned(ctx, 64 + 8 * K, &hp);
  if (rc) return rc;
  int32_t* hn = (int32_t*)hp;
  int32_t* hpairs = (int32_t*)((char*)hp + 64);
  VSL_HIP(ctx, hipMemcpyAsync(hn, dn, 4, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(hpairs, dpairs, 8 * K, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int32_t m = hn[0];
  *n_out = m;
  if (m > 0) std::memcpy(pairs, hpairs, 8 * (size_t)m);
  return VSL_OK;
}

// ------------------------------------------------------------------------------------------------
// Device-resident map: landmark positions, the per-landmark lists of observation descriptors and the
// descriptor pool stay in HBM between frames; a frame is tracked against it with ONE call that chains
// projection, compaction and guided matching on the context's stream and returns only the matches.
// Same kernels, same results as vsl_project_landmarks + vsl_find_matches_landmarks on the same inputs
// (tests/test_vo_gpu.py::test_map_track_equals_host_buffer_path).
struct vsl_map {
  vsl_ctx* ctx = nullptr;
  int cap_lms = 0, cap_refs = 0, cap_pool = 0, cap_kp = 0;
  int n_lms = 0, n_refs = 0, n_pool = 0;
  double* points = nullptr;     // [cap_lms][3]
  int32_t* obs_start = nullptr;  // [cap_lms + 1]
  int32_t* obs_index = nullptr;  // [cap_refs] -> pool
  uint64_t* pool = nullptr;      // [cap_pool][4]
  // per-call scratch
  double* uv = nullptr;       // [cap_lms][2]
  double* out_uv = nullptr;   // [cap_lms][2]
  int32_t* out_idx = nullptr;  // [cap_lms]
  uint8_t* keep = nullptr;    // [cap_lms]
  double* pose_intr = nullptr;  // 16 doubles
  int32_t* counters = nullptr;  // [0] = n_proj, [1] = n_matches
  int32_t* result = nullptr;    // [cap_kp]
  int32_t* pairs = nullptr;     // [cap_kp][2]
  int32_t* gather_ids = nullptr;  // [cap_kp]
  unsigned long long* chain = nullptr;  // [cap_lms / 1024 + 1] running totals of project_compact_kernel
  int chain_cap = 0;
  unsigned int epoch = 0;
  int32_t* mailbox = nullptr;  // pinned host memory the matching kernel writes: [n_proj, tie_count, n_kp, pad | result[cap_kp]]
  int mailbox_cap = 0;
};

namespace {

template <class T>
int map_grow(vsl_ctx* ctx, T** p, size_t old_n, size_t new_n) {
  T* q = nullptr;
  VSL_HIP(ctx, hipMalloc((void**)&q, sizeof(T) * (new_n ? new_n : 1)));
  if (*p && old_n) VSL_HIP(ctx, hipMemcpyAsync(q, *p, sizeof(T) * old_n, hipMemcpyDeviceToDevice, ctx->stream));
  if (*p) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(*p);
  }
  *p = q;
  return VSL_OK;
}

int map_reserve_lms(vsl_map* m, int n, int refs) {
  vsl_ctx* ctx = m->ctx;
  int rc;
  if (n > m->cap_lms) {
    const int c = n + n / 2 + 1024;
    if ((rc = map_grow(ctx, &m->points, 0, 3 * (size_t)c))) return rc;
    if ((rc = map_grow(ctx, &m->obs_start, 0, (size_t)c + 1))) return rc;
    if ((rc = map_grow(ctx, &m->uv, 0, 2 * (size_t)c))) return rc;
    if ((rc = map_grow(ctx, &m->out_uv, 0, 2 * (size_t)c))) return rc;
    if ((rc = map_grow(ctx, &m->out_idx, 0, (size_t)c))) return rc;
    if ((rc = map_grow(ctx, &m->keep, 0, (size_t)c))) return rc;
    m->cap_lms = c;
  }
  if (refs > m->cap_refs) {
    const int c = refs + refs / 2 + 4096;
    if ((rc = map_grow(ctx, &m->obs_index, 0, (size_t)c))) return rc;
    m->cap_refs = c;
  }
  return VSL_OK;
}

int map_reserve_pool(vsl_map* m, int n) {
  if (n <= m->cap_pool) return VSL_OK;
  const int c = n + n / 2 + 4096;
  int rc = map_grow(m->ctx, &m->pool, 4 * (size_t)m->n_pool, 4 * (size_t)c);
  if (rc) return rc;
  m->cap_pool = c;
  return VSL_OK;
}

int map_reserve_kp(vsl_map* m, int n) {
  if (n <= m->cap_kp) return VSL_OK;
  int rc;
  if ((rc = map_grow(m->ctx, &m->result, 0, (size_t)n))) return rc;
  if ((rc = map_grow(m->ctx, &m->pairs, 0, 2 * (size_t)n))) return rc;
  if ((rc = map_grow(m->ctx, &m->gather_ids, 0, (size_t)n))) return rc;
  m->cap_kp = n;
  return VSL_OK;
}

__global__ void map_gather_desc_kernel(const uint64_t* __restrict__ frame_desc, const int32_t* __restrict__ ids, int n,
                                       uint64_t* __restrict__ pool_out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4 * n) return;
  pool_out[t] = frame_desc[4 * (size_t)ids[t >> 2] + (t & 3)];
}

}  // namespace

extern "C" int vsl_map_create(vsl_ctx* ctx, int cap_landmarks, int cap_descriptors, vsl_map** out) {
  if (!ctx || !out || cap_landmarks < 0 || cap_descriptors < 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_map_create: bad arguments");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_map* m = new vsl_map();
  m->ctx = ctx;
  int rc = map_reserve_lms(m, cap_landmarks > 0 ? cap_landmarks : 1, cap_descriptors > 0 ? cap_descriptors : 1);
  if (!rc) rc = map_reserve_pool(m, cap_descriptors > 0 ? cap_descriptors : 1);
  if (!rc) rc = map_grow(ctx, &m->pose_intr, 0, 16);
  if (!rc) rc = map_grow(ctx, &m->counters, 0, 4);
  if (rc) {
    delete m;
    return rc;
  }
  *out = m;
  return VSL_OK;
}

extern "C" void vsl_map_destroy(vsl_map* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  void* ptrs[] = {m->points, m->obs_start, m->obs_index, m->pool, m->uv, m->out_uv, m->out_idx, m->keep,
                  m->pose_intr, m->counters, m->result, m->pairs, m->gather_ids, m->chain};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (m->mailbox) (void)hipHostFree(m->mailbox);
  delete m;
}

extern "C" int vsl_map_append_descriptors(vsl_map* m, int n, const uint64_t* desc, int* first_index) {
  if (!m || n < 0 || (n > 0 && !desc) || !first_index) return VSL_ERR_INVALID;
  vsl_ctx* ctx = m->ctx;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = map_reserve_pool(m, m->n_pool + n);
  if (rc) return rc;
  *first_index = m->n_pool;
  if (n > 0) {
    VSL_HIP(ctx, hipMemcpyAsync(m->pool + 4 * (size_t)m->n_pool, desc, 32 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffer may go away
  }
  m->n_pool += n;
  return VSL_OK;
}

extern "C" int vsl_map_append_descriptors_from_frame(vsl_map* m, vsl_frames* f, int slot, int n, const int32_t* feature_ids,
                                                     int* first_index) {
  if (!m || !f || n < 0 || (n > 0 && !feature_ids) || !first_index || slot < 0 || slot >= f->max_images) return VSL_ERR_INVALID;
  vsl_ctx* ctx = m->ctx;
  for (int i = 0; i < n; i++)
    if (feature_ids[i] < 0 || feature_ids[i] >= f->F) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_map_append_descriptors_from_frame: feature id %d out of range", feature_ids[i]);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = map_reserve_pool(m, m->n_pool + n);
  if (!rc) rc = map_reserve_kp(m, n > f->F ? n : f->F);
  if (rc) return rc;
  if ((rc = vsl_resolve_ties(ctx, f, nullptr))) return rc;  // the descriptors copied must be final
  *first_index = m->n_pool;
  if (n > 0) {
    VSL_HIP(ctx, hipMemcpyAsync(m->gather_ids, feature_ids, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(map_gather_desc_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, ctx->stream,
                       f->kp_desc + 4 * (size_t)slot * f->F, m->gather_ids, n, m->pool + 4 * (size_t)m->n_pool);
    VSL_CHECK_LAUNCH(ctx);
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // feature_ids is the caller's
  }
  m->n_pool += n;
  return VSL_OK;
}

extern "C" int vsl_map_set_landmarks(vsl_map* m, int n, const double* points, const int32_t* obs_start,
                                     const int32_t* obs_pool_index) {
  if (!m || n < 0 || (n > 0 && (!points || !obs_start))) return VSL_ERR_INVALID;
  vsl_ctx* ctx = m->ctx;
  const int refs = n > 0 ? obs_start[n] : 0;
  if (n > 0 && obs_start[0] != 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_map_set_landmarks: obs_start[0] must be 0");
  for (int i = 0; i < n; i++)
    if (obs_start[i + 1] < obs_start[i]) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_map_set_landmarks: obs_start not monotone at %d", i);
  if (refs > 0 && !obs_pool_index) return VSL_ERR_INVALID;
  for (int i = 0; i < refs; i++)
    if (obs_pool_index[i] < 0 || obs_pool_index[i] >= m->n_pool)
      return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_map_set_landmarks: descriptor index %d outside the pool (%d)", obs_pool_index[i], m->n_pool);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = map_reserve_lms(m, n, refs);
  if (rc) return rc;
  if (n > 0) {
    VSL_HIP(ctx, hipMemcpyAsync(m->points, points, 24 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(m->obs_start, obs_start, 4 * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (refs) VSL_HIP(ctx, hipMemcpyAsync(m->obs_index, obs_pool_index, 4 * (size_t)refs, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  m->n_lms = n;
  m->n_refs = refs;
  return VSL_OK;
}

extern "C" int vsl_map_info(const vsl_map* m, int* n_landmarks, int* n_observation_refs, int* n_descriptors) {
  if (!m) return VSL_ERR_INVALID;
  if (n_landmarks) *n_landmarks = m->n_lms;
  if (n_observation_refs) *n_observation_refs = m->n_refs;
  if (n_descriptors) *n_descriptors = m->n_pool;
  return VSL_OK;
}

extern "C" int vsl_map_track(vsl_map* m, vsl_frames* f, int slot, const double* pose7, int cam_model, const double* intr8,
                             int width, int height, double cam_z_threshold, double match_max_dist_2d,
                             int feature_match_threshold, double feature_match_dist_2_best, int32_t* pairs, int* n_pairs,
                             int* n_projected) {
  return vsl_map_track_corners(m, f, slot, pose7, cam_model, intr8, width, height, cam_z_threshold, match_max_dist_2d,
                               feature_match_threshold, feature_match_dist_2_best, pairs, n_pairs, n_projected, nullptr, nullptr);
}

// vsl_map_track + the slot's keypoint positions (corners_xy[2 * max_features], *n_corners) in the same round trip:
// what the host needs for PnP without a second download.
extern "C" int vsl_map_track_corners(vsl_map* m, vsl_frames* f, int slot, const double* pose7, int cam_model, const double* intr8,
                                     int width, int height, double cam_z_threshold, double match_max_dist_2d,
                                     int feature_match_threshold, double feature_match_dist_2_best, int32_t* pairs,
                                     int* n_pairs, int* n_projected, double* corners_xy, int* n_corners) {
  if (!m || !f || !pose7 || !intr8 || !n_pairs || slot < 0 || slot >= f->max_images || cam_model < 0 || cam_model > 3)
    return VSL_ERR_INVALID;
  vsl_ctx* ctx = m->ctx;
  *n_pairs = 0;
  if (n_projected) *n_projected = 0;
  if (n_corners) *n_corners = 0;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = map_reserve_kp(m, f->F);
  if (rc) return rc;
  const int n = m->n_lms;
  if (n == 0) {
    if ((rc = vsl_resolve_ties(ctx, f, nullptr))) return rc;
    if (corners_xy && n_corners)
      return vsl_frames_download_keypoints(ctx, f, slot, f->F, corners_xy, nullptr, nullptr, n_corners);
    return VSL_OK;
  }
  // Two launches and one synchronisation per call (round 3; it was a tie-guard round trip, a pose upload, four kernels,
  // a memset and two copies): the pose travels as a kernel argument, projection + ordered compaction are one kernel,
  // the matching kernel writes its per-keypoint results, the projected count and the frame store's near-tie count
  // straight into pinned host memory, and the match list is compacted on the host (<= F entries).
  const int n_blocks = (n + 1023) / 1024;
  if (n_blocks + 1 > m->chain_cap) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (m->chain) (void)hipFree(m->chain);
    m->chain = nullptr;
    m->chain_cap = 0;
    const int cap = 2 * n_blocks + 64;
    VSL_HIP(ctx, hipMalloc((void**)&m->chain, 8 * (size_t)cap));
    VSL_HIP(ctx, hipMemsetAsync(m->chain, 0, 8 * (size_t)cap, ctx->stream));
    m->chain_cap = cap;
    m->epoch = 0;
  }
  if (3 * f->F + 4 > m->mailbox_cap) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (m->mailbox) (void)hipHostFree(m->mailbox);
    m->mailbox = nullptr;
    m->mailbox_cap = 0;
    VSL_HIP(ctx, hipHostMalloc((void**)&m->mailbox, 4 * (3 * (size_t)f->F + 4), hipHostMallocMapped | hipHostMallocCoherent));
    m->mailbox_cap = 3 * f->F + 4;
  }
  PoseIntr pi;
  for (int i = 0; i < 7; i++) pi.v[i] = pose7[i];
  pi.v[7] = 0;
  for (int i = 0; i < 8; i++) pi.v[8 + i] = intr8[i];
  if (++m->epoch == 0) {  // 2^32 calls later: the chain words of epoch 0 are the freshly cleared ones
    VSL_HIP(ctx, hipMemsetAsync(m->chain, 0, 8 * (size_t)m->chain_cap, ctx->stream));
    m->epoch = 1;
  }
  // chain position = blockIdx while every workgroup is resident from the start, an atomic ticket beyond (kernel header);
  // the ticket word is the last word of the chain allocation (zero between calls)
  unsigned int* ticket = (n_blocks > VO_CHAIN_RESIDENT_BLOCKS || ctx->vo_chain_ticket) ? (unsigned int*)(m->chain + (m->chain_cap - 1)) : nullptr;
  hipLaunchKernelGGL(project_compact_kernel, dim3(n_blocks), dim3(1024), 0, ctx->stream, pi, cam_model, width, height, m->points, n,
                     cam_z_threshold, m->out_uv, m->out_idx, m->counters, m->chain, m->epoch, ticket);
  int32_t* mail = m->mailbox;
  const double max_dist_sq = sqrt_less_threshold(match_max_dist_2d);
  for (int attempt = 0; attempt < 2; attempt++) {
    hipLaunchKernelGGL(find_matches_kernel, dim3((f->F + 3) / 4), dim3(256), 0, ctx->stream, (const double*)nullptr,
                       f->kp_desc + 4 * (size_t)slot * f->F, f->F, m->out_uv, m->out_idx, n, m->obs_start, m->pool,
                       max_dist_sq, feature_match_threshold, feature_match_dist_2_best, mail + 4,
                       f->kp_xy + 2 * (size_t)slot * f->F, m->obs_index, f->kp_count + slot, f->F, m->counters, mail,
                       f->ties_pending ? f->tie_count : (const int32_t*)nullptr, corners_xy ? mail + 4 + f->F : (int32_t*)nullptr);
    VSL_CHECK_LAUNCH(ctx);
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // The descriptors of the slot must be final (rBRIEF near-tie guard, describe.hip).  The guard's count came along
    // with the results: zero (all but ~3 in a million frames) -> done, no second round trip; otherwise resolve the
    // ties the usual way and match once more.
    if (!f->ties_pending) break;
    if (mail[1] == 0) {
      f->ties_settled();
      break;
    }
    if ((rc = vsl_resolve_ties(ctx, f, nullptr))) return rc;
  }
  if (n_projected) *n_projected = mail[0];
  int np = 0;
  const int32_t* res = mail + 4;
  for (int k = 0; k < f->F; k++)
    if (res[k] >= 0) {
      if (!pairs) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_map_track: pairs is null");
      pairs[2 * np] = k;
      pairs[2 * np + 1] = res[k];
      np++;
    }
  *n_pairs = np;
  if (corners_xy && n_corners) {
    const int nk = mail[2] < f->F ? mail[2] : f->F;
    const int32_t* xy = mail + 4 + f->F;
    for (int k = 0; k < 2 * nk; k++) corners_xy[k] = (double)xy[k];
    *n_corners = nk;
  }
  return VSL_OK;
}
