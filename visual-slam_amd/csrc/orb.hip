// orb.hip -- the ORB front end behind compute_bow_vector (include/visnav/keypoints.h:243-254:
// cv::ORB::create(num_features, 1.2, 8, 19, 0, 2, cv::ORB::FAST_SCORE)->detectAndCompute), SURVEY.md 8(f) rank 3.
//
// cv::ORB is [upstream] OpenCV; the arithmetic conventions this file implements are spelled out in
// oracle/orc_orb.cpp (parity with the OpenCV binary is unpinned; parity with that restatement is bit-exact
// and tested).  One image per call -- the reference runs this once per keyframe:
//   pyramid (7 chained bilinear resizes, OpenCV's 8-bit fixed-point weights)  ->  per level: FAST-9/16
//   score image, strict 3x3 non-maximum suppression + border filter + score histogram, retainBest by the
//   histogram cut with an order-preserving compaction (count / scan / emit over 1024-pixel chunks), 7x7 Gaussian blur  ->
//   intensity-centroid orientation (one wavefront per keypoint)  ->  rotated BRIEF tests on the blurred
//   level (one thread per descriptor byte).  cos / sin of the keypoint angles are evaluated by the host's
//   libm between the last two kernels (device cos/sin are not bit-identical to glibc).
#include <cmath>
#include <vector>

#include "vsl_common.h"

namespace {

#define ORB_LEVELS 8
#define ORB_EDGE 19
#define ORB_FAST_THR 20
#define ORB_HALF_PATCH 15

struct OrbPat {
  int xa, ya, xb, yb;
};
__constant__ OrbPat c_orb_pattern[256] = {
#include "rbrief_pattern.inc"
};
__constant__ int c_umax[16];
__constant__ float c_gauss7[7];

__device__ __forceinline__ int d_reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ void orb_resize_kernel(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst, int dw, int dh,
                                  double scale_x, double scale_y) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw) return;
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  if (sx < 0) {
    fx = 0;
    sx = 0;
  }
  if (sx >= sw - 1) {
    fx = 0;
    sx = sw - 1;
  }
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  const int sy = (int)floorf(fy);
  fy -= sy;
  const int a0 = (short)(int)rintf((1.f - fx) * 2048.f), a1 = (short)(int)rintf(fx * 2048.f);
  const int b0 = (short)(int)rintf((1.f - fy) * 2048.f), b1 = (short)(int)rintf(fy * 2048.f);
  const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
  const int sx1 = min(sx + 1, sw - 1);
  const int S0 = src[(size_t)sy0 * sw + sx] * a0 + src[(size_t)sy0 * sw + sx1] * a1;
  const int S1 = src[(size_t)sy1 * sw + sx] * a0 + src[(size_t)sy1 * sw + sx1] * a1;
  const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
  dst[(size_t)dy * dw + dx] = (uint8_t)min(max(v, 0), 255);
}

struct OrbLevels {
  int W[ORB_LEVELS], H[ORB_LEVELS];
  int quota[ORB_LEVELS];
  int seg_base[ORB_LEVELS];  // first keypoint slot of the level's output segment
  int seg_cap[ORB_LEVELS];
  size_t pix_off[ORB_LEVELS];  // offset of the level in the pyramid-shaped buffers
  float scale[ORB_LEVELS];
  int chunk_base[ORB_LEVELS + 1];  // 1024-pixel chunks of the level = [chunk_base[l], chunk_base[l + 1])
};

// FAST-9/16 score of every pixel: the largest threshold at which it is still a corner, 0 if it is not one at
// ORB_FAST_THR.  16 x 16 pixel tiles staged in LDS with a 3-pixel apron.
// All levels in one launch: blockIdx.z = level, the grid is sized for level 0 and the workgroups beyond a
// smaller level's extent leave at once.
__global__ __launch_bounds__(256) void orb_fast_kernel(OrbLevels L, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ score_all) {
  __shared__ uint8_t tile[22][24];
  const int l = blockIdx.z, W = L.W[l], H = L.H[l];
  const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
  if (x0 >= W || y0 >= H) return;
  const uint8_t* img = pyr + L.pix_off[l];
  uint8_t* score = score_all + L.pix_off[l];
  for (int t = threadIdx.x; t < 22 * 22; t += 256) {
    const int ty = t / 22, tx = t - ty * 22;
    const int gx = min(max(x0 + tx - 3, 0), W - 1), gy = min(max(y0 + ty - 3, 0), H - 1);
    tile[ty][tx] = img[(size_t)gy * W + gx];
  }
  __syncthreads();
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= W || y >= H) return;
  int out = 0;
  if (x >= 3 && y >= 3 && x < W - 3 && y < H - 3) {
    const int cx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    const int cy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    const int v = tile[ly + 3][lx + 3];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = (int)tile[ly + 3 + cy[k]][lx + 3 + cx[k]] - v;
    int best = -1;
#pragma unroll
    for (int s = 0; s < 16; s++) {
      int mn = 1 << 20, mx = -(1 << 20);
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int dv = d[(s + k) & 15];
        mn = min(mn, dv);
        mx = max(mx, dv);
      }
      best = max(best, max(mn, -mx));
    }
    out = best > ORB_FAST_THR ? best - 1 : 0;
  }
  score[(size_t)y * W + x] = (uint8_t)out;
}

// strict 3x3 maximum + border filter; flags the survivors and histograms their scores
__global__ void orb_nms_kernel(OrbLevels L, const uint8_t* __restrict__ score_all, uint8_t* __restrict__ flag_all,
                               int* __restrict__ hist_all) {
  const int l = blockIdx.z, W = L.W[l], H = L.H[l];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H) return;
  const uint8_t* score = score_all + L.pix_off[l];
  uint8_t* flag = flag_all + L.pix_off[l];
  int* hist = hist_all + 256 * l;
  uint8_t f = 0;
  if (x >= ORB_EDGE && y >= ORB_EDGE && x < W - ORB_EDGE && y < H - ORB_EDGE) {
    const int s = score[(size_t)y * W + x];
    if (s > 0) {
      bool ok = true;
#pragma unroll
      for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++)
          if (dx || dy) ok = ok && (s > score[(size_t)(y + dy) * W + x + dx]);
      if (ok) {
        f = 1;
        atomicAdd(&hist[s], 1);
      }
    }
  }
  flag[(size_t)y * W + x] = f;
}


// retainBest + order-preserving compaction in three small launches over 1024-pixel chunks of all levels:
//   count (kept keypoints per chunk)  ->  scan (exclusive offsets per level, one workgroup)  ->  emit.
// cut = the score of the quota-th best keypoint of the level: everything at or above it is kept (ties included).
__device__ __forceinline__ int orb_level_of_chunk(const OrbLevels& L, int chunk) {
  int l = 0;
  while (l + 1 < ORB_LEVELS && chunk >= L.chunk_base[l + 1]) l++;
  return l;
}

// cut of every level in one small launch: workgroup = level, thread s = suffix count of the scores >= s
__global__ __launch_bounds__(256) void orb_cut_kernel(OrbLevels L, const int* __restrict__ hist_all, int32_t* __restrict__ cuts) {
  __shared__ int h[256];
  __shared__ int cut_s, total_s;
  const int l = blockIdx.x, s = threadIdx.x, quota = L.quota[l];
  h[s] = s ? hist_all[256 * l + s] : 0;
  if (s == 0) cut_s = 0;
  __syncthreads();
  int acc = 0;
  for (int k = 255; k >= s; k--) acc += h[k];
  if (s == 1) total_s = acc;
  if (s >= 1 && acc >= quota) atomicMax(&cut_s, s);
  __syncthreads();
  if (s == 0) cuts[l] = quota == 0 ? 256 : (total_s <= quota ? 0 : cut_s);
}

template <bool EMIT>
__global__ __launch_bounds__(1024) void orb_compact_kernel(OrbLevels L, const uint8_t* __restrict__ score_all,
                                                           const uint8_t* __restrict__ flag_all, const int32_t* __restrict__ cuts,
                                                           int32_t* __restrict__ chunk_count, const int32_t* __restrict__ chunk_offset,
                                                           int32_t* __restrict__ kp_xy, int32_t* __restrict__ kp_sl) {
  __shared__ int wave_tot[16];
  const int chunk = blockIdx.x;
  const int l = orb_level_of_chunk(L, chunk);
  const int cut_s = cuts[l];
  const int W = L.W[l], n_pix = W * L.H[l];
  const int i = (chunk - L.chunk_base[l]) * 1024 + threadIdx.x;
  const uint8_t* score = score_all + L.pix_off[l];
  const bool ok = i < n_pix && flag_all[L.pix_off[l] + i] && score[i] >= cut_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(ok);
  if (lane == 0) wave_tot[wave] = __popcll(m);
  __syncthreads();
  if (!EMIT) {
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; w++) t += wave_tot[w];
      chunk_count[chunk] = t;
    }
    return;
  }
  if (ok) {
    int off = chunk_offset[chunk];
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    const int p = off + __popcll(m & ((1ull << lane) - 1ull));
    if (p < L.seg_cap[l]) {
      const int y = i / W, x = i - y * W;
      kp_xy[2 * (size_t)(L.seg_base[l] + p)] = x;
      kp_xy[2 * (size_t)(L.seg_base[l] + p) + 1] = y;
      kp_sl[L.seg_base[l] + p] = (int)score[i] | (l << 8);
    }
  }
}

// exclusive scan of the chunk counts within each level (one workgroup, levels one after the other)
__global__ __launch_bounds__(1024) void orb_scan_kernel(OrbLevels L, const int32_t* __restrict__ chunk_count,
                                                        int32_t* __restrict__ chunk_offset, int32_t* __restrict__ level_count) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = 0; l < ORB_LEVELS; l++) {
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int c0 = L.chunk_base[l], c1 = L.chunk_base[l + 1];
    for (int cb = c0; cb < c1; cb += 1024) {
      const int c = cb + threadIdx.x;
      const int v = c < c1 ? chunk_count[c] : 0;
      // inclusive scan inside the wave
      int x = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
      }
      if (lane == 63) wave_tot[wave] = x;
      __syncthreads();
      int off = base_s;
      for (int w = 0; w < wave; w++) off += wave_tot[w];
      if (c < c1) chunk_offset[c] = off + x - v;
      __syncthreads();
      if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; w++) t += wave_tot[w];
        base_s += t;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) level_count[l] = min(base_s, L.seg_cap[l]);
    __syncthreads();
  }
}

__global__ void orb_blur_rows_kernel(OrbLevels L, const uint8_t* __restrict__ pyr, float* __restrict__ tmp_all) {
  const int l = blockIdx.z, W = L.W[l], H = L.H[l];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H) return;
  const uint8_t* src = pyr + L.pix_off[l];
  float* tmp = tmp_all + L.pix_off[l];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 7; i++) s = s + c_gauss7[i] * (float)src[(size_t)y * W + d_reflect101(x + i - 3, W)];
  tmp[(size_t)y * W + x] = s;
}

__global__ void orb_blur_cols_kernel(OrbLevels L, const float* __restrict__ tmp_all, uint8_t* __restrict__ dst_all) {
  const int l = blockIdx.z, W = L.W[l], H = L.H[l];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W || y >= H) return;
  const float* tmp = tmp_all + L.pix_off[l];
  uint8_t* dst = dst_all + L.pix_off[l];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 7; i++) s = s + c_gauss7[i] * tmp[(size_t)d_reflect101(y + i - 3, H) * W + x];
  const int v = (int)rintf(s);
  dst[(size_t)y * W + x] = (uint8_t)min(max(v, 0), 255);
}

__device__ __forceinline__ float d_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI),
              p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.220446049250313e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)2.220446049250313e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// one wavefront per keypoint slot: intensity centroid over the radius-15 disc, exact integer moments
__global__ __launch_bounds__(256) void orb_angle_kernel(OrbLevels L, const uint8_t* __restrict__ pyr, const int32_t* __restrict__ kp_xy,
                                                        const int32_t* __restrict__ kp_sl, const int32_t* __restrict__ level_count,
                                                        int n_slots, float* __restrict__ angle) {
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slot >= n_slots) return;
  int l = 0;
  while (l + 1 < ORB_LEVELS && slot >= L.seg_base[l + 1]) l++;
  if (slot - L.seg_base[l] >= level_count[l]) return;  // wave-uniform
  const int W = L.W[l];
  const uint8_t* center = pyr + L.pix_off[l] + (size_t)kp_xy[2 * (size_t)slot + 1] * W + kp_xy[2 * (size_t)slot];
  int m_01 = 0, m_10 = 0;
  // rows v = -15 .. 15 over the lanes (31 rows), each lane walks its row's columns
  if (lane < 31) {
    const int v = lane - ORB_HALF_PATCH;
    const int d = c_umax[v < 0 ? -v : v];
    int row_sum = 0;
    for (int u = -d; u <= d; ++u) {
      const int val = center[u + v * W];
      row_sum += val;
      m_10 += u * val;
    }
    m_01 = v * row_sum;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m_01 += __shfl_xor(m_01, o);
    m_10 += __shfl_xor(m_10, o);
  }
  if (lane == 0) angle[slot] = d_fast_atan2((float)m_01, (float)m_10);
  (void)kp_sl;
}

// one thread per (keypoint slot, descriptor byte)
__global__ __launch_bounds__(256) void orb_describe_kernel(OrbLevels L, const uint8_t* __restrict__ blurred, const int32_t* __restrict__ kp_xy,
                                                           const int32_t* __restrict__ level_count, int n_slots,
                                                           const float* __restrict__ cs, uint8_t* __restrict__ desc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int slot = t >> 5, j = t & 31;
  if (slot >= n_slots) return;
  int l = 0;
  while (l + 1 < ORB_LEVELS && slot >= L.seg_base[l + 1]) l++;
  if (slot - L.seg_base[l] >= level_count[l]) return;
  const int W = L.W[l];
  const uint8_t* bc = blurred + L.pix_off[l] + (size_t)kp_xy[2 * (size_t)slot + 1] * W + kp_xy[2 * (size_t)slot];
  const float a = cs[2 * (size_t)slot], b = cs[2 * (size_t)slot + 1];
  int byte = 0;
#pragma unroll
  for (int bit = 0; bit < 8; bit++) {
    const OrbPat p = c_orb_pattern[8 * j + bit];
    const float xa = (float)p.xa * a - (float)p.ya * b, ya = (float)p.xa * b + (float)p.ya * a;
    const float xb = (float)p.xb * a - (float)p.yb * b, yb = (float)p.xb * b + (float)p.yb * a;
    const int t0 = bc[(int)rintf(ya) * W + (int)rintf(xa)], t1 = bc[(int)rintf(yb) * W + (int)rintf(xb)];
    byte |= (t0 < t1) << bit;
  }
  desc[32 * (size_t)slot + j] = (uint8_t)byte;
}

bool g_orb_tables_ready[16] = {false};

int orb_upload_tables(vsl_ctx* ctx) {
  if (ctx->device >= 0 && ctx->device < 16 && g_orb_tables_ready[ctx->device]) return VSL_OK;
  int umax[16] = {0};
  const int hp = ORB_HALF_PATCH;
  const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
  for (int v = 0; v <= vmax; v++) umax[v] = (int)std::lrint(std::sqrt((double)hp * hp - v * v));
  for (int v = hp, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
  float k[7];
  double kd[7], sum = 0;
  for (int i = 0; i < 7; i++) {
    const double x = i - 3;
    kd[i] = std::exp(-x * x / (2.0 * 2.0 * 2.0));
    sum += kd[i];
  }
  for (int i = 0; i < 7; i++) k[i] = (float)(kd[i] / sum);
  VSL_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_umax), umax, sizeof(umax)));
  VSL_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_gauss7), k, sizeof(k)));
  if (ctx->device >= 0 && ctx->device < 16) g_orb_tables_ready[ctx->device] = true;
  return VSL_OK;
}

}  // namespace

// kp5: (x, y in level-0 pixels, angle in degrees, response, octave) per keypoint; desc32: 32 bytes each.
extern "C" int vsl_orb_detect_describe(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch, int nfeatures, int cap,
                                       float* kp5, uint8_t* desc32, int* n_out) {
  if (!ctx || !img || !n_out || w < 64 || h < 64 || pitch < (size_t)w || nfeatures < 1 || cap < 0 || (cap > 0 && (!kp5 || !desc32)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_orb_detect_describe: bad arguments (w, h >= 64 required)");
  *n_out = 0;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = orb_upload_tables(ctx);
  if (rc) return rc;
  OrbLevels L;
  size_t total_pix = 0;
  int n_slots = 0;
  {
    const float factor = (float)(1.0 / 1.2f);
    float ndesired = (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)ORB_LEVELS)));
    int sum = 0;
    for (int l = 0; l < ORB_LEVELS; l++) {
      const float s = (float)std::pow((double)1.2f, (double)l);
      L.scale[l] = s;
      L.W[l] = (int)std::lrintf((float)w / s);
      L.H[l] = (int)std::lrintf((float)h / s);
      if (l < ORB_LEVELS - 1) {
        L.quota[l] = (int)std::lrintf(ndesired);
        sum += L.quota[l];
        ndesired *= factor;
      } else {
        L.quota[l] = nfeatures - sum > 0 ? nfeatures - sum : 0;
      }
      L.pix_off[l] = total_pix;
      total_pix += (size_t)L.W[l] * L.H[l];
      total_pix = (total_pix + 255) & ~(size_t)255;
      L.seg_base[l] = n_slots;
      L.seg_cap[l] = 2 * L.quota[l] + 64;  // retainBest keeps every keypoint tied with the last one
      n_slots += L.seg_cap[l];
    }
    L.chunk_base[0] = 0;
    for (int l = 0; l < ORB_LEVELS; l++) L.chunk_base[l + 1] = L.chunk_base[l] + (L.W[l] * L.H[l] + 1023) / 1024;
  }
  const int n_chunks = L.chunk_base[ORB_LEVELS];
  // scratch: pyramid | score | flag | blurred (u8, total_pix each) | tmp (f32) | hist | level_count | kp_xy | kp_sl | angle | cs | desc
  void* d = nullptr;
  const size_t bytes = 4 * total_pix + 4 * total_pix + 4 * (256 * ORB_LEVELS + 16) + (size_t)n_slots * (8 + 4 + 4 + 8 + 32) +
                       8 * (size_t)n_chunks + 1024;
  rc = vsl_ctx_dscratch(ctx, bytes, &d);
  if (rc) return rc;
  uint8_t* pyr = (uint8_t*)d;
  uint8_t* score = pyr + total_pix;
  uint8_t* flag = score + total_pix;
  uint8_t* blurred = flag + total_pix;
  float* tmp = (float*)(blurred + total_pix);
  int* hist = (int*)(tmp + total_pix);
  int32_t* level_count = hist + 256 * ORB_LEVELS;
  int32_t* cuts = level_count + ORB_LEVELS;
  int32_t* kp_xy = level_count + 16;
  int32_t* kp_sl = kp_xy + 2 * (size_t)n_slots;
  float* angle = (float*)(kp_sl + n_slots);
  float* cs = angle + n_slots;
  int32_t* chunk_count = (int32_t*)(cs + 2 * (size_t)n_slots);
  int32_t* chunk_offset = chunk_count + n_chunks;
  uint8_t* ddesc = (uint8_t*)(chunk_offset + n_chunks);
  hipStream_t st = ctx->stream;
  VSL_HIP(ctx, hipMemcpy2DAsync(pyr, w, img, pitch, w, h, hipMemcpyHostToDevice, st));
  VSL_HIP(ctx, hipMemsetAsync(hist, 0, sizeof(int) * (256 * ORB_LEVELS + 16), st));
  for (int l = 1; l < ORB_LEVELS; l++)
    hipLaunchKernelGGL(orb_resize_kernel, dim3((L.W[l] + 255) / 256, L.H[l]), dim3(256), 0, st, pyr + L.pix_off[l - 1], L.W[l - 1],
                       L.H[l - 1], pyr + L.pix_off[l], L.W[l], L.H[l], (double)L.W[l - 1] / L.W[l], (double)L.H[l - 1] / L.H[l]);
  {
    const int W = L.W[0], H = L.H[0];  // level 0 is the largest
    hipLaunchKernelGGL(orb_fast_kernel, dim3((W + 15) / 16, (H + 15) / 16, ORB_LEVELS), dim3(256), 0, st, L, (const uint8_t*)pyr, score);
    hipLaunchKernelGGL(orb_nms_kernel, dim3((W + 255) / 256, H, ORB_LEVELS), dim3(256), 0, st, L, (const uint8_t*)score, flag, hist);
    hipLaunchKernelGGL(orb_blur_rows_kernel, dim3((W + 255) / 256, H, ORB_LEVELS), dim3(256), 0, st, L, (const uint8_t*)pyr, tmp);
    hipLaunchKernelGGL(orb_blur_cols_kernel, dim3((W + 255) / 256, H, ORB_LEVELS), dim3(256), 0, st, L, (const float*)tmp, blurred);
  }
  hipLaunchKernelGGL(orb_cut_kernel, dim3(ORB_LEVELS), dim3(256), 0, st, L, (const int*)hist, cuts);
  hipLaunchKernelGGL(orb_compact_kernel<false>, dim3(n_chunks), dim3(1024), 0, st, L, score, flag, (const int32_t*)cuts, chunk_count,
                     (const int32_t*)chunk_offset, kp_xy, kp_sl);
  hipLaunchKernelGGL(orb_scan_kernel, dim3(1), dim3(1024), 0, st, L, (const int32_t*)chunk_count, chunk_offset, level_count);
  hipLaunchKernelGGL(orb_compact_kernel<true>, dim3(n_chunks), dim3(1024), 0, st, L, score, flag, (const int32_t*)cuts, chunk_count,
                     (const int32_t*)chunk_offset, kp_xy, kp_sl);
  hipLaunchKernelGGL(orb_angle_kernel, dim3((n_slots + 3) / 4), dim3(256), 0, st, L, pyr, kp_xy, kp_sl, level_count, n_slots, angle);
  VSL_CHECK_LAUNCH(ctx);
  // host: cos / sin of every angle with libm (fp32 radians -> double cos -> fp32, like the oracle)
  std::vector<float> h_angle(n_slots), h_cs(2 * (size_t)n_slots, 0.f);
  int32_t h_count[ORB_LEVELS];
  VSL_HIP(ctx, hipMemcpyAsync(h_angle.data(), angle, sizeof(float) * n_slots, hipMemcpyDeviceToHost, st));
  VSL_HIP(ctx, hipMemcpyAsync(h_count, level_count, sizeof(h_count), hipMemcpyDeviceToHost, st));
  VSL_HIP(ctx, hipStreamSynchronize(st));
  int total = 0;
  for (int l = 0; l < ORB_LEVELS; l++) {
    for (int i = 0; i < h_count[l]; i++) {
      const int slot = L.seg_base[l] + i;
      const float rad = h_angle[slot] * (float)(M_PI / 180.0);
      h_cs[2 * (size_t)slot] = (float)std::cos((double)rad);
      h_cs[2 * (size_t)slot + 1] = (float)std::sin((double)rad);
    }
    total += h_count[l];
  }
  VSL_HIP(ctx, hipMemcpyAsync(cs, h_cs.data(), sizeof(float) * 2 * n_slots, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(orb_describe_kernel, dim3((n_slots * 32 + 255) / 256), dim3(256), 0, st, L, blurred, kp_xy, level_count, n_slots, cs,
                     ddesc);
  VSL_CHECK_LAUNCH(ctx);
  std::vector<int32_t> h_xy(2 * (size_t)n_slots), h_sl(n_slots);
  std::vector<uint8_t> h_desc(32 * (size_t)n_slots);
  VSL_HIP(ctx, hipMemcpyAsync(h_xy.data(), kp_xy, sizeof(int32_t) * 2 * n_slots, hipMemcpyDeviceToHost, st));
  VSL_HIP(ctx, hipMemcpyAsync(h_sl.data(), kp_sl, sizeof(int32_t) * n_slots, hipMemcpyDeviceToHost, st));
  VSL_HIP(ctx, hipMemcpyAsync(h_desc.data(), ddesc, 32 * (size_t)n_slots, hipMemcpyDeviceToHost, st));
  VSL_HIP(ctx, hipStreamSynchronize(st));
  int n = 0;
  for (int l = 0; l < ORB_LEVELS && n < cap; l++)
    for (int i = 0; i < h_count[l] && n < cap; i++) {
      const int slot = L.seg_base[l] + i;
      float* k = kp5 + 5 * (size_t)n;
      k[0] = (float)h_xy[2 * (size_t)slot] * L.scale[l];
      k[1] = (float)h_xy[2 * (size_t)slot + 1] * L.scale[l];
      k[2] = h_angle[slot];
      k[3] = (float)(h_sl[slot] & 255);
      k[4] = (float)l;
      std::memcpy(desc32 + 32 * (size_t)n, h_desc.data() + 32 * (size_t)slot, 32);
      n++;
    }
  *n_out = n;
  if (n < total) return vsl_fail(ctx, VSL_ERR_CAPACITY, "vsl_orb_detect_describe: %d keypoints, capacity %d", total, cap);
  return VSL_OK;
}

// compute_bow_vector (include/visnav/keypoints.h:243-254): ORB front end + vocabulary transform.
extern "C" int vsl_compute_bow_vector(vsl_ctx* ctx, const vsl_voc* voc, const uint8_t* img, int w, int h, size_t pitch,
                                      int num_features, int levelsup, int cap, uint32_t* word_ids, double* word_vals, int* nnz,
                                      uint32_t* fv_node, uint32_t* fv_feat, int* fv_n) {
  if (!ctx || !voc || !nnz || !fv_n || cap < 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_compute_bow_vector: bad arguments");
  const int kcap = 2 * num_features + 64 * ORB_LEVELS;
  std::vector<float> kp(5 * (size_t)kcap);
  std::vector<uint8_t> desc(32 * (size_t)kcap);
  int n = 0;
  int rc = vsl_orb_detect_describe(ctx, img, w, h, pitch, num_features, kcap, kp.data(), desc.data(), &n);
  if (rc) return rc;
  if (n > cap) return vsl_fail(ctx, VSL_ERR_CAPACITY, "vsl_compute_bow_vector: %d features, output capacity %d", n, cap);
  return vsl_bow_transform(ctx, voc, desc.data(), n, levelsup, word_ids, word_vals, nnz, fv_node, fv_feat, fv_n);
}
