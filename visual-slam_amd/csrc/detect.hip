// detect.hip -- K1 (min-eigenvalue response), K2a (candidate extraction), K2b (ordered greedy
// min-distance selection).  Together they replace the cv::goodFeaturesToTrack call and the border
// filter of visnav::detectKeypoints (include/visnav/keypoints.h:133-150).
//
// Arithmetic contract (identical, operation for operation, to oracle/orc_keypoints.cpp; the whole
// library is compiled with -ffp-contract=off so no multiply-add is fused):
//   s = (float)(1/3060); s2 = 2s
//   row filter at row yr:  rx = I(x+1) - I(x-1);  ry = ((s*I(x-1)) + (s2*I(x))) + (s*I(x+1))
//   dx = (rx(y-1) + rx(y+1))*s + rx(y)*s2;   dy = ry(y+1) - ry(y-1)              (fp32)
//   cov = (dx*dx, dx*dy, dy*dy)                                                   (fp32)
//   R(x,y) = ((double)c(x-1,y) + c(x,y)) + c(x+1,y);  A = (R(x,y-1) + R(x,y)) + R(x,y+1)  (fp64)
//   a = (float)A_xx*0.5f, b = (float)A_xy, c = (float)A_yy*0.5f
//   response = (a + c) - sqrtf((a-c)*(a-c) + b*b)                                 (fp32)
// BORDER_REFLECT_101 is applied to the image for the derivative and to the cov image for the box
// sum (a lane or row outside the image evaluates the cov of its mirror position).
#include "vsl_common.h"

#define K1_ROWS 16
#define K1_COLS 62  // output columns per wave: 64 lanes minus one halo lane on each side

__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

struct RowF {
  float rx, ry;
};

__device__ __forceinline__ RowF rowfilt(const uint8_t* __restrict__ img, int w, int yr, int xm, int xe, int xp,
                                        float s, float s2) {
  const uint8_t* row = img + (size_t)yr * w;
  const float l = (float)row[xm], m = (float)row[xe], r = (float)row[xp];
  RowF o;
  o.rx = r - l;
  float t = s * l;
  t = t + s2 * m;
  t = t + s * r;
  o.ry = t;
  return o;
}

__global__ void detect_init_kernel(int32_t* meta, int first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    meta[(size_t)(first + i) * VSL_META_STRIDE + VSL_META_MAX] = INT32_MIN;
    meta[(size_t)(first + i) * VSL_META_STRIDE + VSL_META_NCAND] = 0;
  }
}

// K1.  grid = (ceil(w/62), ceil(h/64), n_images), block = 256: wave v of a block owns the 16-row
// strip (blockIdx.y*4 + v) of the 62-column strip blockIdx.x.  One image column per lane; rows are
// walked top to bottom with the row-filter results and the fp64 row sums held in registers, so an
// image byte is loaded ~1.3 times and nothing is staged through LDS; the +-1 column neighbours of
// the cov values come from cross-lane shuffles.
__global__ __launch_bounds__(256) void min_eig_response_kernel(const uint8_t* __restrict__ images,
                                                               float* __restrict__ response,
                                                               int32_t* __restrict__ meta, int w, int h,
                                                               int first) {
  const int slot = first + blockIdx.z;
  const uint8_t* __restrict__ img = images + (size_t)slot * w * h;
  float* __restrict__ resp = response + (size_t)slot * w * h;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int x0 = blockIdx.x * K1_COLS;
  const int y0 = (blockIdx.y * 4 + wave) * K1_ROWS;
  if (y0 >= h) return;
  const float s = (float)(1.0 / (4.0 * 3.0 * 255.0));
  const float s2 = 2.0f * s;
  const int x = x0 - 1 + lane;
  int xe = reflect101(min(x, w), w);  // lanes right of the halo are parked on a valid column
  const int xm = reflect101(xe - 1, w), xp = reflect101(xe + 1, w);

  double Rxx[3], Rxy[3], Ryy[3];  // row sums of rows q-2, q-1, q
  RowF f0, f1, f2;                // row filters of rows ye-1, ye, ye+1
  int prev_ye = -100;
  float vmax = -INFINITY;
  const int y_end = min(h, y0 + K1_ROWS);
  for (int q = y0 - 1; q <= y_end; q++) {
    const int ye = reflect101(q, h);
    if (ye == prev_ye + 1 && ye + 1 < h) {
      f0 = f1;
      f1 = f2;
      f2 = rowfilt(img, w, ye + 1, xm, xe, xp, s, s2);
    } else {
      f0 = rowfilt(img, w, reflect101(ye - 1, h), xm, xe, xp, s, s2);
      f1 = rowfilt(img, w, ye, xm, xe, xp, s, s2);
      f2 = rowfilt(img, w, reflect101(ye + 1, h), xm, xe, xp, s, s2);
    }
    prev_ye = ye;
    const float dx = (f0.rx + f2.rx) * s + f1.rx * s2;
    const float dy = f2.ry - f0.ry;
    const float cxx = dx * dx, cxy = dx * dy, cyy = dy * dy;
    const float lxx = __shfl_up(cxx, 1), lxy = __shfl_up(cxy, 1), lyy = __shfl_up(cyy, 1);
    const float rxx = __shfl_down(cxx, 1), rxy = __shfl_down(cxy, 1), ryy = __shfl_down(cyy, 1);
    Rxx[0] = Rxx[1]; Rxx[1] = Rxx[2];
    Rxy[0] = Rxy[1]; Rxy[1] = Rxy[2];
    Ryy[0] = Ryy[1]; Ryy[1] = Ryy[2];
    Rxx[2] = ((double)lxx + (double)cxx) + (double)rxx;
    Rxy[2] = ((double)lxy + (double)cxy) + (double)rxy;
    Ryy[2] = ((double)lyy + (double)cyy) + (double)ryy;
    if (q >= y0 + 1) {
      const int y = q - 1;
      const double Axx = (Rxx[0] + Rxx[1]) + Rxx[2];
      const double Axy = (Rxy[0] + Rxy[1]) + Rxy[2];
      const double Ayy = (Ryy[0] + Ryy[1]) + Ryy[2];
      const float a = (float)Axx * 0.5f, b = (float)Axy, c = (float)Ayy * 0.5f;
      const float d = a - c;
      float t = d * d;
      const float bb = b * b;
      t = t + bb;
      // sqrtf is correctly rounded under hipcc's default flags; __fsqrt_rn is the approximate native sqrt
      const float val = (a + c) - sqrtf(t);
      if (lane >= 1 && lane <= K1_COLS && x < w) {
        resp[(size_t)y * w + x] = val;
        vmax = fmaxf(vmax, val);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
  if (lane == 0) atomicMax(&meta[(size_t)slot * VSL_META_STRIDE + VSL_META_MAX], vsl_float_to_ordered(vmax));
}

// K2a.  Candidates = strict interior pixels whose thresholded response is non-zero and equals the
// 3x3 maximum of the thresholded response (goodFeaturesToTrack: threshold(TOZERO) + dilate + compare).
// Key = (order-preserving fp32 bits << 32) | pixel index: a descending sort on the key is the
// reference's order (value descending, equal values by address descending).
// A workgroup owns a 64 x 32 tile (a lane walks 8 rows with a sliding 3x3 window: 3 loads per
// pixel), gathers its candidates in LDS and reserves its output range with ONE global atomic; the
// per-image counters sit on separate 128-byte lines (VSL_META_STRIDE), because thousands of
// returning atomics on neighbouring words serialise in one L2 channel (measured: 3.1 ms per 128
// images with one atomic per wave on adjacent counters).
#define K2A_ROWS 8
__global__ __launch_bounds__(256) void candidates_kernel(const float* __restrict__ response,
                                                         int32_t* __restrict__ meta, uint64_t* __restrict__ cand,
                                                         int w, int h, size_t cand_cap, int first, double quality) {
  const int slot = first + blockIdx.z;
  const float* __restrict__ resp = response + (size_t)slot * w * h;
  const float maxv = vsl_ordered_to_float(meta[(size_t)slot * VSL_META_STRIDE + VSL_META_MAX]);
  const float thr = (float)((double)maxv * quality);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane;
  const int y0 = blockIdx.y * (4 * K2A_ROWS) + wave * K2A_ROWS;
  __shared__ uint64_t list[64 * 4 * K2A_ROWS];
  __shared__ int n_list, g_base;
  if (threadIdx.x == 0) n_list = 0;
  __syncthreads();
  const bool col_ok = x >= 1 && x < w - 1;
  if (col_ok && y0 < h - 1) {
    // thresholded 3-wide rows of the window: t[r][c]
    float t[3][3];
    auto load_row = [&](int yy, float* o) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float v = resp[(size_t)yy * w + (x - 1 + c)];
        o[c] = v;
      }
    };
    const int ys = max(y0, 1);
    load_row(ys - 1, t[0]);
    load_row(ys, t[1]);
    const int ye = min(h - 1, y0 + K2A_ROWS);
    for (int y = ys; y < ye; y++) {
      load_row(y + 1, t[2]);
      const float val = t[1][1];
      if (val > thr && val != 0.f) {
        bool is_c = true;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const float nv = t[r][c] > thr ? t[r][c] : 0.f;
            is_c = is_c && !(nv > val);
          }
        if (is_c) {
          const uint32_t ob = (uint32_t)vsl_float_to_ordered(val) ^ 0x80000000u;
          const int p = atomicAdd(&n_list, 1);
          list[p] = ((uint64_t)ob << 32) | (uint32_t)(y * w + x);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        t[0][c] = t[1][c];
        t[1][c] = t[2][c];
      }
    }
  }
  __syncthreads();
  const int n = n_list;
  if (n == 0) return;
  if (threadIdx.x == 0) g_base = atomicAdd(&meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND], n);
  __syncthreads();
  const int base = g_base;
  for (int i = threadIdx.x; i < n; i += 256)
    if ((size_t)(base + i) < cand_cap) cand[(size_t)slot * cand_cap + base + i] = list[i];
}

// ------------------------------------------------------------------------------------------ K2b
// One 1024-thread workgroup per image.  The candidate keys are sorted (descending) in LDS in chunks
// of at most SEL_CHUNK keys (a most-significant-digit radix select finds the chunk boundary when an
// image has more candidates than that), and consumed in rank order, 1024 at a time, by an exact
// parallel restatement of the reference's sequential greedy loop:
//   a candidate is dropped if an ALREADY ACCEPTED corner lies in the 3x3 neighbouring 8-px cells at
//   squared distance < 64; among the survivors of one batch, a candidate waits for every
//   higher-ranked survivor within that distance to be decided, is rejected if one of them was
//   accepted, and is accepted otherwise (the lowest undecided rank always decides, so the rounds
//   terminate and the result is the sequential one).  Acceptance stops at num_features.
#define SEL_THREADS 1024
#define SEL_CHUNK 8192
#define SEL_EMPTY 0xFFFFFFFFu

struct SelShared {
  int wave_tot[16];
  int hist[256];
  unsigned long long sel_prefix;
  int sel_k;
  int n_chunk;
};

// Exclusive prefix count of `p` over the workgroup (thread order) and the total.
__device__ __forceinline__ int block_scan(bool p, int* wave_tot, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(p);
  const int within = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) wave_tot[wave] = __popcll(m);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const int v = wave_tot[w];
    off += (w < wave) ? v : 0;
    tot += v;
  }
  total = tot;
  return off + within;
}

// Static LDS: sized for up to SEL_MAX_CELLS 8x8-px cells (752x480 needs 5640); gfx950 lets one
// workgroup own (almost) all 160 KiB of its CU's LDS.
#define SEL_MAX_CELLS 6144
#define SEL_LDS_BYTES (SEL_CHUNK * 8 + SEL_MAX_CELLS * 2 * 4 + SEL_THREADS * 4 * 3 + SEL_MAX_CELLS * 4 + 2048)
static_assert(SEL_LDS_BYTES <= 160 * 1024, "selection kernel LDS budget");

__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const uint64_t* __restrict__ cand_all,
                                                             const int32_t* __restrict__ meta,
                                                             int32_t* __restrict__ kp_xy, int32_t* __restrict__ kp_count,
                                                             int w, int h, size_t cand_cap, int F, int first,
                                                             int num_features, int border) {
  __shared__ __align__(16) unsigned char smem[SEL_LDS_BYTES];
  const int slot = first + blockIdx.x;
  const uint64_t* __restrict__ cand = cand_all + (size_t)slot * cand_cap;
  const int n_cand = min(meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND], (int)cand_cap);
  const int gw = (w + 7) / 8, gh = (h + 7) / 8, cells = gw * gh;
  // LDS carve-up (all regions 8-byte aligned)
  uint64_t* keys = (uint64_t*)smem;                 // SEL_CHUNK sorted keys
  SelShared* sh = (SelShared*)(keys + SEL_CHUNK);
  uint32_t* acc = (uint32_t*)(sh + 1);              // cells * 2 accepted corners, packed x | y << 16
  int* head = (int*)(acc + 2 * (size_t)cells);      // cells: batch list heads (-1 = empty)
  int* next = head + cells;                         // SEL_THREADS
  uint32_t* cxy = (uint32_t*)(next + SEL_THREADS);  // SEL_THREADS: batch candidate position
  int* state = (int*)(cxy + SEL_THREADS);           // SEL_THREADS: 0 undecided, 1 accepted, 2 rejected
  const int tid = threadIdx.x;

  for (int i = tid; i < 2 * cells; i += SEL_THREADS) acc[i] = SEL_EMPTY;
  for (int i = tid; i < cells; i += SEL_THREADS) head[i] = -1;
  __syncthreads();

  int n_acc = 0, n_out = 0;
  unsigned long long hi = ~0ull;  // keys >= hi have been consumed
  int remaining = n_cand;
  int32_t* out = kp_xy + (size_t)slot * F * 2;

  while (remaining > 0 && n_acc < num_features) {
    // ---- choose the chunk [lo, hi): all remaining keys, or the SEL_CHUNK largest of them
    unsigned long long lo = 0;
    if (remaining > SEL_CHUNK) {
      unsigned long long prefix = 0;
      int k = SEL_CHUNK;  // the k-th largest key below hi, most significant byte first
      for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += SEL_THREADS) sh->hist[i] = 0;
        __syncthreads();
        const unsigned long long pmask = shift == 56 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < n_cand; i += SEL_THREADS) {
          const unsigned long long key = cand[i];
          if (key < hi && (key & pmask) == prefix) atomicAdd(&sh->hist[(int)((key >> shift) & 255)], 1);
        }
        __syncthreads();
        if (tid == 0) {
          int kk = k, d = 255;
          for (; d > 0; d--) {
            if (sh->hist[d] >= kk) break;
            kk -= sh->hist[d];
          }
          sh->sel_k = kk;
          sh->sel_prefix = prefix | ((unsigned long long)d << shift);
        }
        __syncthreads();
        k = sh->sel_k;
        prefix = sh->sel_prefix;
        __syncthreads();
      }
      lo = prefix;  // = the SEL_CHUNK-th largest remaining key (keys are unique)
    }
    // ---- gather the chunk into LDS and pad to a power of two
    if (tid == 0) sh->n_chunk = 0;
    __syncthreads();
    for (int i = tid; i < n_cand; i += SEL_THREADS) {
      const unsigned long long key = cand[i];
      if (key >= lo && key < hi) {
        const int p = atomicAdd(&sh->n_chunk, 1);
        if (p < SEL_CHUNK) keys[p] = key;
      }
    }
    __syncthreads();
    const int n_chunk = min(sh->n_chunk, SEL_CHUNK);
    int N = 1024;
    while (N < n_chunk) N <<= 1;
    for (int i = n_chunk + tid; i < N; i += SEL_THREADS) keys[i] = 0ull;
    __syncthreads();
    // ---- bitonic sort, descending
    for (int k = 2; k <= N; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (N >> 1); t += SEL_THREADS) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const uint64_t a = keys[i], b = keys[l];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) {
            keys[i] = b;
            keys[l] = a;
          }
        }
        __syncthreads();
      }
    }
    // ---- greedy, SEL_THREADS ranks per batch
    for (int r0 = 0; r0 < n_chunk && n_acc < num_features; r0 += SEL_THREADS) {
      const int r = r0 + tid;
      int px = 0, py = 0, cell = 0, cx = 0, cy = 0;
      bool alive = false;
      if (r < n_chunk) {
        const uint32_t pix = (uint32_t)(keys[r] & 0xFFFFFFFFull);
        py = (int)(pix / (uint32_t)w);
        px = (int)(pix - (uint32_t)py * (uint32_t)w);
        cx = px >> 3;
        cy = py >> 3;
        cell = cy * gw + cx;
        alive = true;
        for (int yy = max(0, cy - 1); yy <= min(gh - 1, cy + 1); yy++)
          for (int xx = max(0, cx - 1); xx <= min(gw - 1, cx + 1); xx++) {
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++) {
              const uint32_t a = acc[2 * (yy * gw + xx) + sidx];
              if (a != SEL_EMPTY) {
                const int dx = px - (int)(a & 0xFFFF), dy = py - (int)(a >> 16);
                if (dx * dx + dy * dy < 64) alive = false;
              }
            }
          }
      }
      cxy[tid] = (uint32_t)px | ((uint32_t)py << 16);
      state[tid] = alive ? 0 : 2;
      if (alive) next[tid] = atomicExch(&head[cell], tid);
      __syncthreads();
      bool undecided = alive;
      while (true) {
        int ns = 0;
        if (undecided) {
          bool rej = false, blocked = false;
          for (int yy = max(0, cy - 1); yy <= min(gh - 1, cy + 1); yy++)
            for (int xx = max(0, cx - 1); xx <= min(gw - 1, cx + 1); xx++)
              for (int u = head[yy * gw + xx]; u >= 0; u = next[u]) {
                if (u < tid) {
                  const uint32_t q = cxy[u];
                  const int dx = px - (int)(q & 0xFFFF), dy = py - (int)(q >> 16);
                  if (dx * dx + dy * dy < 64) {
                    const int su = ((volatile int*)state)[u];
                    rej = rej || (su == 1);
                    blocked = blocked || (su == 0);
                  }
                }
              }
          ns = rej ? 2 : (blocked ? 0 : 1);
        }
        if (ns) {
          ((volatile int*)state)[tid] = ns;
          undecided = false;
        }
        if (__syncthreads_count(undecided) == 0) break;
      }
      const bool accepted = alive && state[tid] == 1;
      int total = 0;
      const int rank = n_acc + block_scan(accepted, sh->wave_tot, total);
      const bool keep = accepted && rank < num_features;
      const bool inb = keep && px >= border && px < w - border && py >= border && py < h - border;
      int total_out = 0;
      const int pos = n_out + block_scan(inb, sh->wave_tot, total_out);
      if (inb) {
        out[2 * pos] = px;
        out[2 * pos + 1] = py;
      }
      if (keep) {
        const uint32_t packed = (uint32_t)px | ((uint32_t)py << 16);
        if (atomicCAS(&acc[2 * cell], SEL_EMPTY, packed) != SEL_EMPTY) atomicCAS(&acc[2 * cell + 1], SEL_EMPTY, packed);
      }
      if (alive) head[cell] = -1;
      n_acc = min(num_features, n_acc + total);
      n_out += total_out;
      __syncthreads();
    }
    hi = lo;
    remaining -= n_chunk;
  }
  if (tid == 0) kp_count[slot] = n_out;
}

int vsl_launch_detect(vsl_ctx* ctx, vsl_frames* f, int first, int n, int num_features) {
  if (n <= 0) return VSL_OK;
  if (num_features < 1 || num_features > f->F)
    return vsl_fail(ctx, VSL_ERR_INVALID, "num_features %d not in [1, %d]", num_features, f->F);
  static_assert(sizeof(SelShared) <= 2048, "SelShared fits its LDS slot");
  if (((f->w + 7) / 8) * ((f->h + 7) / 8) > SEL_MAX_CELLS)
    return vsl_fail(ctx, VSL_ERR_CAPACITY, "image %dx%d has more than %d 8x8 cells (selection kernel LDS limit)", f->w, f->h, SEL_MAX_CELLS);
  const int w = f->w, h = f->h;
  {
    VslStage st(ctx, VSL_STAGE_RESPONSE);
    hipLaunchKernelGGL(detect_init_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, f->meta, first, n);
    hipLaunchKernelGGL(min_eig_response_kernel, dim3((w + K1_COLS - 1) / K1_COLS, (h + 4 * K1_ROWS - 1) / (4 * K1_ROWS), n),
                       dim3(256), 0, ctx->stream, f->images, f->response, f->meta, w, h, first);
    hipLaunchKernelGGL(candidates_kernel, dim3((w + 63) / 64, (h + 4 * K2A_ROWS - 1) / (4 * K2A_ROWS), n), dim3(256), 0,
                       ctx->stream, f->response, f->meta, f->cand, w, h, f->cand_cap, first, 0.01);
    VSL_CHECK_LAUNCH(ctx);
  }
  {
    VslStage st(ctx, VSL_STAGE_SELECT);
    hipLaunchKernelGGL(select_kernel, dim3(n), dim3(SEL_THREADS), 0, ctx->stream, f->cand, f->meta, f->kp_xy,
                       f->kp_count, w, h, f->cand_cap, f->F, first, num_features, 19);
    VSL_CHECK_LAUNCH(ctx);
  }
  return VSL_OK;
}
