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
// (the kernel evaluates the last two lines as  t = fma(0.25, fl((X-Y)^2), fl(b*b)),  fma(0.5, fl(X+Y), -sqrt(t))  with
// X = (float)A_xx, Y = (float)A_yy: halving is exact, so these two explicit FMAs round exactly the sums above)
// BORDER_REFLECT_101 is applied to the image for the derivative and to the cov image for the box
// sum (a lane or row outside the image evaluates the cov of its mirror position).
#include <type_traits>

#include "vsl_common.h"

// rows per wave strip; (K1_ROWS + 4) % 6 is 3 or 4 (that many opening steps, then the row loop unrolled by six).  Every
// strip recomputes 4 halo rows, so taller strips waste less; 60 rows: 480 rows = 8 strips = 2 workgroups of 4 waves with
// no idle wave, and 1024 images are exactly 13 rounds of 8 workgroups per compute unit (measured per 1024 images,
// older kernel: 29 rows 0.771 ms, 32: 0.740, 41: 0.712, 44: 0.716; this kernel: 41 rows 0.654, 60 rows 0.638)
#ifndef K1_ROWS
#define K1_ROWS 60
#endif
#ifndef K1_ROWS_SMALL
#define K1_ROWS_SMALL 29  // strips of launches of fewer than 8 images (17 strips of a 480-row image instead of 8)
#endif
// goodFeaturesToTrack's qualityLevel as the reference passes it (keypoints.h:138): threshold = max response * 0.01
#define VSL_QUALITY_LEVEL 0.01
#define K1_WLIST 384  // LDS candidate slots per wave strip (60 x 60 pixels); overflow goes straight to global memory
#define K1_COLS 60  // owned columns per wave: 64 lanes minus two halo lanes on each side

__device__ __forceinline__ int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

struct RowF {
  float rx, ry;
};

struct RowRaw {
  unsigned l, m, r;
};

// yr is wave-uniform: buffer loads through the image's descriptor take the row base as the scalar offset and the
// lane's column as the 32-bit vector offset -- no per-load 64-bit address arithmetic on the VALU (K1 is VALU-bound)
typedef __amdgpu_buffer_rsrc_t ImgSrd;
__device__ __forceinline__ RowRaw rowload(ImgSrd img, int w, int yr, unsigned xm, unsigned xe, unsigned xp) {
  const int row = yr * w;
  RowRaw o;
  o.l = __builtin_amdgcn_raw_buffer_load_b8(img, (int)xm, row, 0);
  o.m = __builtin_amdgcn_raw_buffer_load_b8(img, (int)xe, row, 0);
  o.r = __builtin_amdgcn_raw_buffer_load_b8(img, (int)xp, row, 0);
  return o;
}

__device__ __forceinline__ RowF rowfilt(RowRaw p, float s, float s2) {
  const float l = (float)p.l, m = (float)p.m, r = (float)p.r;
  RowF o;
  o.rx = r - l;
  float t = s * l;
  t = t + s2 * m;
  t = t + s * r;
  o.ry = t;
  return o;
}

// value of the lane below / above (wave-wide shift by one lane, a full-rate DPP move; lane 0 / 63
// receive 0 and are halo lanes whose results are never used)
__device__ __forceinline__ float from_lane_below(float v) {  // lane i <- lane i-1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float v) {  // lane i <- lane i+1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// K1 is bound by VALU issue slots.  Measured on gfx950 at 8 waves per SIMD (tools/probes/valu_rate.hip, valu_rate2.hip),
// cycles per wave-instruction per SIMD: v_add_f32 / v_fma_f32 / v_mov_b32 / integer add 2.3; every DPP form (a move or
// an ALU operation with a DPP source), v_cvt_*, v_cmp_*, v_add_f64, v_pk_*_f32 and v_max3_f32 4.2; v_sqrt_f32 /
// v_rsq_f32 8.2.  So each product is widened to fp64 ONCE and its neighbours' copies are fetched as fp64 -- through LDS
// memory where the LDS pipe has room, by two DPP moves per direction otherwise (see the row step).
// the fp64 value of the lane below as two DPP moves
__device__ __forceinline__ double dpp_below_f64(double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x138, 0xf, 0xf, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// ... and of the lane above
__device__ __forceinline__ double dpp_above_f64(double v) {
  const uint64_t b = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x130, 0xf, 0xf, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x130, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// one-instruction maximum of three finite floats (fmaxf chains compile to v_max_f32 pairs with
// canonicalisation; the responses here are never NaN)
__device__ __forceinline__ float fmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ float fmax2(float a, float b) {  // no NaN canonicalisation moves
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Correctly rounded fp32 square root (what sqrtf is under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt) in
// six issue slots: y = v_rsq_f32(t) (1 ulp), r = t * y (within ~2 ulp of the root), then one Markstein correction
// r' = fma(fma(-r, r, t), y / 2, r): the residual t - r^2 is exact in the fused multiply-add, the corrected value is
// within 2^-44 relative of the root before its single rounding.  That this IS the correctly rounded root for every
// float in [2^-100, 2^128) on gfx950 is not argued but checked exhaustively (tools/probes/sqrt_exhaustive.hip, and
// tests/test_keypoints_gpu.py::test_fast_sqrt_is_correctly_rounded_everywhere through vsl_diag_sqrt_check): 0
// mismatches with sqrtf over all 2^31 bit patterns of that range.  v_sqrt_f32 + the two neighbour residuals used before
// cost two more v_cmp / v_cndmask pairs (4.2 + 3.1 cycles each against 2.3 for an fma).
//  * t = 0 (flat patches): rsq sees K1_SQRT_FLOOR instead (no infinity), r = 0 * y = 0, residual 0, result 0.
//  * 0 < t < 2^-100 (cannot arise from the sums above other than through cancellation to the last bits): wave-uniform
//    branch to the library expansion; a wave that only holds zeros leaves it after one more compare.
#define K1_SQRT_FLOOR 0x1p-100f
__device__ __forceinline__ float fast_sqrt_rn(float t) {
  const float y = __builtin_amdgcn_rsqf(__builtin_fmaxf(t, K1_SQRT_FLOOR));
  const float r = t * y;
  const float e = __builtin_fmaf(-r, r, t);
  return __builtin_fmaf(e, 0.5f * y, r);
}
__device__ __forceinline__ float sqrt_rn(float t) {
  float r = fast_sqrt_rn(t);
  const bool small = t < K1_SQRT_FLOOR;
  if (__builtin_amdgcn_ballot_w64(small) != 0ull) {  // keeps the library expansion off the common path
    asm volatile("" ::: "memory");  // not speculatable: stops the compiler from flattening the branch
    const bool rare = small && t > 0.f;
    if (__builtin_amdgcn_ballot_w64(rare) != 0ull) {
      asm volatile("" ::: "memory");
      if (rare) r = sqrtf(t);
    }
  }
  return r;
}

// diagnostic behind vsl_diag_sqrt_check: compares fast_sqrt_rn with sqrtf on every float bit pattern in [lo, hi]
__global__ void sqrt_check_kernel(uint32_t lo, uint32_t hi, unsigned long long* bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long n = 0;
  for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
    const float t = __builtin_bit_cast(float, (uint32_t)b);
    n += __builtin_bit_cast(uint32_t, fast_sqrt_rn(t)) != __builtin_bit_cast(uint32_t, sqrtf(t));
  }
  if (n) atomicAdd(bad, n);
}

__global__ void detect_init_kernel(int32_t* meta, int first, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    meta[(size_t)(first + i) * VSL_META_STRIDE + VSL_META_MAX] = INT32_MIN;
    meta[(size_t)(first + i) * VSL_META_STRIDE + VSL_META_NCAND] = 0;
  }
}

// K1 + K2a fused.  One workgroup of 256 threads per tile (ceil(w/60) x ceil(h/(4*K1_ROWS)) tiles per image, 1-D grid, see
// below): wave v owns the K1_ROWS-row strip (tile_y*4 + v) of the 60-column strip tile_x.
// One image column per lane; rows
// are walked top to bottom with the row-filter results, the fp64 row sums and three response rows held
// in registers; column neighbours come from DPP lane shifts.  Nothing but the image is read and -- in
// the normal pipeline -- nothing but the candidate list is written: the response image (4 bytes per
// pixel) never goes to memory.
//
// Candidates.  goodFeaturesToTrack keeps a pixel iff its thresholded response is non-zero and equals the
// 3x3 maximum of the thresholded response, with threshold = 0.01 * (global maximum) -- unknown until the
// whole image has been processed.  For a positive threshold that is equivalent to
//      response > threshold   and   no 3x3 neighbour has a larger raw response,
// so this kernel emits every pixel with response > 0 and no larger neighbour ("provisional
// candidates", a superset), and the selection kernel drops those at or below the threshold once the
// maximum is known.  (A response image whose maximum is <= 0 yields no corners here; OpenCV could only
// differ on an image whose responses are all negative.)
// Key = (order-preserving fp32 bits << 32) | y << 16 | x (ordered like the pixel index y * w + x, and the selection
// kernel gets the position back without a division): a descending sort on the key is the
// reference's order (value descending, equal values by address descending).
template <bool STORE_RESPONSE, int ROWS>
__global__ __launch_bounds__(256) void min_eig_response_kernel(const uint8_t* __restrict__ images,
                                                               float* __restrict__ response, int32_t* __restrict__ meta,
                                                               uint64_t* __restrict__ cand, size_t cand_cap, int w, int h,
                                                               int first, int wlist_cap, int n_images, int tiles_x, int tiles_y) {
  __shared__ uint64_t list[4][K1_WLIST];  // wave-private lists: appended with a scalar counter, no atomics
  __shared__ int wave_n[4], wave_max[4], g_base;
  __shared__ double xbuf[4][2][66];  // lane exchange of two of the three products: slot lane + 1 of the wave's row (slots 0 / 65 are
                                     // read by the halo lanes 0 / 63 only, whose sums are never used)
  // XCD-aware 1-D grid: workgroups are dealt round-robin over the 8 XCDs, each with its own L2, and the 39 strips of a
  // 752 x 480 image overlap in halo rows and share 128-byte lines between neighbouring column strips.  With the plain
  // (x, y, image) grid every XCD fetched most of every image (FETCH_SIZE, calibrated: 3.75 x the image bytes per
  // launch); here workgroup b works for XCD b % 8, which takes the images congruent to it modulo 8, all tiles of one
  // image consecutively -- the same trick as the describe kernel.  Speed only: any placement computes the same.
  const int tiles = tiles_x * tiles_y;
  int img_i, tile;
  if (n_images >= 8) {
    const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3;
    img_i = (sl / tiles) * 8 + xcd;
    tile = sl - (sl / tiles) * tiles;
  } else {
    img_i = blockIdx.x / tiles;
    tile = blockIdx.x - img_i * tiles;
  }
  if (img_i >= n_images) return;  // workgroup-uniform (the padded tail of the last group of eight)
  const int bx = tile % tiles_x, by = tile / tiles_x;
  const int slot = first + img_i;
  const ImgSrd img = __builtin_amdgcn_make_buffer_rsrc((void*)(images + (size_t)slot * w * h), 0, w * h, 0x00020000);
  float* __restrict__ resp = response + (size_t)slot * w * h;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // row bookkeeping below stays scalar
  const int xs = bx * K1_COLS;
  const int y0 = (by * 4 + wave) * ROWS;
  int n_wave = 0;  // wave-uniform
  int own_max = INT32_MIN;  // ordered bits of the largest response of this wave's strip
  if (y0 < h) {
    const float s = (float)(1.0 / (4.0 * 3.0 * 255.0));
    const float s2 = 2.0f * s;
    const int x = xs - 2 + lane;
    const unsigned xe = (unsigned)reflect101(min(max(x, -1), w), w);  // out-of-range lanes are parked on a valid column
    const unsigned xm = (unsigned)reflect101((int)xe - 1, w), xp = (unsigned)reflect101((int)xe + 1, w);
    const bool own_col = lane >= 2 && lane < 2 + K1_COLS && x < w;
    const bool cand_col = own_col && x >= 1 && x < w - 1;
    const unsigned long long cand_lanes = __builtin_amdgcn_ballot_w64(cand_col);

    // Register state of the walk down the strip (roles rotated by NAME, no register-to-register moves):
    //  * two generations of fp64 row sums R(x, q) and one PAIR sum.  Every fp64 sum here is exact (each product is a
    //    multiple of 2^-47 below 1; nine of them fit 53 bits), so the order is free and consecutive rows share a pair:
    //      even step q:  B = R(q-1) + R(q),  A(q-1) = R(q-2) + B        odd step q+1:  A(q) = B + R(q+1)
    //    -- three fp64 additions per product per two rows instead of four;
    //  * two response rows (period 2, tied to the parity) and three rows of horizontal 3-maxima H (period 3);
    //  * the row filters of three consecutive rows (period 3).  The row loop is therefore unrolled by six.
    struct R3 {
      double xx, xy, yy;
    };
    R3 ra = {0, 0, 0}, rb = ra, bs = ra;
    float va = 0.f, vb = 0.f, hA = 0.f, hB = 0.f, hC = 0.f;
    RowF fA = {0.f, 0.f}, fB = fA, fC = fA;  // row filters of three consecutive rows
    int prev_ye = -100, pre_row = -100;
    RowRaw pre = {0u, 0u, 0u};
    float vmax = -3.0e38f;
    const int y_end = min(h, y0 + ROWS);
    // one step: q = row whose row sums are produced (into the slot of row q-2, dead by then).  Afterwards v_dn is the
    // response of row q-1 and h_dn its horizontal 3-maximum; the candidate row is q-2 (v_mid; H rows q-3, q-2, q-1).
    // STEADY (compile-time): the caller guarantees rows q-1 .. q+1 are inside the image and that the previous
    // step was row q-1, so f0 / f1 / `pre` are already what this step needs -- no reload path, hence no join
    // whose register copies would land on the common path.
    // EVEN (compile-time): parity of the step index (see above); r2 / r1 hold R(q-2) / R(q-1).
    auto step = [&](auto steady_tag, auto even_tag, int q, R3& r2, R3& r1, float& v_mid_slot, float& v_dn_slot, float& h_up,
                    float& h_mid, float& h_dn, RowF& f0, RowF& f1, RowF& f2) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      constexpr bool EVEN = decltype(even_tag)::value;
      // on entry (steady state) f0 / f1 hold rows ye-1 / ye from the previous step and f2 is the dead slot
      if (STEADY) {
        f2 = rowfilt(pre, s, s2);
        pre = rowload(img, w, min(q + 2, h - 1), xm, xe, xp);
      } else {
        const int ye = reflect101(min(max(q, -1), h), h);
        if (ye == prev_ye + 1 && ye + 1 < h) {
          if (pre_row != ye + 1) pre = rowload(img, w, ye + 1, xm, xe, xp);  // scalar branch, not taken in steady state
          f2 = rowfilt(pre, s, s2);
        } else {
          f0 = rowfilt(rowload(img, w, reflect101(ye - 1, h), xm, xe, xp), s, s2);
          f1 = rowfilt(rowload(img, w, ye, xm, xe, xp), s, s2);
          f2 = rowfilt(rowload(img, w, reflect101(ye + 1, h), xm, xe, xp), s, s2);
        }
        // bytes of the row the next step will filter: in flight during this step's arithmetic
        pre_row = min(ye + 2, h - 1);
        pre = rowload(img, w, pre_row, xm, xe, xp);
        prev_ye = ye;
      }
      const float dx = (f0.rx + f2.rx) * s + f1.rx * s2;
      const float dy = f2.ry - f0.ry;
      // fp64 row sums R(x, q) = (left + centre) + right of the three product images: each product is widened once
      // and its two neighbours' copies are fetched as fp64 (exact either way)
      const double dxx = (double)(dx * dx), dxy = (double)(dx * dy), dyy = (double)(dy * dy);
      // Lane exchange.  dxx and dxy go through LDS memory: one 8-byte write into the wave's own row, the two neighbours'
      // copies come back with one ds_read2_b64 (no VALU slot; LDS operations of one wave execute in order, so the
      // wave-private rows need no barrier -- the compiler barriers keep the reads between this step's and the next
      // step's writes).  dyy takes four DPP moves: the LDS pipe would saturate with all three (measured per 1024
      // images: DPP below + ds_bpermute above for all three 0.653 ms, LDS memory for all three 0.641, this mix 0.636,
      // DPP both ways for all three 0.712; tools/probes/valu_rate2.hip: a ds_write_b64 + ds_read2_b64 pair costs 14
      // CU-cycles of LDS time, four ds_bpermute_b32 24, a DPP move 4.2 VALU cycles).
      R3 cur;
      {
        xbuf[wave][0][lane + 1] = dxx;
        xbuf[wave][1][lane + 1] = dxy;
        asm volatile("" ::: "memory");
        const double lxx = xbuf[wave][0][lane], uxx = xbuf[wave][0][lane + 2];
        const double lxy = xbuf[wave][1][lane], uxy = xbuf[wave][1][lane + 2];
        asm volatile("" ::: "memory");
        cur.xx = (lxx + dxx) + uxx;
        cur.xy = (lxy + dxy) + uxy;
        cur.yy = (dpp_below_f64(dyy) + dyy) + dpp_above_f64(dyy);
      }
      // response of row y = q - 1 from the row sums of rows q-2, q-1, q
      const int y = q - 1;
      double Axx, Axy, Ayy;
      if (EVEN) {
        bs.xx = r1.xx + cur.xx;
        bs.xy = r1.xy + cur.xy;
        bs.yy = r1.yy + cur.yy;
        Axx = r2.xx + bs.xx;
        Axy = r2.xy + bs.xy;
        Ayy = r2.yy + bs.yy;
      } else {
        Axx = bs.xx + cur.xx;
        Axy = bs.xy + cur.xy;
        Ayy = bs.yy + cur.yy;
      }
      r2 = cur;  // the slot of row q-2 is dead
      // lambda_min = (a + c) - sqrt((a - c)^2 + b^2) with a = X/2, c = Y/2.  Halving is exact (X, Y are zero or above
      // 2^-25: never denormal), so a - c = fl(X - Y)/2 and a + c = fl(X + Y)/2: the difference is halved before it is
      // squared like the oracle does, the sum inside the last fused multiply-add (its product is exact).
      const float X = (float)Axx, b = (float)Axy, Y = (float)Ayy;
      const float xpy = X + Y, d = 0.5f * (X - Y);
      float t = d * d;
      t = t + b * b;
      const float v_dn = __builtin_fmaf(0.5f, xpy, -sqrt_rn(t));
      float h_tmp;
      // horizontal 3-maximum of the new response row: two v_max_f32 with a DPP source (bound_ctrl:0 feeds 0.0 into
      // lanes 0 / 63, halo lanes whose H is never used by an owned column).  The two wait states in front are NOT optional:
      // v_dn comes straight out of a v_fma, a VALU write of a VGPR must be two wait states old before a DPP operation
      // reads it (the ISA's list of manually inserted wait states), and the compiler's hazard recogniser does not see a
      // DPP read inside inline assembly.  Without them one-image launches (a lone wave per SIMD issues back to back; in a
      // batch the other waves' instructions space the two out) read stale neighbours at some lanes, emitted false local
      // maxima, and some of those displaced real corners: wrong on 5 of 120 synthetic frames, 5-6 runs of 6.
      asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                   "v_max_f32_dpp %1, %2, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                   : "=&v"(h_tmp), "=&v"(h_dn)
                   : "v"(v_dn));
      if (own_col && y >= y0 && y < y_end) {
        if (STORE_RESPONSE) resp[y * w + x] = v_dn;
        vmax = fmax2(vmax, v_dn);
      }
      // candidate test for row yc = q - 2: v_mid > 0 and no larger value among its 8 neighbours  <=>  v_mid > 0 and
      // v_mid >= the maximum of the 3 x 3 block (itself included) = max3 of the three H rows.  The sign test waits
      // inside the branch: rows without any local maximum (most rows) never pay for it.
      const float v_mid = v_mid_slot;
      const int yc = q - 2;
      if (yc >= y0 && yc < y_end && yc >= 1 && yc < h - 1) {  // scalar
        const float m9 = fmax3(h_up, h_mid, h_dn);
        const bool ge0 = v_mid >= m9;
        if ((__builtin_amdgcn_ballot_w64(ge0) & cand_lanes) != 0ull) {
          const bool ge = ge0 && v_mid > 0.f;
          const unsigned long long mask = __builtin_amdgcn_ballot_w64(ge) & cand_lanes;
          if (cand_col && ge) {
            const uint32_t ob = (uint32_t)vsl_float_to_ordered(v_mid) ^ 0x80000000u;
            const uint64_t key = ((uint64_t)ob << 32) | ((uint32_t)yc << 16) | (uint32_t)x;
            const int p = n_wave + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (__builtin_expect(p < wlist_cap, 1)) {
              list[wave][p] = key;
            } else {  // more than K1_WLIST candidates in one 60 x 41 strip (plateaus): rare direct append
              const int g = atomicAdd(&meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND], 1);
              if ((size_t)g < cand_cap) cand[(size_t)slot * cand_cap + g] = key;
            }
          }
          n_wave += __popcll(mask);
        }
      }
      v_dn_slot = v_dn;
    };
    const int q_first = y0 - 2, q_last = y_end + 1;
    typedef std::integral_constant<bool, true> Steady;
    typedef std::integral_constant<bool, false> Generic;
    typedef std::integral_constant<bool, true> Even;
    typedef std::integral_constant<bool, false> Odd;
    // step i of the strip (i = q - q_first): parity i % 2, H / row-filter roles i % 3
#define K1_STEP_E0(T, q) step(T{}, Even{}, q, ra, rb, va, vb, hA, hB, hC, fA, fB, fC)
#define K1_STEP_O1(T, q) step(T{}, Odd{}, q, rb, ra, vb, va, hB, hC, hA, fB, fC, fA)
#define K1_STEP_E2(T, q) step(T{}, Even{}, q, ra, rb, va, vb, hC, hA, hB, fC, fA, fB)
#define K1_STEP_O3(T, q) step(T{}, Odd{}, q, rb, ra, vb, va, hA, hB, hC, fA, fB, fC)
#define K1_STEP_E4(T, q) step(T{}, Even{}, q, ra, rb, va, vb, hB, hC, hA, fB, fC, fA)
#define K1_STEP_O5(T, q) step(T{}, Odd{}, q, rb, ra, vb, va, hC, hA, hB, fC, fA, fB)
    // K1_OPEN opening steps (the first one reloads everything), then the loop unrolled by six starting at role K1_OPEN
    constexpr int K1_OPEN = (ROWS + 4) % 6;
    static_assert(K1_OPEN == 3 || K1_OPEN == 4, "opening sequence written for 3 or 4 steps");
#define K1_STRIP(T)                                \
  K1_STEP_E0(Generic, q_first);                    \
  K1_STEP_O1(T, q_first + 1);                      \
  K1_STEP_E2(T, q_first + 2);                      \
  if constexpr (K1_OPEN == 3) {                    \
    for (int q = q_first + 3; q <= q_last; q += 6) { \
      K1_STEP_O3(T, q);                            \
      K1_STEP_E4(T, q + 1);                        \
      K1_STEP_O5(T, q + 2);                        \
      K1_STEP_E0(T, q + 3);                        \
      K1_STEP_O1(T, q + 4);                        \
      K1_STEP_E2(T, q + 5);                        \
    }                                              \
  } else {                                         \
    K1_STEP_O3(T, q_first + 3);                    \
    for (int q = q_first + 4; q <= q_last; q += 6) { \
      K1_STEP_E4(T, q);                            \
      K1_STEP_O5(T, q + 1);                        \
      K1_STEP_E0(T, q + 2);                        \
      K1_STEP_O1(T, q + 3);                        \
      K1_STEP_E2(T, q + 4);                        \
      K1_STEP_O3(T, q + 5);                        \
    }                                              \
  }
    if (q_first >= 1 && q_last + 1 < h) {  // scalar: a strip whose halo rows are all interior
      K1_STRIP(Steady)
    } else {
      K1_STRIP(Generic)
    }
#undef K1_STRIP
#undef K1_STEP_E0
#undef K1_STEP_O1
#undef K1_STEP_E2
#undef K1_STEP_O3
#undef K1_STEP_E4
#undef K1_STEP_O5
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    own_max = vsl_float_to_ordered(vmax);
    if (lane == 0) atomicMax(&meta[(size_t)slot * VSL_META_STRIDE + VSL_META_MAX], own_max);
  }
  if (lane == 0) wave_max[wave] = own_max;
  __syncthreads();
  // Early quality cut: the image maximum is at least the maximum of this workgroup's four strips, and the selection
  // kernel drops every candidate at or below (float)(max * 0.01) (monotone in max), so a candidate at or below the
  // threshold of the workgroup's OWN maximum can never survive -- it is dropped here, before it costs 8 bytes of
  // traffic each way (15.5k -> ~4.7k keys per synthetic frame, 17k -> ~9k on EuRoC frames).  Exact: only keys the
  // selection kernel would discard are removed.
  int kept = min(n_wave, wlist_cap);
  {
    const float wg_max = vsl_ordered_to_float(max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3])));
    if (wg_max > 0.f && kept > 0) {
      const float thr = (float)((double)wg_max * VSL_QUALITY_LEVEL);
      const uint32_t floor_hi = (uint32_t)vsl_float_to_ordered(thr) ^ 0x80000000u;
      const int have = kept;
      kept = 0;
      for (int i0 = 0; i0 < have; i0 += 64) {  // in-place, order-preserving compaction of the wave's own list
        const int i = i0 + lane;
        const uint64_t key = i < have ? list[wave][i] : 0ull;
        const bool keep = i < have && (uint32_t)(key >> 32) > floor_hi;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) list[wave][kept + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = key;
        kept += __popcll(m);
      }
    }
  }
  if (lane == 0) wave_n[wave] = kept;
  __syncthreads();
  const int c0 = wave_n[0], c1 = wave_n[1], c2 = wave_n[2], c3 = wave_n[3];
  const int n = c0 + c1 + c2 + c3;
  if (n == 0) return;
  if (threadIdx.x == 0) g_base = atomicAdd(&meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND], n);
  __syncthreads();
  const int base = g_base + (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
  const int mine = wave_n[wave];
  for (int i = lane; i < mine; i += 64)
    if ((size_t)(base + i) < cand_cap) cand[(size_t)slot * cand_cap + base + i] = list[wave][i];
}

// ------------------------------------------------------------------------------------------ K2b
// One 1024-thread workgroup per image.  The candidate keys above the quality threshold are sorted (descending) in
// LDS -- a counting sort on the top response bits plus in-bin ranking for the usual case of at most SEL_CHUNK keys,
// the bitonic network for chunks beyond that (a most-significant-digit radix select finds the chunk boundary) and
// for over-full bins -- and consumed in rank order, 1024 at a time, by an exact parallel restatement of the
// reference's sequential greedy loop:
//   a candidate is dropped if an ALREADY ACCEPTED corner lies in the 3x3 neighbouring 8-px cells at
//   squared distance < 64; among the survivors of one batch, a candidate waits for every
//   higher-ranked survivor within that distance to be decided, is rejected if one of them was
//   accepted, and is accepted otherwise (the lowest undecided rank always decides, so the rounds
//   terminate and the result is the sequential one).  Acceptance stops at num_features.
#define SEL_THREADS 1024
#define SEL_CHUNK 8192
#define SEL_EMPTY 0xFFFFFFFFu
#define SEL_BINS 1024      // response bins of the counting sort (= threads of the workgroup)
#define SEL_MAX_BLOCKERS 6  // blockers of a candidate kept in registers, 10 bits each in one 64-bit word (more: the cell lists are walked again)
static_assert(SEL_THREADS <= 1024 && SEL_MAX_BLOCKERS * 10 <= 64, "blocker packing");

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

// Static LDS: sized for up to SEL_MAX_CELLS 8x8-px cells (752x480 needs 96 x 62 = 5952 with the empty ring); gfx950 lets one
// workgroup own (almost) all 160 KiB of its CU's LDS.
#define SEL_MAX_CELLS 6144
// the key array is padded by one key per 32 (bank-conflict relief for the strided layouts of the sort)
#define SEL_KEYS_PADDED (SEL_CHUNK + SEL_CHUNK / 32)
#define SEL_PHYS(i) ((i) + ((i) >> 5))
#define SEL_LDS_BYTES (SEL_KEYS_PADDED * 8 + SEL_MAX_CELLS * 2 * 4 + SEL_THREADS * 4 * 3 + SEL_MAX_CELLS * 4 + 2048)
static_assert(SEL_LDS_BYTES <= 160 * 1024, "selection kernel LDS budget");

// Register-blocked bitonic network on N = 1024 << LOGK keys in LDS (descending): a thread holds the 1 << LOGK
// keys whose indices differ in bits b .. b+LOGK-1, so up to LOGK consecutive sub-steps are compare-exchanges
// between its own registers; between such groups the keys pass through LDS once to change b.
template <int LOGK>
__device__ __forceinline__ void sel_sort_blocked(uint64_t* keys, int tid, int log_n) {
  constexpr int KPT = 1 << LOGK;
  for (int m = 1; m <= log_n; m++) {
    const int k = 1 << m;
    for (int top = m - 1; top >= 0;) {
      const int b = top >= LOGK - 1 ? top - (LOGK - 1) : 0;  // this group: bits top .. b
      const int nb = top - b + 1;
      const int t_hi = tid >> b, t_lo = tid & ((1 << b) - 1);
      const int i0 = (t_hi << (b + LOGK)) | t_lo;  // index of the thread's key r = 0; key r sits at i0 | (r << b)
      uint64_t rk[KPT];
#pragma unroll
      for (int r = 0; r < KPT; r++) rk[r] = keys[SEL_PHYS(i0 | (r << b))];
#pragma unroll
      for (int x = LOGK - 1; x >= 0; x--) {
        if (x < nb) {
#pragma unroll
          for (int r = 0; r < KPT; r++) {
            if ((r & (1 << x)) == 0) {
              const bool desc = ((i0 | (r << b)) & k) == 0;
              const uint64_t u = rk[r], v = rk[r | (1 << x)];
              const bool sw = desc ? (u < v) : (u > v);
              rk[r] = sw ? v : u;
              rk[r | (1 << x)] = sw ? u : v;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < KPT; r++) keys[SEL_PHYS(i0 | (r << b))] = rk[r];
      __syncthreads();
      top = b - 1;
    }
  }
}

// GRID_GLOBAL = false: the accepted-corner grid and the batch list heads live in LDS (images of up to
// SEL_MAX_CELLS cells, e.g. 752 x 480).  GRID_GLOBAL = true: larger images keep the two per-cell arrays
// in a global scratch (grid_scratch, 3 words per cell per image, L2-resident) -- same algorithm, slower.
template <bool GRID_GLOBAL>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const uint64_t* __restrict__ cand_all,
                                                             int32_t* meta,
                                                             int32_t* __restrict__ kp_xy, int32_t* __restrict__ kp_count,
                                                             int w, int h, size_t cand_cap, int F, int first,
                                                             int num_features, int border, double quality,
                                                             uint32_t* __restrict__ grid_scratch, int bucket_cap) {
  __shared__ __align__(16) unsigned char smem[GRID_GLOBAL ? (SEL_KEYS_PADDED * 8 + SEL_THREADS * 4 * 3 + 2048) : SEL_LDS_BYTES];
  const int slot = first + blockIdx.x;
  const uint64_t* __restrict__ cand = cand_all + (size_t)slot * cand_cap;
  const int n_cand_raw = meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND];
  const int n_cand = min(n_cand_raw, (int)cand_cap);
  // 8 x 8-px cells with a ring of empty cells around the image: every 3 x 3 neighbourhood is nine fixed offsets, no
  // clamping, so its nine list heads / eighteen accepted slots are independent loads issued together
  const int gw = (w + 7) / 8 + 2, gh = (h + 7) / 8 + 2, cells = (gw * gh + 1) & ~1;  // (even: the arrays behind stay 8-byte aligned)
  // LDS carve-up (all regions 8-byte aligned)
  uint64_t* keys = (uint64_t*)smem;                 // SEL_CHUNK sorted keys
  SelShared* sh = (SelShared*)(keys + SEL_KEYS_PADDED);
  // cells * 2 accepted corners, packed x | y << 16, then cells batch list heads (-1 = empty); volatile:
  // the global variant must not keep them in registers / stale L1 lines between workgroup barriers
  volatile uint32_t* acc = GRID_GLOBAL ? (volatile uint32_t*)(grid_scratch + (size_t)slot * cells * 3) : (volatile uint32_t*)(sh + 1);
  volatile int* head = (volatile int*)(acc + 2 * (size_t)cells);
  int* next = GRID_GLOBAL ? (int*)(sh + 1) : (int*)(head + cells);  // SEL_THREADS
  // batch member u: node[u] = (position x | y << 16, next member of its cell's list) -- one 8-byte read per list step
  unsigned long long* node = (unsigned long long*)next;  // SEL_THREADS (the space of `next` and the array after it)
  int* state = next + 2 * SEL_THREADS;              // SEL_THREADS: 0 undecided, 1 accepted, 2 rejected
  const int tid = threadIdx.x;

  for (int i = tid; i < 2 * cells; i += SEL_THREADS) acc[i] = SEL_EMPTY;
  for (int i = tid; i < cells; i += SEL_THREADS) head[i] = -1;
  __syncthreads();

  int n_acc = 0, n_out = 0;
  unsigned long long hi = ~0ull;  // keys >= hi have been consumed
  int32_t* out = kp_xy + (size_t)slot * F * 2;
  // quality threshold (goodFeaturesToTrack: eig > maxVal * qualityLevel survives THRESH_TOZERO): the
  // provisional candidates at or below it do not exist as far as the rest of the kernel is concerned
  const float maxv = vsl_ordered_to_float(meta[(size_t)slot * VSL_META_STRIDE + VSL_META_MAX]);
  const float thr = (float)((double)maxv * quality);  // (K1's early cut uses the same expression)
  const unsigned long long floor_key =
      (((unsigned long long)((uint32_t)vsl_float_to_ordered(thr) ^ 0x80000000u)) << 32) | 0xFFFFFFFFull;
  int remaining = 0;
  // Sorting (first chunk, the normal case of at most SEL_CHUNK surviving candidates): a counting sort on the top
  // bits of the response -- SEL_BINS bins between the threshold and the maximum (both known), bin 0 = largest --
  // puts every key into its bin's slot range, and a key's final rank inside its bin is the number of larger keys
  // there (bins hold a handful of keys: the responses of real images spread over ~6 binades = ~850 bins).  Four
  // passes with a handful of workgroup barriers instead of the 35 LDS exchanges of the bitonic network for 8192
  // keys, which stays as the path for later chunks and for bins fuller than bucket_cap (plateaus of equal responses).
  int* bcount = next;                    // the three arrays alias next / cxy / state, which only the greedy uses
  int* bstart = next + SEL_THREADS;
  int* bcursor = next + 2 * SEL_THREADS;
  static_assert(SEL_BINS == SEL_THREADS, "one bin per thread");
  const uint32_t hi_max = (uint32_t)vsl_float_to_ordered(maxv) ^ 0x80000000u;
  const uint32_t hi_floor = (uint32_t)vsl_float_to_ordered(thr) ^ 0x80000000u;
  int bshift = 16;
  while (bshift < 31 && (hi_max >> bshift) - (hi_floor >> bshift) >= (uint32_t)SEL_BINS) bshift++;
  const uint32_t bin_top = hi_max >> bshift;
  bool bucket_sort = false;
  if (maxv > 0.f) {
    bcount[tid] = 0;
    if (tid == 0) sh->n_chunk = 0;
    __syncthreads();
    for (int i = tid; i < n_cand; i += SEL_THREADS) {
      const unsigned long long key = cand[i];
      if (key > floor_key) {
        // a provisional candidate above the image maximum cannot exist; the clamp keeps a corrupted list in range
        const uint32_t kb = min((uint32_t)(key >> 32) >> bshift, bin_top);
        atomicAdd(&bcount[bin_top - kb], 1);
      }
    }
    __syncthreads();
    // exclusive prefix of the bin counts (bin 0 first = descending keys) and the fullest bin
    const int c = bcount[tid];
    int x = c;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    int cmax = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    if (lane == 63) sh->wave_tot[wv] = x;
    if (lane == 0) sh->hist[wv] = cmax;
    __syncthreads();
    int off = 0, fullest = 0;
    remaining = 0;  // number of surviving keys = the total of the bin counts (no per-thread atomic on one LDS word)
#pragma unroll
    for (int k = 0; k < 16; k++) {
      off += k < wv ? sh->wave_tot[k] : 0;
      remaining += sh->wave_tot[k];
      fullest = max(fullest, sh->hist[k]);
    }
    bstart[tid] = off + x - c;
    bcursor[tid] = 0;
    bucket_sort = remaining <= SEL_CHUNK && fullest <= bucket_cap;
    __syncthreads();
  }

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
          if (key < hi && key > floor_key && (key & pmask) == prefix) atomicAdd(&sh->hist[(int)((key >> shift) & 255)], 1);
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
    int n_chunk;
    if (bucket_sort) {
      // ---- scatter into the bins' slot ranges, then rank inside each bin
      for (int i = tid; i < n_cand; i += SEL_THREADS) {
        const unsigned long long key = cand[i];
        if (key > floor_key) {
          const int bin = (int)(bin_top - min((uint32_t)(key >> 32) >> bshift, bin_top));
          const int p = bstart[bin] + atomicAdd(&bcursor[bin], 1);  // (SEL_PHYS evaluates its argument twice)
          keys[SEL_PHYS(p)] = key;
        }
      }
      __syncthreads();
      n_chunk = remaining;
      unsigned long long mykey[SEL_CHUNK / SEL_THREADS];
      int mypos[SEL_CHUNK / SEL_THREADS];
#pragma unroll
      for (int k = 0; k < SEL_CHUNK / SEL_THREADS; k++) {
        const int p = tid + SEL_THREADS * k;
        mypos[k] = -1;
        if (p < n_chunk) {
          const unsigned long long key = keys[SEL_PHYS(p)];
          const int bin = (int)(bin_top - min((uint32_t)(key >> 32) >> bshift, bin_top));
          const int s0 = bstart[bin], e0 = s0 + bcount[bin];
          int larger = 0;
          for (int q = s0; q < e0; q++) larger += keys[SEL_PHYS(q)] > key;
          mykey[k] = key;
          mypos[k] = s0 + larger;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < SEL_CHUNK / SEL_THREADS; k++)
        if (mypos[k] >= 0) keys[SEL_PHYS(mypos[k])] = mykey[k];
      __syncthreads();
      bucket_sort = false;
    } else {
      // ---- gather the chunk into LDS and pad to a power of two
      if (tid == 0) sh->n_chunk = 0;
      __syncthreads();
      for (int i = tid; i < n_cand; i += SEL_THREADS) {
        const unsigned long long key = cand[i];
        if (key >= lo && key < hi && key > floor_key) {
          const int p = atomicAdd(&sh->n_chunk, 1);
          if (p < SEL_CHUNK) keys[SEL_PHYS(p)] = key;
        }
      }
      __syncthreads();
      n_chunk = min(sh->n_chunk, SEL_CHUNK);
      int N = 1024;
      while (N < n_chunk) N <<= 1;
      for (int i = n_chunk + tid; i < N; i += SEL_THREADS) keys[SEL_PHYS(i)] = 0ull;
      __syncthreads();
      // ---- bitonic sort, descending.  With 2, 4 or 8 keys per thread (N = 2048 / 4096 / 8192) the network is
      // register-blocked (sel_sort_blocked): ceil(m / log2(keys per thread)) LDS exchanges for the stage k = 2^m,
      // e.g. 35 workgroup barriers for 8192 keys instead of 91 (one per sub-step).
      if (N >= 2048) {
        if (N == 8192) sel_sort_blocked<3>(keys, tid, 13);
        else if (N == 4096) sel_sort_blocked<2>(keys, tid, 12);
        else sel_sort_blocked<1>(keys, tid, 11);
      } else {
        for (int k = 2; k <= N; k <<= 1) {
          for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (N >> 1); t += SEL_THREADS) {
              const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
              const int l = i | j;
              const uint64_t a = keys[SEL_PHYS(i)], b = keys[SEL_PHYS(l)];
              const bool desc = (i & k) == 0;
              if (desc ? (a < b) : (a > b)) {
                keys[SEL_PHYS(i)] = b;
                keys[SEL_PHYS(l)] = a;
              }
            }
            __syncthreads();
          }
        }
      }
    }
    // ---- greedy, SEL_THREADS ranks per batch
    for (int r0 = 0; r0 < n_chunk && n_acc < num_features; r0 += SEL_THREADS) {
      const int r = r0 + tid;
      int px = 0, py = 0, cell = 0, cx = 0, cy = 0;
      bool alive = false;
      if (r < n_chunk) {
        const uint32_t pix = (uint32_t)(keys[SEL_PHYS(r)] & 0xFFFFFFFFull);
        py = (int)(pix >> 16);
        px = (int)(pix & 0xFFFFu);
        cx = (px >> 3) + 1;
        cy = (py >> 3) + 1;
        cell = cy * gw + cx;
        alive = true;
        uint32_t a18[18];
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const int c2 = cell + (k / 3 - 1) * gw + (k % 3 - 1);
          // both slots of a cell in one 8-byte read (the LDS pipe, not the instruction count, is what this phase waits for)
          const unsigned long long two = *(volatile const unsigned long long*)&acc[2 * c2];
          a18[2 * k] = (uint32_t)two;
          a18[2 * k + 1] = (uint32_t)(two >> 32);
        }
        int near = 0;  // branch-free: eighteen short-circuit tests compile to eighteen exec-mask branches
#pragma unroll
        for (int k = 0; k < 18; k++) {
          const uint32_t a = a18[k];
          const int dx = px - (int)(a & 0xFFFF), dy = py - (int)(a >> 16);
          const unsigned d2 = (unsigned)(dx * dx) + (unsigned)(dy * dy);  // (an empty slot decodes to 65535, 65535: masked below)
          near |= (int)(a != SEL_EMPTY) & (int)(d2 < 64u);
        }
        alive = near == 0;
      }
      state[tid] = alive ? 0 : 2;
      {
        const int nxt = alive ? atomicExch((int*)&head[cell], tid) : -1;
        node[tid] = ((unsigned long long)(uint32_t)nxt << 32) | ((uint32_t)px | ((uint32_t)py << 16));
      }
      __syncthreads();
      // Blockers = higher-ranked survivors of this batch within the minimum distance.  A candidate is
      // decided once all of them are: rejected if one was accepted, accepted otherwise.  The lowest
      // undecided rank never waits, so free-running polling of the LDS state words terminates; no
      // workgroup barrier per dependency level (all 16 waves are resident and keep being scheduled).
      int nb = 0;
      unsigned long long blk = 0ull;  // the last SEL_MAX_BLOCKERS blockers, 10 bits each (batch-local ranks): a register ARRAY
                                      // indexed by nb costs six compare / select pairs per insertion
      if (alive) {
        int h9[9];
#pragma unroll
        for (int k = 0; k < 9; k++) h9[k] = head[cell + (k / 3 - 1) * gw + (k % 3 - 1)];
#pragma unroll
        for (int k = 0; k < 9; k++)
          for (int u = h9[k]; u >= 0;) {
            const unsigned long long nd = node[u];
            const int uu = u;
            u = (int)(nd >> 32);
            if (uu < tid) {
              const uint32_t q = (uint32_t)nd;
              const int dx = px - (int)(q & 0xFFFF), dy = py - (int)(q >> 16);
              if (dx * dx + dy * dy < 64) {
                blk = (blk << 10) | (unsigned long long)(unsigned)uu;
                nb++;
              }
            }
          }
      }
      bool undecided = alive;
      while (__ballot(undecided) != 0ull) {
        if (undecided) {
          bool rej = false, blocked = false;
          if (nb <= SEL_MAX_BLOCKERS) {
#pragma unroll
            for (int k = 0; k < SEL_MAX_BLOCKERS; k++)
              if (k < nb) {
                const int su = ((volatile int*)state)[(int)(blk >> (10 * k)) & 1023];
                rej = rej || (su == 1);
                blocked = blocked || (su == 0);
              }
          } else {  // crowded neighbourhood: walk the cell lists again
            for (int k = 0; k < 9; k++)
                for (int u = head[cell + (k / 3 - 1) * gw + (k % 3 - 1)], un = -1; u >= 0; u = un) {
                  const unsigned long long nd = node[u];
                  un = (int)(nd >> 32);
                  if (u < tid) {
                    const uint32_t q = (uint32_t)nd;
                    const int dx = px - (int)(q & 0xFFFF), dy = py - (int)(q >> 16);
                    if (dx * dx + dy * dy < 64) {
                      const int su = ((volatile int*)state)[u];
                      rej = rej || (su == 1);
                      blocked = blocked || (su == 0);
                    }
                  }
                }
          }
          const int ns = rej ? 2 : (blocked ? 0 : 1);
          if (ns) {
            ((volatile int*)state)[tid] = ns;
            undecided = false;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
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
        if (atomicCAS((uint32_t*)&acc[2 * cell], SEL_EMPTY, packed) != SEL_EMPTY) atomicCAS((uint32_t*)&acc[2 * cell + 1], SEL_EMPTY, packed);
      }
      if (alive) head[cell] = -1;
      n_acc = min(num_features, n_acc + total);
      n_out += total_out;
      __syncthreads();
    }
    hi = lo;
    remaining -= n_chunk;
  }
  if (tid == 0) {
    kp_count[slot] = n_out;
    // every thread read MAX / NCAND before the first barrier: leave them reset for the next detect call
    meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND_LAST] = n_cand_raw;
    meta[(size_t)slot * VSL_META_STRIDE + VSL_META_MAX] = INT32_MIN;
    meta[(size_t)slot * VSL_META_STRIDE + VSL_META_NCAND] = 0;
  }
}

// Diagnostic: the response kernel's square root against the correctly rounded library sqrtf on every float bit
// pattern in [lo_bits, hi_bits] (both must be non-negative floats); *n_mismatch = number of differing results.
extern "C" int vsl_diag_sqrt_check(vsl_ctx* ctx, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* n_mismatch) {
  if (!ctx || !n_mismatch || lo_bits > hi_bits || hi_bits > 0x7f7fffffu)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_diag_sqrt_check: bad range");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, sizeof(unsigned long long), &d);
  if (rc) return rc;
  VSL_HIP(ctx, hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL(sqrt_check_kernel, dim3(4096), dim3(256), 0, ctx->stream, lo_bits, hi_bits, (unsigned long long*)d);
  VSL_CHECK_LAUNCH(ctx);
  VSL_HIP(ctx, hipMemcpyAsync(n_mismatch, d, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}

int vsl_launch_detect(vsl_ctx* ctx, vsl_frames* f, int first, int n, int num_features) {
  if (n <= 0) return VSL_OK;
  if (num_features < 1 || num_features > f->F)
    return vsl_fail(ctx, VSL_ERR_INVALID, "num_features %d not in [1, %d]", num_features, f->F);
  static_assert(sizeof(SelShared) <= 2048, "SelShared fits its LDS slot");
  const int cells = (((f->w + 7) / 8 + 2) * ((f->h + 7) / 8 + 2) + 1) & ~1;  // with the ring of empty cells, even (as select_kernel counts them)
  const bool grid_global = cells > SEL_MAX_CELLS;
  if (grid_global && !f->sel_grid) {
    VSL_HIP(ctx, hipMalloc((void**)&f->sel_grid, sizeof(uint32_t) * 3 * (size_t)cells * f->max_images));
  }
  const int w = f->w, h = f->h;
  {
    VslStage st(ctx, VSL_STAGE_RESPONSE);
    if (f->detect_meta_dirty)
      hipLaunchKernelGGL(detect_init_kernel, dim3((f->max_images + 255) / 256), dim3(256), 0, ctx->stream, f->meta, 0, f->max_images);
    f->detect_meta_dirty = true;  // until the selection kernel (which resets the counters) is in the queue
    // strips of K1_ROWS rows for batches; short strips for launches of a few images (the tracking loop's one-image calls):
    // a lone wave per SIMD issues one instruction per ~7 cycles, so such a launch is as long as its longest wave
    const bool small = n < 8 && !f->store_response;
    const int rows = small ? K1_ROWS_SMALL : K1_ROWS;
    const int tiles_y = (h + 4 * rows - 1) / (4 * rows);
    const int tiles_x = (w + K1_COLS - 1) / K1_COLS;
    const dim3 k1_grid((unsigned)(tiles_x * tiles_y) * (unsigned)(n >= 8 ? 8 * ((n + 7) / 8) : n));
    const int wcap = ctx->k1_list_cap < 0 ? K1_WLIST : min(ctx->k1_list_cap, K1_WLIST);
    if (f->store_response)
      hipLaunchKernelGGL((min_eig_response_kernel<true, K1_ROWS>), k1_grid, dim3(256), 0, ctx->stream, f->images, f->response, f->meta,
                         f->cand, f->cand_cap, w, h, first, wcap, n, tiles_x, tiles_y);
    else if (small)
      hipLaunchKernelGGL((min_eig_response_kernel<false, K1_ROWS_SMALL>), k1_grid, dim3(256), 0, ctx->stream, f->images, f->response, f->meta,
                         f->cand, f->cand_cap, w, h, first, wcap, n, tiles_x, tiles_y);
    else
      hipLaunchKernelGGL((min_eig_response_kernel<false, K1_ROWS>), k1_grid, dim3(256), 0, ctx->stream, f->images, f->response, f->meta,
                         f->cand, f->cand_cap, w, h, first, wcap, n, tiles_x, tiles_y);
    VSL_CHECK_LAUNCH(ctx);
  }
  {
    VslStage st(ctx, VSL_STAGE_SELECT);
    if (grid_global)
      hipLaunchKernelGGL(select_kernel<true>, dim3(n), dim3(SEL_THREADS), 0, ctx->stream, f->cand, f->meta, f->kp_xy,
                         f->kp_count, w, h, f->cand_cap, f->F, first, num_features, 19, VSL_QUALITY_LEVEL, f->sel_grid, ctx->select_bucket_cap);
    else
      hipLaunchKernelGGL(select_kernel<false>, dim3(n), dim3(SEL_THREADS), 0, ctx->stream, f->cand, f->meta, f->kp_xy,
                         f->kp_count, w, h, f->cand_cap, f->F, first, num_features, 19, VSL_QUALITY_LEVEL, (uint32_t*)nullptr,
                         ctx->select_bucket_cap);
    VSL_CHECK_LAUNCH(ctx);
  }
  f->detect_meta_dirty = false;
  return VSL_OK;
}
