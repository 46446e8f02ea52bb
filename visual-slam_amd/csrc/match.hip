// match.hip -- K5: brute-force Hamming matcher with ratio test and cross-check.
//
// Replaces visnav::matchDescriptors / isPQiffQP (include/visnav/keypoints.h:323-369, :278-313).
//
// Reference semantics restated:
//   for each row descriptor, over all column descriptors in index order:
//     best/second-best distance with strict '<'  => lowest index wins ties, and "second" is the
//     second-smallest distance counted with multiplicity; both start at 256 / index 0;
//   row passes iff best < threshold and !(second < best * dist_2_best)        (double compare)
//   (i, j = best(i)) is a match iff row i passes in direction a->b, column j passes in direction
//   b->a and best_{b->a}(j) == i.  Matches are emitted in ascending i.
//
// Kernels in this file, in the order a reader meets them:
//   hamming_best2_kernel  the VALU popcount form (v_xor / v_bcnt on scalar-loaded columns, one row per lane): the
//                         diagnostic third implementation ("match_use_valu"), kept as a cross-check of the other two;
//   hamming_mfma_kernel   int8 matrix-core form, any descriptor count ("match_use_i8"; the default above 2048);
//   hamming_mx_kernel     THE DEFAULT (<= 2048 descriptors per image): the Hamming matrix as exact 0 / +-1 products on
//                         v_mfma_scale_f32_32x32x64_f8f6f4 (FP4, unit block scales), the accumulator started at the
//                         ordered key 512 + row / 2048 so that a result register IS (distance, row) -- see the comment
//                         block above it.  Launches of >= 8 pairs compute the matrix ONCE: forward pass a -> b,
//                         match_select_kernel lists the columns some passing row points at, the <REVERSE> instance
//                         scans those columns only (the reference's own order, keypoints.h:355-362);
//   match_finalize_kernel threshold / ratio / cross-check and the ordered match list.
//
// Algorithmic bytes per pair: (n_a + n_b) * 32 B read + 8 B per match written.
#include "vsl_common.h"

#ifndef VSL_MATCH_WAVES
#define VSL_MATCH_WAVES 8
#endif
// key = (distance << KEY_SHIFT) | index: 10 bits of distance headroom (256 + 256 never overflows), 22 bits of index
#define KEY_SHIFT 22
#define KEY_INIT ((256u << KEY_SHIFT) | 0u)

// v_med3_u32: median of three unsigned values (one VALU op; no clang builtin for the integer form)
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void hamming_best2_kernel(
    const uint64_t* __restrict__ desc, const int32_t* __restrict__ kp_count,
    const int32_t* __restrict__ pair_slots, uint32_t* __restrict__ best_key,
    uint32_t* __restrict__ second_key, int F) {
  const int pair = blockIdx.z, dir = blockIdx.y;
  const int slot_r = pair_slots[2 * pair + dir];      // rows
  const int slot_c = pair_slots[2 * pair + 1 - dir];  // columns
  const int n_r = kp_count[slot_r], n_c = kp_count[slot_c];
  const int row0 = blockIdx.x * 64;
  if (row0 >= n_r) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = row0 + lane;

  uint32_t r[8];
  {
    const uint32_t* p = (const uint32_t*)(desc + ((size_t)slot_r * F + (row < n_r ? row : 0)) * 4);
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = p[k];
  }
  // column range of this wave, a multiple of 4 columns
  int chunk = (n_c + WAVES - 1) / WAVES;
  chunk = (chunk + 3) & ~3;
  const int c0 = wave * chunk;
  const int c1 = min(n_c, c0 + chunk);
  const uint32_t* __restrict__ cbase = (const uint32_t*)(desc + (size_t)slot_c * F * 4);

  uint32_t b = KEY_INIT, s = KEY_INIT;
  int j = c0;
  for (; j + 4 <= c1; j += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t* __restrict__ c = cbase + (size_t)(j + u) * 8;
      uint32_t d = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) d += __builtin_popcount(r[k] ^ c[k]);
      const uint32_t key = (d << KEY_SHIFT) | (uint32_t)(j + u);
      s = umed3(b, key, s);
      b = min(b, key);
    }
  }
  for (; j < c1; j++) {
    const uint32_t* __restrict__ c = cbase + (size_t)j * 8;
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += __builtin_popcount(r[k] ^ c[k]);
    const uint32_t key = (d << KEY_SHIFT) | (uint32_t)j;
    s = umed3(b, key, s);
    b = min(b, key);
  }

  __shared__ uint32_t sb[WAVES][64], ss[WAVES][64];
  sb[wave][lane] = b;
  ss[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && row < n_r) {
    // merge (b, s) pairs in column order: second = min(med3(b1, b2, s1), s2)
#pragma unroll
    for (int w = 1; w < WAVES; w++) {
      const uint32_t b2 = sb[w][lane], s2 = ss[w][lane];
      s = min(umed3(b, b2, s), s2);
      b = min(b, b2);
    }
    const size_t o = ((size_t)pair * 2 + dir) * F + row;
    best_key[o] = b;
    second_key[o] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core variant.  popcount(a ^ b) = |a| + |b| - 2 <a, b> for bit vectors, and the inner products
// of 32 "database" descriptors with 32 "query" descriptors are one 32x32 tile of an int8 GEMM with K =
// 256 (bits expanded to 0/1 bytes): 8 x v_mfma_i32_32x32x32_i8, exact integer arithmetic.  v_bcnt_u32_b32
// issues at a quarter of the VALU rate on gfx950 (the VALU kernel above measures 43 issue slots per
// pair instead of 19), so the MFMA form is ~5x faster per distance -- this is not a GEMM dressed up for
// the matrix cores' sake, the VALU popcount is simply the slow instruction here.
//   * queries sit on the N side: a lane keeps the 32 x K fragment of ITS query in 32 VGPRs for the whole
//     kernel and receives, per tile, the 16 distances of that query to 16 database rows in its
//     accumulator registers -- so best / second-best are lane-local min / med3 updates on packed keys
//     (no cross-lane reduction; the two lane halves of a column are merged once at the end);
//   * database tiles (32 descriptors) are read packed (1 KiB), expanded to bytes by the workgroup
//     (nibble * 0x00204081 & 0x01010101) into a double-buffered LDS tile shared by the 4 waves;
//   * both directions of a pair are two independent grid slices (query / database roles swapped);
//   * query bytes are +1 / -1, database bytes 0 / 1: accumulator = |d| - 2<q, d>, distance = accumulator + |q|;
//     |q| is a per-lane constant and is added to the two surviving keys once, at the end.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

#define MM_PAD_KEY (1023u << KEY_SHIFT)  // rows past the end of the database: above every real key'
#define MM_ROW 272  // LDS bytes per expanded descriptor: 256 + 16 pad (conflict-free ds_read_b128)

__device__ __forceinline__ v4i_t expand16(uint32_t bits16) {
  v4i_t o;
  o.x = (int)((((bits16 >> 0) & 15u) * 0x00204081u) & 0x01010101u);
  o.y = (int)((((bits16 >> 4) & 15u) * 0x00204081u) & 0x01010101u);
  o.z = (int)((((bits16 >> 8) & 15u) * 0x00204081u) & 0x01010101u);
  o.w = (int)((((bits16 >> 12) & 15u) * 0x00204081u) & 0x01010101u);
  return o;
}

#ifndef MM_WAVES
#define MM_WAVES 8  // wavefronts (32 queries each) per workgroup; they share one expanded database tile (per 512 pairs: 4 waves 0.481 ms, 8: 0.450, 16: 0.467 on a slower box where 8 gave 0.470)
#endif
// STAGGER: the second half of the workgroup's waves (wave >= MM_WAVES / 2: the SIMD partners of the first half --
// a workgroup's waves go to SIMDs in cyclic order, so waves w and w + 4 share one) runs every block in the order
// [key updates of the PREVIOUS super tile, matrix instructions of this one] while the first half runs [matrix
// instructions, key updates].  Same work, same registers (the accumulators are consumed before they are overwritten),
// bit-identical results; but the two waves of a SIMD are no longer in lockstep -- while one holds the matrix pipe
// the other issues its v_lshl_add / v_med3 / v_min chain (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).
// The row-key table is a ring of three (the deferred epilogue reads st - 1 while st + 1 is being written).
template <bool STAGGER>
__global__ __launch_bounds__(64 * MM_WAVES) void hamming_mfma_kernel(const uint64_t* __restrict__ desc,
                                                           const int32_t* __restrict__ kp_count,
                                                           const int32_t* __restrict__ pair_slots,
                                                           uint32_t* __restrict__ best_key,
                                                           uint32_t* __restrict__ second_key, int F) {
  // a "super tile" = 64 database descriptors = two MFMA tiles per workgroup barrier
  __shared__ __align__(16) unsigned char tile[2][64 * MM_ROW];
  __shared__ __align__(16) uint32_t rowkey[3][64];  // (256 << KEY_SHIFT) | m, or MM_PAD_KEY past the end; ring of 3
  const int pair = blockIdx.z, dir = blockIdx.y;
  const int slot_q = pair_slots[2 * pair + dir];      // queries (rows of the result)
  const int slot_d = pair_slots[2 * pair + 1 - dir];  // database (columns of the reference's loop)
  const int n_q = kp_count[slot_q], n_d = kp_count[slot_d];
  const int q0 = blockIdx.x * (32 * MM_WAVES);
  if (q0 >= n_q) return;  // workgroup-uniform
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int qc = q0 + wave * 32 + c;
  const uint32_t* __restrict__ qd = (const uint32_t*)(desc + ((size_t)slot_q * F + (qc < n_q ? qc : 0)) * 4);
  const uint32_t* __restrict__ dbase = (const uint32_t*)(desc + (size_t)slot_d * F * 4);

  // query fragment: bit 1 -> -1, bit 0 -> +1, database bytes are 0 / 1, so the accumulator is
  //   sum_k d_k (1 - 2 q_k) = |d| - 2 <q, d>   and   distance = accumulator + |q|
  v4i_t bq[8];
  int pq = 0;
#pragma unroll
  for (int s = 0; s < 8; s++) {
    const uint32_t wq = qd[s];
    pq += __builtin_popcount(wq);
    const v4i_t e = expand16((wq >> (16 * h)) & 0xFFFFu);
    // bytes are 0/1: * 0xFE gives 0x00 / 0xFE without carries between bytes, + 1 per byte -> 0x01 / 0xFF
    bq[s].x = (int)((uint32_t)e.x * 0xFEu + 0x01010101u);
    bq[s].y = (int)((uint32_t)e.y * 0xFEu + 0x01010101u);
    bq[s].z = (int)((uint32_t)e.z * 0xFEu + 0x01010101u);
    bq[s].w = (int)((uint32_t)e.w * 0xFEu + 0x01010101u);
  }

  // tile fill: thread t expands words (t & 7) of database rows (t >> 3) and (t >> 3) + 32 of the super tile
  // tile fill: the 64 rows x 8 words of a super tile are spread over the workgroup's threads (FILL_K words each)
  constexpr int FILL_K = 64 * MM_WAVES >= 512 ? 1 : 512 / (64 * MM_WAVES);  // 2 with four waves, 1 with eight or more
  const bool filler = tid < 512;  // sixteen waves: the first eight fill
  constexpr int FILL_ROWS = 64 / FILL_K;         // rows covered per pass
  const int frow = tid >> 3, fword = tid & 7;
  auto load_words = [&](int st, uint32_t& w0, uint32_t& w1) {
    const int r0 = st * 64 + frow, r1 = r0 + FILL_ROWS;
    w0 = (filler && r0 < n_d) ? dbase[(size_t)r0 * 8 + fword] : 0u;
    w1 = (FILL_K > 1 && r1 < n_d) ? dbase[(size_t)r1 * 8 + fword] : 0u;
  };
  auto store_rows = [&](int buf, int st, uint32_t w0, uint32_t w1) {
    if (!filler) return;
#pragma unroll
    for (int k = 0; k < FILL_K; k++) {
      const uint32_t wd = k ? w1 : w0;
      const int lr = frow + FILL_ROWS * k;
      unsigned char* dst = &tile[buf][lr * MM_ROW + fword * 32];
      *(v4i_t*)dst = expand16(wd & 0xFFFFu);
      *(v4i_t*)(dst + 16) = expand16(wd >> 16);
      const int m = st * 64 + lr;
      if (fword == 0) rowkey[st % 3][lr] = m < n_d ? ((256u << KEY_SHIFT) | (uint32_t)m) : MM_PAD_KEY;
    }
  };
  const int n_st = (n_d + 63) / 64;
  // Keys are tracked WITHOUT |q| (a per-lane constant, it does not change the order):
  //   key' = ((accumulator + 256) << KEY_SHIFT) | m = one v_lshl_add_u32 on the accumulator and the row's
  // constant; |q| - 256 is added once at the end.  accumulator + 256 is in [0, 512], padded rows carry
  // MM_PAD_KEY (distance field 1023, above every real key', below 2^32).
  // Two independent (best, second) trackers -- one per MFMA tile of the super tile -- halve the length of the
  // dependent med3 / min chain a wave has to walk per super tile (4 waves per SIMD cannot hide it); they are
  // merged once, after the last super tile.
  uint32_t b = 0xFFFFFFFFu, sk = 0xFFFFFFFFu, bB = 0xFFFFFFFFu, skB = 0xFFFFFFFFu;
  {
    uint32_t w0, w1;
    load_words(0, w0, w1);
    store_rows(0, 0, w0, w1);
    __syncthreads();
  }
  v16i_t acc0, acc1;
  auto mfma_phase = [&](int buf) {
    acc0 = (v16i_t){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc1 = (v16i_t){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 8; s++) {
      const v4i_t a0 = *(const v4i_t*)&tile[buf][c * MM_ROW + s * 32 + h * 16];
      const v4i_t a1 = *(const v4i_t*)&tile[buf][(c + 32) * MM_ROW + s * 32 + h * 16];
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bq[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bq[s], acc1, 0, 0, 0);
    }
  };
  // accumulator register g*4+j of lane (c, h) belongs to database row 8g + 4h + j of its MFMA tile
  auto key_phase = [&](int rk_slot) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint4 rk = *(const uint4*)&rowkey[rk_slot][32 * half + 8 * g + 4 * h];
        const uint32_t rks[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int av = half ? acc1[4 * g + j] : acc0[4 * g + j];
          const uint32_t key = ((uint32_t)av << KEY_SHIFT) + rks[j];
          if (half) {
            skB = umed3(bB, key, skB);
            bB = min(bB, key);
          } else {
            sk = umed3(b, key, sk);
            b = min(b, key);
          }
        }
      }
    }
  };
  const bool late = STAGGER && wave >= MM_WAVES / 2;  // wave-uniform
  for (int st = 0; st < n_st; st++) {
    const int buf = st & 1;
    uint32_t n0 = 0, n1 = 0;
    if (st + 1 < n_st) load_words(st + 1, n0, n1);
    if (late) {
      if (st > 0) key_phase((st - 1) % 3);
      mfma_phase(buf);
    } else {
      mfma_phase(buf);
      key_phase(st % 3);
    }
    if (st + 1 < n_st) store_rows(buf ^ 1, st + 1, n0, n1);
    __syncthreads();
  }
  if (late && n_st > 0) key_phase((n_st - 1) % 3);
  // merge the two trackers (disjoint database rows): second = min(max(b, bB), sk, skB)
  sk = min(umed3(b, bB, sk), skB);
  b = min(b, bB);
  // merge the two lane halves of a query column (disjoint database rows)
  const uint32_t b2 = (uint32_t)__shfl_xor((int)b, 32), s2 = (uint32_t)__shfl_xor((int)sk, 32);
  sk = min(umed3(b, b2, sk), s2);
  b = min(b, b2);
  if (h == 0 && qc < n_q) {
    // back to (distance << KEY_SHIFT) | m; anything that is not a real row becomes KEY_INIT
    const uint32_t fix = (uint32_t)(pq - 256) << KEY_SHIFT;  // modular arithmetic: distance = accumulator + 256 + (|q| - 256)
    const size_t o = ((size_t)pair * 2 + dir) * F + qc;
    best_key[o] = (b >> KEY_SHIFT) >= 1023u ? KEY_INIT : b + fix;
    second_key[o] = (sk >> KEY_SHIFT) >= 1023u ? KEY_INIT : sk + fix;
  }
}

// ---------------------------------------------------------------------------------------------
// Block-scaled FP4 variant (the default for database sets of <= 2048 descriptors).  Same identity, same tiling, but
// the bits travel as FP4 (e2m1) elements through v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales:
//   database bit 0 / 1 -> 0.0 / 1.0 (nibbles 0x0 / 0x2), query bit 0 / 1 -> +1.0 / -1.0 (0x2 / 0xA);
//   the f32 accumulator |d| - 2<q, d> is an exact small integer.  One instruction covers K = 64 bits in the cycles
//   the int8 form needs for K = 32 (tools/probes/fp4_mfma_probe.hip: exact on random data, 48 vs 52 ticks per
//   dependent instruction), so the matrix-pipe time per tile halves, and so do the LDS bytes per expanded descriptor
//   (128 B) and the registers of the query fragment.
//   Keys stay lane-local and cost no instruction of their own: an accumulator STARTS at 512 + row / 2048 (the C operand
//   of a tile's first instruction: sixteen resident registers holding the row inside the tile, or, in the last super
//   tile, keys from LDS with the full row index and a pad value), so after the four instructions of a tile it holds
//   512 + |d| - 2<q, d> + row / 2048 -- an exact positive f32 in [256, 769) with 11 fraction bits (21 significant bits),
//   whose bit pattern orders like (distance, row).  The two smallest of a tile's 16 keys come out of a two-level tree of
//   v_min3 / v_med3 on the bits (20 instructions), the tile's first row is added to those two, three more instructions
//   fold them into the lane's (best, second): 1.75 instructions per distance (DESIGN.md 8.3.1 has the measurements that
//   led here: the tile loop runs at the vector issue rate).  Converted back to (distance << KEY_SHIFT) | m at the end;
//   m < 2048 is what limits this variant to 2048 database descriptors.
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
#define MX_ROW 144          // LDS bytes per expanded descriptor: 128 + 16 pad
#define MX_KEY_SCALE 2048.0f
#define MX_KEY_BIAS 512.0f   // keeps every accumulator in [256, 769): one sign, no zero, 2^-14 or finer spacing
#define MX_PAD_KEY_F 4096.0f  // rows past the end of the database (all-zero rows: the accumulator stays here): above every real key

//
// Round 3: the distance matrix is no longer computed twice.  The reference scans forward once and checks the reverse
// direction only for rows that pass threshold + ratio (keypoints.h:355-362, :282-312), so
//   launch 1 (REVERSE = false): queries = set a, database = set b -> best / second per row i          (n_a x n_b distances)
//   match_select_kernel:        one workgroup per pair lists the columns j = best(i) of the rows that passed (ascending j,
//                               <= n_b of them; ~a third of the set on the benchmark frames)
//   launch 2 (REVERSE = true):  the SAME tile loop with those columns as queries against set a    (~0.35 n_a x n_b distances)
// Columns nobody points at keep KEY_INIT-free garbage from an earlier launch -- match_finalize only reads bk1[best(i)] of
// passing rows, exactly the entries launch 2 wrote.  Both grids are 1-D and XCD-aware: the hardware deals workgroups
// round-robin to the 8 XCDs, so block id b serves pair 8 (b / 8 / blocks_per_pair) + b % 8 -- all workgroups of a pair
// share one XCD's L2 and the pair's descriptors are fetched from HBM once instead of once per XCD.
// Between the two passes: one workgroup per pair lists the columns j = best(i) of the rows that pass threshold + ratio,
// ascending and without repeats, into the (not yet written) match buffer of the pair; the count goes to match_count.
// (First version: every workgroup of the reverse launch rebuilt this list in LDS -- six times per pair, four of them
// only to find out that they have nothing to do.)
__global__ __launch_bounds__(512) void match_select_kernel(const int32_t* __restrict__ kp_count, const int32_t* __restrict__ pair_slots,
                                                           const uint32_t* __restrict__ best_key,
                                                           const uint32_t* __restrict__ second_key, int F, int threshold,
                                                           double dist_2_best, int32_t* __restrict__ sel_list,
                                                           int32_t* __restrict__ sel_count) {
  __shared__ unsigned char flag[2048];
  __shared__ int sel_wave[8];
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_a = kp_count[pair_slots[2 * pair]], n_b = kp_count[pair_slots[2 * pair + 1]];
  const uint32_t* bk0 = best_key + ((size_t)pair * 2) * F;
  const uint32_t* sk0 = second_key + ((size_t)pair * 2) * F;
  for (int j = tid; j < 2048; j += 512) flag[j] = 0;
  __syncthreads();
  for (int i = tid; i < n_a; i += 512) {
    const uint32_t bk = bk0[i];
    const int d1 = (int)(bk >> KEY_SHIFT), d2 = (int)(sk0[i] >> KEY_SHIFT);
    if (n_b > 0 && d1 < threshold && !((double)d2 < (double)d1 * dist_2_best)) flag[bk & ((1u << KEY_SHIFT) - 1)] = 1;
  }
  __syncthreads();
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) cnt += flag[tid * 4 + u];
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) sel_wave[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    if (w < wave) base += sel_wave[w];
    total += sel_wave[w];
  }
  int pos = base + incl - cnt;
  int32_t* out = sel_list + (size_t)pair * 2 * F;
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (flag[tid * 4 + u]) out[pos++] = tid * 4 + u;
  if (tid == 0) sel_count[pair] = total;
}

// Occupancy target: THREE workgroups per compute unit (6 waves per SIMD = 80 VGPRs, which the kernel fills without
// spilling -- a spilled register per lane is 6 MB of scratch stores per forward launch, and they reach HBM) instead of the
// two the compiler's own choice allows: the tile loop is vector-issue-bound and a third workgroup keeps the issue port
// busy across the other two's barriers.  Measured around it: 5 waves per SIMD 0.278 ms against 0.256 (first pass of
// round 3), 7 (72 VGPRs, 10 spills) 0.230-0.234 against 0.222-0.227, 8 (64 VGPRs, 16 spills) 0.278-0.282.
// -DMX_TIMING=1 builds the in-kernel clocks read by `MX_TIMING=1 tools/match_probe.py` (phase times of a workgroup's
// step, the clock the chip holds, the timeline of a forward launch's workgroups).
#ifndef MX_WAVES_PER_EU
#define MX_WAVES_PER_EU 6
#endif
#ifndef MX_TIMING
#define MX_TIMING 0
#endif
#define MX_OCC __attribute__((amdgpu_waves_per_eu(MX_WAVES_PER_EU, MX_WAVES_PER_EU)))
#if MX_TIMING
__device__ long long mx_dbg[4 * 16];
__device__ long long mx_tl[4 * 3072];
extern "C" int vsl_mx_timing(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx_dbg), sizeof(mx_dbg)); }
extern "C" int vsl_mx_timeline(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mx_tl), sizeof(mx_tl)); }
#endif
template <bool REVERSE>
__global__ __launch_bounds__(64 * MM_WAVES) MX_OCC void hamming_mx_kernel(const uint64_t* __restrict__ desc,
                                                                   const int32_t* __restrict__ kp_count,
                                                                   const int32_t* __restrict__ pair_slots,
                                                                   uint32_t* __restrict__ best_key,
                                                                   uint32_t* __restrict__ second_key, int F, int n_pairs,
                                                                   int blocks_per_pair, const int32_t* __restrict__ sel_list,
                                                                   const int32_t* __restrict__ sel_count, int both_dirs) {
  __shared__ __align__(16) unsigned char tile[2][64 * MX_ROW];
  __shared__ __align__(16) float rowkey[64];  // last super tile only: 512 + m / 2048, or MX_PAD_KEY_F past the end
  __shared__ uint32_t lut[256];                  // byte -> eight FP4 nibbles (bit j -> nibble j: 0x0 / 0x2)
#if MX_TIMING
  const long long wEntry = wall_clock64();
#endif
  const int xj = (int)(blockIdx.x >> 3);
  const int pair = (xj / blocks_per_pair) * 8 + (int)(blockIdx.x & 7u);
  if (pair >= n_pairs) return;  // the pair count is padded to a multiple of 8
  int blk = xj % blocks_per_pair;
  // both_dirs (REVERSE = false only; launches of a few pairs, where three dependent launches cost more than the second
  // full matrix): the second half of a pair's blocks runs the b -> a direction in the same launch, like rounds 1-2
  int dir = REVERSE ? 1 : 0;
  if (!REVERSE && both_dirs) {
    const int half = blocks_per_pair >> 1;
    dir = blk >= half ? 1 : 0;
    blk -= dir * half;
  }
  const int slot_q = pair_slots[2 * pair + dir];
  const int slot_d = pair_slots[2 * pair + 1 - dir];
  int n_q = kp_count[slot_q];
  const int n_d = kp_count[slot_d];
  const int q0 = blk * (32 * MM_WAVES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  // REVERSE: the queries are the columns some passing row points at, listed in ascending order by match_select_kernel
  const int32_t* __restrict__ sel = sel_list + (size_t)pair * 2 * F;
  if (REVERSE) n_q = sel_count[pair];
  if (q0 >= n_q) return;  // workgroup-uniform
  const int qc = q0 + wave * 32 + c;
  const int qrow = REVERSE ? sel[qc < n_q ? qc : 0] : (qc < n_q ? qc : 0);
  const uint32_t* __restrict__ qd = (const uint32_t*)(desc + ((size_t)slot_q * F + qrow) * 4);
  const uint32_t* __restrict__ dbase = (const uint32_t*)(desc + (size_t)slot_d * F * 4);
  if (tid < 256) {
    uint32_t x = (uint32_t)tid;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    lut[tid] = x << 1;
  }
  __syncthreads();
  auto spread = [&](uint32_t w) -> v4i_t {
    v4i_t o;
    o.x = (int)lut[w & 255u];
    o.y = (int)lut[(w >> 8) & 255u];
    o.z = (int)lut[(w >> 16) & 255u];
    o.w = (int)lut[w >> 24];
    return o;
  };
  // query fragment of MFMA s (bits 64 s .. 64 s + 63): lane half h holds word 2 s + h; nibble 0x2 | (bit << 3).
  // An FP4 operand is the first four registers of the instruction's eight-register tuple; the other four are never
  // read, so they stay undefined and the allocator may put anything there (written-out zeros cost 16 registers).
  auto fp4_operand = [](v4i_t t) -> v8i_t { return __builtin_shufflevector(t, t, 0, 1, 2, 3, -1, -1, -1, -1); };
  v8i_t bq[4];
  int pq = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const uint32_t w0 = qd[2 * s], w1 = qd[2 * s + 1];
    pq += __builtin_popcount(w0) + __builtin_popcount(w1);
    const v4i_t e = spread(h ? w1 : w0);
    bq[s] = fp4_operand((v4i_t){(int)(((uint32_t)e.x << 2) | 0x22222222u), (int)(((uint32_t)e.y << 2) | 0x22222222u),
                                (int)(((uint32_t)e.z << 2) | 0x22222222u), (int)(((uint32_t)e.w << 2) | 0x22222222u)});
  }
  // tile fill: thread t expands word (t & 7) of database row (t >> 3) of the 64-row super tile
  const bool filler = tid < 512;
  const int frow = tid >> 3, fword = tid & 7;
  const int n_st = (n_d + 63) / 64;
  auto load_word = [&](int st) -> uint32_t {
    const int r0 = st * 64 + frow;
    return (filler && r0 < n_d) ? dbase[(size_t)r0 * 8 + fword] : 0u;
  };
  auto store_row = [&](int buf, int st, uint32_t wd) {
    if (!filler) return;
    *(v4i_t*)&tile[buf][frow * MX_ROW + fword * 16] = spread(wd);
    // only the last super tile can hold rows past the end of the database; its accumulators start from these keys
    const int m = st * 64 + frow;
    if (fword == 0 && st == n_st - 1) rowkey[frow] = m < n_d ? (MX_KEY_BIAS + (float)m * (1.0f / MX_KEY_SCALE)) : MX_PAD_KEY_F;
  };
  uint32_t b = 0xFFFFFFFFu, sk = 0xFFFFFFFFu;
  // (B <= S) <- the two smallest of {B, S, k1, k2, k3}: lo <= mid are the triple's two smallest; the smallest of all is
  // min(B, lo), the runner-up min(max(B, lo), S, mid) = med3(B, lo, min(S, mid)) because min(S, mid) >= min(B, lo)
  auto fold3 = [](uint32_t& B, uint32_t& S, uint32_t k1, uint32_t k2, uint32_t k3) {
    const uint32_t lo = min(min(k1, k2), k3), mid = umed3(k1, k2, k3);
    S = umed3(B, lo, min(S, mid));
    B = min(B, lo);
  };
  // the two smallest (L <= M) of the 16 keys of one accumulator tile, in 20 instructions: the minimum and the median of
  // five triples (10); the two smallest of the five minima and the 16th key (7); the runner-up of the tile is either the
  // runner-up of those six or the median of the winner's triple -- and every other median is above its own minimum, so
  // min(runner-up of the six, smallest median) is the same thing (2 + 1)
  auto tile_best2 = [&](uint32_t& L, uint32_t& M, const v16f_t& a) {
    uint32_t lo[5], mid[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const uint32_t k1 = __float_as_uint(a[3 * t]), k2 = __float_as_uint(a[3 * t + 1]), k3 = __float_as_uint(a[3 * t + 2]);
      lo[t] = min(min(k1, k2), k3);
      // the median through the float builtin, not the inline-assembly umed3: these operands are matrix-core results,
      // and the wait states a vector read of them needs are inserted by the compiler's hazard recogniser, which does not
      // look inside inline assembly (all keys here are ordinary positive floats: same order as their bit patterns)
      mid[t] = __float_as_uint(__builtin_amdgcn_fmed3f(a[3 * t], a[3 * t + 1], a[3 * t + 2]));
    }
    L = min(min(lo[0], lo[1]), lo[2]);
    M = umed3(lo[0], lo[1], lo[2]);
    fold3(L, M, lo[3], lo[4], __float_as_uint(a[15]));
    M = min(M, min(min(min(mid[0], mid[1]), mid[2]), min(mid[3], mid[4])));
  };
  // accumulator register 4 g + j of a 32 x 32 tile belongs to database row 8 g + 4 h + j of the tile.  Every tile but
  // the last starts from the SAME sixteen registers, 512 + (8 g + 4 h + j) / 2048 -- the matrix instruction reads them
  // as C and writes the accumulator elsewhere, so starting a tile costs nothing -- and the tile's first row
  // (64 st + 32 half) / 2048 is added to the tile's two smallest keys only (exact: all of it fits 21 bits).
  v16f_t start;
#pragma unroll
  for (int i = 0; i < 16; i++) start[i] = MX_KEY_BIAS + (float)(8 * (i >> 2) + 4 * h + (i & 3)) * (1.0f / MX_KEY_SCALE);
  store_row(0, 0, load_word(0));
  __syncthreads();
  float tile_first = 0.0f;  // (64 st) / 2048
  // a wave whose 32 query columns all lie past the end only helps to fill the tiles (the last workgroup of a reverse
  // pass is mostly such waves: reverse launch 68 -> 65 us)
  const bool idle = __builtin_amdgcn_readfirstlane(q0 + wave * 32) >= n_q;
#if MX_TIMING
  long long tA = 0, tB = 0, tC = 0;
  const long long w0 = wall_clock64();
#endif
  for (int st = 0; st < n_st; st++) {
#if MX_TIMING
    const long long s0 = __builtin_amdgcn_s_memtime();
#endif
    const int buf = st & 1;
    uint32_t nw = 0;
    if (st + 1 < n_st) nw = load_word(st + 1);
    // the two 32-row tiles of the super tile one after the other through ONE set of accumulators (a dependent chain of
    // these instructions issues as fast as two interleaved ones, tools/probes/mfma_valu_coissue.hip; the second set
    // cost 16 registers): while this wave folds a tile, the matrix pipe serves the other waves of the SIMD
    // (requesting all four operand quads of a tile ahead of its instructions bought 1 % and cost 12 registers: ten
    // spilled registers per lane, i.e. +100 MB of scratch traffic per launch -- the quads are read one instruction ahead)
#define MX_QUAD(half, s) fp4_operand(*(const v4i_t*)&tile[buf][(c + 32 * (half)) * MX_ROW + (s) * 32 + h * 16])
    if (!idle)
#pragma unroll
    for (int half = 0; half < 2; half++) {
      v16f_t acc;
      if (st + 1 < n_st) {
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(MX_QUAD(half, 0), bq[0], start, 4, 4, 0, 127, 0, 127);
      } else {
        // last super tile: start from the keys in LDS (full row index, or the pad key for rows past the end)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const float4 r = *(const float4*)&rowkey[32 * half + 8 * g + 4 * h];
          acc[4 * g] = r.x, acc[4 * g + 1] = r.y, acc[4 * g + 2] = r.z, acc[4 * g + 3] = r.w;
        }
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(MX_QUAD(half, 0), bq[0], acc, 4, 4, 0, 127, 0, 127);
      }
#pragma unroll
      for (int s = 1; s < 4; s++) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(MX_QUAD(half, s), bq[s], acc, 4, 4, 0, 127, 0, 127);
      uint32_t l, m;
      tile_best2(l, m, acc);
      if (st + 1 < n_st) {
        const float first = tile_first + (float)(32 * half) / MX_KEY_SCALE;
        l = __float_as_uint(__uint_as_float(l) + first), m = __float_as_uint(__uint_as_float(m) + first);
      }
      // (l, m) and (b, sk) are sorted pairs over disjoint rows
      sk = umed3(b, l, min(sk, m));
      b = min(b, l);
    }
    tile_first += 64.0f / MX_KEY_SCALE;
#if MX_TIMING
    asm volatile("" ::"v"(b), "v"(sk));
    const long long s1 = __builtin_amdgcn_s_memtime();
#endif
    if (st + 1 < n_st) store_row(buf ^ 1, st + 1, nw);
#if MX_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long s2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#if MX_TIMING
    const long long s3 = __builtin_amdgcn_s_memtime();
    tA += s1 - s0, tB += s2 - s1, tC += s3 - s2;
#endif
  }
#if MX_TIMING
  if (!REVERSE && tid == 0 && blockIdx.x < 3072) {
    mx_tl[4 * blockIdx.x] = wEntry, mx_tl[4 * blockIdx.x + 1] = w0, mx_tl[4 * blockIdx.x + 2] = wall_clock64();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    mx_tl[4 * blockIdx.x + 3] = hw;
  }
  if (!REVERSE && blockIdx.x == 1000 && lane == 0) {
    mx_dbg[4 * wave] = tA, mx_dbg[4 * wave + 1] = tB, mx_dbg[4 * wave + 2] = tC, mx_dbg[4 * wave + 3] = wall_clock64() - w0;
  }
#endif
  // merge the two lane halves of a query column (disjoint database rows)
  const uint32_t b2 = (uint32_t)__shfl_xor((int)b, 32), s2 = (uint32_t)__shfl_xor((int)sk, 32);
  sk = min(umed3(b, b2, sk), s2);
  b = min(b, b2);
  if (h == 0 && qc < n_q) {
    // float key -> (distance << KEY_SHIFT) | m; anything that is not a real row becomes KEY_INIT
    auto unpack = [&](uint32_t kb) -> uint32_t {
      if (kb >= __float_as_uint(MX_PAD_KEY_F)) return KEY_INIT;  // padded row or untouched tracker
      const int ki = (int)(__uint_as_float(kb) * MX_KEY_SCALE);  // (512 + |d| - 2<q, d>) * 2048 + m, exact
      return ((uint32_t)((ki >> 11) + pq - (int)MX_KEY_BIAS) << KEY_SHIFT) | (uint32_t)(ki & 2047);
    };
    const size_t o = ((size_t)pair * 2 + dir) * F + qrow;
    best_key[o] = unpack(b);
    second_key[o] = unpack(sk);
  }
}

// Ratio test + cross-check + ordered emit.  One workgroup per pair.
__global__ __launch_bounds__(1024) void match_finalize_kernel(
    const int32_t* __restrict__ kp_count, const int32_t* __restrict__ pair_slots,
    const uint32_t* __restrict__ best_key, const uint32_t* __restrict__ second_key,
    int32_t* __restrict__ matches, int32_t* __restrict__ match_count, int F, int threshold,
    double dist_2_best) {
  const int pair = blockIdx.x;
  const int n_a = kp_count[pair_slots[2 * pair]];
  const int n_b = kp_count[pair_slots[2 * pair + 1]];
  const uint32_t* bk0 = best_key + ((size_t)pair * 2) * F;
  const uint32_t* sk0 = second_key + ((size_t)pair * 2) * F;
  const uint32_t* bk1 = bk0 + F;
  const uint32_t* sk1 = sk0 + F;
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < n_a; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    bool ok = false;
    int jbest = 0;
    if (i < n_a && n_b > 0) {
      const uint32_t b = bk0[i];
      const int d1 = (int)(b >> KEY_SHIFT);
      const int d2 = (int)(sk0[i] >> KEY_SHIFT);
      jbest = (int)(b & ((1u << KEY_SHIFT) - 1));
      if (d1 < threshold && !((double)d2 < (double)d1 * dist_2_best)) {
        const uint32_t rb = bk1[jbest];
        const int e1 = (int)(rb >> KEY_SHIFT);
        const int e2 = (int)(sk1[jbest] >> KEY_SHIFT);
        const int ri = (int)(rb & ((1u << KEY_SHIFT) - 1));
        ok = (e1 < threshold) && !((double)e2 < (double)e1 * dist_2_best) && ri == i;
      }
    }
    const unsigned long long m = __ballot(ok);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    if (ok) {
      matches[((size_t)pair * F + off + within) * 2] = i;
      matches[((size_t)pair * F + off + within) * 2 + 1] = jbest;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; w++) t += wave_tot[w];
      base_s += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) match_count[pair] = base_s;
}

int vsl_launch_match(vsl_ctx* ctx, vsl_frames* f, int n_pairs, int threshold, double dist_2_best, int db_bound) {
  // db_bound: an upper bound of the descriptors per set in this launch (the FP4 kernel's keys hold 11 index bits)
  if (n_pairs <= 0) return VSL_OK;
  constexpr int WAVES = VSL_MATCH_WAVES;
  {
    VslStage st(ctx, VSL_STAGE_MATCH);
    // (a row-split variant -- waves share the column stream, no merge -- was measured slower: 0.288 vs
    // 0.255 ms per 128 pairs; the column split keeps more, shorter waves in flight)
    if (ctx->match_use_valu) {
      dim3 grid((f->F + 63) / 64, 2, n_pairs);
      hipLaunchKernelGGL(hamming_best2_kernel<WAVES>, grid, dim3(64 * WAVES), 0, ctx->stream, f->kp_desc,
                         f->kp_count, f->pair_slots, f->best_key, f->second_key, f->F);
    } else {
      dim3 grid((f->F + 32 * MM_WAVES - 1) / (32 * MM_WAVES), 2, n_pairs);
      if (db_bound <= 2048 && !ctx->match_use_i8) {
        const int bpp = (f->F + 32 * MM_WAVES - 1) / (32 * MM_WAVES);
        if (n_pairs < 8 && !ctx->match_two_pass) {
          // a handful of pairs (the per-keyframe stereo match, sim3 / relocalisation calls): ONE launch with both full
          // directions -- the launch is latency-bound (a pair is 12 workgroups on 256 compute units), two more dependent
          // launches cost more than the redundant distances (28 vs 50 us of device time for one pair)
          const dim3 grid2((unsigned)(((n_pairs + 7) / 8) * 8 * 2 * bpp));
          hipLaunchKernelGGL(hamming_mx_kernel<false>, grid2, dim3(64 * MM_WAVES), 0, ctx->stream, f->kp_desc, f->kp_count,
                             f->pair_slots, f->best_key, f->second_key, f->F, n_pairs, 2 * bpp, (const int32_t*)f->matches,
                             (const int32_t*)f->match_count, 1);
        } else {
        const dim3 grid1((unsigned)(((n_pairs + 7) / 8) * 8 * bpp));
        hipLaunchKernelGGL(hamming_mx_kernel<false>, grid1, dim3(64 * MM_WAVES), 0, ctx->stream, f->kp_desc, f->kp_count,
                           f->pair_slots, f->best_key, f->second_key, f->F, n_pairs, bpp, (const int32_t*)f->matches,
                           (const int32_t*)f->match_count, 0);
        hipLaunchKernelGGL(match_select_kernel, dim3(n_pairs), dim3(512), 0, ctx->stream, f->kp_count, f->pair_slots, f->best_key,
                           f->second_key, f->F, threshold, dist_2_best, f->matches, f->match_count);
        hipLaunchKernelGGL(hamming_mx_kernel<true>, grid1, dim3(64 * MM_WAVES), 0, ctx->stream, f->kp_desc, f->kp_count,
                           f->pair_slots, f->best_key, f->second_key, f->F, n_pairs, bpp, (const int32_t*)f->matches,
                           (const int32_t*)f->match_count, 0);
        }
      } else if (ctx->match_no_stagger)
        hipLaunchKernelGGL(hamming_mfma_kernel<false>, grid, dim3(64 * MM_WAVES), 0, ctx->stream, f->kp_desc, f->kp_count,
                           f->pair_slots, f->best_key, f->second_key, f->F);
      else
        hipLaunchKernelGGL(hamming_mfma_kernel<true>, grid, dim3(64 * MM_WAVES), 0, ctx->stream, f->kp_desc, f->kp_count,
                           f->pair_slots, f->best_key, f->second_key, f->F);
    }
    VSL_CHECK_LAUNCH(ctx);
  }
  {
    VslStage st(ctx, VSL_STAGE_MATCH_FIN);
    hipLaunchKernelGGL(match_finalize_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, f->kp_count,
                       f->pair_slots, f->best_key, f->second_key, f->matches, f->match_count, f->F, threshold,
                       dist_2_best);
    VSL_CHECK_LAUNCH(ctx);
  }
  return VSL_OK;
}

// The pair list of a steady-state pipeline never changes (slot 2k <-> 2k+1), so it is uploaded only
// when it differs from the cached copy (synchronously: the caller's array may be reused at once).
int vsl_set_pairs(vsl_ctx* ctx, vsl_frames* f, const int32_t* slot_pairs, int n_pairs) {
  const size_t n = 2 * (size_t)n_pairs;
  if (f->pair_cache.size() >= n && memcmp(f->pair_cache.data(), slot_pairs, n * sizeof(int32_t)) == 0) return VSL_OK;
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  VSL_HIP(ctx, hipMemcpy(f->pair_slots, slot_pairs, n * sizeof(int32_t), hipMemcpyHostToDevice));
  f->pair_cache.assign(slot_pairs, slot_pairs + n);
  return VSL_OK;
}

extern "C" int vsl_frames_match(vsl_ctx* ctx, vsl_frames* f, const int32_t* slot_pairs, int n_pairs, int threshold,
                                double dist_2_best) {
  if (!ctx || !f || (n_pairs > 0 && !slot_pairs) || n_pairs < 0 || n_pairs > f->max_pairs)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_match: bad arguments (n_pairs=%d, max_pairs=%d)", n_pairs, f ? f->max_pairs : -1);
  for (int i = 0; i < 2 * n_pairs; i++)
    if (slot_pairs[i] < 0 || slot_pairs[i] >= f->max_images)
      return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_match: slot %d out of range", slot_pairs[i]);
  if (n_pairs == 0) return VSL_OK;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc = vsl_set_pairs(ctx, f, slot_pairs, n_pairs);
  if (rc) return rc;
  return vsl_launch_match(ctx, f, n_pairs, threshold, dist_2_best, f->F);
}

extern "C" int vsl_match_descriptors(vsl_ctx* ctx, const uint64_t* d1, int n1, const uint64_t* d2, int n2,
                                     int threshold, double dist_2_best, int32_t* pairs, int* n_out) {
  if (!ctx || !n_out || n1 < 0 || n2 < 0 || (n1 > 0 && !d1) || (n2 > 0 && !d2))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_match_descriptors: bad arguments");
  *n_out = 0;
  if (n1 == 0 || n2 == 0) return VSL_OK;  // best stays 256 >= threshold for every row
  if (n1 >= (1 << KEY_SHIFT) || n2 >= (1 << KEY_SHIFT))
    return vsl_fail(ctx, VSL_ERR_CAPACITY, "vsl_match_descriptors: at most %d descriptors per set", (1 << KEY_SHIFT) - 1);
  if (!pairs) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_match_descriptors: pairs is null");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_frames* f = nullptr;
  const int feat = n1 > n2 ? n1 : n2;
  int rc = vsl_ctx_scratch_frames(ctx, ctx->scratch ? ctx->scratch_w : 64, ctx->scratch ? ctx->scratch_h : 64,
                                  feat > (ctx->scratch ? ctx->scratch_feat : 0) ? feat : ctx->scratch_feat, &f);
  if (rc) return rc;
  const int32_t counts[2] = {n1, n2};
  const int32_t slots[2] = {0, 1};
  VSL_HIP(ctx, hipMemcpyAsync(f->kp_desc, d1, 32 * (size_t)n1, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(f->kp_desc + (size_t)f->F * 4, d2, 32 * (size_t)n2, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(f->kp_count, counts, sizeof(counts), hipMemcpyHostToDevice, ctx->stream));
  rc = vsl_set_pairs(ctx, f, slots, 1);
  if (rc) return rc;
  rc = vsl_launch_match(ctx, f, 1, threshold, dist_2_best, n1 > n2 ? n1 : n2);
  if (rc) return rc;
  return vsl_frames_download_matches(ctx, f, 0, n1 < n2 ? n1 : n2, pairs, n_out);
}
