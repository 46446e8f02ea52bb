// bow.hip -- K8 (vocabulary-tree transform) and K9 (batched L1 score) of the DBoW2 path the reference
// uses to score loop-closure candidates.
//
// Replaces, for the ORB vocabulary (k-ary tree, Hamming distance, TF-IDF weights, L1 scoring):
//   ORBVocabulary::loadFromTextFile  thirdparty/DBoW2_ORBSLAM/DBoW2/TemplatedVocabulary.h:1338-1424
//   ORBVocabulary::transform         TemplatedVocabulary.h:1127-1194 and :1218-1259, FORB::distance FORB.cpp:81-101,
//                                    BowVector::addWeight / normalize BowVector.cpp:34-84,
//                                    FeatureVector::addFeature FeatureVector.cpp:30-44
//   ORBVocabulary::score             TemplatedVocabulary.h:1199-1203 -> L1Scoring::score ScoringObject.cpp:23-68
// Call sites in the reference: include/visnav/keypoints.h:253, include/visnav/loop_closure_utils.h:119, :201,
// include/visnav/tracking.h:208.
//
// Bit-exactness: distances are integers; the only floating-point work is sums of doubles, and those
// are performed in the reference's ORDER (weights of one word added one feature at a time; the L1 norm
// and the score accumulated in ascending word id), so the results are bit-identical, not just close.
//
// Documented deviation (same as the oracle): blank lines of the vocabulary file are skipped instead of
// being parsed as a node with an uninitialised descriptor (TemplatedVocabulary.h:1380 `while(!f.eof())`).
//
// Round 3 layout and kernels (measured at the reference's vocabulary shape k = 10, L = 6 -- 1,111,111 nodes):
//   * the tree is stored by CHILD SLOT (the position of a node in its parent's child list, CSR order): the k children of
//     a node are k consecutive 32-byte rows, and a slot carries (first child slot, child count) of ITS node, so one
//     level of the descent is ONE round trip (k rows + k slot records requested together), no dependent index load;
//   * K8a: a group of 16 lanes (32 / 64 for wider trees) per descriptor, lane = child, minimum by DPP row rotations;
//   * K8b: both sorts in one bitonic network whose steps inside a wavefront's 128 keys need no workgroup barrier,
//     wave-level scans, the ordered L1 norm from 16-byte LDS reads;
//   * K9: the query lives in LDS, a wavefront per candidate, four 64-word chunks in flight per wavefront (branch-free
//     binary searches), the matched terms summed in word order through ballots + v_readlane (no LDS round trip);
//   * a device-resident database of BowVectors (vsl_bowdb_*) so that scoring M keyframes moves no candidate bytes
//     over PCIe.
#include <algorithm>
#include <cstdlib>
#include <string>

#include "vsl_common.h"

struct vsl_voc {
  int device = 0;
  int k = 0, L = 0, n_nodes = 0, n_words = 0;
  int n_slots = 0, root_nc = 0, group = 16;  // group: lanes per descriptor in the descent (>= widest child list)
  // device arrays by CHILD SLOT (slot s = position in the CSR child list; the children of a node are consecutive slots)
  uint4* sdesc = nullptr;      // [n_slots][2]  32-byte descriptor rows (byte order as in the file)
  int2* sinfo = nullptr;       // [n_slots]     (first child slot, number of children) of the node in this slot
  uint32_t* snode = nullptr;   // [n_slots]     node id
  uint32_t* sword = nullptr;   // [n_slots]     word id (leaves)
  double* sweight = nullptr;   // [n_slots]     node weight
};

// Device-resident database of BowVectors (one per keyframe): CSR, grown geometrically.
struct vsl_bowdb {
  int device = 0;
  uint32_t* ids = nullptr;
  double* vals = nullptr;
  int64_t* off = nullptr;      // [cap_vecs + 1]
  int64_t cap_entries = 0, n_entries = 0;
  int cap_vecs = 0, n_vecs = 0;
  int max_nnz = 0;             // longest stored vector (the workgroup form of the scoring kernel stages a whole candidate)
};

namespace {

#define BOW_MAX_N 8192
#define BOW_SCORE_U 4  // 64-word chunks of a candidate in flight per wavefront (scoring kernel)
#define BOW_Q_LDS_MAX 8192  // query entries the LDS scoring kernel holds (96 KB); larger queries take the global-memory kernel

// minimum over the G lanes of a descriptor's group: DPP inside a row of 16 lanes, butterflies across rows
template <int G>
__device__ __forceinline__ uint32_t group_min(uint32_t key) {
  key = min(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  key = min(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  key = min(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x141, 0xf, 0xf, false));  // row_half_mirror
  key = min(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x140, 0xf, 0xf, false));  // row_mirror
  if (G >= 32) key = min(key, (uint32_t)__shfl_xor((int)key, 16));
  if (G >= 64) key = min(key, (uint32_t)__shfl_xor((int)key, 32));
  return key;
}

// K8a: G lanes per descriptor walk the tree; lane c evaluates child c of the current node.
// Strict '<' with the first child winning ties (TemplatedVocabulary.h:1238-1249) == minimum of (distance << 8 | child).
// One level = one round trip: the child rows and the children's own (first, count) records are requested together.
template <int G>
__global__ __launch_bounds__(256) void bow_descend_kernel(const uint4* __restrict__ feat, int n, const uint4* __restrict__ sdesc,
                                                          const int2* __restrict__ sinfo, const uint32_t* __restrict__ snode,
                                                          const uint32_t* __restrict__ sword, const double* __restrict__ sweight,
                                                          int root_nc, int L, int levelsup, uint32_t* __restrict__ out_word,
                                                          double* __restrict__ out_w, uint32_t* __restrict__ out_node) {
  const int gid = (int)((blockIdx.x * 256u + threadIdx.x) / G);
  const int c = threadIdx.x & (G - 1);
  if (gid >= n) return;  // whole groups leave together (256 % G == 0)
  const uint4 d0 = feat[2 * (size_t)gid], d1 = feat[2 * (size_t)gid + 1];
  const int nid_level = L - levelsup;
  int first = 0, nc = root_nc, level = 0, best_slot = 0, nid_slot = -1;
  while (nc > 0) {  // isLeaf() == no children; uniform within the group
    ++level;
    uint32_t key = 0xFFFFFFFFu;
    int2 info = make_int2(0, 0);
    if (c < nc) {
      const size_t slot = (size_t)first + c;
      const uint4 a = sdesc[2 * slot], b = sdesc[2 * slot + 1];
      info = sinfo[slot];
      const uint32_t dist = __builtin_popcount(d0.x ^ a.x) + __builtin_popcount(d0.y ^ a.y) + __builtin_popcount(d0.z ^ a.z) +
                            __builtin_popcount(d0.w ^ a.w) + __builtin_popcount(d1.x ^ b.x) + __builtin_popcount(d1.y ^ b.y) +
                            __builtin_popcount(d1.z ^ b.z) + __builtin_popcount(d1.w ^ b.w);
      key = (dist << 8) | (uint32_t)c;
    }
    key = group_min<G>(key);
    const int cb = (int)(key & 0xFFu);
    best_slot = first + cb;
    first = __shfl(info.x, cb, G);
    nc = __shfl(info.y, cb, G);
    if (level == nid_level) nid_slot = best_slot;
  }
  if (c == 0) {
    out_word[gid] = sword[best_slot];
    out_w[gid] = sweight[best_slot];
    out_node[gid] = nid_slot >= 0 ? snode[nid_slot] : 0u;  // root when nid_level <= 0
  }
}

// Bitonic network over N = 2048 * m LDS keys, 1024 threads.  One compute unit runs the whole network, so it is written
// for few vector instructions and few LDS round trips:
//   * a step with distance j <= 64 only moves keys inside a block of 128 that one wavefront owns.  Those steps run in
//     REGISTERS: lane l holds keys l and l + 64 of the block, j = 64 is a compare-exchange inside the lane, j < 64 an
//     exchange with lane l ^ j (ds_bpermute, no memory) -- 56 of the 66 steps of 2048 keys;
//   * the 10 steps with j >= 128 go through LDS: minimum and maximum are both written back, the direction only picks the
//     two addresses (no divergent branch).
// The keys are 32 bits wide whenever (id, feature index) fits -- it does for a million words and <= 2048 features.
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int j) { return (uint32_t)__shfl_xor((int)v, j); }
__device__ __forceinline__ unsigned long long lane_xor(unsigned long long v, int j) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, j), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), j);
  return ((unsigned long long)hi << 32) | lo;
}

// steps j = jmax, jmax / 2, ..., 1 (jmax <= 64) of stage k on the block's two keys per lane; e0 = index of x0
template <class K>
__device__ __forceinline__ void bitonic_wave_steps(K& x0, K& x1, int e0, int k, int jmax) {
  const bool asc = (e0 & k) == 0;  // bit k of e0 and of e0 + 64 agree for every k != 64; k = 64 is handled below
  if (jmax == 64) {
    const K lo = x0 < x1 ? x0 : x1, hi = x0 < x1 ? x1 : x0;
    x0 = asc ? lo : hi;
    x1 = asc ? hi : lo;
    jmax = 32;
  }
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) {
    if (j > jmax) continue;
    const K y0 = lane_xor(x0, j), y1 = lane_xor(x1, j);
    // the lower element of a pair (bit j clear) keeps the minimum of an ascending pair
    const bool low = (e0 & j) == 0;
    const bool asc0 = k == 64 ? true : asc, asc1 = k == 64 ? false : asc;  // stage 64: keys 0..63 ascend, 64..127 descend
    const bool min0 = low == asc0, min1 = low == asc1;
    const K lo0 = x0 < y0 ? x0 : y0, hi0 = x0 < y0 ? y0 : x0;
    const K lo1 = x1 < y1 ? x1 : y1, hi1 = x1 < y1 ? y1 : x1;
    x0 = min0 ? lo0 : hi0;
    x1 = min1 ? lo1 : hi1;
  }
}

template <class K>
__device__ __forceinline__ void bitonic_sort(K* __restrict__ keys, int N) {
  const int lane = threadIdx.x & 63;
  const int blocks = N >> 7;  // blocks of 128 keys; wave w owns blocks w, w + 16, ...
  // stages k = 2 .. 128: entirely inside the blocks
  for (int blk = threadIdx.x >> 6; blk < blocks; blk += 16) {
    const int e0 = (blk << 7) + lane;
    K x0 = keys[e0], x1 = keys[e0 + 64];
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) bitonic_wave_steps<K>(x0, x1, e0, k, k >> 1);
    bitonic_wave_steps<K>(x0, x1, e0, 128, 64);
    keys[e0] = x0;
    keys[e0 + 64] = x1;
  }
  __syncthreads();
  for (int k = 256; k <= N; k <<= 1) {
    for (int j = k >> 1; j >= 128; j >>= 1) {
      for (int t = threadIdx.x; t < (N >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const bool asc = (i & k) == 0;
        const K a0 = keys[i], a1 = keys[l];
        keys[asc ? i : l] = a0 < a1 ? a0 : a1;
        keys[asc ? l : i] = a0 < a1 ? a1 : a0;
      }
      __syncthreads();
    }
    for (int blk = threadIdx.x >> 6; blk < blocks; blk += 16) {
      const int e0 = (blk << 7) + lane;
      K x0 = keys[e0], x1 = keys[e0 + 64];
      bitonic_wave_steps<K>(x0, x1, e0, k, 64);
      keys[e0] = x0;
      keys[e0 + 64] = x1;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// K8b: two workgroups assemble the FeatureVector (block 0) and the BowVector (block 1) of one image -- the two sorts are
// independent and one compute unit per sort is what bounds this kernel.
//   BowVector  : features with weight > 0, grouped by word (ascending id); the value of a word is its
//                weight added once per feature, in feature order (== BowVector::addWeight); L1 norm
//                accumulated in ascending word order (== BowVector::normalize), then divided.
//   FeatureVector: (node, feature) pairs sorted by node, then feature index.
// Keys: (id << shift) | feature index, all-ones = padding (sorts last); K = uint32_t when that fits, else 64 bits.
// SMALL (<= 2048 features, the reference passes <= 1500): weights and word values live in LDS too; otherwise they stay
// in global memory (the LDS belongs to the keys).
template <int NCAP, class K, bool SMALL>
__global__ __launch_bounds__(1024) void bow_assemble_kernel(const uint32_t* __restrict__ f_word,
                                                            const double* __restrict__ f_w,
                                                            const uint32_t* __restrict__ f_node, int n, int shift,
                                                            uint32_t* __restrict__ word_ids, double* __restrict__ word_vals,
                                                            int32_t* __restrict__ counts, uint32_t* __restrict__ fv_node,
                                                            uint32_t* __restrict__ fv_feat) {
  __shared__ __attribute__((aligned(16))) K keys[NCAP];
  __shared__ __attribute__((aligned(16))) double w_lds[SMALL ? NCAP : 2];     // the features' weights
  __shared__ __attribute__((aligned(16))) double vals_lds[SMALL ? NCAP : 2];  // the words' values before normalisation
  __shared__ int wsum[16];
  __shared__ int kept_s;
  __shared__ double norm_s;
  const double* w_s = SMALL ? w_lds : f_w;
  double* vals_s = SMALL ? vals_lds : word_vals;
  const int tid = threadIdx.x;
  const bool bow = blockIdx.x == 1;
  const uint32_t* __restrict__ f_id = bow ? f_word : f_node;
  const K pad = ~(K)0;
  const K imask = (((K)1) << shift) - 1;
  int N = 2048;
  while (N < n) N <<= 1;
  for (int i = tid; i < N; i += 1024) {
    const double w = i < n ? f_w[i] : 0.0;
    if (SMALL) w_lds[i] = w;
    keys[i] = w > 0.0 ? ((((K)f_id[i]) << shift) | (K)i) : pad;
  }
  if (tid == 0) kept_s = 0;
  __syncthreads();
  bitonic_sort<K>(keys, N);
  if (!bow) {
    // ---- FeatureVector: the kept entries sort in front of the padding
    for (int i = tid; i < N; i += 1024) {
      const K key = keys[i];
      if (key != pad) {
        fv_node[i] = (uint32_t)(key >> shift);
        fv_feat[i] = (uint32_t)(key & imask);
        if (i == N - 1 || keys[i + 1] == pad) counts[1] = i + 1;
      }
    }
    if (tid == 0 && keys[0] == pad) counts[1] = 0;
    return;
  }
  // ---- BowVector: a run of equal word ids = one entry; its first element computes the value.  Ordered compaction of
  // the run heads: per-thread contiguous slices, wave scans + one scan over the 16 wave totals.
  for (int i = tid; i < N; i += 1024)
    if (keys[i] != pad && (i == N - 1 || keys[i + 1] == pad)) kept_s = i + 1;
  __syncthreads();
  const int kept = kept_s;
  const int per = N >> 10;
  int heads = 0;
  for (int q = 0; q < per; q++) {
    const int i = tid * per + q;
    heads += i < kept && (i == 0 || (keys[i - 1] >> shift) != (keys[i] >> shift));
  }
  const int incl = wave_inclusive_scan(heads);
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int wave_base = 0, uniq = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const int s = wsum[w];
    if (w < (tid >> 6)) wave_base += s;
    uniq += s;
  }
  int rank = wave_base + incl - heads;
  for (int q = 0; q < per; q++) {
    const int i = tid * per + q;
    if (i >= kept) break;
    const K kw = keys[i] >> shift;
    if (i != 0 && (keys[i - 1] >> shift) == kw) continue;
    double v = w_s[(int)(keys[i] & imask)];  // insert(id, w) ...
    for (int r = i + 1; r < kept && (keys[r] >> shift) == kw; r++)
      v += w_s[(int)(keys[r] & imask)];      // ... then += w per further feature (BowVector.cpp:38-45)
    word_ids[rank] = (uint32_t)kw;
    vals_s[rank] = v;
    rank++;
  }
  if (!SMALL) __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    // ascending word order, ONE accumulator (BowVector.cpp:62-74): 1500 dependent additions -- the operands come two
    // per 16-byte LDS read, eight reads in flight
    double norm = 0.0;
    int i = 0;
    const double2* v2 = reinterpret_cast<const double2*>(vals_s);
    for (; i + 16 <= uniq; i += 16) {
      double2 r[8];
#pragma unroll
      for (int u = 0; u < 8; u++) r[u] = v2[(i >> 1) + u];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        norm += fabs(r[u].x);
        norm += fabs(r[u].y);
      }
    }
    for (; i < uniq; i++) norm += fabs(vals_s[i]);
    norm_s = norm;
    counts[0] = uniq;
  }
  __syncthreads();
  const double norm = norm_s;
  for (int i = tid; i < uniq; i += 1024) word_vals[i] = norm > 0.0 ? vals_s[i] / norm : vals_s[i];
}

// K9: L1 score (ScoringObject.cpp:23-68) of the query against candidates in CSR form; candidate m is vector
// idx[m] (or m when idx is null) of (c_ids, c_vals, off).  The query sits in LDS, padded to a power of two with
// 0xFFFFFFFF so that the lower-bound search is branch-free; a wavefront per candidate, BOW_SCORE_U 64-word chunks in flight;
// matched terms are summed in ascending word order -- the reference's order; unmatched words add nothing there --
// by walking the chunk's ballot and reading the lane's term with v_readlane.
__global__ __launch_bounds__(256) void bow_score_lds_kernel(const uint32_t* __restrict__ q_ids, const double* __restrict__ q_vals,
                                                            int q_nnz, int P, const uint32_t* __restrict__ c_ids,
                                                            const double* __restrict__ c_vals, const int64_t* __restrict__ off,
                                                            const int32_t* __restrict__ idx, int m, double* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* qv = reinterpret_cast<double*>(smem);                 // [q_nnz] (rounded up to even)
  uint32_t* qi = reinterpret_cast<uint32_t*>(qv + ((q_nnz + 1) & ~1));  // [P]
  for (int i = threadIdx.x; i < P; i += 256) qi[i] = i < q_nnz ? q_ids[i] : 0xFFFFFFFFu;
  for (int i = threadIdx.x; i < q_nnz; i += 256) qv[i] = q_vals[i];
  __syncthreads();
  const int cand = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cand >= m) return;
  const int lane = threadIdx.x & 63;
  const int vec = idx ? idx[cand] : cand;
  const int64_t a = off[vec], b = off[vec + 1];
  double score = 0.0;
  // BOW_SCORE_U x 64 words per round.  Measured at M = 100 / 10,000 candidates of ~1500 words: 4 chunks 18 / 64 us;
  // with the next round's words prefetched one round ahead 24 / 64; 8 chunks 22 / 81 (56 VGPRs) -- kept: 4, no prefetch.
  for (int64_t base = a; base < b; base += 64 * BOW_SCORE_U) {
    uint32_t id[BOW_SCORE_U];
    double wv[BOW_SCORE_U], term[BOW_SCORE_U];
    int lo[BOW_SCORE_U];
#pragma unroll
    for (int u = 0; u < BOW_SCORE_U; u++) {
      const int64_t i = base + 64 * u + lane;
      const bool in = i < b;
      id[u] = in ? c_ids[i] : 0xFFFFFFFFu;
      wv[u] = in ? c_vals[i] : 0.0;
      lo[u] = 0;
    }
    for (int step = P >> 1; step >= 1; step >>= 1) {
#pragma unroll
      for (int u = 0; u < BOW_SCORE_U; u++)
        if (qi[lo[u] + step - 1] < id[u]) lo[u] += step;
    }
    unsigned long long mask[BOW_SCORE_U];
#pragma unroll
    for (int u = 0; u < BOW_SCORE_U; u++) {
      const bool hit = id[u] != 0xFFFFFFFFu && qi[lo[u]] == id[u];
      term[u] = 0.0;
      if (hit) {
        const double vi = qv[lo[u]], wi = wv[u];
        term[u] = fabs(vi - wi) - fabs(vi) - fabs(wi);
      }
      mask[u] = __ballot(hit);
    }
#pragma unroll
    for (int u = 0; u < BOW_SCORE_U; u++) {
      unsigned long long mk = mask[u];
      const int tlo = __builtin_bit_cast(int2, term[u]).x, thi = __builtin_bit_cast(int2, term[u]).y;
      while (mk) {
        const int l = __builtin_ctzll(mk);
        mk &= mk - 1;
        const int2 t = make_int2(__builtin_amdgcn_readlane(tlo, l), __builtin_amdgcn_readlane(thi, l));
        score += __builtin_bit_cast(double, t);
      }
    }
  }
  if (lane == 0) scores[cand] = -score / 2.0;
}

// K9 for FEW candidates (m <= BOW_WG_MAX_M = 256: loop-closure / relocalisation queries score ~100 keyframes): one WORKGROUP
// per candidate instead of one wavefront.  The wave-per-candidate kernel above is latency at that size (20 us at M = 100:
// 25 workgroups on 256 compute units, a lone wave per SIMD walking six rounds of 11 dependent LDS searches); here the 1024
// threads of a workgroup take one candidate word each, ONE round of searches, the terms go to LDS and a single wavefront
// adds the matched ones in ascending word order (the reference's order, ScoringObject.cpp:23-68; a query and a
// candidate share tens to hundreds of words, so the ordered part is short).
#define BOW_WG_THREADS 1024
#define BOW_WG_TERMS 4096     // candidate words a workgroup stages
#define BOW_WG_Q_MAX 4096     // query words (48 KB + 16 KB of search keys)
#define BOW_WG_MAX_M 256      // one round of workgroups on the chip (measured: M = 100 20 -> 11 us; M = 1000 21 -> 31 us with this form)
__global__ __launch_bounds__(BOW_WG_THREADS) void bow_score_wg_kernel(const uint32_t* __restrict__ q_ids, const double* __restrict__ q_vals,
                                                                      int q_nnz, int P, const uint32_t* __restrict__ c_ids,
                                                                      const double* __restrict__ c_vals, const int64_t* __restrict__ off,
                                                                      const int32_t* __restrict__ idx, double* __restrict__ scores) {
  __shared__ double qv[BOW_WG_Q_MAX];
  __shared__ uint32_t qi[2 * BOW_WG_Q_MAX];  // padded to the power of two P <= 8192 with 0xFFFFFFFF
  __shared__ double term_s[BOW_WG_TERMS];
  __shared__ unsigned long long hit_s[BOW_WG_TERMS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vec = idx ? idx[blockIdx.x] : (int)blockIdx.x;
  const int64_t a = off[vec];
  const int len = (int)(off[vec + 1] - a);  // <= BOW_WG_TERMS (the host checks)
  // the candidate's words are requested before the query is staged: one round trip for both
  uint32_t id[BOW_WG_TERMS / BOW_WG_THREADS];
  double wv[BOW_WG_TERMS / BOW_WG_THREADS];
#pragma unroll
  for (int u = 0; u < BOW_WG_TERMS / BOW_WG_THREADS; u++) {
    const int i = u * BOW_WG_THREADS + tid;
    id[u] = i < len ? c_ids[a + i] : 0xFFFFFFFFu;
    wv[u] = i < len ? c_vals[a + i] : 0.0;
  }
  for (int i = tid; i < P; i += BOW_WG_THREADS) qi[i] = i < q_nnz ? q_ids[i] : 0xFFFFFFFFu;
  for (int i = tid; i < q_nnz; i += BOW_WG_THREADS) qv[i] = q_vals[i];
  __syncthreads();
  const int rounds = (len + BOW_WG_THREADS - 1) / BOW_WG_THREADS;
#pragma unroll
  for (int u = 0; u < BOW_WG_TERMS / BOW_WG_THREADS; u++) {
    if (u < rounds) {  // workgroup-uniform
      int lo = 0;
      for (int step = P >> 1; step >= 1; step >>= 1)
        if (qi[lo + step - 1] < id[u]) lo += step;
      const bool hit = id[u] != 0xFFFFFFFFu && qi[lo] == id[u];
      double t = 0.0;
      if (hit) {
        const double vi = qv[lo], wi = wv[u];
        t = fabs(vi - wi) - fabs(vi) - fabs(wi);
      }
      term_s[u * BOW_WG_THREADS + tid] = t;
      const unsigned long long mk = __ballot(hit);
      if (lane == 0) hit_s[u * (BOW_WG_THREADS / 64) + wave] = mk;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  double score = 0.0;
  const int chunks = (len + 63) / 64;  // <= 64: lane c holds the hit mask of chunk c, chunks without a hit are never visited
  const unsigned long long my_mask = lane < chunks ? hit_s[lane] : 0ull;
  unsigned long long live = __ballot(my_mask != 0ull);
  const int mlo = (int)(uint32_t)my_mask, mhi = (int)(uint32_t)(my_mask >> 32);
  while (live) {
    const int c = __builtin_ctzll(live);
    live &= live - 1;
    unsigned long long mk = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(mhi, c) << 32) | (uint32_t)__builtin_amdgcn_readlane(mlo, c);
    const double t = term_s[64 * c + lane];
    const int tlo = __builtin_bit_cast(int2, t).x, thi = __builtin_bit_cast(int2, t).y;
    while (mk) {
      const int l = __builtin_ctzll(mk);
      mk &= mk - 1;
      const int2 v = make_int2(__builtin_amdgcn_readlane(tlo, l), __builtin_amdgcn_readlane(thi, l));
      score += __builtin_bit_cast(double, v);
    }
  }
  if (lane == 0) scores[blockIdx.x] = -score / 2.0;
}

// Queries too large for LDS (> BOW_Q_LDS_MAX words): the query stays in global memory.
__global__ __launch_bounds__(64) void bow_score_global_kernel(const uint32_t* __restrict__ q_ids, const double* __restrict__ q_vals,
                                                              int q_nnz, const uint32_t* __restrict__ c_ids,
                                                              const double* __restrict__ c_vals, const int64_t* __restrict__ off,
                                                              const int32_t* __restrict__ idx, double* __restrict__ scores) {
  __shared__ double term[64];
  const int lane = threadIdx.x;
  const int vec = idx ? idx[blockIdx.x] : (int)blockIdx.x;
  const int64_t a = off[vec], b = off[vec + 1];
  double score = 0.0;
  for (int64_t base = a; base < b; base += 64) {
    const int64_t i = base + lane;
    double t = 0.0;
    if (i < b) {
      const uint32_t id = c_ids[i];
      int lo = 0, hi = q_nnz;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (q_ids[mid] < id) lo = mid + 1; else hi = mid;
      }
      if (lo < q_nnz && q_ids[lo] == id) {
        const double vi = q_vals[lo], wi = c_vals[i];
        t = fabs(vi - wi) - fabs(vi) - fabs(wi);
      }
    }
    term[lane] = t;
    __syncthreads();
    if (lane == 0) {
      const int cnt = (int)min((int64_t)64, b - base);
      for (int q = 0; q < cnt; q++)
        if (term[q] != 0.0) score += term[q];  // an unmatched word adds nothing in the reference either
    }
    __syncthreads();
  }
  if (lane == 0) scores[blockIdx.x] = -score / 2.0;
}

template <class T>
int to_device(vsl_ctx* ctx, T** dst, const std::vector<T>& src) {
  VSL_HIP(ctx, hipMalloc((void**)dst, sizeof(T) * (src.size() ? src.size() : 1)));
  if (!src.empty()) VSL_HIP(ctx, hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  return VSL_OK;
}

// ---- text parsing (TemplatedVocabulary.h:1338-1424 reads every line through a stringstream; the 1.1 M lines of the
// ORB vocabulary are parsed here by hand: same tokens, same "a failed extraction leaves zeros" behaviour)
struct LineParser {
  const char* p;
  const char* end;
  bool failed = false;
  void skip_blanks() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
  }
  bool next_int(long* out) {  // `ss >> int`: false (and sticky failure) when the next token is not an integer
    if (failed) return false;
    skip_blanks();
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) {
      neg = *q == '-';
      q++;
    }
    if (q >= end || *q < '0' || *q > '9') {
      failed = true;
      return false;
    }
    long v = 0;
    while (q < end && *q >= '0' && *q <= '9') {
      v = v * 10 + (*q - '0');
      if (v > (1L << 40)) v = 1L << 40;
      q++;
    }
    p = q;
    *out = neg ? -v : v;
    return true;
  }
  bool next_double(double* out) {
    if (failed) return false;
    skip_blanks();
    if (p >= end) {
      failed = true;
      return false;
    }
    char* e = nullptr;
    const double v = std::strtod(p, &e);  // the buffer is NUL-terminated and a line ends in '\n': strtod stops there
    if (e == p || e > end) {
      failed = true;
      return false;
    }
    p = e;
    *out = v;
    return true;
  }
};

}  // namespace

extern "C" int vsl_voc_destroy(vsl_voc* v) {
  if (!v) return VSL_OK;
  (void)hipSetDevice(v->device);
  void* ptrs[] = {v->sdesc, v->sinfo, v->snode, v->sword, v->sweight};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete v;
  return VSL_OK;
}

extern "C" int vsl_voc_load_text(vsl_ctx* ctx, const char* path, vsl_voc** out) {
  if (!ctx || !path || !out) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_voc_load_text: null argument");
  *out = nullptr;
  std::string text;
  {
    FILE* f = std::fopen(path, "rb");
    if (!f) return vsl_fail(ctx, VSL_ERR_IO, "cannot open vocabulary file %s", path);
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz < 0) {
      std::fclose(f);
      return vsl_fail(ctx, VSL_ERR_IO, "cannot size vocabulary file %s", path);
    }
    text.resize((size_t)sz);
    const size_t got = sz ? std::fread(&text[0], 1, (size_t)sz, f) : 0;
    std::fclose(f);
    if (got != (size_t)sz) return vsl_fail(ctx, VSL_ERR_IO, "short read of vocabulary file %s", path);
  }
  const char* cur = text.c_str();
  const char* const end = cur + text.size();
  auto line_end = [&](const char* s) {
    const char* e = (const char*)memchr(s, '\n', (size_t)(end - s));
    return e ? e : end;
  };
  int k = -1, L = -1, n1 = -1, n2 = -1;
  {
    const char* le = line_end(cur);
    LineParser lp{cur, le};
    long a = -1, b = -1, c = -1, d = -1;
    const bool ok = lp.next_int(&a) && lp.next_int(&b) && lp.next_int(&c) && lp.next_int(&d);
    k = (int)a; L = (int)b; n1 = (int)c; n2 = (int)d;
    if (!ok || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)
      return vsl_fail(ctx, VSL_ERR_IO, "vocabulary header '%.*s' is not `k L scoring weighting`", (int)std::min<long>(le - cur, 80), cur);
    cur = le < end ? le + 1 : end;
  }
  if (n1 != 0 || n2 != 0)
    return vsl_fail(ctx, VSL_ERR_INVALID, "only L1_NORM scoring (0) with TF_IDF weighting (0) is implemented; file declares %d %d", n1, n2);
  // node arrays in file order (node 0 = root)
  std::vector<int32_t> parent(1, 0);
  std::vector<uint32_t> desc(8, 0), word_id(1, 0);
  std::vector<double> weight(1, 0.0);
  {
    const size_t guess = text.size() / 100 + 16;
    parent.reserve(guess);
    desc.reserve(8 * guess);
    word_id.reserve(guess);
    weight.reserve(guess);
  }
  int n_words = 0;
  while (cur < end) {
    const char* le = line_end(cur);
    LineParser lp{cur, le};
    lp.skip_blanks();
    if (lp.p >= le) {  // blank line: skipped (documented deviation)
      cur = le < end ? le + 1 : end;
      continue;
    }
    long pid = 0, leaf = 0;
    const int nid = (int)parent.size();
    const bool ok = lp.next_int(&pid) && lp.next_int(&leaf);
    if (!ok || pid < 0 || pid >= nid) return vsl_fail(ctx, VSL_ERR_IO, "vocabulary node %d: bad parent id", nid);
    uint8_t bytes[32] = {0};
    for (int i = 0; i < 32; i++) {  // FORB::fromString, FORB.cpp:118-135
      long v = 0;
      if (lp.next_int(&v)) bytes[i] = (uint8_t)v;
    }
    double w = 0;
    lp.next_double(&w);
    parent.push_back((int32_t)pid);
    uint32_t words[8];
    memcpy(words, bytes, 32);
    desc.insert(desc.end(), words, words + 8);
    weight.push_back(w);
    word_id.push_back(leaf > 0 ? (uint32_t)n_words : 0u);
    if (leaf > 0) n_words++;
    cur = le < end ? le + 1 : end;
  }
  text.clear();
  text.shrink_to_fit();
  const int n_nodes = (int)parent.size();
  // CSR child lists in file order (= the push_back order of the reference's loader)
  std::vector<int32_t> child_start(n_nodes + 1, 0), child_ids(n_nodes > 1 ? n_nodes - 1 : 0);
  for (int i = 1; i < n_nodes; i++) child_start[parent[i] + 1]++;
  int widest = 0;
  for (int i = 0; i < n_nodes; i++) {
    widest = std::max(widest, child_start[i + 1]);
    child_start[i + 1] += child_start[i];
  }
  if (widest > 64) return vsl_fail(ctx, VSL_ERR_IO, "a vocabulary node has %d children (at most 64 are supported)", widest);
  {
    std::vector<int32_t> fill(child_start.begin(), child_start.end() - 1);
    for (int i = 1; i < n_nodes; i++) child_ids[fill[parent[i]]++] = i;
  }
  // slot arrays
  const int n_slots = n_nodes - 1;
  std::vector<uint32_t> sdesc(8 * (size_t)std::max(n_slots, 1)), snode(std::max(n_slots, 1)), sword(std::max(n_slots, 1));
  std::vector<int2> sinfo(std::max(n_slots, 1));
  std::vector<double> sweight(std::max(n_slots, 1));
  for (int s = 0; s < n_slots; s++) {
    const int node = child_ids[s];
    memcpy(&sdesc[8 * (size_t)s], &desc[8 * (size_t)node], 32);
    sinfo[s] = make_int2(child_start[node], child_start[node + 1] - child_start[node]);
    snode[s] = (uint32_t)node;
    sword[s] = word_id[node];
    sweight[s] = weight[node];
  }
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_voc* v = new (std::nothrow) vsl_voc;
  if (!v) return vsl_fail(ctx, VSL_ERR_NOMEM, "out of host memory");
  v->device = ctx->device;
  v->k = k;
  v->L = L;
  v->n_nodes = n_nodes;
  v->n_words = n_words;
  v->n_slots = n_slots;
  v->root_nc = n_nodes > 1 ? child_start[1] : 0;
  v->group = widest <= 16 ? 16 : (widest <= 32 ? 32 : 64);
  int rc = 0;
  uint32_t* sdesc_dev = nullptr;
  if ((rc = to_device(ctx, &sdesc_dev, sdesc)) || (rc = to_device(ctx, &v->sinfo, sinfo)) ||
      (rc = to_device(ctx, &v->snode, snode)) || (rc = to_device(ctx, &v->sword, sword)) ||
      (rc = to_device(ctx, &v->sweight, sweight))) {
    if (sdesc_dev) (void)hipFree(sdesc_dev);
    vsl_voc_destroy(v);
    return rc;
  }
  v->sdesc = reinterpret_cast<uint4*>(sdesc_dev);
  *out = v;
  return VSL_OK;
}

extern "C" int vsl_voc_info(const vsl_voc* v, int* k, int* L, int* n_nodes, int* n_words) {
  if (!v) return VSL_ERR_INVALID;
  if (k) *k = v->k;
  if (L) *L = v->L;
  if (n_nodes) *n_nodes = v->n_nodes;
  if (n_words) *n_words = v->n_words;
  return VSL_OK;
}

// transform of n descriptors that are already on the device (feat_dev) or on the host (desc32); results through ONE
// pinned buffer and one copy
static int bow_transform_impl(vsl_ctx* ctx, const vsl_voc* voc, const uint8_t* desc32, const void* feat_dev, int n, int levelsup,
                              uint32_t* word_ids, double* word_vals, int* nnz, uint32_t* fv_node, uint32_t* fv_feat, int* fv_n) {
  *nnz = 0;
  *fv_n = 0;
  if (n == 0 || voc->n_nodes <= 1) return VSL_OK;  // empty(): TemplatedVocabulary.h:1135
  if (n > BOW_MAX_N) return vsl_fail(ctx, VSL_ERR_CAPACITY, "vsl_bow_transform: at most %d descriptors per call", BOW_MAX_N);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  // device scratch: f_w (8n) | feat (32n) | f_word (4n) | f_node (4n) | out: vals (8n) | ids (4n) | fvn (4n) | fvf (4n) | counts (16)
  const size_t N = (size_t)n;
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, 8 * N + 32 * N + 8 * N + 20 * N + 64 + 256, &d);
  if (rc) return rc;
  uint8_t* base = (uint8_t*)d;
  double* f_w = (double*)base;
  uint32_t* feat = (uint32_t*)(f_w + N);
  uint32_t* f_word = feat + 8 * N;
  uint32_t* f_node = f_word + N;
  // outputs, contiguous: one device-to-host copy
  uint8_t* outp = (uint8_t*)(f_node + N);
  outp += (16 - ((uintptr_t)outp & 15)) & 15;
  int32_t* counts = (int32_t*)outp;
  double* vals = (double*)(outp + 16);
  uint32_t* ids = (uint32_t*)(vals + N);
  uint32_t* fvn = ids + N;
  uint32_t* fvf = fvn + N;
  const size_t out_bytes = 16 + 20 * N;
  void* hp = nullptr;
  if ((rc = vsl_ctx_hpinned(ctx, out_bytes + 32 * N, &hp))) return rc;
  const void* src = feat_dev;
  if (!feat_dev) {
    uint8_t* stage = (uint8_t*)hp + out_bytes;  // pinned staging: the upload is a real asynchronous copy
    memcpy(stage, desc32, 32 * N);
    VSL_HIP(ctx, hipMemcpyAsync(feat, stage, 32 * N, hipMemcpyHostToDevice, ctx->stream));
    src = feat;
  }
  {
    VslStage st(ctx, VSL_STAGE_BOW_TRANSFORM);
    const int G = voc->group;
    const dim3 grid((unsigned)(((size_t)n * G + 255) / 256));
#define BOW_DESCEND(GG)                                                                                                   \
  hipLaunchKernelGGL(bow_descend_kernel<GG>, grid, dim3(256), 0, ctx->stream, (const uint4*)src, n, voc->sdesc, voc->sinfo, \
                     voc->snode, voc->sword, voc->sweight, voc->root_nc, voc->L, levelsup, f_word, f_w, f_node)
    if (G == 16) BOW_DESCEND(16);
    else if (G == 32) BOW_DESCEND(32);
    else BOW_DESCEND(64);
#undef BOW_DESCEND
    // key width: (id << shift) | feature index with all-ones reserved for the padding
    int shift = 11;
    while ((1 << shift) < n) shift++;
    const uint64_t max_id = (uint64_t)std::max(voc->n_nodes, voc->n_words);
    const bool k32 = !ctx->bow_keys64 && max_id + 2 <= (1ull << (32 - shift));
#define BOW_ASSEMBLE(CAP, KT, SH)                                                                                              \
  hipLaunchKernelGGL((bow_assemble_kernel<CAP, KT, (CAP <= 2048)>), dim3(2), dim3(1024), 0, ctx->stream, f_word, f_w, f_node, n, SH, ids, vals, \
                     counts, fvn, fvf)
    if (n <= 2048) {
      if (k32) BOW_ASSEMBLE(2048, uint32_t, shift);
      else BOW_ASSEMBLE(2048, unsigned long long, 32);
    } else {
      if (k32) BOW_ASSEMBLE(BOW_MAX_N, uint32_t, shift);
      else BOW_ASSEMBLE(BOW_MAX_N, unsigned long long, 32);
    }
#undef BOW_ASSEMBLE
    VSL_CHECK_LAUNCH(ctx);
  }
  VSL_HIP(ctx, hipMemcpyAsync(hp, outp, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const uint8_t* h = (const uint8_t*)hp;
  const int32_t* hc = (const int32_t*)h;
  *nnz = hc[0];
  *fv_n = hc[1];
  if (hc[0] > 0) {
    memcpy(word_vals, h + 16, 8 * (size_t)hc[0]);
    memcpy(word_ids, h + 16 + 8 * N, 4 * (size_t)hc[0]);
  }
  if (hc[1] > 0) {
    memcpy(fv_node, h + 16 + 12 * N, 4 * (size_t)hc[1]);
    memcpy(fv_feat, h + 16 + 16 * N, 4 * (size_t)hc[1]);
  }
  return VSL_OK;
}

extern "C" int vsl_bow_transform(vsl_ctx* ctx, const vsl_voc* voc, const uint8_t* desc32, int n, int levelsup,
                                 uint32_t* word_ids, double* word_vals, int* nnz, uint32_t* fv_node, uint32_t* fv_feat,
                                 int* fv_n) {
  if (!ctx || !voc || !nnz || !fv_n || n < 0 || (n > 0 && (!desc32 || !word_ids || !word_vals || !fv_node || !fv_feat)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bow_transform: bad arguments");
  return bow_transform_impl(ctx, voc, desc32, nullptr, n, levelsup, word_ids, word_vals, nnz, fv_node, fv_feat, fv_n);
}

// query upload + kernel + score download; the candidates are already on the device
static int bow_score_launch(vsl_ctx* ctx, const uint32_t* q_ids, const double* q_vals, int q_nnz, const uint32_t* c_ids_dev,
                            const double* c_vals_dev, const int64_t* off_dev, const int32_t* idx_dev, int m, uint8_t* qscratch,
                            double* scores, int max_cand_nnz) {
  // qscratch (device): q_vals (8Q) | scores (8M) | q_ids (4Q)
  const size_t Q = (size_t)q_nnz, M = (size_t)m;
  double* dqv = (double*)qscratch;
  double* dsc = dqv + Q;
  uint32_t* dqi = (uint32_t*)(dsc + M);
  void* hp = nullptr;
  int rc = vsl_ctx_hpinned(ctx, 12 * Q + 8 * M + 64, &hp);
  if (rc) return rc;
  double* hsc = (double*)hp;                 // [M]
  uint8_t* hq = (uint8_t*)(hsc + M);         // q_vals | q_ids staged in pinned memory
  if (Q) {
    memcpy(hq, q_vals, 8 * Q);
    memcpy(hq + 8 * Q, q_ids, 4 * Q);
    VSL_HIP(ctx, hipMemcpyAsync(dqv, hq, 8 * Q, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(dqi, hq + 8 * Q, 4 * Q, hipMemcpyHostToDevice, ctx->stream));
  }
  {
    VslStage st(ctx, VSL_STAGE_BOW_SCORE);
    bool wg_form = false;
    if (m <= BOW_WG_MAX_M && q_nnz <= BOW_WG_Q_MAX && !ctx->bow_no_wg_score) {
      // every candidate must fit a workgroup's staging (the offsets live on the device: the callers pass the bound)
      wg_form = max_cand_nnz >= 0 && max_cand_nnz <= BOW_WG_TERMS;
    }
    if (wg_form) {
      int P = 1;
      while (P < q_nnz + 1) P <<= 1;
      hipLaunchKernelGGL(bow_score_wg_kernel, dim3(m), dim3(BOW_WG_THREADS), 0, ctx->stream, dqi, dqv, q_nnz, P, c_ids_dev, c_vals_dev,
                         off_dev, idx_dev, dsc);
    } else if (q_nnz <= BOW_Q_LDS_MAX) {
      int P = 1;
      while (P < q_nnz + 1) P <<= 1;  // at least one 0xFFFFFFFF sentinel behind the query
      const size_t lds = 8 * (size_t)((q_nnz + 1) & ~1) + 4 * (size_t)P;
      if (lds > 64 * 1024) {
        if (!ctx->bow_score_attr_set) {  // per context = per device (the attribute is a per-device setting)
          VSL_HIP(ctx, hipFuncSetAttribute((const void*)bow_score_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
          ctx->bow_score_attr_set = true;
        }
      }
      hipLaunchKernelGGL(bow_score_lds_kernel, dim3((m + 3) / 4), dim3(256), lds, ctx->stream, dqi, dqv, q_nnz, P, c_ids_dev,
                         c_vals_dev, off_dev, idx_dev, m, dsc);
    } else {
      hipLaunchKernelGGL(bow_score_global_kernel, dim3(m), dim3(64), 0, ctx->stream, dqi, dqv, q_nnz, c_ids_dev, c_vals_dev,
                         off_dev, idx_dev, dsc);
    }
    VSL_CHECK_LAUNCH(ctx);
  }
  VSL_HIP(ctx, hipMemcpyAsync(hsc, dsc, 8 * M, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(scores, hsc, 8 * M);
  return VSL_OK;
}

extern "C" int vsl_bow_score_batch(vsl_ctx* ctx, const uint32_t* q_ids, const double* q_vals, int q_nnz,
                                   const uint32_t* c_ids, const double* c_vals, const int32_t* c_offsets, int m,
                                   double* scores) {
  if (!ctx || q_nnz < 0 || m < 0 || (m > 0 && (!c_offsets || !scores)) || (q_nnz > 0 && (!q_ids || !q_vals)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bow_score_batch: bad arguments");
  if (m == 0) return VSL_OK;
  const int total = c_offsets[m];
  if (total < 0 || (total > 0 && (!c_ids || !c_vals))) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bow_score_batch: bad candidate arrays");
  std::vector<int64_t> off64((size_t)m + 1);
  off64[0] = c_offsets[0];
  for (int i = 0; i < m; i++) {
    if (c_offsets[i + 1] < c_offsets[i]) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bow_score_batch: offsets not monotone");
    off64[i + 1] = c_offsets[i + 1];
  }
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t Q = (size_t)q_nnz, T = (size_t)total, M = (size_t)m;
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, 8 * (Q + M) + 4 * Q + 64 + 8 * T + 8 * (M + 1) + 4 * T + 64, &d);
  if (rc) return rc;
  uint8_t* qs = (uint8_t*)d;
  uint8_t* p = qs + 8 * (Q + M) + 4 * Q;
  p += (16 - ((uintptr_t)p & 15)) & 15;
  double* dcv = (double*)p;
  int64_t* dof = (int64_t*)(dcv + T);
  uint32_t* dci = (uint32_t*)(dof + M + 1);
  if (T) {
    VSL_HIP(ctx, hipMemcpyAsync(dcv, c_vals, 8 * T, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(dci, c_ids, 4 * T, hipMemcpyHostToDevice, ctx->stream));
  }
  VSL_HIP(ctx, hipMemcpyAsync(dof, off64.data(), 8 * (M + 1), hipMemcpyHostToDevice, ctx->stream));
  int max_nnz = 0;
  for (int i = 0; i < m; i++) max_nnz = std::max(max_nnz, c_offsets[i + 1] - c_offsets[i]);
  return bow_score_launch(ctx, q_ids, q_vals, q_nnz, dci, dcv, dof, nullptr, m, qs, scores, max_nnz);
}

// ------------------------------------------------------------------------------------------------ vsl_bowdb
extern "C" int vsl_bowdb_create(vsl_ctx* ctx, int64_t cap_entries, int cap_vectors, vsl_bowdb** out) {
  if (!ctx || !out || cap_entries < 0 || cap_vectors < 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bowdb_create: bad arguments");
  *out = nullptr;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_bowdb* db = new (std::nothrow) vsl_bowdb;
  if (!db) return vsl_fail(ctx, VSL_ERR_NOMEM, "out of host memory");
  db->device = ctx->device;
  db->cap_entries = std::max<int64_t>(cap_entries, 4096);
  db->cap_vecs = std::max(cap_vectors, 64);
  if (hipMalloc((void**)&db->ids, 4 * (size_t)db->cap_entries) != hipSuccess ||
      hipMalloc((void**)&db->vals, 8 * (size_t)db->cap_entries) != hipSuccess ||
      hipMalloc((void**)&db->off, 8 * ((size_t)db->cap_vecs + 1)) != hipSuccess) {
    vsl_bowdb_destroy(db);
    return vsl_fail(ctx, VSL_ERR_NOMEM, "vsl_bowdb_create: device allocation failed");
  }
  const int64_t zero = 0;
  VSL_HIP(ctx, hipMemcpy(db->off, &zero, 8, hipMemcpyHostToDevice));
  *out = db;
  return VSL_OK;
}

extern "C" int vsl_bowdb_destroy(vsl_bowdb* db) {
  if (!db) return VSL_OK;
  (void)hipSetDevice(db->device);
  if (db->ids) (void)hipFree(db->ids);
  if (db->vals) (void)hipFree(db->vals);
  if (db->off) (void)hipFree(db->off);
  delete db;
  return VSL_OK;
}

template <class T>
static int grow_dev(vsl_ctx* ctx, T** p, size_t used, size_t new_cap) {
  T* q = nullptr;
  if (hipMalloc((void**)&q, sizeof(T) * new_cap) != hipSuccess) return vsl_fail(ctx, VSL_ERR_NOMEM, "vsl_bowdb: device allocation failed");
  VSL_HIP(ctx, hipMemcpyAsync(q, *p, sizeof(T) * used, hipMemcpyDeviceToDevice, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(*p);
  *p = q;
  return VSL_OK;
}

extern "C" int vsl_bowdb_append(vsl_ctx* ctx, vsl_bowdb* db, const uint32_t* ids, const double* vals, int nnz, int* index_out) {
  if (!ctx || !db || nnz < 0 || (nnz > 0 && (!ids || !vals))) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bowdb_append: bad arguments");
  for (int i = 1; i < nnz; i++)
    if (ids[i] <= ids[i - 1]) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bowdb_append: word ids must be strictly ascending");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  int rc;
  if (db->n_entries + nnz > db->cap_entries) {
    const int64_t cap = std::max<int64_t>(2 * db->cap_entries, db->n_entries + nnz);
    if ((rc = grow_dev(ctx, &db->ids, (size_t)db->n_entries, (size_t)cap)) || (rc = grow_dev(ctx, &db->vals, (size_t)db->n_entries, (size_t)cap)))
      return rc;
    db->cap_entries = cap;
  }
  if (db->n_vecs + 1 > db->cap_vecs) {
    const int cap = 2 * db->cap_vecs;
    if ((rc = grow_dev(ctx, &db->off, (size_t)db->n_vecs + 1, (size_t)cap + 1))) return rc;
    db->cap_vecs = cap;
  }
  if (nnz) {
    VSL_HIP(ctx, hipMemcpyAsync(db->ids + db->n_entries, ids, 4 * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(db->vals + db->n_entries, vals, 8 * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
  }
  db->n_entries += nnz;
  const int64_t end = db->n_entries;
  VSL_HIP(ctx, hipMemcpyAsync(db->off + db->n_vecs + 1, &end, 8, hipMemcpyHostToDevice, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the sources are the caller's (pageable) arrays
  if (index_out) *index_out = db->n_vecs;
  db->n_vecs++;
  db->max_nnz = std::max(db->max_nnz, nnz);
  return VSL_OK;
}

extern "C" int vsl_bowdb_info(const vsl_bowdb* db, int* n_vectors, int64_t* n_entries) {
  if (!db) return VSL_ERR_INVALID;
  if (n_vectors) *n_vectors = db->n_vecs;
  if (n_entries) *n_entries = db->n_entries;
  return VSL_OK;
}

extern "C" int vsl_bowdb_score(vsl_ctx* ctx, const vsl_bowdb* db, const uint32_t* q_ids, const double* q_vals, int q_nnz,
                               const int32_t* cand_index, int m, double* scores) {
  if (!ctx || !db || q_nnz < 0 || m < 0 || (m > 0 && !scores) || (q_nnz > 0 && (!q_ids || !q_vals)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bowdb_score: bad arguments");
  if (m == 0) return VSL_OK;
  if (!cand_index && m > db->n_vecs) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bowdb_score: %d candidates, %d vectors stored", m, db->n_vecs);
  if (cand_index)
    for (int i = 0; i < m; i++)
      if (cand_index[i] < 0 || cand_index[i] >= db->n_vecs)
        return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bowdb_score: candidate %d = vector %d, %d vectors stored", i, cand_index[i], db->n_vecs);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t Q = (size_t)q_nnz, M = (size_t)m;
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, 8 * (Q + M) + 4 * Q + 4 * M + 64, &d);
  if (rc) return rc;
  uint8_t* qs = (uint8_t*)d;
  int32_t* didx = nullptr;
  if (cand_index) {
    didx = (int32_t*)(qs + 8 * (Q + M) + 4 * Q + ((4 - ((8 * (Q + M) + 4 * Q) & 3)) & 3));
    VSL_HIP(ctx, hipMemcpyAsync(didx, cand_index, 4 * M, hipMemcpyHostToDevice, ctx->stream));
  }
  return bow_score_launch(ctx, q_ids, q_vals, q_nnz, db->ids, db->vals, db->off, didx, m, qs, scores, db->max_nnz);
}
