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
#include <fstream>
#include <sstream>
#include <string>

#include "vsl_common.h"

struct vsl_voc {
  int device = 0;
  int k = 0, L = 0, n_nodes = 0, n_words = 0;
  // device arrays, one entry per node
  uint32_t* desc = nullptr;         // [n_nodes][8]   node descriptor (32 bytes, byte order as in the file)
  int32_t* child_start = nullptr;   // [n_nodes + 1]  CSR into child_ids
  int32_t* child_ids = nullptr;     // [n_nodes - 1]  children in the order the file lists them
  double* weight = nullptr;         // [n_nodes]
  uint32_t* word_id = nullptr;      // [n_nodes]
};

namespace {

#define BOW_MAX_N 8192

// K8a: one wavefront per descriptor walks the tree; lane c evaluates child c of the current node.
// Strict '<' with the first child winning ties == minimum of (distance << 8 | child position).
__global__ __launch_bounds__(256) void bow_descend_kernel(const uint32_t* __restrict__ feat, int n,
                                                          const uint32_t* __restrict__ ndesc,
                                                          const int32_t* __restrict__ child_start,
                                                          const int32_t* __restrict__ child_ids,
                                                          const double* __restrict__ weight,
                                                          const uint32_t* __restrict__ word_id, int L, int levelsup,
                                                          uint32_t* __restrict__ out_word, double* __restrict__ out_w,
                                                          uint32_t* __restrict__ out_node) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (f >= n) return;
  uint32_t d[8];
#pragma unroll
  for (int q = 0; q < 8; q++) d[q] = feat[8 * (size_t)f + q];
  const int nid_level = L - levelsup;
  uint32_t nid = 0;  // root when nid_level <= 0
  int node = 0, level = 0;
  while (true) {
    const int c0 = child_start[node], nc = child_start[node + 1] - c0;
    if (nc == 0) break;  // isLeaf()
    ++level;
    uint32_t key = 0xFFFFFFFFu;
    if (lane < nc) {
      const int cid = child_ids[c0 + lane];
      const uint32_t* cd = ndesc + 8 * (size_t)cid;
      uint32_t dist = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) dist += __builtin_popcount(d[q] ^ cd[q]);
      key = (dist << 8) | (uint32_t)lane;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o));
    node = child_ids[c0 + (int)(key & 0xFF)];
    if (level == nid_level) nid = (uint32_t)node;
  }
  if (lane == 0) {
    out_word[f] = word_id[node];
    out_w[f] = weight[node];
    out_node[f] = nid;
  }
}

__device__ void bitonic_sort_u64(unsigned long long* keys, int N) {
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (N >> 1); t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const unsigned long long a = keys[i], b = keys[l];
        const bool asc = (i & k) == 0;
        if (asc ? (a > b) : (a < b)) {
          keys[i] = b;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
}

// K8b: one workgroup assembles the BowVector and the FeatureVector of one image.
//   BowVector  : features with weight > 0, grouped by word (ascending id); the value of a word is its
//                weight added once per feature, in feature order (== BowVector::addWeight); L1 norm
//                accumulated in ascending word order (== BowVector::normalize), then divided.
//   FeatureVector: (node, feature) pairs sorted by node, then feature index.
__global__ __launch_bounds__(1024) void bow_assemble_kernel(const uint32_t* __restrict__ f_word,
                                                            const double* __restrict__ f_w,
                                                            const uint32_t* __restrict__ f_node, int n,
                                                            uint32_t* __restrict__ word_ids, double* __restrict__ word_vals,
                                                            int32_t* __restrict__ counts, uint32_t* __restrict__ fv_node,
                                                            uint32_t* __restrict__ fv_feat) {
  __shared__ unsigned long long keys[BOW_MAX_N];
  __shared__ double vals_s[BOW_MAX_N];
  __shared__ int scan[1024];
  __shared__ int n_kept;
  __shared__ double norm_s;
  const int tid = threadIdx.x;
  int N = 1024;
  while (N < n) N <<= 1;
  // ---- FeatureVector
  for (int i = tid; i < N; i += 1024)
    keys[i] = (i < n && f_w[i] > 0.0) ? (((unsigned long long)f_node[i] << 32) | (unsigned)i) : ~0ull;
  if (tid == 0) n_kept = 0;
  __syncthreads();
  bitonic_sort_u64(keys, N);
  for (int i = tid; i < N; i += 1024)
    if (keys[i] != ~0ull) {
      fv_node[i] = (uint32_t)(keys[i] >> 32);
      fv_feat[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull);
      atomicAdd(&n_kept, 1);
    }
  __syncthreads();
  const int kept = n_kept;
  __syncthreads();
  // ---- BowVector
  for (int i = tid; i < N; i += 1024)
    keys[i] = (i < n && f_w[i] > 0.0) ? (((unsigned long long)f_word[i] << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  bitonic_sort_u64(keys, N);
  // a run of equal word ids = one BowVector entry; its first element computes the value.  Ordered
  // compaction of the run heads by a block scan over contiguous per-thread slices.
  const int per = N >> 10;
  int heads = 0;
  for (int k = 0; k < per; k++) {
    const int i = tid * per + k;
    heads += i < kept && (i == 0 || (uint32_t)(keys[i - 1] >> 32) != (uint32_t)(keys[i] >> 32));
  }
  scan[tid] = heads;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int add = tid >= d ? scan[tid - d] : 0;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  int rank = scan[tid] - heads;
  const int uniq = scan[1023];
  for (int k = 0; k < per; k++) {
    const int i = tid * per + k;
    if (i >= kept) break;
    const uint32_t w = (uint32_t)(keys[i] >> 32);
    if (i != 0 && (uint32_t)(keys[i - 1] >> 32) == w) continue;
    double v = 0.0;
    bool init = false;
    for (int q = i; q < kept && (uint32_t)(keys[q] >> 32) == w; q++) {
      const double wt = f_w[(uint32_t)(keys[q] & 0xFFFFFFFFull)];
      v = init ? v + wt : wt;  // insert(id, w) then += w (BowVector.cpp:38-45)
      init = true;
    }
    word_ids[rank] = w;
    vals_s[rank] = v;
    rank++;
  }
  __syncthreads();
  if (tid == 0) {
    double norm = 0.0;  // ascending word order, one accumulator (BowVector.cpp:62-74)
    for (int i = 0; i < uniq; i++) norm += fabs(vals_s[i]);
    norm_s = norm;
    counts[0] = uniq;
    counts[1] = kept;
  }
  __syncthreads();
  const double norm = norm_s;
  for (int i = tid; i < uniq; i += 1024) word_vals[i] = norm > 0.0 ? vals_s[i] / norm : vals_s[i];
}

// K9: one wavefront per candidate BowVector.  Lane-parallel lookup of every candidate word in the
// query (binary search), then the matched terms are summed in ascending word order by one lane --
// the reference's order (ScoringObject.cpp:32-59); unmatched entries contribute an exact +0.
__global__ __launch_bounds__(64) void bow_score_kernel(const uint32_t* __restrict__ q_ids, const double* __restrict__ q_vals,
                                                       int q_nnz, const uint32_t* __restrict__ c_ids,
                                                       const double* __restrict__ c_vals, const int32_t* __restrict__ c_off,
                                                       double* __restrict__ scores) {
  __shared__ double term[64];
  const int m = blockIdx.x, lane = threadIdx.x;
  const int a = c_off[m], b = c_off[m + 1];
  double score = 0.0;
  for (int base = a; base < b; base += 64) {
    const int i = base + lane;
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
      const int cnt = min(64, b - base);
      for (int q = 0; q < cnt; q++) score += term[q];
    }
    __syncthreads();
  }
  if (lane == 0) scores[m] = -score / 2.0;
}

template <class T>
int to_device(vsl_ctx* ctx, T** dst, const std::vector<T>& src) {
  VSL_HIP(ctx, hipMalloc((void**)dst, sizeof(T) * (src.size() ? src.size() : 1)));
  if (!src.empty()) VSL_HIP(ctx, hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  return VSL_OK;
}

}  // namespace

extern "C" int vsl_voc_destroy(vsl_voc* v) {
  if (!v) return VSL_OK;
  (void)hipSetDevice(v->device);
  void* ptrs[] = {v->desc, v->child_start, v->child_ids, v->weight, v->word_id};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete v;
  return VSL_OK;
}

extern "C" int vsl_voc_load_text(vsl_ctx* ctx, const char* path, vsl_voc** out) {
  if (!ctx || !path || !out) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_voc_load_text: null argument");
  *out = nullptr;
  std::ifstream f(path);
  if (!f.is_open()) return vsl_fail(ctx, VSL_ERR_IO, "cannot open vocabulary file %s", path);
  std::string line;
  std::getline(f, line);
  int k = -1, L = -1, n1 = -1, n2 = -1;
  {
    std::stringstream ss(line);
    ss >> k >> L >> n1 >> n2;
    if (ss.fail() || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)
      return vsl_fail(ctx, VSL_ERR_IO, "vocabulary header '%s' is not `k L scoring weighting`", line.c_str());
  }
  if (n1 != 0 || n2 != 0)
    return vsl_fail(ctx, VSL_ERR_INVALID, "only L1_NORM scoring (0) with TF_IDF weighting (0) is implemented; file declares %d %d", n1, n2);
  std::vector<int32_t> parent(1, 0);
  std::vector<uint32_t> desc(8, 0), word_id(1, 0);
  std::vector<double> weight(1, 0.0);
  std::vector<uint8_t> is_leaf(1, 0);
  int n_words = 0;
  while (std::getline(f, line)) {
    if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
    std::stringstream ss(line);
    int pid = 0, leaf = 0;
    ss >> pid >> leaf;
    const int nid = (int)parent.size();
    if (ss.fail() || pid < 0 || pid >= nid) return vsl_fail(ctx, VSL_ERR_IO, "vocabulary node %d: bad parent id", nid);
    uint8_t bytes[32] = {0};
    for (int i = 0; i < 32; i++) {
      int v = 0;
      ss >> v;
      if (!ss.fail()) bytes[i] = (uint8_t)v;
    }
    double w = 0;
    ss >> w;
    parent.push_back(pid);
    uint32_t words[8];
    memcpy(words, bytes, 32);
    desc.insert(desc.end(), words, words + 8);
    weight.push_back(w);
    is_leaf.push_back(leaf > 0);
    word_id.push_back(leaf > 0 ? (uint32_t)n_words : 0u);
    if (leaf > 0) n_words++;
  }
  const int n_nodes = (int)parent.size();
  std::vector<int32_t> child_start(n_nodes + 1, 0), child_ids(n_nodes > 1 ? n_nodes - 1 : 0);
  for (int i = 1; i < n_nodes; i++) child_start[parent[i] + 1]++;
  for (int i = 0; i < n_nodes; i++) child_start[i + 1] += child_start[i];
  {
    std::vector<int32_t> fill(child_start.begin(), child_start.end() - 1);
    for (int i = 1; i < n_nodes; i++) child_ids[fill[parent[i]]++] = i;  // file order = push_back order
  }
  for (int i = 0; i < n_nodes; i++)
    if (child_start[i + 1] - child_start[i] > 64)
      return vsl_fail(ctx, VSL_ERR_IO, "vocabulary node %d has more than 64 children", i);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_voc* v = new (std::nothrow) vsl_voc;
  if (!v) return vsl_fail(ctx, VSL_ERR_NOMEM, "out of host memory");
  v->device = ctx->device;
  v->k = k;
  v->L = L;
  v->n_nodes = n_nodes;
  v->n_words = n_words;
  int rc = 0;
  if ((rc = to_device(ctx, &v->desc, desc)) || (rc = to_device(ctx, &v->child_start, child_start)) ||
      (rc = to_device(ctx, &v->child_ids, child_ids)) || (rc = to_device(ctx, &v->weight, weight)) ||
      (rc = to_device(ctx, &v->word_id, word_id))) {
    vsl_voc_destroy(v);
    return rc;
  }
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

extern "C" int vsl_bow_transform(vsl_ctx* ctx, const vsl_voc* voc, const uint8_t* desc32, int n, int levelsup,
                                 uint32_t* word_ids, double* word_vals, int* nnz, uint32_t* fv_node, uint32_t* fv_feat,
                                 int* fv_n) {
  if (!ctx || !voc || !nnz || !fv_n || n < 0 || (n > 0 && (!desc32 || !word_ids || !word_vals || !fv_node || !fv_feat)))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bow_transform: bad arguments");
  *nnz = 0;
  *fv_n = 0;
  if (n == 0 || voc->n_nodes <= 1) return VSL_OK;  // empty(): TemplatedVocabulary.h:1135
  if (n > BOW_MAX_N) return vsl_fail(ctx, VSL_ERR_CAPACITY, "vsl_bow_transform: at most %d descriptors per call", BOW_MAX_N);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  // device scratch: feat (32n) | f_word (4n) | f_node (4n) | ids (4n) | fvn (4n) | fvf (4n) | f_w (8n) | vals (8n) | counts
  const size_t N = (size_t)n;
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, 32 * N + 5 * 4 * N + 2 * 8 * N + 64 + 64, &d);
  if (rc) return rc;
  uint8_t* base = (uint8_t*)d;
  double* f_w = (double*)base;
  double* vals = f_w + N;
  uint32_t* feat = (uint32_t*)(vals + N);
  uint32_t* f_word = feat + 8 * N;
  uint32_t* f_node = f_word + N;
  uint32_t* ids = f_node + N;
  uint32_t* fvn = ids + N;
  uint32_t* fvf = fvn + N;
  int32_t* counts = (int32_t*)(fvf + N);
  VSL_HIP(ctx, hipMemcpyAsync(feat, desc32, 32 * N, hipMemcpyHostToDevice, ctx->stream));
  {
    VslStage st(ctx, VSL_STAGE_BOW_TRANSFORM);
    hipLaunchKernelGGL(bow_descend_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, feat, n, voc->desc,
                       voc->child_start, voc->child_ids, voc->weight, voc->word_id, voc->L, levelsup, f_word, f_w, f_node);
    hipLaunchKernelGGL(bow_assemble_kernel, dim3(1), dim3(1024), 0, ctx->stream, f_word, f_w, f_node, n, ids, vals,
                       counts, fvn, fvf);
    VSL_CHECK_LAUNCH(ctx);
  }
  int32_t hc[2] = {0, 0};
  VSL_HIP(ctx, hipMemcpyAsync(hc, counts, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *nnz = hc[0];
  *fv_n = hc[1];
  if (hc[0] > 0) {
    VSL_HIP(ctx, hipMemcpy(word_ids, ids, 4 * (size_t)hc[0], hipMemcpyDeviceToHost));
    VSL_HIP(ctx, hipMemcpy(word_vals, vals, 8 * (size_t)hc[0], hipMemcpyDeviceToHost));
  }
  if (hc[1] > 0) {
    VSL_HIP(ctx, hipMemcpy(fv_node, fvn, 4 * (size_t)hc[1], hipMemcpyDeviceToHost));
    VSL_HIP(ctx, hipMemcpy(fv_feat, fvf, 4 * (size_t)hc[1], hipMemcpyDeviceToHost));
  }
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
  for (int i = 0; i < m; i++)
    if (c_offsets[i + 1] < c_offsets[i]) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_bow_score_batch: offsets not monotone");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t Q = (size_t)q_nnz, T = (size_t)total, M = (size_t)m;
  void* d = nullptr;
  int rc = vsl_ctx_dscratch(ctx, 8 * (Q + T + M) + 4 * (Q + T + M + 1) + 64, &d);
  if (rc) return rc;
  double* dqv = (double*)d;
  double* dcv = dqv + Q;
  double* dsc = dcv + T;
  uint32_t* dqi = (uint32_t*)(dsc + M);
  uint32_t* dci = dqi + Q;
  int32_t* dof = (int32_t*)(dci + T);
  if (Q) {
    VSL_HIP(ctx, hipMemcpyAsync(dqv, q_vals, 8 * Q, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(dqi, q_ids, 4 * Q, hipMemcpyHostToDevice, ctx->stream));
  }
  if (T) {
    VSL_HIP(ctx, hipMemcpyAsync(dcv, c_vals, 8 * T, hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(dci, c_ids, 4 * T, hipMemcpyHostToDevice, ctx->stream));
  }
  VSL_HIP(ctx, hipMemcpyAsync(dof, c_offsets, 4 * (M + 1), hipMemcpyHostToDevice, ctx->stream));
  {
    VslStage st(ctx, VSL_STAGE_BOW_SCORE);
    hipLaunchKernelGGL(bow_score_kernel, dim3(m), dim3(64), 0, ctx->stream, dqi, dqv, q_nnz, dci, dcv, dof, dsc);
    VSL_CHECK_LAUNCH(ctx);
  }
  VSL_HIP(ctx, hipMemcpyAsync(scores, dsc, 8 * M, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}
