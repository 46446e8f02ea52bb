// vsl_common.h -- internal definitions shared by the HIP translation units of libvslam_hip.so.
// Written for gfx950 (MI355X) only: 64-lane wavefronts are assumed everywhere.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vslam_hip.h"

#define VSL_WAVE 64
#define VSL_META_STRIDE 32  // ints: 128 bytes per image
#define VSL_META_MAX 0      // order-preserving int encoding of the fp32 response maximum
#define VSL_META_NCAND 1    // number of corner candidates
#define VSL_META_NEXACT 3   // rBRIEF bits whose rotated coordinates need the exact (integer) rounding
// MAX / NCAND / NEXACT are consumed and reset by the kernel that reads them (selection, exact bits), so a
// detect / describe call needs no separate clearing launch; the last values stay readable here:
#define VSL_META_NCAND_LAST 4
#define VSL_META_NEXACT_LAST 5
#define VSL_EXACT_CAP 16384 // capacity of that per-image list
#ifndef VSL_TILE_LX
#define VSL_TILE_LX 7  // log2 width / height of the keypoint tiles of the batched describe kernel (describe.hip)
#endif
#ifndef VSL_TILE_LY
#define VSL_TILE_LY 7
#endif
#define VSL_TIE_OVERFLOW_FLAG 0x40000000u  // set in tie_count by exact_bits_kernel when an image's list overflowed

struct vsl_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  char err[512] = {0};
  // per-stage profiling
  bool profiling = false;
  struct StageEv {
    hipEvent_t a, b;
    int stage;
  };
  std::vector<StageEv> pending;
  std::vector<hipEvent_t> ev_pool;
  double stage_ms[VSL_STAGE_COUNT] = {0};
  int64_t stage_launches[VSL_STAGE_COUNT] = {0};
  // scratch frame store used by the host-buffer entry points (lazily created)
  struct vsl_frames* scratch = nullptr;
  int scratch_w = 0, scratch_h = 0, scratch_feat = 0;
  // generic device / pinned scratch
  void* dscratch = nullptr;
  size_t dscratch_cap = 0;
  void* hpinned = nullptr;
  size_t hpinned_cap = 0;
  // job lists of the block-cyclic-reduction solver (chol.hip), kept on the device while (n, bandwidth) stay the same
  void* bcr_jobs = nullptr;
  size_t bcr_jobs_cap = 0;
  int bcr_key_n = -1, bcr_key_bw = -1;
  // device arena lent to vsl_bundle_adjust calls on this context (one allocation reused across solves)
  void* ba_arena = nullptr;
  size_t ba_arena_cap = 0;
  bool ba_arena_busy = false;
  // pinned, device-mapped mailbox of the fused local-BA iteration (ba_fused.hip): kernels post their scalars there
  int64_t last_ba_s_elems = 0;  // layout of the reduced camera system of the last general-path solve set up on this context
  int last_ba_banded = 0, last_ba_bw = 0;
  void* ba_pin = nullptr;   // [plan of the solve | mailbox]: one copy carries the plan to the device
  size_t ba_pin_cap = 0;    // bytes
  bool select_attr_set = false;
  bool bow_score_attr_set = false;  // per context, hence per device: hipFuncSetAttribute is a per-device setting
  double* status_word = nullptr;    // 64 device bytes allocated with the context: the flag of status exchanges between ranks (never null in a live context)
  double tie_eps = 1e-12;  // rBRIEF near-tie guard band (describe.hip)
  bool match_use_i8 = false;            // diagnostic: int8 matrix-core matcher even where the FP4 one applies (<= 2048 features)
  bool match_no_stagger = false;        // diagnostic: all waves of a matcher workgroup in the same phase order (the pre-stagger kernel)
  bool match_two_pass = false;          // diagnostic: forward + reverse passes of the FP4 matcher even for launches of fewer than 8 pairs (which default to one launch with both full directions)
  bool match_use_valu = false;          // diagnostic: VALU popcount matcher instead of the MFMA one
  bool force_generic_describe = false;  // diagnostic: use the f64 kernel for every describe call
  int select_bucket_cap = 128;          // diagnostic: fullest response bin the counting sort of the selection kernel accepts (0: always the bitonic network)
  bool chol_no_bcr = false;             // diagnostic: long narrow bands by the (two-ended) band Cholesky instead of block cyclic reduction
  bool chol_one_ended = false;          // diagnostic: narrow-band Cholesky by one workgroup from the top only (no two-ended split)
  bool chol_no_fused = false;           // diagnostic: band Cholesky as one launch per panel step instead of the fused single-launch kernel
  bool ba_force_dense = false;          // diagnostic: dense reduced camera system even where the band form applies
  bool ba_schur_atomics = false;        // diagnostic: large-system Schur complement by fp64 atomics (one wavefront per landmark) instead of the per-block gather
  bool ba_no_fused = false;             // diagnostic: local windows by the operator-by-operator kernels of ba.hip instead of the fused iteration (ba_fused.hip)
  bool ba_schur_entries = false;        // diagnostic: single-entry ownership in the small-system Schur kernel instead of 3 x 3 sub-blocks
  int describe_tile_min_images = 96;   // describe launches of at least this many images use the shared-tile kernel (measured break-even ~64 images; diagnostic: 1 forces it, 0 disables it)
  bool ba_no_cyclic = false;            // diagnostic: large reduced camera systems in the linear band form (reverse Cuthill-McKee order) even when the cyclic form is narrower
  bool ba_host_lm = false;              // diagnostic: the fused local iteration with the Levenberg-Marquardt decision on the HOST (one synchronisation per iteration) instead of on the device (ba_fused.hip baf_decide_kernel)
  bool bow_no_wg_score = false;         // diagnostic: wave-per-candidate scoring kernel also for few candidates
  bool bow_keys64 = false;              // diagnostic: 64-bit sort keys in the BowVector assembly even where (id, feature) fits 32 bits
  int vo_chain_ticket = 0;              // diagnostic: vsl_map_track draws chain positions from the atomic ticket at every map size
  int pending_desc_max = 64;            // unresolved describe launches a frame store queues before it settles them itself (diagnostic: tests lower it)
  int exact_list_cap = VSL_EXACT_CAP;   // diagnostic: per-image exact-rounding list entries the describe kernels use (tests shrink it to hit the overflow fallback)
  int k1_list_cap = -1;                // diagnostic: per-wave LDS candidate slots in K1 (tests shrink it to hit the overflow path)
};

int vsl_fail(vsl_ctx* ctx, int code, const char* fmt, ...);

// ba_fused.hip: the local-window LM loop in four launches per iteration; *handled = 0 when the problem does not fit it
int vsl_ba_fused_solve(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary, int* handled);
// ba.hip: dc = -(S^-1 rhs) for n <= 128 by one workgroup (enqueued); *ok_flag (device-visible memory) = 1 / 0
int vsl_ba_chol_small_launch(vsl_ctx* ctx, int n, const double* S, const double* rhs, double* dc, int* ok_flag);

#define VSL_HIP(ctx, call)                                                                        \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return vsl_fail((ctx), VSL_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,            \
                      hipGetErrorString(e_));                                                     \
  } while (0)

#define VSL_CHECK_LAUNCH(ctx)                                                                     \
  do {                                                                                            \
    hipError_t e_ = hipGetLastError();                                                            \
    if (e_ != hipSuccess)                                                                         \
      return vsl_fail((ctx), VSL_ERR_HIP, "%s:%d kernel launch -> %s", __FILE__, __LINE__,        \
                      hipGetErrorString(e_));                                                     \
  } while (0)

// Grow-only device / pinned-host scratch owned by the context.
int vsl_ctx_dscratch(vsl_ctx* ctx, size_t bytes, void** out);
int vsl_ctx_hpinned(vsl_ctx* ctx, size_t bytes, void** out);

// RAII stage timer: records events around a stage when profiling is enabled.
struct VslStage {
  vsl_ctx* ctx;
  int idx = -1;
  VslStage(vsl_ctx* c, int stage);
  ~VslStage();
};

// ---------------------------------------------------------------------------------------------
// Device-resident frame store (see include/vslam_hip.h "device-resident batched frame store").
// SoA per image slot; F = max_features; every array is dense so a range of slots is one launch.
struct vsl_frames {
  int device = 0;
  int max_images = 0, w = 0, h = 0, F = 0, max_pairs = 0;
  size_t cand_cap = 0;          // candidate capacity per image (w*h: every pixel may be a candidate)
  uint8_t* images = nullptr;    // [max_images][h][w]
  float* response = nullptr;    // [max_images][h][w]            K1 output
  int32_t* meta = nullptr;      // [max_images][VSL_META_STRIDE]: per-image counters on their own 128-B lines
  uint64_t* cand = nullptr;     // [max_images][cand_cap]        (fp32 bits << 32 | pixel index)
  int32_t* kp_xy = nullptr;     // [max_images][F][2]            selected corners, response-descending
  int32_t* kp_count = nullptr;  // [max_images]
  int32_t* kp_moments = nullptr;  // [max_images][F][2]          (m01, m10), exact
  double* kp_angle = nullptr;   // [max_images][F]
  uint64_t* kp_desc = nullptr;  // [max_images][F][4]
  // matcher
  int32_t* pair_slots = nullptr;   // [max_pairs][2]
  uint32_t* best_key = nullptr;    // [max_pairs][2][F]   (distance << 22 | index), direction 0: a->b
  uint32_t* second_key = nullptr;  // [max_pairs][2][F]
  int32_t* matches = nullptr;      // [max_pairs][F][2]
  int32_t* match_count = nullptr;  // [max_pairs]
  uint32_t* exact_list = nullptr;  // [max_images][VSL_EXACT_CAP]: (keypoint << 8) | bit, see describe.hip
  int32_t* tile_off = nullptr;     // [max_images][tiles + 1]: keypoints of an image by 64 x 64 tile (describe.hip), null when the tile kernel does not apply
  uint32_t* tile_ent = nullptr;    // [max_images][F]: (keypoint << 12) | (y in tile << 6) | x in tile, tile-major
  int tiles_x = 0, tiles = 0;
  uint32_t* sel_grid = nullptr;    // [max_images][cells][3]: selection grid of images too large for LDS (lazy)
  // rBRIEF near-tie records (see describe.hip)
  int32_t* tie_count = nullptr;    // [1]
  int32_t* tie_rec = nullptr;      // [tie_cap][4]  (slot, keypoint, bit, unused)
  int tie_cap = 0;
  bool ties_pending = false, ties_from_angles = false;
  bool detect_meta_dirty = true;    // MAX / NCAND not in their reset state (first use, or a failed launch)
  bool describe_meta_dirty = true;  // same for NEXACT
  // slot ranges described by the fast kernels since the last vsl_resolve_ties, in launch order: an exact-list overflow
  // in ANY of them is redone by the generic f64 kernel there (several asynchronous launches may precede one resolve)
  struct DescRange {
    int first, n, rotate;
  };
  std::vector<DescRange> pending_desc;
  // the guard's count was read as zero (or the descriptors are not handed out): nothing queued is left to verify.
  // Every fast path that skips vsl_resolve_ties ends here, so stale ranges do not pile up frame after frame.
  void ties_settled() {
    ties_pending = false;
    pending_desc.clear();
  }
  int exact_fallbacks = 0;            // how often that fallback ran (diagnostic)
  bool store_response = false;  // K1 writes the fp32 response image only for the parity hook
  std::vector<int32_t> pair_cache;  // host copy of pair_slots (skip the upload when unchanged)
};

int vsl_frames_alloc(vsl_ctx* ctx, int max_images, int w, int h, int F, int max_pairs, vsl_frames** out);

// internal launchers (asynchronous on ctx->stream)
int vsl_launch_detect(vsl_ctx* ctx, vsl_frames* f, int first, int n, int num_features);
int vsl_launch_describe(vsl_ctx* ctx, vsl_frames* f, int first, int n, int rotate_features,
                        int from_angles);
int vsl_launch_match(vsl_ctx* ctx, vsl_frames* f, int n_pairs, int threshold, double dist_2_best, int db_bound);
int vsl_resolve_ties(vsl_ctx* ctx, vsl_frames* f, int* n_resolved);
int vsl_set_pairs(vsl_ctx* ctx, vsl_frames* f, const int32_t* slot_pairs, int n_pairs);

// dense fp64 Cholesky solve on the device (chol.hip): S x = b in place, *ok_dev = 0 if not SPD
int vsl_chol_solve_dev(vsl_ctx* ctx, double* S, double* b, int n, int* ok_dev);
// the same on LAPACK-style lower band storage (chol.hip, "BAND FORM"): S = storage + bws, ld = bws = bw + VSL_CHOL_NB
#define VSL_CHOL_NB 32
int vsl_chol_solve_band_dev(vsl_ctx* ctx, double* S, double* b, int n, int ld, int bw, int* ok_dev, int cyclic = 0, double* neg_out = nullptr);
bool vsl_chol_bcr_cyclic_layout(int n, int bw, int* B_out, int* nblk_out);  // chol.hip: is there a ring of blocks for this cyclic band?

// scratch store of the host-buffer API
int vsl_ctx_scratch_frames(vsl_ctx* ctx, int w, int h, int feat, vsl_frames** out);

// order-preserving float <-> int mapping for atomicMax on fp32 values of any sign
__host__ __device__ inline int32_t vsl_float_to_ordered(float f) {
  int32_t i;
  memcpy(&i, &f, 4);
  return i >= 0 ? i : (int32_t)(i ^ 0x7fffffff);
}
__host__ __device__ inline float vsl_ordered_to_float(int32_t i) {
  int32_t j = i >= 0 ? i : (int32_t)(i ^ 0x7fffffff);
  float f;
  memcpy(&f, &j, 4);
  return f;
}
