/*
 * vslam_hip.h -- C ABI of the MI355X-native (gfx950) visual-SLAM hot path.
 *
 * The reference (yunjinli/visual-slam) has no plugin/FFI layer: its "operator
 * API" is the set of free functions of the include/visnav headers that src/slam.cpp
 * calls directly.  Every entry point below is what a thin C++ wrapper with the
 * reference's own signature binds to (the wrappers live in
 * the include/visnav_amd headers; the reference-side patch is shown in INTEGRATION.md).
 * Each declaration cites the reference interface it replaces (file:line,
 * relative to the reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - unless a function name ends in `_dev`, every pointer is a HOST pointer
 *     owned by the caller, and the call is synchronous w.r.t. its outputs;
 *   - return value: VSL_OK (0) or a negative VSL_ERR_* code; nothing throws;
 *   - a vsl_ctx owns one HIP stream plus its scratch; calls on DIFFERENT
 *     contexts may run concurrently from different host threads (the reference
 *     has up to three threads inside this path: main, opt_thread,
 *     global_ba_thread -- src/slam.cpp:1557, :1780); one context must not be
 *     used from two threads at once;
 *   - there is NO CPU fallback: without a HIP device vsl_ctx_create fails with
 *     VSL_ERR_NO_DEVICE and nothing else can be called.
 *
 * Descriptor layout: 256 bits as 4 x uint64_t, bit i in word i/64 at position
 * i%64 -- the libstdc++ layout of std::bitset<256>
 * (include/visnav/common_types.h:120), so
 * reinterpret_cast<const uint64_t*>(vector<bitset<256>>::data()) is the
 * argument.
 */
#ifndef VSLAM_HIP_H
#define VSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSL_OK 0
#define VSL_ERR_INVALID (-1)   /* bad argument (null pointer, negative size, ...) */
#define VSL_ERR_HIP (-2)       /* a HIP runtime call failed; see vsl_last_error  */
#define VSL_ERR_NOMEM (-3)     /* host or device allocation failed               */
#define VSL_ERR_CAPACITY (-4)  /* caller-provided output capacity too small      */
#define VSL_ERR_NO_DEVICE (-5) /* no HIP device / extension unusable             */
#define VSL_ERR_IO (-6)        /* file could not be read / parsed                */
#define VSL_ERR_NUMERIC (-7)   /* linear solve failed (non-finite / not SPD)     */

typedef struct vsl_ctx vsl_ctx;

/* ---------------------------------------------------------------- context */

const char* vsl_version(void);
/* Human-readable text of the last failure on this context (never NULL). */
const char* vsl_last_error(const vsl_ctx* ctx);
/* Number of HIP devices visible to the process (0 if none / no runtime). */
int vsl_device_count(void);

/* Create a context on `device` with its own non-blocking HIP stream. */
int vsl_ctx_create(int device, vsl_ctx** out);
/* Create a context that enqueues on an existing hipStream_t (e.g. the stream a
 * caller times with its own events).  The stream is borrowed, not owned. */
int vsl_ctx_create_on_stream(int device, void* hip_stream, vsl_ctx** out);
int vsl_ctx_destroy(vsl_ctx* ctx);
int vsl_ctx_synchronize(vsl_ctx* ctx);
/* The hipStream_t this context enqueues on. */
void* vsl_ctx_stream(vsl_ctx* ctx);
/* Device-side ordering between contexts of one device (no host round trip): vsl_event_record marks the work
 * enqueued on `ctx` so far; work enqueued on a context after vsl_ctx_wait_event starts only after the marked
 * work has completed (an event never recorded is not waited for).  The upload / compute hand-off of a streaming
 * pipeline -- the reference's load -> detect order, src/slam.cpp:1122-1128. */
typedef struct vsl_event vsl_event;
int vsl_event_create(vsl_ctx* ctx, vsl_event** out);
int vsl_event_destroy(vsl_event* e);
int vsl_event_record(vsl_event* e, vsl_ctx* ctx);
int vsl_ctx_wait_event(vsl_ctx* ctx, vsl_event* e);

/* Per-stage device timing with hipEvents on the context's stream.  When
 * enabled, every kernel stage is bracketed by events; vsl_ctx_stage_ms returns
 * the accumulated milliseconds and launch count per stage since the last
 * reset (this synchronizes the stream). */
enum {
  VSL_STAGE_RESPONSE = 0, /* K1 min-eigenvalue response + candidates          */
  VSL_STAGE_SELECT = 1,   /* K2 sort + greedy min-distance selection          */
  VSL_STAGE_DESCRIBE = 2, /* K3+K4 orientation + rBRIEF-256                   */
  VSL_STAGE_MATCH = 3,    /* K5 Hamming best/second-best, both directions     */
  VSL_STAGE_MATCH_FIN = 4,/* K5 ratio test + cross-check + ordered emit       */
  VSL_STAGE_BA_LIN = 5,   /* K6 residual + Jacobian blocks                    */
  VSL_STAGE_BA_SCHUR = 6, /* K7 Schur-complement reduction                    */
  VSL_STAGE_BA_SOLVE = 7, /* reduced camera system Cholesky + back-subst.     */
  VSL_STAGE_BOW_TRANSFORM = 8,
  VSL_STAGE_BOW_SCORE = 9,
  VSL_STAGE_BA_FINISH = 10, /* fused local BA: sum of the per-workgroup partials + damping   */
  VSL_STAGE_BA_STEP = 11,   /* fused local BA: back-substitution, model change, candidate    */
  VSL_STAGE_COUNT = 12
};
int vsl_ctx_set_profiling(vsl_ctx* ctx, int enabled);
int vsl_ctx_stage_ms(vsl_ctx* ctx, int stage, double* total_ms, int64_t* launches);
int vsl_ctx_reset_profiling(vsl_ctx* ctx);

/* ------------------------------------------------- keypoints (host buffers) */

/* Replaces visnav::detectKeypointsAndDescriptors (include/visnav/keypoints.h:223-229)
 * = detectKeypoints (:133-150) -> computeAngles (:152-189) -> computeDescriptors
 * (:191-221).  img: 8-bit gray, `pitch` bytes per row.  Outputs, for
 * i < *n_out <= cap: corners_xy[2i], [2i+1] (integer-valued doubles, order =
 * descending corner response), angles[i] (radians), desc[4i..4i+3].
 * Detector semantics = cv::goodFeaturesToTrack(img, num_features, 0.01, 8,
 * noArray(), 3, false) followed by InBounds(x, y, 19). */
int vsl_detect_describe(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch,
                        int num_features, int rotate_features, int cap,
                        double* corners_xy, double* angles, uint64_t* desc, int* n_out);

/* Replaces visnav::detectKeypoints (include/visnav/keypoints.h:133-150). */
int vsl_detect_keypoints(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch,
                         int num_features, int cap, double* corners_xy, int* n_out);

/* Replaces visnav::computeAngles (include/visnav/keypoints.h:152-189).
 * corners_xy: n points (truncated to int like the reference, :159-160). */
int vsl_compute_angles(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch,
                       const double* corners_xy, int n, int rotate_features, double* angles);

/* Replaces visnav::computeDescriptors (include/visnav/keypoints.h:191-221):
 * descriptors from caller-provided corners and angles. */
int vsl_compute_descriptors(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch,
                            const double* corners_xy, const double* angles, int n,
                            uint64_t* desc);

/* Raw K1 output for parity tests: fp32 min-eigenvalue response (w*h floats,
 * dense rows), the quantity cv::cornerMinEigenVal(img, 3, 3) produces inside
 * goodFeaturesToTrack (called at include/visnav/keypoints.h:138). */
int vsl_min_eig_response(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch,
                         float* response);

/* --------------------------------------------------- matcher (host buffers) */

/* Replaces visnav::matchDescriptors (include/visnav/keypoints.h:323-369,
 * helper isPQiffQP :278-313): best / second-best Hamming with strict-<
 * (lowest index wins ties), reject best >= threshold, reject
 * second < best * dist_2_best (double), mutual cross-check with the same
 * tests.  pairs: (i, j) with ascending i, capacity 2*min(n1,n2) int32. */
int vsl_match_descriptors(vsl_ctx* ctx, const uint64_t* d1, int n1, const uint64_t* d2, int n2,
                          int threshold, double dist_2_best, int32_t* pairs, int* n_out);

/* --------------------------- per-frame landmark projection / guided matching */
/* (the "next" callers of the matcher, run on every frame)
 * Replaces visnav::project_landmarks (include/visnav/vo_utils.h:48-81): p_c = T_w_c^-1 * p; kept iff
 * p_c.z >= cam_z_threshold and the projection lies in [0, width] x [0, height].  Landmarks are visited
 * in the given order; proj_idx[i] = index into `points` of the i-th kept landmark (capacity n). */
int vsl_project_landmarks(vsl_ctx* ctx, const double* pose7, int cam_model, const double* intr8,
                          int width, int height, const double* points, int n,
                          double cam_z_threshold, double* proj_uv, int32_t* proj_idx, int* n_out);
/* Replaces visnav::find_matches_landmarks (include/visnav/vo_utils.h:83-167): for every keypoint, the
 * projected landmarks within match_max_dist_2d; landmark distance = min Hamming distance to the
 * landmark's observation descriptors obs_desc[lm_obs_start[l] .. lm_obs_start[l+1]); best / second by
 * std::partial_sort semantics; threshold and ratio tests.  pairs: (keypoint, landmark index) in keypoint
 * order, capacity 2*n_kp. */
int vsl_find_matches_landmarks(vsl_ctx* ctx, const double* kp_xy, const uint64_t* kp_desc, int n_kp,
                               const double* proj_uv, const int32_t* proj_lm, int n_proj,
                               const int32_t* lm_obs_start, int n_lms, const uint64_t* obs_desc,
                               double match_max_dist_2d, int feature_match_threshold,
                               double feature_match_dist_2_best, int32_t* pairs, int* n_out);

/* ------------------------------------- device-resident batched frame store */
/*
 * The throughput path: B images live in HBM; detect/describe runs over a range
 * of slots per call (one launch set for the whole range), stereo / temporal
 * matching runs over a list of (slot_a, slot_b) pairs per call.  Nothing
 * returns to the host until a download call.  All enqueue on ctx's stream.
 */
typedef struct vsl_frames vsl_frames;
typedef struct vsl_map vsl_map;

int vsl_frames_create(vsl_ctx* ctx, int max_images, int w, int h, int max_features,
                      int max_pairs, vsl_frames** out);
int vsl_frames_destroy(vsl_frames* f);
/* Device pointer to slot 0 of the image store: max_images dense w*h u8 images,
 * image k at base + k*w*h.  A caller may fill it directly (e.g. from a device
 * tensor); `_dev` marks the pointer as a device pointer. */
void* vsl_frames_images_dev(vsl_frames* f);
/* Copy n host images (row pitch `pitch`, image stride `img_stride` bytes) into
 * slots [first, first+n), enqueued on ctx's stream (any context of the device: a
 * dedicated upload context overlaps the copy with another context's kernels).  A
 * dense batch (pitch == w, img_stride == w*h) is ONE transfer; it is asynchronous
 * when `imgs` is pinned host memory, and `imgs` must stay valid until the stream
 * has passed the copy. */
int vsl_frames_upload(vsl_ctx* ctx, vsl_frames* f, int first, int n, const uint8_t* imgs,
                      size_t pitch, size_t img_stride);
/* Page-locks / releases a caller-owned host buffer (hipHostRegister): uploads from it are asynchronous DMA transfers
 * instead of staged copies.  Optional; the buffer must stay valid until it is unregistered. */
int vsl_host_register(vsl_ctx* ctx, void* ptr, size_t bytes);
int vsl_host_unregister(vsl_ctx* ctx, void* ptr);
/* detectKeypointsAndDescriptors over slots [first, first+n) (asynchronous). */
int vsl_frames_detect_describe(vsl_ctx* ctx, vsl_frames* f, int first, int n, int num_features,
                               int rotate_features);
/* Exactness guard of the rBRIEF stage (see DESIGN.md "Bit-exact descriptors"): synchronizes and
 * re-evaluates, with the host's libm exactly as include/visnav/keypoints.h:206-214 does, every
 * descriptor bit whose rotated sample coordinate lies within the guard band of a rounding
 * boundary (expected ~3e-6 bits per frame); *n_resolved (nullable) = number of bits checked.
 * Download calls do this implicitly; call it between vsl_frames_detect_describe and
 * vsl_frames_match when the match must see the patched bits. */
int vsl_frames_resolve_ties(vsl_ctx* ctx, vsl_frames* f, int* n_resolved);
/* Diagnostic: how many times this store redid described ranges with the f64 kernel because an image's exact-rounding
 * list overflowed (0 in normal operation; tests shrink "exact_list_cap" to get there). */
int vsl_frames_exact_fallbacks(const vsl_frames* f);
/* Diagnostic knob: width of that guard band (default 1e-12; tests widen it to exercise the path). */
int vsl_ctx_set_tie_eps(vsl_ctx* ctx, double eps);
/* Diagnostic knobs for the parity tests (results never change, only which kernel path produces them):
 *   "match_use_valu" (0/1)          popcount matcher instead of the matrix-core one
 *   "match_use_i8" (0/1)            int8 matrix-core matcher where the block-scaled FP4 one would run (<= 2048 features)
 *   "match_two_pass" (0/1)          FP4 matcher as forward + reverse passes also for launches of fewer than 8 pairs
 *   "match_no_stagger" (0/1)        matrix-core matcher with every wave of a workgroup in the same phase order
 *   "force_generic_describe" (0/1)  f64 describe kernel for every call
 *   "describe_tile_min_images" (default 96) describe launches of at least this many images use the shared-tile
 *                                   kernel (image widths that are multiples of 16); 1 = always, 0 = never
 *   "k1_list_cap" (0..384, -1 = all)         per-wave LDS candidate slots of the response kernel (overflow path)
 *   "chol_no_fused" (0/1)           band Cholesky as one launch per panel step instead of the single-launch kernel
 *   "chol_no_bcr" (0/1)             long narrow bands by the band Cholesky instead of block cyclic reduction
 *   "chol_one_ended" (0/1)          narrow-band Cholesky eliminated from the top only instead of from both ends
 *   "ba_force_dense" (0/1)          large bundle adjustment with the dense reduced camera system (no band ordering)
 *   "ba_schur_atomics" (0/1)        large-system Schur complement by fp64 atomics instead of the per-block gather
 *   "ba_no_fused" (0/1)             local windows (<= 21 free cameras) by the operator-by-operator kernels instead of the
 *                                   fused four-launch iteration (ba_fused.hip)
 *   "ba_schur_entries" (0/1)        small-system Schur kernel with single-entry ownership instead of 3 x 3 sub-blocks
 *   "bow_no_wg_score" (0/1)         L1 scoring of <= 256 candidates by the wave-per-candidate kernel instead of the
 *                                   workgroup-per-candidate one
 *   "bow_keys64" (0/1)              vocabulary transform with 64-bit (id, feature) sort keys where 32 bits would do
 *   "exact_list_cap" (0..16384)     per-image exact-rounding list entries of the describe kernels; an overflow is
 *                                   detected at the next synchronisation and the range is redone by the f64 kernel
 *   "vo_chain_ticket" (0/1)         vsl_map_track: chain positions of the projection workgroups from an atomic ticket
 *                                   (the rule for maps above 262 k landmarks) at every map size
 *   "pending_desc_max" (default 64) describe launches a frame store queues without a resolve before it resolves them
 *                                   itself (the launch that crosses the limit included)
 *   "select_bucket_cap" (default 128) fullest response bin the selection kernel's counting sort accepts; 0 = always
 *                                   the bitonic network */
int vsl_ctx_set_diagnostic(vsl_ctx* ctx, const char* name, int value);

/* matchDescriptors for n_pairs (slot_a, slot_b) pairs; slot_pairs is a HOST
 * array of 2*n_pairs slot indices; results land in pair slots [0, n_pairs)
 * (asynchronous). */
int vsl_frames_match(vsl_ctx* ctx, vsl_frames* f, const int32_t* slot_pairs, int n_pairs,
                     int threshold, double dist_2_best);
/* Synchronize and copy one image's keypoints out (same layout as
 * vsl_detect_describe). */
int vsl_frames_download_keypoints(vsl_ctx* ctx, vsl_frames* f, int slot, int cap,
                                  double* corners_xy, double* angles, uint64_t* desc, int* n_out);
/* Synchronize and copy one pair's match list out (capacity 2*max_features). */
int vsl_frames_download_matches(vsl_ctx* ctx, vsl_frames* f, int pair, int cap_pairs,
                                int32_t* pairs, int* n_out);
/* Synchronize and copy only the per-image keypoint counts / per-pair match
 * counts (cheap completeness check for benchmarks). */
int vsl_frames_download_counts(vsl_ctx* ctx, vsl_frames* f, int n_images, int32_t* n_keypoints,
                               int n_pairs, int32_t* n_matches);

/* Diagnostic: number of corner candidates (3x3 local maxima above the quality threshold) each of
 * the first n_images slots produced in its last detect call -- the input size of the selection
 * stage, needed to price its traffic. */
int vsl_frames_download_candidate_counts(vsl_ctx* ctx, vsl_frames* f, int n_images, int32_t* n_candidates);

/* Diagnostic: the response kernel evaluates sqrt as rsq + one fused correction step; this compares that sequence
 * with the correctly rounded sqrtf on EVERY float bit pattern in [lo_bits, hi_bits] (non-negative floats) on the
 * device and returns the number of differing results (0 over [2^-100, FLT_MAX] and at 0 is what the kernel relies
 * on; the GPU tests run the whole range). */
int vsl_diag_sqrt_check(vsl_ctx* ctx, uint32_t lo_bits, uint32_t hi_bits, unsigned long long* n_mismatch);

/* ------------------------------------------------ device-resident map (per-frame tracking) */
/*
 * The single-stream path: what project_landmarks + find_matches_landmarks (include/visnav/vo_utils.h:48-167)
 * need stays in HBM between frames -- landmark positions, each landmark's list of observation descriptors,
 * and the descriptor pool -- and a frame whose keypoints are already in a frame-store slot is tracked
 * against it with ONE call (projection, compaction and guided matching chained on the stream); only the
 * matches return.  Results are those of vsl_project_landmarks followed by vsl_find_matches_landmarks on
 * the same landmark order.
 *   pool:   descriptors of keyframe observations, append-only; appended from the host or copied
 *           device-to-device out of a frame-store slot (feature_ids index that slot's keypoints).
 *   table:  vsl_map_set_landmarks replaces the landmark table: n points (3 doubles each, in the order the
 *           caller iterates its map -- that order is the order of the reference's loop, vo_utils.h:60) and
 *           a CSR list of pool indices per landmark (obs_start[n+1], obs_pool_index[obs_start[n]]).
 *   track:  pairs = (feature id in the slot, landmark index in the table), ascending feature id, capacity
 *           2 * max_features int32; *n_projected (nullable) = landmarks that passed the visibility test.
 */
int vsl_map_create(vsl_ctx* ctx, int cap_landmarks, int cap_descriptors, vsl_map** out);
void vsl_map_destroy(vsl_map* map);
int vsl_map_append_descriptors(vsl_map* map, int n, const uint64_t* desc, int* first_index);
int vsl_map_append_descriptors_from_frame(vsl_map* map, vsl_frames* f, int slot, int n, const int32_t* feature_ids,
                                          int* first_index);
int vsl_map_set_landmarks(vsl_map* map, int n, const double* points, const int32_t* obs_start,
                          const int32_t* obs_pool_index);
int vsl_map_info(const vsl_map* map, int* n_landmarks, int* n_observation_refs, int* n_descriptors);
int vsl_map_track(vsl_map* map, vsl_frames* f, int slot, const double* pose7, int cam_model, const double* intr8, int width,
                  int height, double cam_z_threshold, double match_max_dist_2d, int feature_match_threshold,
                  double feature_match_dist_2_best, int32_t* pairs, int* n_pairs, int* n_projected);
/* The same call, which also hands back the slot's keypoint positions (corners_xy: 2 * max_features doubles, *n_corners)
 * in the same round trip -- the positions the caller's PnP needs, without a second download. */
int vsl_map_track_corners(vsl_map* map, vsl_frames* f, int slot, const double* pose7, int cam_model, const double* intr8,
                          int width, int height, double cam_z_threshold, double match_max_dist_2d, int feature_match_threshold,
                          double feature_match_dist_2_best, int32_t* pairs, int* n_pairs, int* n_projected, double* corners_xy,
                          int* n_corners);

/* -------------------------------------------------------- bundle adjustment */
/*
 * Replaces visnav::bundle_adjustment (include/visnav/map_utils.h:337-421) and
 * visnav::global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748):
 * the caller flattens Cameras / Landmarks / Corners to the arrays below
 * (include/visnav_amd/bundle_adjustment.h does that for the reference types).
 * Cost functor = BundleAdjustmentReprojectionCostFunctor
 * (include/visnav/reprojection.h:81-105) with the camera models of
 * include/visnav/camera_models.h, pose update = T * exp(delta)
 * (include/visnav/local_parameterization_se3.hpp:43-50), loss = Huber
 * (map_utils.h:384-387), solver = Levenberg-Marquardt with Schur elimination
 * of the landmark blocks (map_utils.h:406-411: SPARSE_SCHUR, 20 iterations).
 */
enum { VSL_CAM_DS = 0, VSL_CAM_PINHOLE = 1, VSL_CAM_EUCM = 2, VSL_CAM_KB4 = 3 };

typedef struct vsl_ba_problem {
  int32_t n_cams;           /* camera poses (left and right are separate blocks)   */
  int32_t n_lms;            /* landmarks                                           */
  int32_t n_obs;            /* observations = residual blocks                      */
  int32_t cam_model[2];     /* VSL_CAM_* of intrinsics[0], intrinsics[1]           */
  double* poses;            /* [7*n_cams] qx qy qz qw tx ty tz (T_w_c), in/out     */
  const uint8_t* cam_fixed; /* [n_cams] 1 = SetParameterBlockConstant              */
  const int32_t* cam_intr;  /* [n_cams] 0/1: which intrinsics block (fcid.cam_id)  */
  const double* intr;       /* [16] two 8-vectors fx fy cx cy p1..p4 (constant)    */
  double* points;           /* [3*n_lms] in/out                                    */
  const int32_t* obs_cam;   /* [n_obs]                                             */
  const int32_t* obs_lm;    /* [n_obs]                                             */
  const double* obs_uv;     /* [2*n_obs] detected corner (p_2d)                    */
} vsl_ba_problem;

typedef struct vsl_ba_options {
  int32_t use_huber;          /* BundleAdjustmentOptions::use_huber (map_utils.h:326) */
  double huber_parameter;     /* :329, pixels                                         */
  int32_t max_num_iterations; /* :332                                                 */
  int32_t verbosity;          /* 0 silent, 1 one line, 2 per-iteration table (stderr) */
} vsl_ba_options;

typedef struct vsl_ba_summary {
  double initial_cost, final_cost;
  int32_t iterations;            /* LM iterations run (successful + unsuccessful) */
  int32_t successful_steps;
  int32_t termination;           /* 0 no-convergence(max iters) 1 function tol 2 gradient tol
                                    3 parameter tol 4 failure                      */
  double linearize_ms, schur_ms, solve_ms; /* device time per stage, summed; 0 unless vsl_ctx_set_profiling(ctx, 1) */
  double total_ms;                         /* host wall time of the call */
} vsl_ba_summary;

int vsl_bundle_adjust(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt,
                      vsl_ba_summary* summary);

/* One linearization at the current poses/points, for parity tests and for the
 * multi-GPU global-BA path (each rank reduces its landmark range; S and g are
 * then summed with an all-reduce).  Cameras are numbered by free-camera index
 * c (fixed cameras skipped, ascending camera id); n_free = #free cameras.
 *   S  [(6*n_free)^2] row-major, = sum_c F^T F  -  sum_l (F^T E) (E^T E)^-1 (E^T F)
 *   g  [6*n_free]             , = sum F^T r     -  sum_l (F^T E) (E^T E)^-1 (E^T r)
 *   cost = 1/2 sum rho(|r|^2)
 * computed WITHOUT Jacobi scaling or LM damping (mu = 0), with the Huber
 * corrector applied (residual and Jacobian scaled by sqrt(rho')).
 * lm_first/lm_count restrict the reduction to landmarks [lm_first,
 * lm_first+lm_count) and their observations (lm_count < 0: all). */
int vsl_ba_linearize(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt,
                     int lm_first, int lm_count, double* S, double* g, double* cost, int* n_free);

/* Residual r (2) and Jacobian blocks J_pose (2x6 row-major, tangent
 * (upsilon, omega) of T*exp(delta)) and J_point (2x3) per observation, raw
 * (no robust scaling) -- parity hook for K6. */
int vsl_ba_residuals_jacobians(vsl_ctx* ctx, const vsl_ba_problem* prob, double* r, double* J_pose,
                               double* J_point);

/* ---- step-wise session: multi-GPU global bundle adjustment ----
 * One process per GPU.  Every rank holds all camera poses and owns the landmark range
 * [lm_first, lm_first + lm_count) with its observations; per LM iteration the ranks SUM-all-reduce
 * the packed partial reduced camera system and five scalars (RCCL over xGMI via torch.distributed;
 * the loop is visual-slam_amd/ba_dist.py).  `_dev` arguments are DEVICE pointers; those calls are
 * asynchronous on the context's stream.  Replaces, for ~500-keyframe maps, the single Ceres solve of
 * global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748). */
typedef struct vsl_ba_session vsl_ba_session;
int vsl_ba_session_create(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt,
                          int lm_first, int lm_count, vsl_ba_session** out);
int vsl_ba_session_destroy(vsl_ba_session* s);
/* n = 6 * free cameras; owned landmarks / observations; cameras. */
int vsl_ba_session_dims(const vsl_ba_session* s, int* n, int* n_lms_own, int* n_obs_own, int* n_cams);
/* use_scale = 0: the Jacobi-scaling pass (column norms, cost); 1: after an accepted step.  Large systems run in the
 * recompute form (visual-slam_amd/csrc/ba_large.h: nothing is stored per observation) -- there the call with
 * use_scale = 1 has nothing to do, vsl_ba_session_reduce_dev evaluates the observations at the current point itself.
 * Diagnostic "ba_no_fused" / VSL_BA_NO_FUSED: the chain over stored residual / Jacobian blocks instead. */
int vsl_ba_session_linearize(vsl_ba_session* s, int use_scale);
/* out_dev[n + 1] = [diag(H_part) | cost_part] */
int vsl_ba_session_hdiag_cost_dev(vsl_ba_session* s, double* out_dev);
int vsl_ba_session_set_scale_dev(vsl_ba_session* s, const double* hdiag_full_dev);
/* packB_dev[n*n + 3n + 2] = [S_part | rhs_part | diag(H_part) | g_c part | cost_part | 0];
 * gmax_l_dev[1] (nullable) = max |gradient| over the owned landmark columns */
int vsl_ba_session_reduce_dev(vsl_ba_session* s, double radius, double* packB_dev, double* gmax_l_dev);
/* packC_dev[8] = [bad, model_part, step2, x2, cand_cost_part, step2_cams, x2_cams, 0] */
int vsl_ba_session_step_dev(vsl_ba_session* s, const double* packB_full_dev, double radius,
                            int refresh_diag, double* packC_dev);
int vsl_ba_session_accept(vsl_ba_session* s);
int vsl_ba_session_download(vsl_ba_session* s, double* poses, double* points_own);
/* Layout of the reduced camera system inside packB: *s_elems doubles -- n * n when dense; in band form (cameras
 * renumbered into a narrow band by reverse Cuthill-McKee on the covisibility graph of the FULL problem, identically
 * on every rank; *banded = 1) or in CYCLIC band form (cameras as they came, the band closes around the loop -- the wrap
 * blocks sit in the leading slots of the first rows; *banded = 2) n * (bandwidth + 33) + 64 -- followed by rhs / diag H /
 * g_c (n each), cost, 0.  Wherever this header says "n*n" for packB read *s_elems. */
int vsl_ba_session_layout(const vsl_ba_session* s, int64_t* s_elems, int* banded, int* bandwidth);

/* The whole Levenberg-Marquardt loop over a session, host code in C++.  Collectives go through ONE caller-supplied
 * function: in-place all-reduce of `count` doubles at DEVICE pointer buf, op 0 = SUM, 1 = MAX, ordered on hip_stream
 * (the context's stream: an RCCL caller enqueues ncclAllReduce on it and returns; a host-hopping caller synchronises
 * it first); returns 0 on success.  world = 1: allreduce may be NULL.  poses_out [7 * n_cams] and points_all_out
 * [3 * n_lms of the FULL problem] (host, nullable) receive the result on every rank. */
typedef int (*vsl_allreduce_fn)(void* user, double* buf, int64_t count, int op, void* hip_stream);
int vsl_ba_session_solve(vsl_ba_session* s, vsl_allreduce_fn allreduce, void* user, int world, int max_iters,
                         int verbosity, double* poses_out, double* points_all_out, vsl_ba_summary* summary);
/* visnav::global_bundle_adjustment over `world` ranks (one process per GPU): landmark ranges balanced by observation
 * count, rank `rank` owns one, prob->poses / prob->points updated in place on every rank. */
int vsl_global_bundle_adjust(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt,
                             vsl_allreduce_fn allreduce, void* user, int rank, int world, vsl_ba_summary* summary);
/* Diagnostic: storage of the reduced camera system in the last solve the general path (vsl_bundle_adjust beyond the
 * local window, sessions, vsl_global_bundle_adjust) set up on this context: doubles of S, *banded = 0 dense / 1 band (cameras
 * in reverse Cuthill-McKee order) / 2 cyclic band (cameras as they came, the band closes around the loop), bandwidth.
 * Diagnostic "ba_no_cyclic" / VSL_BA_NO_CYCLIC keep form 1 where form 2 would be taken. */
int vsl_ctx_last_ba_layout(vsl_ctx* ctx, int64_t* s_elems, int* banded, int* bandwidth);
/* Plain synchronous copy on the context's device (kind 0 host->device, 1 device->host, 2 device->device): for
 * callers that hold device pointers handed out by this library (all-reduce callbacks). */
int vsl_ctx_memcpy(vsl_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
/* visnav::bundle_adjustment with BundleAdjustmentOptions::optimize_intrinsics = true (include/visnav/map_utils.h:324,
 * :397-403: the two intrinsics blocks are not set constant; wired to a hidden GUI variable, src/slam.cpp:304, :1545):
 * poses, landmarks AND the two 8-parameter intrinsics blocks are optimised jointly (same restated Ceres policy; the
 * parameters a model does not use keep their values).  intr_io [16] in/out; prob->intr is ignored. */
int vsl_bundle_adjust_intrinsics(vsl_ctx* ctx, const vsl_ba_problem* prob, const vsl_ba_options* opt, double* intr_io,
                                 vsl_ba_summary* summary);
/* The solver behind the reduced camera system, on its own (the place Ceres' SPARSE_SCHUR hands S to a Cholesky,
 * include/visnav/map_utils.h:406-411, loop_closure_utils.h:735): solves S x = b for a symmetric positive definite S
 * given as a HOST row-major n x n array of which only the lower triangle is read.  half_bandwidth < 0: dense
 * factorisation; otherwise S is taken to be zero for |i - j| > half_bandwidth and is factorised in band storage
 * (single-launch kernel for half_bandwidth <= 512 unless the "chol_no_fused" diagnostic is set).  x (host, n doubles)
 * receives the solution.  VSL_ERR_NUMERIC if S is not positive definite. */
int vsl_spd_solve(vsl_ctx* ctx, const double* S, const double* b, int n, int half_bandwidth, double* x);
/* The same for a CYCLIC band (non-zeros where min(|i - j|, n - |i - j|) <= half_bandwidth: the reduced camera system of a
 * closed loop ordered along the trajectory): block cyclic reduction over a ring of blocks.  VSL_ERR_INVALID when the system
 * has no such block layout (fewer than 8 blocks of >= half_bandwidth + 1 unknowns). */
int vsl_spd_solve_cyclic(vsl_ctx* ctx, const double* S, const double* b, int n, int half_bandwidth, double* x);
/* Host-only (no device needed): 1 and the ring layout -- *block unknowns per solver block (a multiple of 32, <= 256),
 * *n_blocks >= 8 blocks of floor / ceil (n / *n_blocks) >= half_bandwidth + 1 unknowns each -- when the cyclic solver takes
 * such a system, 0 when it does not (the bundle adjustment then keeps the linear band form). */
int vsl_bcr_cyclic_layout(int n, int half_bandwidth, int* block, int* n_blocks);

/* ------------------------------------------------------------ pose graph optimisation */
/*
 * The numerical core of visnav::pose_graph_optimization (include/visnav/loop_closure_utils.h:446-587): one
 * residual block r = log(T_w_c^-1 * T_w_n) - upsilon_omega per edge (PoseGraphRelativePoseCostFunctor,
 * include/visnav/reprojection.h:107-126) on SE3 blocks with the tangent parameterisation T * exp(delta),
 * HuberLoss(huber_parameter), Levenberg-Marquardt with the same restated Ceres policy as vsl_bundle_adjust
 * (options: use_huber, huber_parameter, max_num_iterations, verbosity; summary: costs, iterations,
 * termination).  Which edges exist (spanning tree, covisibility above the essential threshold, the loop
 * constraint) is the caller's graph bookkeeping; the reference fixes the current keyframe when
 * LoopClosureOptions::set_current_kf_fixed is set -> node_fixed.
 */
typedef struct vsl_pgo_problem {
  int32_t n_nodes, n_edges;
  double* poses;             /* [n_nodes][7] qx qy qz qw tx ty tz (T_w_c), optimised in place */
  const uint8_t* node_fixed; /* [n_nodes] */
  const int32_t* edge_a;     /* [n_edges] the functor's T_w_c */
  const int32_t* edge_b;     /* [n_edges] the functor's T_w_n */
  const double* edge_meas;   /* [n_edges][6] upsilon, omega */
} vsl_pgo_problem;
/* The normal equations are kept dense, or -- when the graph is narrow in the nodes' own order (keyframes in time order:
 * odometry + covisibility edges, and the loop edge that closes the ring) -- in linear / cyclic band storage and solved by
 * the band / ring solvers of the reduced camera system (diagnostic "ba_force_dense" / VSL_PGO_DENSE: always dense). */
int vsl_pose_graph_optimize(vsl_ctx* ctx, const vsl_pgo_problem* prob, const vsl_ba_options* opt, vsl_ba_summary* summary);
/* Test hook: H = J^T J (n x n row-major, n = 6 x free nodes in node order), g = J^T r and the cost of the
 * robustified problem at the given poses. */
int vsl_pgo_linearize(vsl_ctx* ctx, const vsl_pgo_problem* prob, const vsl_ba_options* opt, double* H, double* g, double* cost,
                      int* n_free);

/* --------------------------------------------------------------- DBoW2 path */
/*
 * Replaces, for loop-closure candidate scoring:
 *   ORBVocabulary::loadFromTextFile (thirdparty/DBoW2_ORBSLAM/DBoW2/TemplatedVocabulary.h:1338-1424)
 *   ORBVocabulary::transform        (TemplatedVocabulary.h:1127-1194, :1218-1259; FORB::distance FORB.cpp:81-101)
 *   ORBVocabulary::score            (TemplatedVocabulary.h:1199-1203 -> L1Scoring::score, ScoringObject.cpp:23-68)
 * Only TF_IDF weighting + L1 scoring (what ORBvoc.txt declares and the only
 * combination the reference exercises) is implemented; other header values
 * make vsl_voc_load_text fail with VSL_ERR_INVALID.
 */
typedef struct vsl_voc vsl_voc;

int vsl_voc_load_text(vsl_ctx* ctx, const char* path, vsl_voc** out);
int vsl_voc_destroy(vsl_voc* voc);
/* k, L, number of nodes (root included), number of words. */
int vsl_voc_info(const vsl_voc* voc, int* k, int* L, int* n_nodes, int* n_words);

/* desc32: n descriptors of 32 bytes in DBoW2/cv::Mat byte order (MSB-first
 * within each byte, include/visnav/converter.h:23-33).  Outputs: the BowVector
 * as (word_ids ascending, word_vals L1-normalised) with *nnz entries
 * (capacity n), and the FeatureVector flattened as (fv_node[i], fv_feat[i])
 * sorted by (node, feature index), *fv_n entries (capacity n). */
int vsl_bow_transform(vsl_ctx* ctx, const vsl_voc* voc, const uint8_t* desc32, int n, int levelsup,
                      uint32_t* word_ids, double* word_vals, int* nnz, uint32_t* fv_node,
                      uint32_t* fv_feat, int* fv_n);

/* The ORB front end of compute_bow_vector (include/visnav/keypoints.h:243-254:
 * cv::ORB::create(num_features, 1.2, 8, 19, 0, 2, cv::ORB::FAST_SCORE)->detectAndCompute).  cv::ORB is upstream
 * OpenCV: parity with its binary is unpinned; the arithmetic conventions are those written down in
 * oracle/orc_orb.cpp, against which this is bit-exact.  kp5: (x, y in level-0 pixels, angle in degrees,
 * response, octave) per keypoint, desc32: 32 bytes per keypoint in cv::Mat order; cap >= 2 * num_features + 512
 * is always enough (retainBest keeps every keypoint tied with the last one of a level). */
int vsl_orb_detect_describe(vsl_ctx* ctx, const uint8_t* img, int w, int h, size_t pitch, int num_features, int cap,
                            float* kp5, uint8_t* desc32, int* n_out);
/* compute_bow_vector: that front end followed by vsl_bow_transform (outputs as there, capacity cap). */
int vsl_compute_bow_vector(vsl_ctx* ctx, const vsl_voc* voc, const uint8_t* img, int w, int h, size_t pitch,
                           int num_features, int levelsup, int cap, uint32_t* word_ids, double* word_vals, int* nnz,
                           uint32_t* fv_node, uint32_t* fv_feat, int* fv_n);

/* L1 score of one query BowVector against m candidates given in CSR form
 * (c_offsets[m+1]); ids ascending within each vector. */
int vsl_bow_score_batch(vsl_ctx* ctx, const uint32_t* q_ids, const double* q_vals, int q_nnz,
                        const uint32_t* c_ids, const double* c_vals, const int32_t* c_offsets,
                        int m, double* scores);

/* A device-resident database of BowVectors -- what the reference keeps per keyframe in Camera::bow_vector
 * (include/visnav/common_types.h:204-221) and scores one pair at a time with voc->score
 * (loop_closure_utils.h:119, :201; tracking.h:208).  Vectors are appended once (word ids strictly ascending,
 * *index_out = position); vsl_bowdb_score scores one query against the vectors cand_index[0..m) (null: vectors
 * 0..m-1) with no candidate bytes crossing PCIe.  Same arithmetic and summation order as vsl_bow_score_batch. */
typedef struct vsl_bowdb vsl_bowdb;
int vsl_bowdb_create(vsl_ctx* ctx, int64_t cap_entries, int cap_vectors, vsl_bowdb** out);
int vsl_bowdb_destroy(vsl_bowdb* db);
int vsl_bowdb_append(vsl_ctx* ctx, vsl_bowdb* db, const uint32_t* ids, const double* vals, int nnz, int* index_out);
int vsl_bowdb_info(const vsl_bowdb* db, int* n_vectors, int64_t* n_entries);
int vsl_bowdb_score(vsl_ctx* ctx, const vsl_bowdb* db, const uint32_t* q_ids, const double* q_vals, int q_nnz,
                    const int32_t* cand_index, int m, double* scores);

/* Bit-order converters of include/visnav/converter.h:23-33 and :50-61
 * (bitset<256> word layout <-> 32-byte MSB-first row).  Pure host helpers. */
void vsl_desc_bitset_to_bytes(const uint64_t* desc, int n, uint8_t* desc32);
void vsl_desc_bytes_to_bitset(const uint8_t* desc32, int n, uint64_t* desc);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_H */
