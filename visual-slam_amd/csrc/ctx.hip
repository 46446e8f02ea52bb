// ctx.hip -- context, error reporting, stage timing, device frame store allocation and transfers.
#include <cmath>

#include "vsl_common.h"

static thread_local char g_noctx_err[512] = "";

int vsl_fail(vsl_ctx* ctx, int code, const char* fmt, ...) {
  char* dst = ctx ? ctx->err : g_noctx_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* vsl_version(void) { return "vslam_hip 0.1 (gfx950)"; }

extern "C" const char* vsl_last_error(const vsl_ctx* ctx) { return ctx ? ctx->err : g_noctx_err; }

extern "C" int vsl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int ctx_create_impl(int device, void* stream, bool borrow, vsl_ctx** out) {
  if (!out) return vsl_fail(nullptr, VSL_ERR_INVALID, "vsl_ctx_create: out is null");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return vsl_fail(nullptr, VSL_ERR_NO_DEVICE, "no HIP device visible (there is no CPU fallback)");
  if (device < 0 || device >= n) return vsl_fail(nullptr, VSL_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return vsl_fail(nullptr, VSL_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return vsl_fail(nullptr, VSL_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  vsl_ctx* c = new (std::nothrow) vsl_ctx;
  if (!c) return vsl_fail(nullptr, VSL_ERR_NOMEM, "out of host memory");
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return vsl_fail(nullptr, VSL_ERR_HIP, "hipSetDevice(%d) failed", device);
  }
  if (borrow) {
    c->stream = (hipStream_t)stream;
    c->owns_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return vsl_fail(nullptr, VSL_ERR_HIP, "hipStreamCreate failed");
    }
    c->owns_stream = true;
  }
  // the flag word of rank status exchanges lives as long as the context: a rank whose later allocations fail can still
  // enter the agreement collective with a valid buffer (ba.hip, vsl_global_bundle_adjust)
  if (hipMalloc((void**)&c->status_word, 64) != hipSuccess) {
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return vsl_fail(nullptr, VSL_ERR_NOMEM, "hipMalloc of the context's status word failed");
  }
  *out = c;
  return VSL_OK;
}

extern "C" int vsl_ctx_create(int device, vsl_ctx** out) { return ctx_create_impl(device, nullptr, false, out); }

extern "C" int vsl_ctx_create_on_stream(int device, void* hip_stream, vsl_ctx** out) {
  return ctx_create_impl(device, hip_stream, true, out);
}

extern "C" int vsl_ctx_destroy(vsl_ctx* ctx) {
  if (!ctx) return VSL_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) vsl_frames_destroy(ctx->scratch);
  for (auto& p : ctx->pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->dscratch) (void)hipFree(ctx->dscratch);
  if (ctx->status_word) (void)hipFree(ctx->status_word);
  if (ctx->bcr_jobs) (void)hipFree(ctx->bcr_jobs);
  if (ctx->ba_arena) (void)hipFree(ctx->ba_arena);
  if (ctx->ba_pin) (void)hipHostFree(ctx->ba_pin);
  if (ctx->hpinned) (void)hipHostFree(ctx->hpinned);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return VSL_OK;
}

extern "C" int vsl_ctx_synchronize(vsl_ctx* ctx) {
  if (!ctx) return VSL_ERR_INVALID;
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}

extern "C" void* vsl_ctx_stream(vsl_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int vsl_ctx_dscratch(vsl_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->dscratch_cap) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->dscratch) (void)hipFree(ctx->dscratch);
    ctx->dscratch = nullptr;
    ctx->dscratch_cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    VSL_HIP(ctx, hipMalloc(&ctx->dscratch, cap));
    ctx->dscratch_cap = cap;
  }
  *out = ctx->dscratch;
  return VSL_OK;
}

int vsl_ctx_hpinned(vsl_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->hpinned_cap) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->hpinned) (void)hipHostFree(ctx->hpinned);
    ctx->hpinned = nullptr;
    ctx->hpinned_cap = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    VSL_HIP(ctx, hipHostMalloc(&ctx->hpinned, cap, hipHostMallocDefault));
    ctx->hpinned_cap = cap;
  }
  *out = ctx->hpinned;
  return VSL_OK;
}

// ------------------------------------------------------------------------------ stage profiling
VslStage::VslStage(vsl_ctx* c, int stage) : ctx(c) {
  if (!c->profiling) return;
  vsl_ctx::StageEv ev;
  ev.stage = stage;
  for (hipEvent_t* e : {&ev.a, &ev.b}) {
    if (!c->ev_pool.empty()) {
      *e = c->ev_pool.back();
      c->ev_pool.pop_back();
    } else if (hipEventCreate(e) != hipSuccess) {
      return;
    }
  }
  (void)hipEventRecord(ev.a, c->stream);
  c->pending.push_back(ev);
  idx = (int)c->pending.size() - 1;
}

VslStage::~VslStage() {
  if (idx >= 0) (void)hipEventRecord(ctx->pending[idx].b, ctx->stream);
}

static int drain_profiling(vsl_ctx* ctx) {
  if (ctx->pending.empty()) return VSL_OK;
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->stage_ms[p.stage] += ms;
      ctx->stage_launches[p.stage] += 1;
    }
    ctx->ev_pool.push_back(p.a);
    ctx->ev_pool.push_back(p.b);
  }
  ctx->pending.clear();
  return VSL_OK;
}

extern "C" int vsl_ctx_set_profiling(vsl_ctx* ctx, int enabled) {
  if (!ctx) return VSL_ERR_INVALID;
  int rc = drain_profiling(ctx);
  ctx->profiling = enabled != 0;
  return rc;
}

extern "C" int vsl_ctx_stage_ms(vsl_ctx* ctx, int stage, double* total_ms, int64_t* launches) {
  if (!ctx || stage < 0 || stage >= VSL_STAGE_COUNT) return VSL_ERR_INVALID;
  int rc = drain_profiling(ctx);
  if (rc) return rc;
  if (total_ms) *total_ms = ctx->stage_ms[stage];
  if (launches) *launches = ctx->stage_launches[stage];
  return VSL_OK;
}

extern "C" int vsl_ctx_reset_profiling(vsl_ctx* ctx) {
  if (!ctx) return VSL_ERR_INVALID;
  int rc = drain_profiling(ctx);
  for (int i = 0; i < VSL_STAGE_COUNT; i++) {
    ctx->stage_ms[i] = 0;
    ctx->stage_launches[i] = 0;
  }
  return rc;
}

// ---------------------------------------------------------------------------------- frame store
template <class T>
static hipError_t dalloc(T** p, size_t n) {
  return hipMalloc((void**)p, n * sizeof(T));
}

int vsl_frames_alloc(vsl_ctx* ctx, int max_images, int w, int h, int F, int max_pairs, vsl_frames** out) {
  if (!ctx || !out) return VSL_ERR_INVALID;
  *out = nullptr;
  if (max_images <= 0 || w < 40 || h < 40 || w > 65535 || h > 65535 || F <= 0 || max_pairs < 0 || (int64_t)w * h >= (1ll << 31) ||
      F >= (1 << 22))  // (corner candidates carry their position as y << 16 | x)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_create: bad sizes images=%d w=%d h=%d F=%d pairs=%d", max_images, w, h, F, max_pairs);
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_frames* f = new (std::nothrow) vsl_frames;
  if (!f) return vsl_fail(ctx, VSL_ERR_NOMEM, "out of host memory");
  f->device = ctx->device;
  f->max_images = max_images;
  f->w = w;
  f->h = h;
  f->F = F;
  f->max_pairs = max_pairs;
  f->cand_cap = (size_t)w * h;
  f->tie_cap = 4096;
  const size_t px = (size_t)w * h, M = (size_t)max_images, P = (size_t)(max_pairs > 0 ? max_pairs : 1);
  bool ok = true;
  ok = ok && dalloc(&f->images, M * px) == hipSuccess;
  ok = ok && dalloc(&f->response, M * px) == hipSuccess;
  ok = ok && dalloc(&f->meta, M * VSL_META_STRIDE) == hipSuccess;
  ok = ok && dalloc(&f->cand, M * f->cand_cap) == hipSuccess;
  ok = ok && dalloc(&f->kp_xy, M * F * 2) == hipSuccess;
  ok = ok && dalloc(&f->kp_count, M) == hipSuccess;
  ok = ok && dalloc(&f->kp_moments, M * F * 2) == hipSuccess;
  ok = ok && dalloc(&f->kp_angle, M * F) == hipSuccess;
  ok = ok && dalloc(&f->kp_desc, M * F * 4) == hipSuccess;
  ok = ok && dalloc(&f->pair_slots, P * 2) == hipSuccess;
  ok = ok && dalloc(&f->best_key, P * 2 * F) == hipSuccess;
  ok = ok && dalloc(&f->second_key, P * 2 * F) == hipSuccess;
  ok = ok && dalloc(&f->matches, P * F * 2) == hipSuccess;
  ok = ok && dalloc(&f->match_count, P) == hipSuccess;
  ok = ok && dalloc(&f->exact_list, M * VSL_EXACT_CAP) == hipSuccess;
  // keypoint lists by 64 x 64 tile for the batched describe kernel (16-byte image segments, <= 1024 tiles in LDS counters)
  const int tiles_x = (w + (1 << VSL_TILE_LX) - 1) >> VSL_TILE_LX, tiles_y = (h + (1 << VSL_TILE_LY) - 1) >> VSL_TILE_LY;
  if (w % 16 == 0 && F < (1 << (32 - VSL_TILE_LX - VSL_TILE_LY)) && tiles_x * tiles_y <= 1024) {
    f->tiles_x = tiles_x;
    f->tiles = tiles_x * tiles_y;
    ok = ok && dalloc(&f->tile_off, M * (size_t)(f->tiles + 1)) == hipSuccess;
    ok = ok && dalloc(&f->tile_ent, M * F) == hipSuccess;
  }
  ok = ok && dalloc(&f->tie_count, 1) == hipSuccess;
  ok = ok && dalloc(&f->tie_rec, (size_t)f->tie_cap * 4) == hipSuccess;
  if (!ok) {
    vsl_frames_destroy(f);
    return vsl_fail(ctx, VSL_ERR_NOMEM, "device allocation failed for frame store (%d images %dx%d)", max_images, w, h);
  }
  (void)hipMemsetAsync(f->kp_count, 0, M * sizeof(int32_t), ctx->stream);
  (void)hipMemsetAsync(f->match_count, 0, P * sizeof(int32_t), ctx->stream);
  (void)hipMemsetAsync(f->tie_count, 0, sizeof(int32_t), ctx->stream);
  (void)hipMemsetAsync(f->meta, 0, M * VSL_META_STRIDE * sizeof(int32_t), ctx->stream);
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *out = f;
  return VSL_OK;
}

extern "C" int vsl_frames_create(vsl_ctx* ctx, int max_images, int w, int h, int max_features, int max_pairs,
                                 vsl_frames** out) {
  return vsl_frames_alloc(ctx, max_images, w, h, max_features, max_pairs, out);
}

extern "C" int vsl_frames_destroy(vsl_frames* f) {
  if (!f) return VSL_OK;
  (void)hipSetDevice(f->device);
  (void)hipDeviceSynchronize();
  void* ptrs[] = {f->images, f->response, f->meta, f->cand, f->kp_xy, f->kp_count,
                  f->kp_moments, f->kp_angle, f->kp_desc, f->pair_slots, f->best_key, f->second_key,
                  f->matches, f->match_count, f->tie_count, f->tie_rec, f->sel_grid, f->exact_list, f->tile_off, f->tile_ent};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete f;
  return VSL_OK;
}

extern "C" void* vsl_frames_images_dev(vsl_frames* f) { return f ? (void*)f->images : nullptr; }

extern "C" int vsl_frames_upload(vsl_ctx* ctx, vsl_frames* f, int first, int n, const uint8_t* imgs, size_t pitch,
                                 size_t img_stride) {
  if (!ctx || !f || !imgs || first < 0 || n < 0 || first + n > f->max_images || pitch < (size_t)f->w)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_upload: bad arguments");
  if (n == 0) return VSL_OK;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  const size_t img_bytes = (size_t)f->w * f->h;
  if (pitch == (size_t)f->w && img_stride == img_bytes) {
    // dense batch: ONE transfer (a 2-D copy per image costs ~5-10 us of submission each: 1024 images would be
    // 5-10 ms of host time, more than the kernels that consume them); asynchronous when `imgs` is pinned
    VSL_HIP(ctx, hipMemcpyAsync(f->images + (size_t)first * img_bytes, imgs, (size_t)n * img_bytes, hipMemcpyHostToDevice,
                                ctx->stream));
    return VSL_OK;
  }
  for (int i = 0; i < n; i++) {
    VSL_HIP(ctx, hipMemcpy2DAsync(f->images + (size_t)(first + i) * img_bytes, f->w, imgs + (size_t)i * img_stride,
                                  pitch, f->w, f->h, hipMemcpyHostToDevice, ctx->stream));
  }
  return VSL_OK;
}

// Page-lock a caller-owned host buffer (a decoded image, a ring of frames) so that vsl_frames_upload from it is a real
// asynchronous DMA instead of a staged copy through the runtime's own pinned buffer.
extern "C" int vsl_host_register(vsl_ctx* ctx, void* ptr, size_t bytes) {
  if (!ctx || !ptr || bytes == 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_host_register: bad arguments");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  VSL_HIP(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return VSL_OK;
}
extern "C" int vsl_host_unregister(vsl_ctx* ctx, void* ptr) {
  if (!ctx || !ptr) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_host_unregister: bad arguments");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  VSL_HIP(ctx, hipHostUnregister(ptr));
  return VSL_OK;
}

// Cross-stream ordering without a host round trip (an upload context feeding a compute context, the compute context
// handing the buffer back: bench.py's streaming mode; the reference's order is load -> detect, src/slam.cpp:1122-1128).
struct vsl_event {
  int device = 0;
  hipEvent_t ev = nullptr;
  bool recorded = false;
};

extern "C" int vsl_event_create(vsl_ctx* ctx, vsl_event** out) {
  if (!ctx || !out) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_event_create: null argument");
  *out = nullptr;
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  vsl_event* e = new (std::nothrow) vsl_event;
  if (!e) return vsl_fail(ctx, VSL_ERR_NOMEM, "out of host memory");
  e->device = ctx->device;
  hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
  if (rc != hipSuccess) {
    delete e;
    return vsl_fail(ctx, VSL_ERR_HIP, "hipEventCreate -> %s", hipGetErrorString(rc));
  }
  *out = e;
  return VSL_OK;
}

extern "C" int vsl_event_destroy(vsl_event* e) {
  if (!e) return VSL_OK;
  (void)hipSetDevice(e->device);
  (void)hipEventDestroy(e->ev);
  delete e;
  return VSL_OK;
}

extern "C" int vsl_event_record(vsl_event* e, vsl_ctx* ctx) {
  if (!e || !ctx || e->device != ctx->device) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_event_record: bad arguments");
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  VSL_HIP(ctx, hipEventRecord(e->ev, ctx->stream));
  e->recorded = true;
  return VSL_OK;
}

extern "C" int vsl_ctx_wait_event(vsl_ctx* ctx, vsl_event* e) {
  if (!e || !ctx || e->device != ctx->device) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_ctx_wait_event: bad arguments");
  if (!e->recorded) return VSL_OK;  // nothing marked yet: nothing to wait for
  VSL_HIP(ctx, hipSetDevice(ctx->device));
  VSL_HIP(ctx, hipStreamWaitEvent(ctx->stream, e->ev, 0));
  return VSL_OK;
}

int vsl_ctx_scratch_frames(vsl_ctx* ctx, int w, int h, int feat, vsl_frames** out) {
  if (ctx->scratch && (ctx->scratch_w != w || ctx->scratch_h != h || ctx->scratch_feat < feat)) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    vsl_frames_destroy(ctx->scratch);
    ctx->scratch = nullptr;
  }
  if (!ctx->scratch) {
    int rc = vsl_frames_alloc(ctx, 2, w, h, feat, 1, &ctx->scratch);
    if (rc) return rc;
    ctx->scratch_w = w;
    ctx->scratch_h = h;
    ctx->scratch_feat = feat;
  }
  *out = ctx->scratch;
  return VSL_OK;
}

extern "C" int vsl_frames_download_counts(vsl_ctx* ctx, vsl_frames* f, int n_images, int32_t* n_keypoints,
                                          int n_pairs, int32_t* n_matches) {
  if (!ctx || !f || n_images < 0 || n_images > f->max_images || n_pairs < 0 || n_pairs > f->max_pairs)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_download_counts: bad arguments");
  if (n_images > 0 && n_keypoints)
    VSL_HIP(ctx, hipMemcpyAsync(n_keypoints, f->kp_count, sizeof(int32_t) * n_images, hipMemcpyDeviceToHost, ctx->stream));
  if (n_pairs > 0 && n_matches)
    VSL_HIP(ctx, hipMemcpyAsync(n_matches, f->match_count, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VSL_OK;
}

extern "C" int vsl_frames_download_candidate_counts(vsl_ctx* ctx, vsl_frames* f, int n_images, int32_t* n_candidates) {
  if (!ctx || !f || n_images < 0 || n_images > f->max_images || !n_candidates)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_download_candidate_counts: bad arguments");
  if (n_images == 0) return VSL_OK;
  std::vector<int32_t> m((size_t)n_images * VSL_META_STRIDE);
  VSL_HIP(ctx, hipMemcpyAsync(m.data(), f->meta, sizeof(int32_t) * m.size(), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n_images; i++) n_candidates[i] = m[(size_t)i * VSL_META_STRIDE + VSL_META_NCAND_LAST];
  return VSL_OK;
}

extern "C" int vsl_frames_download_keypoints(vsl_ctx* ctx, vsl_frames* f, int slot, int cap, double* corners_xy,
                                             double* angles, uint64_t* desc, int* n_out) {
  if (!ctx || !f || slot < 0 || slot >= f->max_images || !n_out || cap < 0)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_download_keypoints: bad arguments");
  // ONE round trip: the near-tie count, the keypoint count and the slot's full-capacity arrays travel in the
  // same batch of copies into pinned memory (72 KB at 1500 features), one synchronisation.  Only when a tie
  // was flagged (about once per 3e5 frames) or a diagnostic check is on does the slow path run.
  const size_t F = (size_t)f->F;
  void* hp = nullptr;
  int rc = vsl_ctx_hpinned(ctx, 64 + F * (8 + 8 + 32), &hp);
  if (rc) return rc;
  int32_t* hdr = (int32_t*)hp;  // [0] tie count, [1] keypoint count
  int32_t* hxy = (int32_t*)((char*)hp + 64);
  int32_t* hmom = hxy + 2 * F;
  uint64_t* hdesc = (uint64_t*)(hmom + 2 * F);
  const size_t base = (size_t)slot * F;
  const bool ties = f->ties_pending;
  hdr[0] = 0;
  if (ties) VSL_HIP(ctx, hipMemcpyAsync(&hdr[0], f->tie_count, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(&hdr[1], f->kp_count + slot, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipMemcpyAsync(hxy, f->kp_xy + base * 2, sizeof(int32_t) * 2 * F, hipMemcpyDeviceToHost, ctx->stream));
  if (angles) VSL_HIP(ctx, hipMemcpyAsync(hmom, f->kp_moments + base * 2, sizeof(int32_t) * 2 * F, hipMemcpyDeviceToHost, ctx->stream));
  if (desc) VSL_HIP(ctx, hipMemcpyAsync(hdesc, f->kp_desc + base * 4, sizeof(uint64_t) * 4 * F, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ties) {
    if (hdr[0] != 0) {  // flagged samples: host libm re-evaluation + device patch, then fetch the patched descriptors
      if ((rc = vsl_resolve_ties(ctx, f, nullptr))) return rc;
      if (desc) {
        VSL_HIP(ctx, hipMemcpyAsync(hdesc, f->kp_desc + base * 4, sizeof(uint64_t) * 4 * F, hipMemcpyDeviceToHost, ctx->stream));
        VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
    } else {
      f->ties_settled();
    }
  }
  const int n = hdr[1];
  *n_out = n;
  if (n > cap) return vsl_fail(ctx, VSL_ERR_CAPACITY, "keypoint capacity %d < %d", cap, n);
  if (n == 0) return VSL_OK;
  if (corners_xy)
    for (int i = 0; i < 2 * n; i++) corners_xy[i] = (double)hxy[i];
  if (angles) {
    // The orientation is atan2 of two exact integer moments (include/visnav/keypoints.h:184).  The
    // device keeps its own atan2 (kp_angle, within 2 ulp) for device-side consumers; what is handed
    // to the host is evaluated with the host's libm so that it is bit-identical to the reference's
    // own call on this machine.  A slot described with rotate_features = 0 has moments (0, 0) and
    // atan2(0, 0) = 0, the reference's value.
    for (int i = 0; i < n; i++) angles[i] = atan2((double)hmom[2 * i], (double)hmom[2 * i + 1]);
  }
  if (desc) std::memcpy(desc, hdesc, sizeof(uint64_t) * 4 * (size_t)n);
  return VSL_OK;
}

extern "C" int vsl_frames_download_matches(vsl_ctx* ctx, vsl_frames* f, int pair, int cap_pairs, int32_t* pairs,
                                           int* n_out) {
  if (!ctx || !f || pair < 0 || pair >= f->max_pairs || !n_out || cap_pairs < 0)
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_frames_download_matches: bad arguments");
  // one round trip: the count and the pair's full-capacity list into pinned memory together
  const size_t F = (size_t)f->F;
  void* hp = nullptr;
  int rc = vsl_ctx_hpinned(ctx, 64 + 8 * F, &hp);
  if (rc) return rc;
  int32_t* hdr = (int32_t*)hp;
  int32_t* hpairs = (int32_t*)((char*)hp + 64);
  VSL_HIP(ctx, hipMemcpyAsync(hdr, f->match_count + pair, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (pairs)
    VSL_HIP(ctx, hipMemcpyAsync(hpairs, f->matches + (size_t)pair * F * 2, sizeof(int32_t) * 2 * F, hipMemcpyDeviceToHost, ctx->stream));
  VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = hdr[0];
  *n_out = n;
  if (n > cap_pairs) return vsl_fail(ctx, VSL_ERR_CAPACITY, "match capacity %d < %d", cap_pairs, n);
  if (n > 0 && pairs) std::memcpy(pairs, hpairs, sizeof(int32_t) * 2 * (size_t)n);
  return VSL_OK;
}

// ------------------------------------------------------------------- descriptor byte-order helpers
// include/visnav/converter.h:23-33: bit i of the bitset -> byte i/8, bit 7 - i%8.
extern "C" void vsl_desc_bitset_to_bytes(const uint64_t* desc, int n, uint8_t* desc32) {
  for (int k = 0; k < n; k++)
    for (int b = 0; b < 32; b++) {
      const uint8_t src = (uint8_t)(desc[4 * (size_t)k + b / 8] >> (8 * (b % 8)));
      uint8_t r = src;
      r = (uint8_t)(((r & 0xF0) >> 4) | ((r & 0x0F) << 4));
      r = (uint8_t)(((r & 0xCC) >> 2) | ((r & 0x33) << 2));
      r = (uint8_t)(((r & 0xAA) >> 1) | ((r & 0x55) << 1));
      desc32[32 * (size_t)k + b] = r;
    }
}

// include/visnav/converter.h:50-61
extern "C" void vsl_desc_bytes_to_bitset(const uint8_t* desc32, int n, uint64_t* desc) {
  for (int k = 0; k < n; k++) {
    uint64_t wds[4] = {0, 0, 0, 0};
    for (int b = 0; b < 32; b++) {
      uint8_t r = desc32[32 * (size_t)k + b];
      r = (uint8_t)(((r & 0xF0) >> 4) | ((r & 0x0F) << 4));
      r = (uint8_t)(((r & 0xCC) >> 2) | ((r & 0x33) << 2));
      r = (uint8_t)(((r & 0xAA) >> 1) | ((r & 0x55) << 1));
      wds[b / 8] |= (uint64_t)r << (8 * (b % 8));
    }
    memcpy(desc + 4 * (size_t)k, wds, 32);
  }
}
