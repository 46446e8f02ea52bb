// visnav_amd/rccl_world.h -- the collective behind the multi-GPU visnav::global_bundle_adjustment: one process per
// GPU, RCCL (ncclAllReduce on the solver's HIP stream, no host hop) over xGMI.  Include it BEFORE
// bundle_adjustment.h and link -lrccl -lamdhip64; a build without it keeps global_bundle_adjustment single-GPU.
//
// The reference's global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748) is one ceres::Solve in one
// process; here every rank runs the same call on the same map, owns a contiguous landmark range (balanced by
// observation count) and all-reduces the packed partial reduced camera system once per LM iteration
// (vsl_global_bundle_adjust, include/vslam_hip.h).  Environment:
//   VISNAV_AMD_WORLD / VISNAV_AMD_RANK   (default: WORLD_SIZE / RANK of a torchrun-style launcher, else 1 / 0)
//   VISNAV_AMD_NCCL_ID_FILE              rendezvous file (default <TMPDIR or /tmp>/visnav_amd.<uid>/nccl_id.<MASTER_PORT or 0>,
//                                        directory mode 0700)
//   VISNAV_AMD_DEVICE                    GPU of this rank -- the SAME rule as the solver context (device_select.h:
//                                        VISNAV_AMD_DEVICE, else LOCAL_RANK, else the rank, modulo the device count), so the
//                                        communicator and the solver's buffers / stream are on one device
// Rendezvous protocol (stale-proof: a file left by a crashed or earlier run can never be taken for this run's):
//   every rank r > 0 draws a random nonce and keeps <file>.hello.<r> in place until it has its id;
//   rank 0 first removes any old <file> and <file>.hello.*, waits for the world-1 hellos, then publishes
//   {magic, world, nonces[], ncclUniqueId} atomically (write + rename); rank r accepts the file only if it carries ITS
//   nonce; rank 0 removes everything once the communicator exists (on every path, the forced one-rank case writes nothing).
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../vslam_hip.h"
#include "device_select.h"
#include "file_rendezvous.h"

#define VISNAV_AMD_HAVE_RCCL 1

namespace visnav {
namespace amd {

class RcclWorld {
 public:
  static RcclWorld& instance() {
    static RcclWorld w;
    return w;
  }
  int rank() const { return rank_; }
  int world() const { return world_; }
  int device() const { return device_; }
  bool enabled() const { return comm_ != nullptr; }

  // vsl_allreduce_fn: in-place all-reduce of `count` doubles on the solver's stream
  static int allreduce(void* user, double* buf, int64_t count, int op, void* hip_stream) {
    RcclWorld* w = static_cast<RcclWorld*>(user);
    if (hip_stream != w->checked_stream_) {  // once per stream: the solver's stream must live on the communicator's GPU
      hipDevice_t sdev = -1;
      if (hipStreamGetDevice(static_cast<hipStream_t>(hip_stream), &sdev) != hipSuccess || (int)sdev != w->device_) {
        std::fprintf(stderr,
                     "visnav_amd: the solver's stream is on device %d but the RCCL communicator of rank %d is on device %d "
                     "(both follow device_select.h: VISNAV_AMD_DEVICE, LOCAL_RANK, RANK -- was the context created with an explicit index?)\n",
                     (int)sdev, w->rank_, w->device_);
        return 2;
      }
      w->checked_stream_ = hip_stream;
    }
    const ncclResult_t r = ncclAllReduce(buf, buf, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMax, w->comm_,
                                         static_cast<hipStream_t>(hip_stream));
    if (r != ncclSuccess) {
      std::fprintf(stderr, "visnav_amd: ncclAllReduce failed: %s\n", ncclGetErrorString(r));
      return 1;
    }
    return 0;
  }

 private:
  static int env_int(const char* a, const char* b, int dflt) {
    const char* v = std::getenv(a);
    if (!v && b) v = std::getenv(b);
    return v ? std::atoi(v) : dflt;
  }
  RcclWorld() {
    world_ = env_int("VISNAV_AMD_WORLD", "WORLD_SIZE", 1);
    rank_ = env_int("VISNAV_AMD_RANK", "RANK", 0);
    const bool force = std::getenv("VISNAV_AMD_FORCE_RCCL") != nullptr;  // a one-rank communicator (tests on a one-GPU box)
    if (world_ <= 1 && !force) return;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) die("no HIP device");
    device_ = device_index_for(ndev);
    if (hipSetDevice(device_) != hipSuccess) die("hipSetDevice failed");
    ncclUniqueId id;
    if (world_ <= 1) {
      if (ncclGetUniqueId(&id) != ncclSuccess) die("ncclGetUniqueId failed");
    } else {
      const char* p = std::getenv("VISNAV_AMD_NCCL_ID_FILE");
      const std::string path = p ? std::string(p) : rendezvous_default_path();
      if (rank_ == 0 && ncclGetUniqueId(&id) != ncclSuccess) die("ncclGetUniqueId failed");
      std::string err;
      if (!file_rendezvous(path, rank_, world_, &id, sizeof(id), 120, &err)) die(err.c_str());
      cleanup_path_ = rank_ == 0 ? path : std::string();
    }
    const ncclResult_t r = ncclCommInitRank(&comm_, world_, id, rank_);
    // everybody has read the id once the communicator exists -- removed on failure too
    if (!cleanup_path_.empty()) file_rendezvous_cleanup(cleanup_path_, world_);
    if (r != ncclSuccess) die("ncclCommInitRank failed");
  }
  ~RcclWorld() {
    if (comm_) ncclCommDestroy(comm_);
  }
  [[noreturn]] static void die(const char* what) {
    std::fprintf(stderr, "visnav_amd: RCCL set-up: %s\n", what);
    std::abort();
  }
  int rank_ = 0, world_ = 1, device_ = 0;
  ncclComm_t comm_ = nullptr;
  void* checked_stream_ = reinterpret_cast<void*>(-1);
  std::string cleanup_path_;
};

}  // namespace amd
}  // namespace visnav
