// visnav_amd/rccl_world.h -- the collective behind the multi-GPU visnav::global_bundle_adjustment: one process per
// GPU, RCCL (ncclAllReduce on the solver's HIP stream, no host hop) over xGMI.  Include it BEFORE
// bundle_adjustment.h and link -lrccl -lamdhip64; a build without it keeps global_bundle_adjustment single-GPU.
//
// The reference's global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748) is one ceres::Solve in one
// process; here every rank runs the same call on the same map, owns a contiguous landmark range (balanced by
// observation count) and all-reduces the packed partial reduced camera system once per LM iteration
// (vsl_global_bundle_adjust, include/vslam_hip.h).  Environment:
//   VISNAV_AMD_WORLD / VISNAV_AMD_RANK   (default: WORLD_SIZE / RANK of a torchrun-style launcher, else 1 / 0)
//   VISNAV_AMD_NCCL_ID_FILE              rendezvous: rank 0 writes its ncclUniqueId there (atomically), the others wait
//                                        for it (default /tmp/visnav_amd_nccl_id.<MASTER_PORT or 0>)
//   VISNAV_AMD_DEVICE                    GPU of this rank (default LOCAL_RANK, else the rank)
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "../vslam_hip.h"

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
  bool enabled() const { return comm_ != nullptr; }

  // vsl_allreduce_fn: in-place all-reduce of `count` doubles on the solver's stream
  static int allreduce(void* user, double* buf, int64_t count, int op, void* hip_stream) {
    RcclWorld* w = static_cast<RcclWorld*>(user);
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
    int dev = env_int("VISNAV_AMD_DEVICE", "LOCAL_RANK", rank_);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) die("no HIP device");
    dev %= ndev;
    if (hipSetDevice(dev) != hipSuccess) die("hipSetDevice failed");
    std::string path;
    if (const char* p = std::getenv("VISNAV_AMD_NCCL_ID_FILE")) {
      path = p;
    } else {
      const char* port = std::getenv("MASTER_PORT");
      path = std::string("/tmp/visnav_amd_nccl_id.") + (port ? port : "0");
    }
    ncclUniqueId id;
    if (rank_ == 0) {
      if (ncclGetUniqueId(&id) != ncclSuccess) die("ncclGetUniqueId failed");
      const std::string tmp = path + ".tmp";
      FILE* f = std::fopen(tmp.c_str(), "wb");
      if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) die("cannot write the rendezvous file");
      std::fclose(f);
      if (std::rename(tmp.c_str(), path.c_str()) != 0) die("cannot publish the rendezvous file");
    } else {
      bool got = false;
      for (int tries = 0; tries < 6000 && !got; tries++) {  // up to 60 s
        FILE* f = std::fopen(path.c_str(), "rb");
        if (f) {
          got = std::fread(&id, sizeof(id), 1, f) == 1;
          std::fclose(f);
        }
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
      if (!got) die("timed out waiting for rank 0's ncclUniqueId");
    }
    if (ncclCommInitRank(&comm_, world_, id, rank_) != ncclSuccess) die("ncclCommInitRank failed");
    if (rank_ == 0 && world_ > 1) {  // everybody has read the id once the communicator exists
      std::remove(path.c_str());
    }
  }
  ~RcclWorld() {
    if (comm_) ncclCommDestroy(comm_);
  }
  [[noreturn]] static void die(const char* what) {
    std::fprintf(stderr, "visnav_amd: RCCL set-up: %s\n", what);
    std::abort();
  }
  int rank_ = 0, world_ = 1;
  ncclComm_t comm_ = nullptr;
};

}  // namespace amd
}  // namespace visnav
