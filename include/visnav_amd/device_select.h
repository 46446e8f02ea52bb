// visnav_amd/device_select.h -- ONE rule for "which GPU does this process use", shared by the per-thread solver
// context (keypoints.h: amd::ctx()) and the RCCL communicator (rccl_world.h).  One process per GPU:
//   VISNAV_AMD_DEVICE  explicit device index, else
//   LOCAL_RANK         the launcher's local rank (torchrun-style), else
//   VISNAV_AMD_RANK / RANK, else 0;
// always modulo the number of visible devices (several ranks may rehearse on a one-GPU box).
// No HIP call in here: the device count is an argument, so the rule is testable without a GPU
// (tests/cpp/device_select_test.cpp).
#pragma once
#include <cstdlib>

namespace visnav {
namespace amd {

inline int device_index_for(int n_devices) {
  static const char* const order[] = {"VISNAV_AMD_DEVICE", "LOCAL_RANK", "VISNAV_AMD_RANK", "RANK"};
  int dev = 0;
  for (const char* name : order) {
    const char* v = std::getenv(name);
    if (v && *v) {
      dev = std::atoi(v);
      break;
    }
  }
  if (dev < 0) dev = 0;
  return n_devices > 0 ? dev % n_devices : dev;
}

}  // namespace amd
}  // namespace visnav
