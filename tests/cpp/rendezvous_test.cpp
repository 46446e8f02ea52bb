// CPU test of the two host-side rules of the multi-GPU C++ path (no HIP, no RCCL):
//   rendezvous_test device <n_devices>              prints the device index device_select.h picks from the environment
//   rendezvous_test meet <file> <rank> <world> <s>  runs the file rendezvous; rank 0 publishes the string <s>, every rank
//                                                   prints the payload it ends up with
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "visnav_amd/device_select.h"
#include "visnav_amd/file_rendezvous.h"

int main(int argc, char** argv) {
  if (argc == 3 && !std::strcmp(argv[1], "device")) {
    std::printf("%d\n", visnav::amd::device_index_for(std::atoi(argv[2])));
    return 0;
  }
  if (argc == 6 && !std::strcmp(argv[1], "meet")) {
    const int rank = std::atoi(argv[3]), world = std::atoi(argv[4]);
    char payload[128];
    std::memset(payload, 0, sizeof(payload));
    if (rank == 0) std::snprintf(payload, sizeof(payload), "%s", argv[5]);
    std::string err;
    if (!visnav::amd::file_rendezvous(argv[2], rank, world, payload, sizeof(payload), 20, &err)) {
      std::fprintf(stderr, "rank %d: %s\n", rank, err.c_str());
      return 1;
    }
    std::printf("rank %d got %s\n", rank, payload);
    if (rank == 0) {
      // (the real caller cleans up after ncclCommInitRank, i.e. after everybody has read; here: give the others 1 s)
      std::this_thread::sleep_for(std::chrono::milliseconds(1000));
      visnav::amd::file_rendezvous_cleanup(argv[2], world);
    }
    return 0;
  }
  return 2;
}
