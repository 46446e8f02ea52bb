// visnav_amd/file_rendezvous.h -- how the ranks of one multi-GPU job agree on rank 0's ncclUniqueId through the file
// system (no MPI, no TCP store).  Plain C++ (no HIP / RCCL), so the protocol is tested on the CPU
// (tests/cpp/rendezvous_test.cpp, tests/test_dist_cpu.py).
//
// Stale-proof: a file left behind by a crashed or earlier run can never be taken for this run's.
//   every rank r > 0 draws a random nonce and keeps <file>.hello.<r> in place until it has the payload;
//   rank 0 first removes any old <file> and <file>.hello.*, waits for the world-1 hellos, then publishes
//   {magic, world, nonces[world], payload} atomically (write + rename);
//   rank r accepts <file> only if it carries ITS nonce (an old file cannot: the nonce is drawn per process);
//   rank 0 removes everything once the payload has served (file_rendezvous_cleanup), on every path.
#pragma once
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace visnav {
namespace amd {

namespace rendezvous_detail {
constexpr uint64_t kMagic = 0x76736c5f6e63636cull;  // "vsl_nccl"
inline bool write_atomically(const std::string& path, const void* data, size_t bytes) {
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(data, 1, bytes, f) == bytes;
  std::fclose(f);
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
    std::remove(tmp.c_str());
    return false;
  }
  return true;
}
inline bool read_all(const std::string& path, void* data, size_t bytes) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  const bool ok = std::fread(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok;
}
inline std::string hello(const std::string& path, int r) { return path + ".hello." + std::to_string(r); }
}  // namespace rendezvous_detail

// <TMPDIR or /tmp>/visnav_amd.<uid>/nccl_id.<MASTER_PORT or 0>; the per-user directory (mode 0700) keeps the name
// out of a world-writable place
inline std::string rendezvous_default_path() {
  const char* tmp = std::getenv("TMPDIR");
  const std::string dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/visnav_amd." + std::to_string((long)getuid());
  (void)mkdir(dir.c_str(), 0700);
  // a directory somebody else planted under the shared tmp, a symlink, or one that others can write to is refused:
  // the empty path makes the rendezvous fail with a message instead of trusting files in it
  struct stat sb;
  if (lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & 0777) != 0700) {
    std::fprintf(stderr, "visnav_amd: %s is not a directory owned by uid %ld with mode 0700 -- refusing it for the rendezvous\n",
                 dir.c_str(), (long)getuid());
    return std::string();
  }
  const char* port = std::getenv("MASTER_PORT");
  return dir + "/nccl_id." + (port ? port : "0");
}

inline void file_rendezvous_cleanup(const std::string& path, int world) {
  std::remove(path.c_str());
  for (int r = 1; r < world; r++) std::remove(rendezvous_detail::hello(path, r).c_str());
}

// Rank 0 passes the payload in, every other rank receives it.  False + *err on a timeout or an I/O failure.
inline bool file_rendezvous(const std::string& path, int rank, int world, void* payload, size_t payload_bytes, int timeout_s,
                            std::string* err) {
  using namespace rendezvous_detail;
  const size_t bytes = 16 + 8 * (size_t)world + payload_bytes;
  std::vector<unsigned char> blob(bytes);
  const long max_tries = 100L * timeout_s;  // 10 ms per try
  auto fail = [&](const char* what) {
    if (err) *err = what;
    return false;
  };
  if (path.empty()) return fail("no usable rendezvous path (the per-user directory was refused; set VISNAV_AMD_NCCL_ID_FILE)");
  if (rank == 0) {
    file_rendezvous_cleanup(path, world);  // nothing of an earlier run survives rank 0's start
    std::vector<uint64_t> nonces(world, 0);
    for (int r = 1; r < world; r++) {
      bool got = false;
      for (long tries = 0; tries < max_tries && !got; tries++) {
        got = read_all(hello(path, r), &nonces[r], 8) && nonces[r] != 0;
        if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
      if (!got) return fail("timed out waiting for the other ranks' hello files");
    }
    const uint64_t w64 = (uint64_t)world;
    std::memcpy(blob.data(), &kMagic, 8);
    std::memcpy(blob.data() + 8, &w64, 8);
    std::memcpy(blob.data() + 16, nonces.data(), 8 * (size_t)world);
    std::memcpy(blob.data() + 16 + 8 * (size_t)world, payload, payload_bytes);
    if (!write_atomically(path, blob.data(), bytes)) return fail("cannot publish the rendezvous file");
    return true;
  }
  std::random_device rd;
  uint64_t nonce = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 17) ^
                   (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  if (nonce == 0) nonce = 1;
  bool got = false;
  for (long tries = 0; tries < max_tries && !got; tries++) {
    uint64_t cur = 0;
    if (!read_all(hello(path, rank), &cur, 8) || cur != nonce)  // (re)publish: rank 0 clears old hellos when it starts
      if (!write_atomically(hello(path, rank), &nonce, 8)) return fail("cannot write the hello file");
    if (read_all(path, blob.data(), bytes)) {
      uint64_t magic = 0, w64 = 0, mine = 0;
      std::memcpy(&magic, blob.data(), 8);
      std::memcpy(&w64, blob.data() + 8, 8);
      std::memcpy(&mine, blob.data() + 16 + 8 * (size_t)rank, 8);
      got = magic == kMagic && w64 == (uint64_t)world && mine == nonce;  // this run's file, not a stale one
    }
    if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  std::remove(hello(path, rank).c_str());
  if (!got)
    return fail("timed out waiting for rank 0's rendezvous file (is rank 0 running with the same VISNAV_AMD_NCCL_ID_FILE / MASTER_PORT?)");
  std::memcpy(payload, blob.data() + 16 + 8 * (size_t)world, payload_bytes);
  return true;
}

}  // namespace amd
}  // namespace visnav
