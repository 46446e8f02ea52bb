// pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths this library uses
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel streams a
// buffer far larger than the Infinity Cache ONCE with one access width:
//   calib_read_b8    one byte per lane per load   (K1's image loads: buffer_load_ubyte, 64 B per wave-instruction)
//   calib_read_b32   one dword per lane           (describe's window loads, candidate keys)
//   calib_read_b128  16 bytes per lane            (the guide's reference pattern)
//   calib_write_b64  8 bytes per lane             (K1's candidate keys)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); tools/pmc_summary.py divides the
// known bytes (printed here) by the counter to get the correction factor of each width.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/pmc_calib tools/probes/pmc_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ void calib_read_b8(const uint8_t* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ void calib_read_b32(const uint32_t* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ void calib_read_b128(const uint4* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0xFFFFFFFFu) *sink = acc;
}
__global__ void calib_write_b64(uint64_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i;
}

int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB: four times the Infinity Cache
  void* buf = nullptr;
  unsigned* sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, bytes);
  (void)hipDeviceSynchronize();
  const dim3 grid(256 * 16), block(256);
  hipLaunchKernelGGL(calib_read_b8, grid, block, 0, 0, (const uint8_t*)buf, bytes, sink);
  hipLaunchKernelGGL(calib_read_b32, grid, block, 0, 0, (const uint32_t*)buf, bytes / 4, sink);
  hipLaunchKernelGGL(calib_read_b128, grid, block, 0, 0, (const uint4*)buf, bytes / 16, sink);
  hipLaunchKernelGGL(calib_write_b64, grid, block, 0, 0, (uint64_t*)buf, bytes / 8);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  std::printf("pmc_calib: every kernel moved %zu bytes\n", bytes);
  return 0;
}
