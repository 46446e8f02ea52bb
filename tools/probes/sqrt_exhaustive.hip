// Developer probe: is  r = t * rsq(max(t, FLOOR));  r' = fma(fma(-r, r, t), 0.5 * rsq, r)  the correctly rounded
// square root for EVERY float t in [FLOOR, 2^127]?  Compared bit for bit with sqrtf (correctly rounded under hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt).  Prints the number of mismatches per binade group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ float fast_sqrt(float t, float floor_) {
  const float tm = __builtin_fmaxf(t, floor_);
  const float y = __builtin_amdgcn_rsqf(tm);
  const float r = t * y;
  const float h = 0.5f * y;
  const float e = __builtin_fmaf(-r, r, t);
  return __builtin_fmaf(e, h, r);
}

__global__ void check(uint32_t lo, uint32_t hi, float floor_, unsigned long long* bad, uint32_t* first) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long n = 0;
  for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
    const float t = __builtin_bit_cast(float, (uint32_t)b);
    const float a = fast_sqrt(t, floor_);
    const float ref = sqrtf(t);
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, ref)) {
      n++;
      atomicMin(first, (uint32_t)b);
    }
  }
  if (n) atomicAdd(bad, n);
}

int main() {
  unsigned long long* bad;
  uint32_t* first;
  hipMalloc(&bad, 8);
  hipMalloc(&first, 4);
  const float floor_ = 0x1p-92f;
  // zero, then [2^-92, 2^127] in chunks of 8 binades
  for (int e = -92; e < 128; e += 8) {
    const int e1 = e + 8 > 128 ? 128 : e + 8;
    const uint32_t lo = (uint32_t)(e + 127) << 23, hi = (e1 == 128 ? 0x7f7fffffu : (((uint32_t)(e1 + 127) << 23) - 1u));
    hipMemset(bad, 0, 8);
    hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, lo, hi, floor_, bad, first);
    unsigned long long hb;
    uint32_t hf;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    std::printf("2^%d .. 2^%d: %llu mismatches of %u (first bits %08x)\n", e, e1, hb, hi - lo + 1, hf);
  }
  hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, 0u, 0u, floor_, bad, first);
  unsigned long long hb;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  std::printf("t = 0: %llu mismatches\n", hb);
  return 0;
}
