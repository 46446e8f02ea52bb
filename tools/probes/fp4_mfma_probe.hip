// fp4_mfma_probe.hip -- checks, with exact integer data, that v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1)
// operands computes the 0/1 x +-1 inner products the Hamming matcher needs, and times it against the int8 form.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/fp4_mfma_probe tools/probes/fp4_mfma_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// 32 bits -> 32 FP4 nibbles (bit j -> nibble j): database form 0 -> 0x0 (0.0), 1 -> 0x2 (1.0);
// query form 0 -> 0x2 (+1.0), 1 -> 0xA (-1.0)
__host__ __device__ inline void spread_bits(uint32_t bits, bool query, uint32_t out[4]) {
  for (int w = 0; w < 4; w++) {
    uint32_t v = 0;
    for (int j = 0; j < 8; j++) {
      const uint32_t b = (bits >> (8 * w + j)) & 1u;
      const uint32_t nib = query ? (b ? 0xAu : 0x2u) : (b ? 0x2u : 0x0u);
      v |= nib << (4 * j);
    }
    out[w] = v;
  }
}

__global__ void probe(const uint32_t* A /*[32 rows][2 halves] bits*/, const uint32_t* B /*[32 cols][2 halves]*/, float* D,
                      int reps, long long* cycles) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  uint32_t a4[4], b4[4];
  spread_bits(A[2 * r + h], false, a4);
  spread_bits(B[2 * r + h], true, b4);
  v8i_t a = {(int)a4[0], (int)a4[1], (int)a4[2], (int)a4[3], 0, 0, 0, 0};
  v8i_t b = {(int)b4[0], (int)b4[1], (int)b4[2], (int)b4[3], 0, 0, 0, 0};
  v16f_t c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; i++)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4 /*A fp4*/, 4 /*B fp4*/, 0, 127 /*2^0*/, 0, 127);
  const long long t1 = __builtin_amdgcn_s_memtime();
  for (int g = 0; g < 16; g++) D[lane * 16 + g] = c[g];
  if (lane == 0) cycles[0] = t1 - t0;
}

__global__ void probe_i8(int reps, long long* cycles, int* sink) {
  v4i_t a = {0x01010101, 0x01000100, 0x00010001, 0x01010000}, b = {0x01FF01FF, (int)0xFF01FF01u, 0x0101FFFF, (int)0xFFFF0101u};
  v16i_t c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; i++) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  const long long t1 = __builtin_amdgcn_s_memtime();
  sink[threadIdx.x] = c[0] + c[5];
  if (threadIdx.x == 0) cycles[1] = t1 - t0;
}

int main() {
  std::vector<uint32_t> A(64), B(64);
  srand(5);
  for (auto& v : A) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  for (auto& v : B) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  uint32_t *dA, *dB;
  float* dD;
  long long* dc;
  int* ds;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 64 * 16 * 4); hipMalloc(&dc, 16); hipMalloc(&ds, 256);
  hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1, dc);
  std::vector<float> D(64 * 16);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  // expected: D[row m][col c] = sum_k d_k (1 - 2 q_k) over the 64 bits; C/D layout col = lane & 31,
  // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  int bad = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int g = 0; g < 16; g++) {
      const int col = lane & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
      int e = 0;
      for (int hh = 0; hh < 2; hh++)
        for (int j = 0; j < 32; j++) {
          const int d = (A[2 * row + hh] >> j) & 1, q = (B[2 * col + hh] >> j) & 1;
          e += d * (1 - 2 * q);
        }
      if ((float)e != D[lane * 16 + g]) {
        if (bad < 5) printf("mismatch lane %d reg %d: got %g expected %d\n", lane, g, D[lane * 16 + g], e);
        bad++;
      }
    }
  printf("fp4 32x32x64 exactness: %s (%d mismatches of 1024)\n", bad ? "FAIL" : "ok", bad);
  const int reps = 4096;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, reps, dc);
  hipLaunchKernelGGL(probe_i8, dim3(1), dim3(64), 0, 0, reps, dc, ds);
  long long cyc[2];
  hipMemcpy(cyc, dc, 16, hipMemcpyDeviceToHost);
  printf("cycles per instruction (one wave, dependent chain): fp4 32x32x64 %.1f, i8 32x32x32 %.1f  (s_memtime ticks)\n",
         (double)cyc[0] / reps, (double)cyc[1] / reps);
  return bad ? 1 : 0;
}
