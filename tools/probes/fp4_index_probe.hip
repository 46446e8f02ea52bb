// Developer probe: can the block-scaled FP4 MFMA deliver a row INDEX into the accumulator?  A-side fragment of row r:
// k-half 0 encodes L = m & 63 as a sum of FP4 values with block scale 2^-11, k-half 1 encodes H = m >> 6 with block scale
// 2^-5; B side all 1.0; C = 256.  Expected D[row][col] = 256 + m / 2048 exactly, for every column.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/fp4_index_probe tools/probes/fp4_index_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

// v in [0, 63] as up to 12 e2m1 nibbles (6.0 = 0x7, 4.0 = 0x6, 3.0 = 0x5, 2.0 = 0x4, 1.0 = 0x2), the rest 0
__host__ __device__ inline void encode(int v, uint32_t out[4]) {
  uint32_t nib[32] = {0};
  int n = 0;
  for (; v >= 6; v -= 6) nib[n++] = 0x7;
  if (v == 5) { nib[n++] = 0x6; nib[n++] = 0x2; }
  else if (v == 4) nib[n++] = 0x6;
  else if (v == 3) nib[n++] = 0x5;
  else if (v == 2) nib[n++] = 0x4;
  else if (v == 1) nib[n++] = 0x2;
  for (int w = 0; w < 4; w++) {
    out[w] = 0;
    for (int j = 0; j < 8; j++) out[w] |= nib[8 * w + j] << (4 * j);
  }
}

__global__ void probe(const int* m_of_row, float* D) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int m = m_of_row[r];
  uint32_t a4[4];
  encode(h ? (m >> 6) : (m & 63), a4);
  const v8i_t a = {(int)a4[0], (int)a4[1], (int)a4[2], (int)a4[3], 0, 0, 0, 0};
  const v8i_t b = {0x22222222, 0x22222222, 0x22222222, 0x22222222, 0, 0, 0, 0};
  v16f_t c;
  for (int g = 0; g < 16; g++) c[g] = 256.0f;
  const int scale_a = h ? (127 - 5) : (127 - 11);
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, scale_a, 0, 127);
  for (int g = 0; g < 16; g++) D[lane * 16 + g] = c[g];
}

int main() {
  std::vector<int> m(32);
  for (int r = 0; r < 32; r++) m[r] = (r * 67 + 5) % 2048;
  m[0] = 0; m[1] = 2047; m[2] = 63; m[3] = 64; m[4] = 1983;
  int* dm; float* dD;
  hipMalloc(&dm, 128); hipMalloc(&dD, 64 * 16 * 4);
  hipMemcpy(dm, m.data(), 128, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dm, dD);
  std::vector<float> D(64 * 16);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; lane++)
    for (int g = 0; g < 16; g++) {
      const int row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5);
      const float e = 256.0f + m[row] / 2048.0f;
      if (D[lane * 16 + g] != e) {
        if (bad < 8) std::printf("lane %d reg %d row %d: got %.6f expected %.6f\n", lane, g, row, D[lane * 16 + g], e);
        bad++;
      }
    }
  std::printf("index through the block-scaled FP4 MFMA: %d mismatches of 1024\n", bad);
  return bad != 0;
}
