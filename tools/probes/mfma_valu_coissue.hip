// mfma_valu_coissue.hip -- do FP4 block-scaled matrix instructions of one wave and v_min / v_med3 instructions of ANOTHER
// wave on the same SIMD overlap?  One 512-thread workgroup: waves 0-3 (one per SIMD) run matrix instructions, waves 4-7
// (their SIMD partners) run a vector stream; each role alone, then together.  Wall clock per role in s_memtime ticks
// (100 MHz) and the ratio together / alone.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu_coissue tools/probes/mfma_valu_coissue.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// mode bit 0: matrix waves work; bit 1: vector waves work; chains: independent accumulators per matrix wave (1 or 2)
template <int CHAINS, bool I8>
__global__ __launch_bounds__(512) void probe(int mode, int reps, long long* ticks, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (mode & 1) {
      v8i_t a = {0x22222222, 0x02020202, 0x20202020, 0x22002200, 0, 0, 0, 0}, b = {0x2A2A2A2A, (int)0xA2A2A2A2u, 0x22AA22AA, (int)0xAAAA2222u, 0, 0, 0, 0};
      if (I8) {
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        typedef int v16i_t __attribute__((ext_vector_type(16)));
        const v4i_t a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
        v16i_t d0, d1;
        for (int k = 0; k < 16; k++) d0[k] = lane, d1[k] = k;
        for (int i = 0; i < reps; i++) {
          d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, d0, 0, 0, 0);
          if (CHAINS == 2) d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, d1, 0, 0, 0);
        }
        sink[threadIdx.x] = (float)(d0[3] + d1[7]);
      } else {
        v16f_t c0, c1;
        for (int i = 0; i < 16; i++) c0[i] = 512.0f + lane, c1[i] = 600.0f + i;
        for (int i = 0; i < reps; i++) {
          c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
          if (CHAINS == 2) c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
        }
        sink[threadIdx.x] = c0[3] + c1[7];
      }
    }
  } else if (mode & 2) {
    uint32_t b[4], s[4], k = 0x9E3779B9u * (uint32_t)(threadIdx.x + 1);
    for (int j = 0; j < 4; j++) b[j] = s[j] = 0xFFFFFFFFu;
    for (int i = 0; i < reps; i++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {  // 4 x (med3 + min) per repetition = 8 vector instructions + 1 for the key
        const uint32_t key = k ^ (uint32_t)(i * 4 + u);
        s[u] = umed3(b[u], key, s[u]);
        b[u] = min(b[u], key);
      }
      k = k * 3u + 1u;
    }
    sink[threadIdx.x] = (float)(b[0] ^ b[1] ^ b[2] ^ b[3] ^ s[0] ^ s[1] ^ s[2] ^ s[3]);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) ticks[wave] = t1 - t0;
}

// second experiment: NM matrix waves and NV vector waves per SIMD in one workgroup (wave groups of four), the same
// amount of work per wave; time of the slowest wave.  If the two pipes overlap, (NM, NV) costs max(NM * m, NV * v).
__global__ __launch_bounds__(1024) void mix(int nm, int nv, int reps, long long* ticks, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = wave >> 2;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (grp < nm) {
    v8i_t a = {0x22222222, 0x02020202, 0x20202020, 0x22002200, 0, 0, 0, 0}, b = {0x2A2A2A2A, (int)0xA2A2A2A2u, 0x22AA22AA, (int)0xAAAA2222u, 0, 0, 0, 0};
    v16f_t c0, c1;
    for (int i = 0; i < 16; i++) c0[i] = 512.0f + lane, c1[i] = 600.0f + i;
    for (int i = 0; i < reps; i++) {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
    }
    sink[threadIdx.x] = c0[3] + c1[7];
  } else if (grp < nm + nv) {
    uint32_t b[4], s[4], k = 0x9E3779B9u * (uint32_t)(threadIdx.x + 1);
    for (int j = 0; j < 4; j++) b[j] = s[j] = 0xFFFFFFFFu;
    for (int i = 0; i < reps; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {  // 16 tracker instructions per repetition, keys from a cheap recurrence (8 more)
        k += 0x9E3779B9u;
        s[u & 3] = umed3(b[u & 3], k, s[u & 3]);
        b[u & 3] = min(b[u & 3], k);
      }
    }
    sink[threadIdx.x] = (float)(b[0] ^ b[1] ^ b[2] ^ b[3] ^ s[0] ^ s[1] ^ s[2] ^ s[3]);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) ticks[wave] = t1 - t0;
}

static void run_mix(int nm, int nv, int reps) {
  long long* ticks;
  float* sink;
  hipMalloc(&ticks, 16 * sizeof(long long));
  hipMalloc(&sink, 1024 * sizeof(float));
  long long h[16];
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(mix, dim3(1), dim3(64 * 4 * (nm + nv)), 0, 0, nm, nv, reps, ticks, sink);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  long long tm = 0, tv = 0;
  for (int w = 0; w < 4 * (nm + nv); w++) (w < 4 * nm ? tm : tv) = h[w] > (w < 4 * nm ? tm : tv) ? h[w] : (w < 4 * nm ? tm : tv);
  printf("matrix waves per SIMD %d, vector waves per SIMD %d: slowest matrix wave %8lld  slowest vector wave %8lld  (per repetition %.1f / %.1f ticks)\n", nm, nv, tm, tv,
         (double)tm / reps, (double)tv / reps);
  hipFree(ticks);
  hipFree(sink);
}

template <int CHAINS, bool I8>
static void run(const char* name, int reps) {
  long long* ticks;
  float* sink;
  hipMalloc(&ticks, 8 * sizeof(long long));
  hipMalloc(&sink, 512 * sizeof(float));
  long long h[3][8];
  for (int mode = 1; mode <= 3; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL((probe<CHAINS, I8>), dim3(1), dim3(512), 0, 0, mode, reps, ticks, sink);
      hipDeviceSynchronize();
    }
    hipMemcpy(h[mode - 1], ticks, sizeof(h[0]), hipMemcpyDeviceToHost);
  }
  const double m_alone = (double)h[0][0], v_alone = (double)h[1][4], m_both = (double)h[2][0], v_both = (double)h[2][4];
  printf("%-34s matrix alone %6.0f  vector alone %6.0f  together: matrix %6.0f (x%.2f)  vector %6.0f (x%.2f)  max/sum of alone %.0f/%.0f\n", name,
         m_alone, v_alone, m_both, m_both / m_alone, v_both, v_both / v_alone, m_alone > v_alone ? m_alone : v_alone, m_alone + v_alone);
  hipFree(ticks);
  hipFree(sink);
}

int main() {
  // repetitions chosen so that both roles alone take a similar time: a matrix repetition is 32 (64) cycles of the
  // matrix pipe, a vector repetition 13 instructions x 4 cycles
  run<1, false>("fp4 32x32x64, 1 chain", 20000);
  run<2, false>("fp4 32x32x64, 2 chains", 20000);
  run<2, true>("int8 32x32x32, 2 chains", 20000);
  const int cfg[][2] = {{1, 0}, {2, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}, {1, 2}, {1, 3}, {2, 2}};
  for (auto& c : cfg) run_mix(c[0], c[1], 10000);
  return 0;
}
