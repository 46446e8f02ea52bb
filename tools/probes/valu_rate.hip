// Developer probe: issue cost of the VALU instructions K1 is made of, at 8 waves per SIMD (all CUs busy).
// Prints cycles per wave-instruction per SIMD assuming the clock printed (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
#define LOOPS 2000

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, uint64_t* clk) {
  float a0 = threadIdx.x * 1.0f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < LOOPS; i++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      if (OP == 0) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 1) asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));
      if (OP == 4) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      if (OP == 6) asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 7) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 8) asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte0 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte0 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 9) asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 10) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_ge_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
      if (OP == 11) asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 12) asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      if (OP == 14) asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 15) asm volatile("v_med3_u32 %0, %0, %1, %2\n v_med3_u32 %1, %1, %2, %3\n v_med3_u32 %2, %2, %3, %0\n v_med3_u32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 16) asm volatile("v_min3_u32 %0, %0, %1, %2\n v_min3_u32 %1, %1, %2, %3\n v_min3_u32 %2, %2, %3, %0\n v_min3_u32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 17) asm volatile("v_med3_u32 %0, %0, %1, %2\n v_min_u32 %1, %1, %2\n v_med3_u32 %2, %2, %3, %0\n v_min_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 18) asm volatile("v_max_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_min_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 19) asm volatile("v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 20) asm volatile("v_med3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 22) asm volatile("v_max_i32 %0, %0, %1\n v_min_i32 %1, %1, %2\n v_max_i32 %2, %2, %3\n v_min_i32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 23) asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_max_u16 %1, %1, %2\n v_pk_min_u16 %2, %2, %3\n v_pk_max_u16 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 24) asm volatile("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 25) asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 26) asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 27) asm volatile("v_and_b32 %0, %0, %1\n v_lshlrev_b32 %1, 1, %2\n v_and_b32 %2, %2, %3\n v_lshlrev_b32 %3, 1, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 28) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
      if (OP == 29) asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 30) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %1, %1, %2, %3\n v_dot4_u32_u8 %2, %2, %3, %0\n v_dot4_u32_u8 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 31) asm volatile("v_mul_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_sub_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 13) asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3);
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int OP>
void run(const char* name, float* out, uint64_t* clk, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(2 * blocks);
  hipMemcpy(h.data(), clk, sizeof(uint64_t) * 2 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int b = 0; b < blocks; b++) {
    cyc += h[2 * b];
    real += h[2 * b + 1];
  }
  const double ghz = cyc / real * 0.1;  // s_memrealtime ticks at 100 MHz
  // waves per SIMD = blocks * 4 / (256 CUs * 4 SIMDs); instructions per wave = LOOPS * REP
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  const double cycles = ms * 1e-3 * ghz * 1e9;
  std::printf("%-16s %8.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD\n", name, ms, ghz,
              cycles / (waves_per_simd * LOOPS * REP));
}

int main() {
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  float* out;
  uint64_t* clk;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clk, sizeof(uint64_t) * 2 * blocks);
  run<0>("v_add_f32", out, clk, blocks);
  run<7>("v_fma_f32", out, clk, blocks);
  run<11>("v_add_u32", out, clk, blocks);
  run<1>("v_add_f64", out, clk, blocks);
  run<12>("v_fma_f64", out, clk, blocks);
  run<2>("v_cvt_f64_f32", out, clk, blocks);
  run<3>("v_cvt_f32_f64", out, clk, blocks);
  run<8>("v_cvt_f32_ubyte0", out, clk, blocks);
  run<4>("v_mov_b32_dpp", out, clk, blocks);
  run<5>("v_pk_mul_f32", out, clk, blocks);
  run<13>("v_pk_add_f32", out, clk, blocks);
  run<6>("v_sqrt_f32", out, clk, blocks);
  run<9>("v_max3_f32", out, clk, blocks);
  run<10>("v_cmp+v_cndmask", out, clk, blocks);
  run<14>("v_min_u32", out, clk, blocks);
  run<15>("v_med3_u32", out, clk, blocks);
  run<16>("v_min3_u32", out, clk, blocks);
  run<17>("v_med3+v_min_u32", out, clk, blocks);
  run<18>("v_max+v_min_u32", out, clk, blocks);
  run<22>("v_max+v_min_i32", out, clk, blocks);
  run<19>("v_min+v_max_f32", out, clk, blocks);
  run<20>("v_med3+v_min3_f32", out, clk, blocks);
  run<23>("v_pk_min/max_u16", out, clk, blocks);
  run<24>("v_rndne_f32", out, clk, blocks);
  run<25>("v_cvt_i32_f32", out, clk, blocks);
  run<26>("v_mad_u32_u24", out, clk, blocks);
  run<27>("v_and/v_lshlrev_b32", out, clk, blocks);
  run<28>("v_cmp_lt_u32", out, clk, blocks);
  run<29>("v_fract_f32", out, clk, blocks);
  run<30>("v_dot4_u32_u8", out, clk, blocks);
  run<31>("v_mul/v_sub_f32", out, clk, blocks);
  return 0;
}
