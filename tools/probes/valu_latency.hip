// Developer probe: what ONE wavefront per SIMD can issue -- a dependent chain against four independent chains of
// the same instruction (cycles per instruction), and the same with 3 waves per SIMD.  Decides how much
// instruction-level parallelism a low-occupancy kernel needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define LOOPS 4000
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, uint64_t* clk) {
  float a0 = threadIdx.x + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < LOOPS; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (OP == 0) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(a0));
      if (OP == 1) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0\n v_add_f64 %0, %0, %0" : "+v"(d0));
      if (OP == 3) asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
      if (OP == 4) asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0\n v_rsq_f32 %0, %0" : "+v"(a0));
      if (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(d0), "+v"(a1));
      if (OP == 6) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1" : "+v"(a0), "+v"(a1));
      if (OP == 7) asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1" : "+v"(d0), "+v"(d1));
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3);
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, float* out, uint64_t* clk, int blocks) {
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, clk);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(blocks);
  hipMemcpy(h.data(), clk, 8 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (auto v : h) cyc += v;
  // s_memtime runs at the constant 100 MHz reference: convert with the 2.4 GHz peak clock
  std::printf("%-44s %d waves/SIMD: %.2f cycles per instruction per wave (at 2.4 GHz)\n", name, blocks / 256,
              cyc / blocks * 24.0 / (LOOPS * 64.0));
}
int main() {
  float* out;
  uint64_t* clk;
  hipMalloc(&out, 4 * 256 * 256 * 8);
  hipMalloc(&clk, 8 * 256 * 8);
  for (int w : {1, 3}) {
    run<0>("v_add_f32 dependent chain", out, clk, 256 * w);
    run<6>("v_add_f32 two chains", out, clk, 256 * w);
    run<1>("v_add_f32 four chains", out, clk, 256 * w);
    run<2>("v_add_f64 dependent chain", out, clk, 256 * w);
    run<7>("v_add_f64 two chains", out, clk, 256 * w);
    run<3>("v_add_f64 four chains", out, clk, 256 * w);
    run<4>("v_rsq_f32 dependent chain", out, clk, 256 * w);
    run<5>("cvt f64<->f32 dependent chain", out, clk, 256 * w);
  }
  return 0;
}
