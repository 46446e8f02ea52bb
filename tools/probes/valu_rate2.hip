// Developer probe (round 2, second sheet): issue cost of the candidate replacements considered for K1 -- an ALU op with
// a DPP source used as a lane shift, rsq / rcp, integer converts -- and the rate of the LDS instructions a lane exchange
// through LDS memory would use (ds_write_b64 + ds_read2_b64 against ds_bpermute_b32), at 8 waves per SIMD.
// Prints cycles per wave-instruction per SIMD (LDS rows: per wave-instruction per CU / 4, i.e. the same normalisation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
#define LOOPS 2000

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, uint64_t* clk) {
  __shared__ double ex[4][66];
  float a0 = threadIdx.x * 1.0f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  typedef double dpair __attribute__((ext_vector_type(2)));
  dpair q1 = {1.0, 2.0}, q2 = {3.0, 4.0};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t la = (uint32_t)(uintptr_t)&ex[wave][lane];  // LDS byte address of this lane's slot
  const int bp = ((lane + 1) & 63) << 2;
  ex[wave][lane] = d0;
  ex[wave][(lane + 2) % 66] = d1;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < LOOPS; i++) {
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
      if (OP == 0) asm volatile("v_add_u32_dpp %0, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %1, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %2, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %3, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0));
      if (OP == 1) asm volatile("v_or_b32_dpp %0, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_or_b32_dpp %1, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_or_b32_dpp %2, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_or_b32_dpp %3, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(0));
      if (OP == 2) asm volatile("v_max_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_max_f32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_max_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_max_f32_dpp %3, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 3) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 4) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 5) asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 6) asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 7) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %1, %2\n v_cmp_ge_f32 vcc, %2, %3\n v_cmp_ge_f32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
      if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
      if (OP == 9) asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(bp));
      if (OP == 10) asm volatile("ds_write_b64 %4, %0\n ds_read2_b64 %1, %4 offset0:0 offset1:2\n ds_write_b64 %4, %0 offset:8\n ds_read2_b64 %2, %4 offset0:0 offset1:2\n s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(q1), "+v"(q2), "+v"(d3) : "v"(la));
      if (OP == 11) asm volatile("ds_write_b64 %4, %0\n ds_read_b64 %1, %4 offset:8\n ds_write_b64 %4, %2\n ds_read_b64 %3, %4 offset:16\n s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(la));
      if (OP == 12) asm volatile("ds_write_b32 %4, %0\n ds_read2_b32 %1, %4 offset0:0 offset1:2\n ds_write_b32 %4, %2\n ds_read_b32 %3, %4 offset:4\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(d1), "+v"(a2), "+v"(a3) : "v"(la));
      if (OP == 13) asm volatile("v_mul_legacy_f32 %0, %0, %1\n v_mul_legacy_f32 %1, %1, %2\n v_mul_legacy_f32 %2, %2, %3\n v_mul_legacy_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 14) asm volatile("v_sub_u32 %0, %0, %1\n v_sub_u32 %1, %1, %2\n v_sub_u32 %2, %2, %3\n v_sub_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      if (OP == 15) asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte2 %1, %1\n v_cvt_f32_ubyte3 %2, %2\n v_cvt_f32_ubyte0 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      // a K1-like mix: 40 cheap VALU slots + 3 ds_write_b64 + 3 ds_read2_b64 per "row step" (4 reps = one step)
      if (OP == 16) {
        asm volatile("ds_write_b64 %4, %0\n ds_read2_b64 %1, %4 offset0:0 offset1:2\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\ns_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(q1), "+v"(q2), "+v"(d3) : "v"(la));
      }
      // the same VALU work alone
      if (OP == 17) {
        asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(la));
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3 + q1.x + q1.y + q2.x + q2.y);
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int OP>
void run(const char* name, float* out, uint64_t* clk, int blocks, double per_rep4) {
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
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  const double cycles = ms * 1e-3 * ghz * 1e9;
  // per_rep4 = instructions of interest per unrolled group of four
  std::printf("%-34s %8.3f ms  clock %.2f GHz  %.2f cycles per wave-instruction per SIMD\n", name, ms, ghz,
              cycles / (waves_per_simd * LOOPS * (REP / 4) * per_rep4));
}

int main() {
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  float* out;
  uint64_t* clk;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clk, sizeof(uint64_t) * 2 * blocks);
  run<4>("v_mov_b32", out, clk, blocks, 4);
  run<0>("v_add_u32_dpp (x + 0) wave_shr", out, clk, blocks, 4);
  run<1>("v_or_b32_dpp (x | 0) wave_shr", out, clk, blocks, 4);
  run<2>("v_max_f32_dpp wave_shr", out, clk, blocks, 4);
  run<3>("v_add_f32_dpp row_shr", out, clk, blocks, 4);
  run<5>("v_rsq_f32", out, clk, blocks, 4);
  run<6>("v_cvt_f32_i32", out, clk, blocks, 4);
  run<15>("v_cvt_f32_ubyte1..3", out, clk, blocks, 4);
  run<7>("v_cmp_ge_f32", out, clk, blocks, 4);
  run<8>("v_cndmask_b32", out, clk, blocks, 4);
  run<13>("v_mul_legacy_f32", out, clk, blocks, 4);
  run<14>("v_sub_u32", out, clk, blocks, 4);
  run<9>("ds_bpermute_b32", out, clk, blocks, 4);
  run<10>("ds_write_b64 + ds_read2_b64 (pair)", out, clk, blocks, 2);
  run<11>("ds_write_b64 + ds_read_b64 (pair)", out, clk, blocks, 2);
  run<12>("ds_write_b32 + ds_read(2)_b32 (pair)", out, clk, blocks, 2);
  run<17>("12 v_add_f64 (group)", out, clk, blocks, 1);
  run<16>("12 v_add_f64 + w64 + r2_64 (group)", out, clk, blocks, 1);
  return 0;
}
