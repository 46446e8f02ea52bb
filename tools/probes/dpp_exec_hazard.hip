// Developer probe: does a DPP read see a lane that an SALU write to EXEC re-enabled a few instructions earlier?
// (Yes, always: 0 wrong lanes for every distance, EXEC cleared or halved -- this was the first suspect of the K1 fault
// of DESIGN 8.2 and is NOT its cause; the cause was the VALU-write -> DPP-read wait states.)
// One wave per SIMD (the regime of one-image launches).  EXEC is cleared (or halved), a branch over an empty body is
// taken, EXEC is restored, N wait states follow, then v_mov_b32_dpp wave_shr:1 with zero-fill.  Lane i must receive
// lane i-1's value; the probe counts lanes that received 0 instead, for N = 0 .. 6.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/dpp_exec_hazard tools/probes/dpp_exec_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int PARTIAL>
__global__ __launch_bounds__(64) void k(unsigned* bad, int reps) {
  const int lane = threadIdx.x;
  unsigned wrong = 0;
  for (int it = 0; it < reps; it++) {
    int v = lane * 7 + it + 1, got = -1;
    asm volatile(
        "s_mov_b64 s[10:11], exec\n\t"
        "s_mov_b64 s[12:13], %2\n\t"
        "s_and_b64 exec, exec, s[12:13]\n\t"
        "s_cbranch_execz 1f\n\t"
        "v_add_u32 %1, %1, 0\n\t"
        "1:\n\t"
        "s_or_b64 exec, exec, s[10:11]\n\t"
        ".rept %3\n\t s_nop 0\n\t .endr\n\t"
        "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
        "s_nop 1"
        : "+v"(got), "+v"(v)
        : "s"(PARTIAL ? 0x00000000ffffffffull : 0ull), "n"(N)
        : "s10", "s11", "s12", "s13");
    if (lane > 0 && got != (lane - 1) * 7 + it + 1) wrong++;
  }
  atomicAdd(bad, wrong);
}

template <int N, int PARTIAL>
void run(unsigned* d) {
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL((k<N, PARTIAL>), dim3(1024), dim3(64), 0, 0, d, 2000);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  std::printf("EXEC %s, %d wait state(s) between the restore and the DPP read: %u wrong lanes of %u\n", PARTIAL ? "halved " : "cleared", N, h,
              1024u * 63u * 2000u);
}

int main() {
  unsigned* d;
  hipMalloc(&d, 4);
  run<0, 0>(d); run<1, 0>(d); run<2, 0>(d); run<3, 0>(d); run<4, 0>(d); run<5, 0>(d); run<6, 0>(d);
  run<0, 1>(d); run<1, 1>(d); run<2, 1>(d); run<4, 1>(d);
  return 0;
}
