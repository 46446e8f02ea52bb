// dpp_chol.h -- the DPP building blocks of the small dense Cholesky factorisations (ba_fused.hip: baf_chol_kernel;
// chol.hip: the 224-blocks of the block-cyclic-reduction solver).  A wavefront keeps a 16 x 16 diagonal block one row per
// lane of each 16-lane DPP row (d[16]) and one panel row per lane beside it (p[16]); a pivot step is the broadcast of the
// pivot (v_mov_b64_dpp row_newbcast), rsq + two Newton steps, and per remaining column ONE v_fmac_f64_dpp for the diagonal
// block and one for the panel row: the neighbour lane's factor entry arrives as the DPP operand -- no LDS, no v_readlane,
// no barrier.  fp64 instructions accept no other DPP control than row_newbcast on gfx950; inline assembly that reads a
// fresh VALU result through DPP carries its own two wait states.
#pragma once
#include <hip/hip_runtime.h>

namespace {

template <int J>
__device__ __forceinline__ double cs_bcast(double v) {  // lane J of every 16-lane row -> all lanes of the row
  double r;
  // (two wait states between a VALU write and a DPP read of the same register: the compiler does not see into the asm)
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
  return r;
}
template <int K, bool NOP>
__device__ __forceinline__ void cs_fmac_bcast(double& acc, double bsrc, double own) {  // acc -= bsrc[lane K of the row] * own
  if (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(own), "n"(K));
  else
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(own), "n"(K));
}

// Pivot step J of a panel: d = the diagonal block's row of this lane, p = its panel row.  The updates of a column are
// split (round 4): only column J + 1 -- the next pivot's column -- is updated at once; the others are DEFERRED into the
// next step, where they are issued between the pivot's broadcast and the use of its reciprocal square root, i.e. under the
// latency of rsq and its two Newton steps instead of in front of it (a wavefront issues in order, and one wavefront per
// SIMD runs the panel: 16 columns x (chain ~150 cycles + 2 (15 - J) multiply-adds) became ~max of the two).  Every entry
// still receives its updates in column order: results are bit-identical to the undeferred form.
template <int J>
struct CsCol {
  // (ljp, pjp): column J - 1 of the factor (this lane's rows of the diagonal block / of the panel), not yet applied to
  // columns J + 1 .. 15
  static __device__ __forceinline__ void run(double (&d)[16], double (&p)[16], double* inv_out, bool& good, double ljp = 0.0,
                                             double pjp = 0.0) {
    const double dj = cs_bcast<J>(d[J]);
    if (!(dj > 0.0) || !isfinite(dj)) good = false;
    // 1 / sqrt by the hardware estimate (2^-23 relative) and two Newton steps (an IEEE sqrt + divide is ~200 dependent
    // instructions); a step is three dependent operations -- iv^2, fma(-dj / 2, iv^2, 3/2), product -- not four
    double iv = __builtin_amdgcn_rsq(dj);
    const double hj = 0.5 * dj;
    if constexpr (J > 0) upd<J - 1, J + 1>(d, p, ljp, pjp);
    iv = iv * fma(-hj, iv * iv, 1.5);
    iv = iv * fma(-hj, iv * iv, 1.5);
    if (inv_out) inv_out[J] = iv;  // (lane J of wavefront 0 only: the pointer is null elsewhere)
    const double lj = d[J] * iv;  // lane J: sqrt(pivot); lanes below: l(i, J)
    const double pj = p[J] * iv;
    d[J] = lj;
    p[J] = pj;
    if constexpr (J < 15) {
      cs_fmac_bcast<J + 1, true>(d[J + 1], lj, lj);   // d(i, J + 1) -= l(J + 1, J) l(i, J)
      cs_fmac_bcast<J + 1, false>(p[J + 1], lj, pj);  // p(r, J + 1) -= l(J + 1, J) l(r, J)
    }
    CsCol<J + 1>::run(d, p, inv_out, good, lj, pj);
  }
  template <int C, int K>  // column C of the factor into columns K .. 15
  static __device__ __forceinline__ void upd(double (&d)[16], double (&p)[16], double lj, double pj) {
    if constexpr (K < 16) {
      cs_fmac_bcast<K, false>(d[K], lj, lj);  // (lj was written a whole step ago: no wait states needed)
      cs_fmac_bcast<K, false>(p[K], lj, pj);
      upd<C, K + 1>(d, p, lj, pj);
    }
  }
};
template <>
struct CsCol<16> {
  static __device__ __forceinline__ void run(double (&)[16], double (&)[16], double*, bool&, double = 0.0, double = 0.0) {}
};

template <int J>
struct CsBack {  // step J (descending) of the transposed 16 x 16 solve: c[k] = l(k, lane) of the diagonal block
  static __device__ __forceinline__ void run(double& tp, double& x, const double (&c)[16], double ivl, int dr) {
    const double xj = tp * ivl;          // meaningful in lane J
    if (dr == J) x = xj;
    cs_fmac_bcast<J, true>(tp, xj, c[J]);  // t(lane) -= x_J l(J, lane)  (lanes >= J: c[J] = 0 or already solved)
    if constexpr (J > 0) CsBack<J - 1>::run(tp, x, c, ivl, dr);
  }
};

}  // namespace
