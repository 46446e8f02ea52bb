// chol.hip -- dense fp64 Cholesky solve for the reduced camera system of large (global) bundle
// adjustment: S x = b with S = (6C) x (6C), C up to ~1000 cameras.
//
// Ceres hands this system to a sparse Cholesky (SPARSE_SCHUR, include/visnav/map_utils.h:408); on an
// MI355X the dense factorisation of a 6000^2 matrix is ~72 GFLOP of fp64 and the matrix (288 MB)
// stays resident in HBM, so a plain blocked right-looking factorisation is used:
//   for each 32-column panel k:  (1) factor the 32x32 diagonal block and invert the factor in the registers
//                                    of one wavefront (column broadcasts through v_readlane),
//                                (2) panel below it = rows times Linv^T (a small dense product),
//                                (3) rank-32 update of the trailing lower triangle, 64x64 tiles,
//                                    4x4 register blocking, operands staged through LDS.
// Forward / backward substitution reuse the panel structure and the inverted diagonal blocks: one launch
// per panel and direction, no serial triangular solve.
//
// BAND FORM.  Every kernel addresses the matrix as A[i * ld + c] and visits only rows within `bw` of the panel, so
// the same code factors a dense matrix (ld = n, bw = n) and a symmetric BAND matrix in LAPACK-style lower band
// storage: row i keeps its entries c in [i - bws, i] contiguously, bws = bw + CH_NB, and the caller passes
// A = storage + bws, ld = bws  (then A[i * ld + c] = storage[i * (bws + 1) + (c - i + bws)]).  The extra CH_NB
// columns of explicit zeros are the ragged corner of a panel (rows up to k + nb + bw - 1 against columns from k), and
// they stay zero: a Cholesky factor has no fill outside the band.  The reduced camera system of a 500-keyframe loop
// is such a matrix once the cameras are ordered along the trajectory (ba.hip: reverse Cuthill-McKee on the
// covisibility graph): ~n bw^2 instead of n^3 / 3 operations and n (bws + 1) instead of n^2 doubles.
// Everything is deterministic
// (no atomics).  Written here rather than calling rocSOLVER: librocsolver.so is a 0.9 GB load that
// takes minutes to page in on a fresh machine.
#include <algorithm>
#include <vector>

#include "vsl_common.h"
#include "dpp_chol.h"

#define CH_NB 32

// (1) factor A[k:k+nb, k:k+nb] in place (lower) and invert the triangular factor, in the registers of ONE
// wavefront: lane i holds row i, a column is broadcast lane by lane through v_readlane (scalar operands),
// no LDS and no barriers.  Linv (row-major CH_NB x CH_NB, zero above the diagonal, identity-padded when
// nb < CH_NB) turns the panel solve and both substitutions into small dense products with no serial
// dependency chain.  ok = 0 if the block is not positive definite.
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, src_lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), src_lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ A, int ld, int k, int nb, int* __restrict__ ok,
                                                       double* __restrict__ Linv) {
  if (!*ok) return;
  const int lane = threadIdx.x & 31;  // lanes 32..63 mirror 0..31 (their results are discarded)
  double r[CH_NB];
#pragma unroll
  for (int c = 0; c < CH_NB; c++)
    r[c] = (lane < nb && c < nb && c <= lane) ? A[(size_t)(k + lane) * ld + k + c] : (lane == c ? 1.0 : 0.0);  // lower triangle only (band storage has no upper part)
  bool good = true;
  double dinv[CH_NB];  // 1 / L_jj, wave-uniform
#pragma unroll
  for (int j = 0; j < CH_NB; j++) {
    const double d = lane_bcast(r[j], j);
    good = good && (d > 0.0) && isfinite(d);
    const double sj = sqrt(d);
    dinv[j] = 1.0 / sj;
    const double lij = (lane == j) ? sj : r[j] * dinv[j];  // rows above the diagonal hold junk that is never read
    r[j] = lij;
#pragma unroll
    for (int c = j + 1; c < CH_NB; c++) r[c] -= lij * lane_bcast(lij, c);
  }
  if (!good) {
    if (threadIdx.x == 0) *ok = 0;
    return;
  }
  if (threadIdx.x < nb) {
#pragma unroll
    for (int c = 0; c < CH_NB; c++)
      if (c <= lane && c < nb) A[(size_t)(k + lane) * ld + k + c] = r[c];
  }
  // column `lane` of the inverse: forward substitution of L x = e_lane; L_rp comes from lane r's registers
  double x[CH_NB];
#pragma unroll
  for (int rr = 0; rr < CH_NB; rr++) {
    double s = (rr == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < CH_NB; p++)
      if (p < rr) s -= lane_bcast(r[p], rr) * x[p];
    x[rr] = (rr < lane) ? 0.0 : s * dinv[rr];
  }
  if (threadIdx.x < CH_NB) {
#pragma unroll
    for (int rr = 0; rr < CH_NB; rr++) Linv[rr * CH_NB + lane] = x[rr];
  }
}

// (2) rows i >= k+nb:  A[i, k:k+nb] <- A[i, k:k+nb] * L_kk^-T = sum_p A[i][p] Linv[c][p]
// 256 threads = 8 rows x 32 columns per step, 64 rows per workgroup.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int n, int ld, int k, int nb, const int* __restrict__ ok,
                                                         const double* __restrict__ Linv) {
  // n = one past the last row this panel reaches (min(matrix rows, k + nb + bw))
  __shared__ double Li[CH_NB][CH_NB + 1];
  __shared__ double R[64][CH_NB + 1];
  if (!*ok) return;
  const int i0 = k + nb + blockIdx.x * 64;
  for (int t = threadIdx.x; t < CH_NB * CH_NB; t += 256) Li[t / CH_NB][t % CH_NB] = Linv[t];
  for (int t = threadIdx.x; t < 64 * CH_NB; t += 256) {
    const int r = t / CH_NB, c = t % CH_NB;
    R[r][c] = (i0 + r < n && c < nb) ? A[(size_t)(i0 + r) * ld + k + c] : 0.0;
  }
  __syncthreads();
  const int c = threadIdx.x % CH_NB;
  for (int r = threadIdx.x / CH_NB; r < 64; r += 8) {
    double s = 0;
#pragma unroll
    for (int p = 0; p < CH_NB; p++) s += R[r][p] * Li[c][p];  // Linv is zero above its diagonal
    if (i0 + r < n && c < nb) A[(size_t)(i0 + r) * ld + k + c] = s;
  }
}

// (3) trailing update, lower triangle: C[i][j] -= sum_p P[i][p] P[j][p], i, j >= k+nb.
// grid.x enumerates tile pairs (ti >= tj) of 64x64 tiles; 256 threads, 4x4 outputs each.
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, int n, int ld, int k, int nb, int T,
                                                          const int* __restrict__ ok) {
  if (!*ok) return;
  // decode the (ti, tj) pair of this workgroup from its linear id over the lower triangle
  int ti = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ti++;
  while (ti * (ti + 1) / 2 > (int)blockIdx.x) ti--;
  const int tj = blockIdx.x - ti * (ti + 1) / 2;
  (void)T;
  const int base = k + nb;
  const int i0 = base + ti * 64, j0 = base + tj * 64;
  __shared__ double Pi[64][CH_NB + 1];
  __shared__ double Pj[64][CH_NB + 1];
  for (int t = threadIdx.x; t < 64 * CH_NB; t += 256) {
    const int r = t / CH_NB, c = t % CH_NB;
    Pi[r][c] = (i0 + r < n && c < nb) ? A[(size_t)(i0 + r) * ld + k + c] : 0.0;
    Pj[r][c] = (j0 + r < n && c < nb) ? A[(size_t)(j0 + r) * ld + k + c] : 0.0;
  }
  __syncthreads();
  const int tr = (threadIdx.x / 16) * 4, tc = (threadIdx.x % 16) * 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0;
#pragma unroll 8
  for (int p = 0; p < CH_NB; p++) {
    double x[4], y[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      x[a] = Pi[tr + a][p];
      y[a] = Pj[tc + a][p];
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] += x[a] * y[b];
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int i = i0 + tr + a, j = j0 + tc + b;
      if (i < n && j < n && j <= i) A[(size_t)i * ld + j] -= acc[a][b];
    }
}

// forward, panel k:  y_k = Linv_kk b_k  (every workgroup computes it, workgroup 0 stores it in y), then
// b[i] -= L[i, k:k+nb] . y_k for the rows i >= k+nb of this workgroup.  b_k itself is only read.
__global__ __launch_bounds__(256) void chol_fwd_kernel(const double* __restrict__ A, double* __restrict__ b, double* __restrict__ y,
                                                       const double* __restrict__ Linv, int n, int ld, int k, int nb) {
  __shared__ double bk[CH_NB], yk[CH_NB];
  if (threadIdx.x < CH_NB) bk[threadIdx.x] = threadIdx.x < nb ? b[k + threadIdx.x] : 0.0;
  __syncthreads();
  if (threadIdx.x < CH_NB) {
    double s = 0;
    for (int p = 0; p < CH_NB; p++) s += Linv[threadIdx.x * CH_NB + p] * bk[p];
    yk[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x < nb) y[k + threadIdx.x] = s;
  }
  __syncthreads();
  const int i = k + nb + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* row = A + (size_t)i * ld + k;
  double s = 0;
  for (int p = 0; p < nb; p++) s += row[p] * yk[p];
  b[i] -= s;
}

// backward, panel k:  x_k = Linv_kk^T y_k  (stored into b by workgroup 0), then y[c] -= sum_r L[k+r][c] x_k[r]
// for the columns c < k of this workgroup.
__global__ __launch_bounds__(256) void chol_bwd_kernel(const double* __restrict__ A, double* __restrict__ b, double* __restrict__ y,
                                                       const double* __restrict__ Linv, int c_first, int ld, int k, int nb) {
  __shared__ double yk[CH_NB], xk[CH_NB];
  if (threadIdx.x < CH_NB) yk[threadIdx.x] = threadIdx.x < nb ? y[k + threadIdx.x] : 0.0;
  __syncthreads();
  if (threadIdx.x < CH_NB) {
    double s = 0;
    for (int p = 0; p < CH_NB; p++) s += Linv[p * CH_NB + threadIdx.x] * yk[p];
    xk[threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x < nb) b[k + threadIdx.x] = s;
  }
  __syncthreads();
  const int c = c_first + blockIdx.x * 256 + threadIdx.x;  // columns [c_first, k): the band reaches no further left
  if (c >= k) return;
  double s = 0;
  for (int r = 0; r < nb; r++) s += A[(size_t)(k + r) * ld + c] * xk[r];
  y[c] -= s;
}

// ---------------------------------------------------------------------------------------------------------------
// Narrow bands (bw <= CBF_MAXBW): the whole solve -- factorisation, forward substitution (the right-hand side rides
// along as one more row of the matrix) and backward substitution -- in ONE launch of ONE workgroup.  A panel step
// touches only the bw rows below the diagonal block, ~bw^2 / 2 x 32 multiply-adds (0.8 MFLOP at bw = 221): too little
// to spread over the chip, and as separate launches the 188 panels of a 1000-camera system are 940 launches of 5-22 us
// (measured ~10 ms of a 14.7 ms LM iteration).  One step here, on a window staged in LDS (rows k .. k + 32 + m of the
// 32 panel columns, plus the right-hand side as the last row):
//   * four sub-steps of 8 columns: every wavefront factors the 8 x 8 diagonal sub-block redundantly in its lanes 0..7
//     (v_readlane column broadcasts), every thread solves ITS row against it and, after a barrier, subtracts its
//     contribution from the remaining columns of its row -- the diagonal-block rows are rows like any other, there is no
//     one-wavefront serial phase (a row-per-lane 32 x 32 factor on one wavefront measured 27k cycles per panel, the
//     32-column row solve behind it 26k: 60 KB of unrolled code);
//   * the m x m window update in 16 x 16 tiles of v_mfma_f64_16x16x4_f64 (faster here than register tiles on the
//     vector ALU, see the loop); rows of the band are written back coalesced;
// nothing to synchronise between workgroups, no flags, no spinning.
//
// TWO-ENDED FORM.  The 188 panel steps of a 1000-camera system are a dependent chain on one compute unit; a band
// matrix can be eliminated from both ends at once without any waiting between the two workgroups: workgroup 0 takes
// the unknowns [0, s) top-down, workgroup 1 the unknowns [N - s', N) bottom-up -- the same algorithm on the
// index-reversed view i' = N - 1 - i of the same storage (entry (i', j') of the view is A[N-1-j'][N-1-i']: a reversed
// panel is 32 storage rows x contiguous columns, so its accesses coalesce along the window-row direction instead of
// along the panel) -- and both stop at a separator of bw .. bw + 31 unknowns in the middle.  Unknowns on the two sides
// of it are more than bw apart, hence not coupled: the separator block receives the sum of the two Schur updates (the
// top one in place, the bottom one in a scratch block), is factored last by the single-ended kernel, and the two
// backward substitutions run outwards from it, again one workgroup each.  Half the chain, three more small launches.
// Cutting the band into more pieces does not pay at N / bw ~ 27: an interior piece has to carry its upper separator as
// bw extra rows through every step (3 x the update work) and the separator system has twice the bandwidth.
typedef double v4d_t __attribute__((ext_vector_type(4)));
#define CBF_THREADS 768  // 12 wavefronts = 3 per SIMD (with 9 -- one thread per window row was enough -- one SIMD carried 3 and the window update, bound by its matrix unit, ran at that SIMD's pace)
#define CBF_MAXBW 512
#define CBF_SUB 8
#define CBF_TCH 6  // window tiles per wavefront whose old values are in flight together

struct CbfView {
  double* A;       // band storage of the FULL system, pre-offset: A[i * ld + c], c <= i
  int ld;
  int N;           // unknowns of the full system (index reversal i -> N - 1 - i)
  int n;           // unknowns of this view: the chunk it eliminates followed by the separator
  int bw;
  int sep0, sepn;  // reversed view, factor phase: view rows [sep0, sep0 + sepn) are the separator -- their mutual entries
  double* M3;      //   accumulate in M3[(i - sep0) * sepn + (j - sep0)] and their right-hand side in b3[i - sep0]
  double* b3;      //   (sepn = 0: everything lives in A / b)
  double* b;       // right-hand side / solution of the full system
  double* y;       // per view: intermediate vector and 1 / pivot, by view index
  double* dinv;
  const double* pend0 = nullptr;  // plain view only: right-hand-side updates still to be subtracted when an entry is
  const double* pend1 = nullptr;  //   staged (block cyclic reduction: two slots per block)
};

// SEP (compile time): the step's window reaches the separator rows of a reversed view, whose mutual entries and
// right-hand side live in the scratch block; every other step addresses the band storage only
template <bool REV, bool SEP>
__device__ __forceinline__ double* cbf_at(const CbfView& v, int i, int j) {  // entry (i, j) of the view, j <= i
  if (!REV) return v.A + (size_t)i * v.ld + j;
  if (SEP && j >= v.sep0) return v.M3 + (size_t)(i - v.sep0) * v.sepn + (j - v.sep0);
  return v.A + (size_t)(v.N - 1 - j) * v.ld + (v.N - 1 - i);
}
template <bool REV, bool SEP>
__device__ __forceinline__ double* cbf_rhs(const CbfView& v, int i) {
  if (!REV) return v.b + i;
  if (SEP && i >= v.sep0) return v.b3 + (i - v.sep0);
  return v.b + (v.N - 1 - i);
}

struct CbfShared {
  double W[CH_NB + CBF_MAXBW + 1][CH_NB + 1];  // window rows x 32 panel columns (row stride 33: conflict-free)
  double red[CBF_THREADS / 32][CH_NB + 1];
  double xw[CBF_MAXBW];   // backward substitution: the solution entries the panel reaches
  double dinv_s[CH_NB];   // 1 / pivot of the panel's columns (a global store inside a sub-step would make its barrier wait for the round trip)
  double Ld_s[CBF_SUB][CBF_SUB + 1];  // the factored 8 x 8 diagonal sub-block of the current sub-step and
  double inv_s[CBF_SUB];              //   1 / its pivots, published by wavefront 0
  int fail_s;
};

// Visits this thread's elements of a rows x 32 block EIGHT at a time -- fn(row[8], c[8], ok[8]) -- in an order whose
// fastest index is contiguous in memory: the panel column in the plain view (idx -> (idx >> 5, idx & 31)), the window
// row in the reversed one (wave w takes columns w, w + 9, ...; its lanes 64 consecutive rows).  Eight independent
// accesses per call: as a one-element loop the staging was 14 dependent global round trips per thread.
template <bool REV, typename F>
__device__ __forceinline__ void cbf_visit(int rows, F&& fn) {
  const int tid = threadIdx.x;
  int row[8], c[8];
  bool ok[8];
  if (!REV) {
    const int total = rows * CH_NB;
    for (int i0 = tid; i0 < total; i0 += 8 * CBF_THREADS) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int idx = i0 + u * CBF_THREADS;
        ok[u] = idx < total;
        row[u] = idx >> 5;
        c[u] = idx & 31;
      }
      fn(row, c, ok);
    }
  } else {
    const int lane = tid & 63, wave = tid >> 6, RI = (rows + 63) >> 6;
    constexpr int NCC = (CH_NB + CBF_THREADS / 64 - 1) / (CBF_THREADS / 64);  // columns per wavefront
    for (int cc = 0; cc < NCC; cc += 2)
      for (int rr0 = 0; rr0 < RI; rr0 += 4) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
          c[u] = wave + (CBF_THREADS / 64) * (cc + (u >> 2));
          row[u] = (63 - lane) + 64 * (rr0 + (u & 3));  // lanes ascend in ADDRESS (descending view rows): the coalescer merges ascending lanes only (descending: 23k instead of 14k cycles per staging)
          ok[u] = c[u] < CH_NB && row[u] < rows;
        }
        fn(row, c, ok);
      }
  }
}

// W <- the panel of step k: diagonal block (lower triangle; completed with the identity when UNIT), the m rows below it
// and, when with_rhs, the right-hand side entries of the panel columns as one more row
template <bool REV, bool SEP, bool UNIT>
__device__ __forceinline__ void cbf_stage(const CbfView& v, double (*W)[CH_NB + 1], int k, int nb, int m, bool with_rhs) {
  const int rows = CH_NB + m + (with_rhs ? 1 : 0);
  cbf_visit<REV>(rows, [&](const int* row, const int* c, const bool* ok) {
    const double* ptr[8];
    double val[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      ptr[u] = nullptr;
      val[u] = 0.0;
      if (ok[u]) {
        const int r = row[u], cc = c[u];
        if (r < CH_NB) {
          if (r < nb && cc <= r)
            ptr[u] = cbf_at<REV, SEP>(v, k + r, k + cc);
          else if (UNIT && r == cc)
            val[u] = 1.0;
        } else if (r < CH_NB + m) {
          if (cc < nb) ptr[u] = cbf_at<REV, SEP>(v, k + nb + r - CH_NB, k + cc);
        } else if (cc < nb) {
          ptr[u] = cbf_rhs<REV, SEP>(v, k + cc);
          if (!REV && v.pend0) val[u] = -(v.pend0[k + cc] + v.pend1[k + cc]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (ptr[u]) val[u] += *ptr[u];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (ok[u]) W[row[u]][c[u]] = val[u];
  });
}

#ifdef CBF_TIMING
#define CBF_STAMP(q) { const long long t2 = __builtin_amdgcn_s_memtime(); tph[q] += t2 - tt; tt = t2; }
#define CBF_TARGS , long long* tph, long long& tt
#define CBF_TPASS , tph, tt
#else
#define CBF_STAMP(q)
#define CBF_TARGS
#define CBF_TPASS
#endif

// The panel in W (rows x 32: diagonal block, the rows below it, the right-hand side as the last row) is factored in
// place: four sub-steps of eight columns.  On return W holds the panel of L (and the solved right-hand-side row),
// sh.dinv_s the reciprocal pivots; sh.fail_s is set if a pivot was not positive.  Ends with a workgroup barrier.
template <int NTHR = CBF_THREADS>
__device__ __forceinline__ void cbf_panel_factor(double (*W)[CH_NB + 1], int rows, CbfShared& sh CBF_TARGS) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // The 8 x 8 diagonal sub-block of a sub-step is factored by wavefront 0 ALONE and published through LDS (every
  // wavefront factoring it redundantly -- ~400 dependent instructions -- cost three wavefronts' worth of issue slots
  // per SIMD and sub-step: 19k of the 86k cycles of a panel step).  For sub-steps 1..3 wavefront 0 does it inside the
  // previous sub-step's update phase, right after the tile that completes that sub-block, so no barrier is added.
  auto factor8 = [&](int j0) {  // wavefront 0 only; lanes 0..7 = the rows of the sub-block
    double dr[CBF_SUB];
#pragma unroll
    for (int c = 0; c < CBF_SUB; c++) dr[c] = lane < CBF_SUB ? W[j0 + lane][j0 + c] : (lane == c ? 1.0 : 0.0);
    bool good = true;
#pragma unroll
    for (int c = 0; c < CBF_SUB; c++) {
      const double d = lane_bcast(dr[c], c);
      if (!(d > 0.0) || !isfinite(d)) good = false;  // wave-uniform
      // 1 / sqrt(d) by the hardware estimate and two Newton steps, sqrt(d) = d / sqrt(d): the IEEE sqrt and divide
      // expansions are ~100 dependent fp64 instructions (~1000 cycles) per pivot, and the 32 pivots of a panel are
      // sequential -- measured 47k of the 100k cycles of a panel step
      double iv = __builtin_amdgcn_rsq(d);
      iv = iv * (1.5 - 0.5 * d * iv * iv);
      iv = iv * (1.5 - 0.5 * d * iv * iv);
      if (lane == 0) sh.inv_s[c] = iv;
      dr[c] = (lane == c) ? d * iv : dr[c] * iv;  // lanes > c: l(lane, c)
#pragma unroll
      for (int q = c + 1; q < CBF_SUB; q++) dr[q] -= dr[c] * lane_bcast(dr[c], q);  // meaningful for lanes >= q
    }
    if (lane < CBF_SUB) {
#pragma unroll
      for (int c = 0; c < CBF_SUB; c++) sh.Ld_s[lane][c] = dr[c];  // l(lane, c) for c <= lane (the rest is never read)
    }
    if (!good && lane == 0) sh.fail_s = 1;
  };
  if (wave == 0) factor8(0);
  __syncthreads();
  CBF_STAMP(4)
#pragma unroll 1
  for (int j0 = 0; j0 < CH_NB; j0 += CBF_SUB) {
    const bool act = tid < rows && tid >= j0;
    // x L_d^T = a: this row's entries in the sub-panel (for a row of the diagonal sub-block: that row of L_d)
    double x[CBF_SUB];
#pragma unroll
    for (int c = 0; c < CBF_SUB; c++) {
      double t = act ? W[tid][j0 + c] : 0.0;
#pragma unroll
      for (int q = 0; q < c; q++) t -= x[q] * sh.Ld_s[c][q];
      x[c] = t * sh.inv_s[c];
    }
    CBF_STAMP(5)
    if (act) {
#pragma unroll
      for (int c = 0; c < CBF_SUB; c++)
        if (tid >= j0 + CBF_SUB || c <= tid - j0) W[tid][j0 + c] = x[c];
    }
    if (tid < CBF_SUB) sh.dinv_s[j0 + tid] = sh.inv_s[tid];
    __syncthreads();
    CBF_STAMP(6)
    // the remaining panel columns lose X . L_d^T: W[r][cc] -= sum_q W[r][j0 + q] W[cc][j0 + q] for rows r >= j0 + 8 and
    // columns cc in [j0 + 8, 32) -- a rank-8 update in 16 x 16 tiles on the matrix unit (two v_mfma_f64_16x16x4_f64 per
    // tile; as one row per thread with broadcast LDS reads of the factor rows this phase was LDS-bound: 17k of the
    // 87k cycles of a panel step).  Entries above the diagonal of the diagonal block receive garbage; nothing reads them.
    // Tile 0 (wavefront 0's first) holds the next 8 x 8 diagonal sub-block.
    if (j0 + CBF_SUB < CH_NB) {
      const int r_first = (j0 + CBF_SUB) & ~15, c_first = j0 + CBF_SUB < 16 ? 0 : 16;
      const int n_rt = (rows - r_first + 15) >> 4, n_ct = (CH_NB - c_first) >> 4;
      const int kq = lane >> 4, l16 = lane & 15;
      for (int t = wave; t < n_rt * n_ct; t += NTHR / 64) {
        const int rt = t / n_ct, ct = t - rt * n_ct;
        const int R0 = r_first + 16 * rt, C0 = c_first + 16 * ct;
        const int ra = min(R0 + l16, rows - 1), rb = C0 + l16;
        v4d_t acc = {0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[ra][j0 + kq], W[rb][j0 + kq], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[ra][j0 + 4 + kq], W[rb][j0 + 4 + kq], acc, 0, 0, 0);
        const int cc = C0 + l16;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int r = R0 + kq + 4 * q;
          if (r < rows && r >= j0 + CBF_SUB && cc >= j0 + CBF_SUB) W[r][cc] -= acc[q];
        }
        if (t == 0) factor8(j0 + CBF_SUB);  // wavefront 0 (t == wave == 0): its own LDS writes above are ordered before these reads
      }
    }
    CBF_STAMP(7)
    __syncthreads();
    CBF_STAMP(8)
  }
}
// One panel step of the view: factorisation of columns [k, k + 32), forward substitution, window update.
// Returns false (workgroup-uniform) if a pivot is not positive.
template <bool REV, bool SEP>
__device__ __forceinline__ bool cbf_step(const CbfView& v, int k, CbfShared& sh CBF_TARGS) {
  double(*W)[CH_NB + 1] = sh.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = v.n, bw = v.bw;
  {
    const int nb = min(CH_NB, n - k);
    const int m = min(n - k - nb, bw);
    const int rows = CH_NB + m + 1;  // diagonal block, panel, right-hand side
    const int base = k + nb;
    cbf_stage<REV, SEP, true>(v, W, k, nb, m, true);
    __syncthreads();
    CBF_STAMP(0)
    cbf_panel_factor(W, rows, sh CBF_TPASS);
    CBF_STAMP(1)
    if (sh.fail_s) return false;  // workgroup-uniform
    // write the factored rows back to the band (coalesced), y_k, and b of the window rows
    cbf_visit<REV>(CH_NB + m, [&](const int* row, const int* c, const bool* ok) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        if (!ok[u]) continue;
        const int r = row[u], cc = c[u];
        if (r < CH_NB) {
          if (r < nb && cc <= r) *cbf_at<REV, SEP>(v, k + r, k + cc) = W[r][cc];
        } else if (cc < nb) {
          *cbf_at<REV, SEP>(v, k + nb + r - CH_NB, k + cc) = W[r][cc];
        }
      }
    });
    if (tid < nb) {
      v.y[k + tid] = W[rows - 1][tid];
      v.dinv[k + tid] = sh.dinv_s[tid];
    }
    if (tid < m) {
      double dot = 0.0;
#pragma unroll
      for (int c = 0; c < CH_NB; c++) dot += W[CH_NB + tid][c] * W[rows - 1][c];
      *cbf_rhs<REV, SEP>(v, k + nb + tid) -= dot;
    }
    CBF_STAMP(9)
    if (m > 0) {
      // window update A[i][j] -= sum_c P[i][c] P[j][c], j <= i (P = the solved panel, W rows 32 ..): 16 x 16 tiles of
      // the lower triangle, CBF_THREADS / 64 apart per wavefront, eight v_mfma_f64_16x16x4_f64 each in two chains.
      // Plain view: lane l feeds P[i0 + (l & 15)][c0 + (l >> 4)] and P[j0 + (l & 15)][c0 + (l >> 4)] and receives rows
      // (l >> 4) + 4 reg, column l & 15 of the tile, so a register is four 128-byte row segments of the band storage;
      // reversed view: the operands trade places and the lane receives the TRANSPOSED tile, i.e. view rows
      // i0 + (l & 15) -- contiguous storage columns -- of view columns j0 + (l >> 4) + 4 reg.  All old values of a
      // chunk of tiles are requested before its matrix instructions.  Measured alternatives, cycles per panel step at
      // bw = 221: 4 x 4 register tiles on the vector ALU with uncoalesced accesses 45k, the same with 512-byte row
      // segments 90k (LDS-read bound), MFMA one chain 40k, two chains 35k.
      const int T = (m + 15) >> 4, ntiles = T * (T + 1) / 2;
      const int kq = lane >> 4, l16 = lane & 15;
      int ti = 0, tj = wave;  // tile index -> (ti, tj) by integer stepping (an fp64 sqrt per tile costs more than the tile)
      for (int tile0 = wave; tile0 < ntiles; tile0 += CBF_TCH * (CBF_THREADS / 64)) {
        double old[CBF_TCH][4];
        int tis[CBF_TCH], tjs[CBF_TCH];
#pragma unroll
        for (int u = 0; u < CBF_TCH; u++) {
          while (tj > ti) {
            tj -= ti + 1;
            ti++;
          }
          tis[u] = ti;
          tjs[u] = tj;
          const bool have = tile0 + u * (CBF_THREADS / 64) < ntiles;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int i = REV ? 16 * ti + 15 - l16 : 16 * ti + kq + 4 * q;  // (reversed: lanes ascend in address)
            const int j = REV ? 16 * tj + kq + 4 * q : 16 * tj + l16;
            old[u][q] = (have && i < m && j <= i) ? *cbf_at<REV, SEP>(v, base + i, base + j) : 0.0;
          }
          tj += CBF_THREADS / 64;
        }
#pragma unroll
        for (int u = 0; u < CBF_TCH; u++) {
          const bool have = tile0 + u * (CBF_THREADS / 64) < ntiles;
          const int ra = CH_NB + min(16 * tis[u] + (REV ? 15 - l16 : l16), m - 1), rb = CH_NB + min(16 * tjs[u] + l16, m - 1);
          const int r0 = REV ? rb : ra, r1 = REV ? ra : rb;
          v4d_t acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int c0 = 0; c0 < CH_NB / 2; c0 += 4) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[r0][c0 + kq], W[r1][c0 + kq], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[r0][CH_NB / 2 + c0 + kq], W[r1][CH_NB / 2 + c0 + kq], acc2, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int i = REV ? 16 * tis[u] + 15 - l16 : 16 * tis[u] + kq + 4 * q;
            const int j = REV ? 16 * tjs[u] + kq + 4 * q : 16 * tjs[u] + l16;
            if (have && i < m && j <= i) *cbf_at<REV, SEP>(v, base + i, base + j) = old[u][q] - (acc[q] + acc2[q]);
          }
        }
      }
    }
    __syncthreads();
    CBF_STAMP(2)
  }
  return true;
}

// Panel steps k in [0, k_end) of the view (k_end a multiple of 32, or the view's n): factorisation and forward
// substitution.  Returns false (workgroup-uniform) if a pivot is not positive.
template <bool REV>
__device__ __forceinline__ bool cbf_factor(const CbfView& v, int k_end, CbfShared& sh) {
  if (threadIdx.x == 0) sh.fail_s = 0;
  __syncthreads();
#ifdef CBF_TIMING
  long long tph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tt = __builtin_amdgcn_s_memtime();
#endif
  bool good = true;
  for (int k = 0; k < k_end && good; k += CH_NB) {
    bool sep_step = false;
    if (REV) sep_step = v.sepn > 0 && k + CH_NB + v.bw > v.sep0;  // the window reaches the separator rows
    if (REV && sep_step)
      good = cbf_step<REV, REV>(v, k, sh CBF_TPASS);
    else
      good = cbf_step<REV, false>(v, k, sh CBF_TPASS);
  }
#ifdef CBF_TIMING
  if (threadIdx.x == 0) printf("chol_band_fused ticks (block %d): stage %lld rest-of-factor %lld update %lld | factor8 %lld solve %lld store+barrier %lld rowupdate %lld barrier %lld writeback+b %lld\n", (int)blockIdx.x, tph[0], tph[1], tph[2], tph[4], tph[5], tph[6], tph[7], tph[8], tph[9]);
#endif
  return good;
}

// Backward substitution L^T x = y over the view's panels [0, k_end), last panel first; x overwrites the right-hand side.
// The solution entries beyond k_end (the separator, for a chunk) must already be in place (in b: no scratch here).
template <bool REV>
__device__ __forceinline__ void cbf_backward(const CbfView& v, int k_end, CbfShared& sh) {
  double(*W)[CH_NB + 1] = sh.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31;
  const int n = v.n, bw = v.bw;
  const int n_panels = (k_end + CH_NB - 1) / CH_NB;
  for (int pi = n_panels - 1; pi >= 0; pi--) {
    const int k = pi * CH_NB;
    const int nb = min(CH_NB, n - k);
    const int m = min(n - k - nb, bw);
    // stage the factored panel (diagonal block + the m rows below it) and the solution entries it reaches
    cbf_stage<REV, false, false>(v, W, k, nb, m, false);
    if (tid < m) sh.xw[tid] = *cbf_rhs<REV, false>(v, k + nb + tid);
    __syncthreads();
    {
      const int c = tid & 31, part = tid >> 5;  // CBF_THREADS / 32 parts x 32 columns
      constexpr int NP = CBF_THREADS / 32;
      double sum = 0.0;
      for (int t = part; t < m; t += NP) sum += W[CH_NB + t][c] * sh.xw[t];
      sh.red[part][c] = sum;
    }
    __syncthreads();
    if (wave == 0) {
      double sacc = 0.0;
#pragma unroll
      for (int part = 0; part < CBF_THREADS / 32; part++) sacc += sh.red[part][l32];
      double val = l32 < nb ? v.y[k + l32] - sacc : 0.0;
      const double dv = l32 < nb ? v.dinv[k + l32] : 1.0;
      // column `l32` of the diagonal factor: rT[c] = L[c][l32], c >= l32
      double rT[CH_NB];
#pragma unroll
      for (int c = 0; c < CH_NB; c++) rT[c] = (c < nb && l32 < nb && c >= l32) ? W[c][l32] : 0.0;
      double xv = 0.0;
#pragma unroll
      for (int c = CH_NB - 1; c >= 0; c--) {
        const double xc = lane_bcast(val, c) * lane_bcast(dv, c);
        if (l32 == c) xv = xc;
        if (l32 < c) val -= rT[c] * xc;
      }
      if (lane < nb) *cbf_rhs<REV, false>(v, k + lane) = xv;
    }
    __syncthreads();
  }
}

// the whole solve of one band system by one workgroup
__global__ __launch_bounds__(CBF_THREADS) void chol_band_fused_kernel(CbfView v, int* __restrict__ ok) {
  __shared__ CbfShared sh;
  if (!*ok) return;
  if (!cbf_factor<false>(v, v.n, sh)) {
    if (threadIdx.x == 0) *ok = 0;
    return;
  }
  cbf_backward<false>(v, v.n, sh);
}

// two-ended form, phase 1: workgroup 0 factors the top chunk of `top`, workgroup 1 the (reversed) bottom chunk of `rev`
__global__ __launch_bounds__(CBF_THREADS) void cbf2_factor_kernel(CbfView top, int k_end_top, CbfView rev, int k_end_rev,
                                                                  int* __restrict__ ok) {
  __shared__ CbfShared sh;
  const bool good = blockIdx.x == 0 ? cbf_factor<false>(top, k_end_top, sh) : cbf_factor<true>(rev, k_end_rev, sh);
  if (!good && threadIdx.x == 0) *ok = 0;
}

// phase 2a: the separator system = separator block of the storage (top update applied in place) + the bottom update
// (M3, in reversed coordinates: view index a = sepn - 1 - u), into a band storage of its own; likewise its right-hand side
__global__ void cbf2_mid_build_kernel(const double* __restrict__ A, int ld, int s, int sepn, const double* __restrict__ M3,
                                      const double* __restrict__ b, const double* __restrict__ b3, double* __restrict__ Am,
                                      int ldm, double* __restrict__ bm, const int* __restrict__ ok) {
  if (!*ok) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= sepn * sepn) return;
  const int u = idx / sepn, w = idx - u * sepn;
  if (w == 0) bm[u] = b[s + u] + b3[sepn - 1 - u];
  if (w > u) return;
  // (u, w), u >= w, is (a, c) = (sepn - 1 - u, sepn - 1 - w) in reversed coordinates with a <= c: stored as M3[c][a]
  Am[(size_t)u * ldm + w] = A[(size_t)(s + u) * ld + s + w] + M3[(size_t)(sepn - 1 - w) * sepn + (sepn - 1 - u)];
}

__global__ void cbf2_scatter_kernel(const double* __restrict__ xm, double* __restrict__ b, int s, int sepn, const int* __restrict__ ok) {
  if (!*ok) return;
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < sepn) b[s + u] = xm[u];
}

// phase 3: both backward substitutions, outwards from the separator
__global__ __launch_bounds__(CBF_THREADS) void cbf2_backward_kernel(CbfView top, int k_end_top, CbfView rev, int k_end_rev,
                                                                    const int* __restrict__ ok) {
  __shared__ CbfShared sh;
  if (!*ok) return;
  if (blockIdx.x == 0)
    cbf_backward<false>(top, k_end_top, sh);
  else
    cbf_backward<true>(rev, k_end_rev, sh);
}

#define BCR_MAXB 256  // block size limit of the cyclic reduction: static LDS Us[256][33] + Lp[32][257] = 133 KB (dynamic LDS above 64 KiB is refused by the runtime)
int vsl_chol_solve_bcr_dev(vsl_ctx* ctx, double* S, double* b, int n, int ld, int bw, int* ok_dev, int cyclic, double* neg_out = nullptr);

// Solves S x = b in place (S destroyed, b <- x).  *ok_dev = 1 on success, 0 if S is not SPD.
// Dense: ld = n, bw = n.  Band: S = storage + bws, ld = bws = bw + CH_NB (see the file header); bw = the largest
// i - c of a non-zero entry.
__global__ void chol_negate_kernel(int n, const double* __restrict__ x, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -x[i];
}
int vsl_chol_solve_band_impl(vsl_ctx* ctx, double* S, double* b, int n, int ld, int bw, int* ok_dev);
// neg_out (nullable): -x as well (the bundle adjustment's step is the negated solution: saves its launch on the BCR path)
int vsl_chol_solve_band_dev(vsl_ctx* ctx, double* S, double* b, int n, int ld, int bw, int* ok_dev, int cyclic, double* neg_out) {
  if (cyclic) return vsl_chol_solve_bcr_dev(ctx, S, b, n, ld, bw, ok_dev, 1, neg_out);  // (the caller has checked the layout)
  if (ld != n && !ctx->chol_no_fused && !ctx->chol_no_bcr && (bw + 1 + 31) / 32 * 32 <= BCR_MAXB &&
      n >= 8 * ((bw + 1 + 31) / 32 * 32))  // long narrow band: block cyclic reduction over the whole chip
    return vsl_chol_solve_bcr_dev(ctx, S, b, n, ld, bw, ok_dev, 0, neg_out);
  const int rc = vsl_chol_solve_band_impl(ctx, S, b, n, ld, bw, ok_dev);
  if (rc == VSL_OK && neg_out) {
    hipLaunchKernelGGL(chol_negate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, b, neg_out);
    VSL_CHECK_LAUNCH(ctx);
  }
  return rc;
}

int vsl_chol_solve_band_impl(vsl_ctx* ctx, double* S, double* b, int n, int ld, int bw, int* ok_dev) {
  const int one = 1;
  const int n_panels = (n + CH_NB - 1) / CH_NB;
  if (ld != n && bw <= CBF_MAXBW && !ctx->chol_no_fused) {  // narrow band: one launch for the whole solve, or two-ended
    VSL_HIP(ctx, hipMemcpyAsync(ok_dev, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    const bool two_ended = !ctx->chol_one_ended && bw + CH_NB - 1 <= CBF_MAXBW && n >= 8 * (bw + CH_NB);
    if (!two_ended) {
      void* ws = nullptr;
      int rc = vsl_ctx_dscratch(ctx, sizeof(double) * (2 * (size_t)n + 16), &ws);
      if (rc) return rc;
      CbfView v = {S, ld, n, n, bw, 0, 0, nullptr, nullptr, b, (double*)ws, (double*)ws + n};
      hipLaunchKernelGGL(chol_band_fused_kernel, dim3(1), dim3(CBF_THREADS), 0, ctx->stream, v, ok_dev);
      VSL_CHECK_LAUNCH(ctx);
      return VSL_OK;
    }
    // top chunk [0, s), bottom chunk [n - sp, n) (both whole panels), separator [s, s + sepn) with bw <= sepn <= bw + 31
    const int s = ((n - bw) / 2) / CH_NB * CH_NB;
    const int sp = (n - s - bw) / CH_NB * CH_NB;
    const int sepn = n - s - sp;
    const int bwm = sepn - 1, ldm = bwm + CH_NB;  // the separator system, dense, in band storage of its own
    const size_t mid_elems = (size_t)sepn * (ldm + 1) + 64;
    const size_t need = 4 * (size_t)n + (size_t)sepn * sepn + 4 * (size_t)sepn + mid_elems + 64;
    void* ws = nullptr;
    int rc = vsl_ctx_dscratch(ctx, sizeof(double) * need, &ws);
    if (rc) return rc;
    double* p = (double*)ws;
    double *y_top = p, *dinv_top = p + n, *y_rev = p + 2 * (size_t)n, *dinv_rev = p + 3 * (size_t)n;
    p += 4 * (size_t)n;
    double* M3 = p;             // sepn x sepn, followed by b3: cleared together
    double* b3 = p + (size_t)sepn * sepn;
    p += (size_t)sepn * sepn + sepn;
    double *bm = p, *ym = p + sepn, *dinvm = p + 2 * (size_t)sepn;
    p += 3 * (size_t)sepn;
    double* Am_store = p;       // band storage of the separator system: cleared (explicit zeros outside the lower triangle)
    VSL_HIP(ctx, hipMemsetAsync(M3, 0, sizeof(double) * ((size_t)sepn * sepn + sepn), ctx->stream));
    VSL_HIP(ctx, hipMemsetAsync(Am_store, 0, sizeof(double) * mid_elems, ctx->stream));
    CbfView top = {S, ld, n, s + sepn, bw, 0, 0, nullptr, nullptr, b, y_top, dinv_top};
    CbfView rev = {S, ld, n, sp + sepn, bw, sp, sepn, M3, b3, b, y_rev, dinv_rev};
    hipLaunchKernelGGL(cbf2_factor_kernel, dim3(2), dim3(CBF_THREADS), 0, ctx->stream, top, s, rev, sp, ok_dev);
    double* Am = Am_store + ldm;
    hipLaunchKernelGGL(cbf2_mid_build_kernel, dim3((sepn * sepn + 255) / 256), dim3(256), 0, ctx->stream, S, ld, s, sepn, M3, b,
                       b3, Am, ldm, bm, ok_dev);
    CbfView mid = {Am, ldm, sepn, sepn, bwm, 0, 0, nullptr, nullptr, bm, ym, dinvm};
    hipLaunchKernelGGL(chol_band_fused_kernel, dim3(1), dim3(CBF_THREADS), 0, ctx->stream, mid, ok_dev);
    hipLaunchKernelGGL(cbf2_scatter_kernel, dim3((sepn + 255) / 256), dim3(256), 0, ctx->stream, bm, b, s, sepn, ok_dev);
    rev.sepn = 0;  // the separator's solution is read from b now
    hipLaunchKernelGGL(cbf2_backward_kernel, dim3(2), dim3(CBF_THREADS), 0, ctx->stream, top, s, rev, sp, ok_dev);
    VSL_CHECK_LAUNCH(ctx);
    return VSL_OK;
  }
  void* ws = nullptr;  // inverse diagonal blocks + the intermediate vector y
  int rc = vsl_ctx_dscratch(ctx, sizeof(double) * ((size_t)n_panels * CH_NB * CH_NB + (size_t)n + 8), &ws);
  if (rc) return rc;
  double* Linv = (double*)ws;
  double* y = Linv + (size_t)n_panels * CH_NB * CH_NB;
  VSL_HIP(ctx, hipMemcpyAsync(ok_dev, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0, pi = 0; k < n; k += CH_NB, pi++) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    double* Li = Linv + (size_t)pi * CH_NB * CH_NB;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, ctx->stream, S, ld, k, nb, ok_dev, Li);
    const int m = std::min(n - k - nb, bw);  // rows below the diagonal block that the band reaches
    if (m > 0) {
      const int n_end = k + nb + m;
      hipLaunchKernelGGL(chol_panel_kernel, dim3((m + 63) / 64), dim3(256), 0, ctx->stream, S, n_end, ld, k, nb, ok_dev, Li);
      const int T = (m + 63) / 64;
      hipLaunchKernelGGL(chol_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, ctx->stream, S, n_end, ld, k, nb, T, ok_dev);
    }
  }
  VSL_CHECK_LAUNCH(ctx);
  // (a failed factorisation leaves Linv / y undefined; the caller looks at *ok_dev before using b)
  for (int k = 0, pi = 0; k < n; k += CH_NB, pi++) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    const int m = std::min(n - k - nb, bw);
    hipLaunchKernelGGL(chol_fwd_kernel, dim3(m > 0 ? (m + 255) / 256 : 1), dim3(256), 0, ctx->stream, S, b, y,
                       Linv + (size_t)pi * CH_NB * CH_NB, k + nb + std::max(m, 0), ld, k, nb);
  }
  for (int pi = n_panels - 1; pi >= 0; pi--) {
    const int k = pi * CH_NB;
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    const int c_first = std::max(0, k + nb - 1 - (bw + CH_NB - 1));  // row k + nb - 1 reaches back to column (k + nb - 1) - bw at most
    const int cols = k - c_first;
    hipLaunchKernelGGL(chol_bwd_kernel, dim3(cols > 0 ? (cols + 255) / 256 : 1), dim3(256), 0, ctx->stream, S, b, y,
                       Linv + (size_t)pi * CH_NB * CH_NB, c_first, ld, k, nb);
  }
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}

int vsl_chol_solve_dev(vsl_ctx* ctx, double* S, double* b, int n, int* ok_dev) {
  return vsl_chol_solve_band_dev(ctx, S, b, n, n, n, ok_dev);
}

// Test / diagnostic entry point: the solver on a caller-provided system (include/vslam_hip.h).
extern "C" int vsl_spd_solve(vsl_ctx* ctx, const double* S, const double* b, int n, int half_bandwidth, double* x) {
  if (!ctx || !S || !b || !x || n <= 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_spd_solve: bad argument");
  const bool band = half_bandwidth >= 0 && half_bandwidth < n - 1;
  const int bw = band ? half_bandwidth : n, bws = bw + CH_NB;
  const size_t elems = band ? (size_t)n * (bws + 1) + 64 : (size_t)n * n;
  double *dS = nullptr, *db = nullptr;
  int* dok = nullptr;
  VSL_HIP(ctx, hipMalloc((void**)&dS, sizeof(double) * elems));
  int rc = VSL_OK, ok = 0;
  hipError_t e = hipMalloc((void**)&db, sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc((void**)&dok, sizeof(int));
  if (e == hipSuccess) {
    if (band) {
      // row i keeps columns [i - bws, i] at storage[i * (bws + 1) ...]; entries beyond the half bandwidth stay zero
      std::vector<double> st(elems, 0.0);
      for (int i = 0; i < n; i++)
        for (int c = std::max(0, i - bw); c <= i; c++) st[(size_t)i * (bws + 1) + (c - i + bws)] = S[(size_t)i * n + c];
      e = hipMemcpyAsync(dS, st.data(), sizeof(double) * elems, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    } else {
      e = hipMemcpyAsync(dS, S, sizeof(double) * elems, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(db, b, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      rc = band ? vsl_chol_solve_band_dev(ctx, dS + bws, db, n, bws, bw, dok) : vsl_chol_solve_band_dev(ctx, dS, db, n, n, n, dok);
      if (rc == VSL_OK) {
        e = hipMemcpyAsync(&ok, dok, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(x, db, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
    }
  }
  (void)hipFree(dS);
  (void)hipFree(db);
  (void)hipFree(dok);
  if (e != hipSuccess) return vsl_fail(ctx, VSL_ERR_HIP, "vsl_spd_solve: %s", hipGetErrorString(e));
  if (rc != VSL_OK) return rc;
  if (!ok) return vsl_fail(ctx, VSL_ERR_NUMERIC, "vsl_spd_solve: matrix is not positive definite");
  return VSL_OK;
}

// Host-only query (no device needed): the ring layout the cyclic solver would use for n unknowns of half bandwidth
// half_bandwidth -- *block = the kernels' block size, *n_blocks the ring length; returns 0 when there is none
// (vsl_spd_solve_cyclic / the bundle adjustment then keep the linear band form).
bool vsl_chol_bcr_cyclic_layout(int n, int bw, int* B_out, int* nblk_out);
extern "C" int vsl_bcr_cyclic_layout(int n, int half_bandwidth, int* block, int* n_blocks) {
  int B = 0, nb = 0;
  if (n <= 0 || half_bandwidth < 0 || !vsl_chol_bcr_cyclic_layout(n, half_bandwidth, &B, &nb)) return 0;
  if (block) *block = B;
  if (n_blocks) *n_blocks = nb;
  return 1;
}

// Test / diagnostic entry point for the CYCLIC band form: S (dense, row-major, symmetric) has non-zeros only where the
// cyclic distance min(|i - j|, n - |i - j|) <= half_bandwidth.
bool vsl_chol_bcr_cyclic_layout(int n, int bw, int* B_out, int* nblk_out);
extern "C" int vsl_spd_solve_cyclic(vsl_ctx* ctx, const double* S, const double* b, int n, int half_bandwidth, double* x) {
  if (!ctx || !S || !b || !x || n <= 0 || half_bandwidth < 0) return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_spd_solve_cyclic: bad argument");
  const int bw = half_bandwidth, bws = bw + CH_NB;
  int Bq, nq;
  if (!vsl_chol_bcr_cyclic_layout(n, bw, &Bq, &nq))
    return vsl_fail(ctx, VSL_ERR_INVALID, "vsl_spd_solve_cyclic: %d unknowns with half bandwidth %d have no cyclic block layout (>= 8 blocks of >= bw + 1 and <= %d unknowns)", n, bw, BCR_MAXB);
  const size_t elems = (size_t)n * (bws + 1) + 64;
  double *dS = nullptr, *db = nullptr;
  int* dok = nullptr;
  VSL_HIP(ctx, hipMalloc((void**)&dS, sizeof(double) * elems));
  int rc = VSL_OK, ok = 0;
  hipError_t e = hipMalloc((void**)&db, sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc((void**)&dok, sizeof(int));
  if (e == hipSuccess) {
    // row i keeps columns [i - bws, i]; a column below zero is column c + n of the wrap-around corner
    std::vector<double> st(elems, 0.0);
    for (int i = 0; i < n; i++)
      for (int c = i - bw; c <= i; c++) st[(size_t)i * (bws + 1) + (c - i + bws)] = S[(size_t)i * n + (c < 0 ? c + n : c)];
    e = hipMemcpyAsync(dS, st.data(), sizeof(double) * elems, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db, b, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      rc = vsl_chol_solve_bcr_dev(ctx, dS + bws, db, n, bws, bw, dok, 1);
      if (rc == VSL_OK) {
        e = hipMemcpyAsync(&ok, dok, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(x, db, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      }
    }
  }
  (void)hipFree(dS);
  (void)hipFree(db);
  (void)hipFree(dok);
  if (e != hipSuccess) return vsl_fail(ctx, VSL_ERR_HIP, "vsl_spd_solve_cyclic: %s", hipGetErrorString(e));
  if (rc != VSL_OK) return rc;
  if (!ok) return vsl_fail(ctx, VSL_ERR_NUMERIC, "vsl_spd_solve_cyclic: matrix is not positive definite");
  return VSL_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// BLOCK CYCLIC REDUCTION of a long narrow band (n >= 8 blocks of B >= bw + 1 unknowns).  The band Cholesky above is a
// chain of n / 32 dependent panel steps on one compute unit (two with the two-ended form); cutting the band into
// blocks of B unknowns makes it block tridiagonal, and eliminating every other block is independent work:
//   level 0: blocks 1, 3, 5, ... are factored at once (one workgroup each, the single-workgroup kernel on a dense B x B
//   block), their Schur updates D_p -= U1^T U1, D_q -= U2^T U2 and the fill K(q, p) = -U2^T U1 between their two
//   neighbours (U1 = L_e^-1 K(e, p), U2 = L_e^-1 K(q, e)^T) are dense B x B products spread over the chip; the
//   remaining blocks 0, 2, 4, ... are block tridiagonal again, and so on: log2(n / B) levels whose critical path is one
//   dense B x B Cholesky each instead of n / 32 panel steps -- at 2.7 x the flops.  It is the Cholesky factorisation
//   under a nested-dissection ordering of the blocks: no pivoting issues for an SPD matrix, no atomics, reproducible.
// The right-hand side rides along (z_e = L_e^-1 b_e, b_p -= U1^T z_e, b_q -= U2^T z_e) and the solution is recovered
// level by level in reverse: x_e = L_e^-T (z_e - U1 x_p - U2 x_q).
struct BcrJob {
  int e, p, q;      // eliminated block and its two active neighbours (q = -1: none below)
  int kp, kq;       // indices of K(e, p) and K(q, e) in this level's coupling array
  int knew;         // index of K(q, p) in the next level's coupling array (-1: none)
  int u;            // slot of U1 / U2
};

// band storage -> dense blocks: D[i] (lower triangle, identity-padded past the block's unknowns), C[i] = block (i + 1, i),
// b[i].  Block i holds the unknowns [off[i], off[i + 1]) (<= B of them).  CYCLIC band (round 4): the storage of row i
// keeps columns [i - bws, i]; for i < bws the leading slots -- columns "below zero" -- hold the entries (i, j - n) of the
// wrap-around corner (a camera loop ordered along its trajectory: first and last keyframes see each other), and
// C[nblk - 1] = block (0, nblk - 1) closes the ring of blocks.
__global__ void bcr_extract_kernel(const double* __restrict__ A, int ld, int n, int bws, const double* __restrict__ b, int B,
                                   int nblk, const int* __restrict__ off, int cyclic, double* __restrict__ D,
                                   double* __restrict__ C, double* __restrict__ bb, double* __restrict__ pend,
                                   int* __restrict__ ok_out) {
  const int blk = blockIdx.y;
  const size_t BB = (size_t)B * B;
  const int o0 = off[blk], sz = off[blk + 1] - o0;
  const int nxt = blk + 1 < nblk ? blk + 1 : 0;  // rows of the coupling block: the next block (the first one for the last)
  const int o1 = off[nxt], sz1 = off[nxt + 1] - o1;
  const bool has_c = blk + 1 < nblk || cyclic;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * B; idx += gridDim.x * blockDim.x) {
    const int r = idx / B, c = idx - r * B;
    double d = 0.0;
    if (r < sz && c < sz) {
      if (c <= r) d = A[(size_t)(o0 + r) * ld + o0 + c];  // r - c <= B - 1 <= bws: inside the row's slots (zeros beyond the band)
    } else if (r == c) {
      d = 1.0;
    }
    D[blk * BB + idx] = d;
    if (has_c) {
      double v = 0.0;
      if (r < sz1 && c < sz) {
        const int gr = o1 + r;
        const int gc = blk + 1 < nblk ? o0 + c : o0 + c - n;  // the wrap coupling reads the slots of the columns "below zero"
        if (gr - gc <= bws) v = A[(size_t)gr * ld + gc];
      }
      C[blk * BB + idx] = v;
    }
  }
  if (blockIdx.x == 0)
    for (int r = threadIdx.x; r < B; r += blockDim.x) {
      bb[(size_t)blk * B + r] = r < sz ? b[o0 + r] : 0.0;
      pend[((size_t)2 * blk) * B + r] = 0.0;
      pend[((size_t)2 * blk + 1) * B + r] = 0.0;
    }
  if (blk == 0 && blockIdx.x == 0 && threadIdx.x == 0) *ok_out = 1;  // (the factorisations clear it; this kernel does not read it)
}

// c0 += c1^T (ring of two blocks: both couplings join the same pair); out = in (the wrap coupling of an odd ring moves on)
__global__ void bcr_add_transposed_kernel(int B, double* __restrict__ c0, const double* __restrict__ c1, const int* __restrict__ ok) {
  if (!*ok) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * B) return;
  const int r = idx / B, c = idx - r * B;
  c0[idx] += c1[(size_t)c * B + r];
}
__global__ void bcr_copy_block_kernel(int B, const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ ok) {
  if (!*ok) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < B * B) out[idx] = in[idx];
}

// Dense B x B block (B <= 256) with its trailing matrix in REGISTERS: the 16 x 16 tiles of the lower triangle are
// dealt to the 12 wavefronts once (<= 12 tiles = 96 VGPRs each, in the accumulator layout of v_mfma_f64_16x16x4_f64) and
// stay there -- a panel step updates them with the matrix unit straight from the panel in LDS, and the tiles of the next
// panel's two tile columns are written into the LDS window when their turn comes.  No global round trip between the
// steps (the band kernel on such a block spent 88k of its 389k cycles staging panels and 102k in the window update's
// load -> MFMA -> store chain); global traffic is one read of the block and one write of its factor.
// y = L^-1 (b - pending updates) rides along as the window's last row; L (lower), 1 / pivots and y go to global memory.
// The 32-column panel of bcr_chol_registers factored with the DPP scheme of dpp_chol.h (round 4) instead of
// cbf_panel_factor's four 8-column sub-steps (factor8 by one wavefront with v_readlane broadcasts + a row solve through
// LDS per sub-step): two 16-column halves, each ONE pass in which every wavefront that has panel rows factors the
// diagonal block in its DPP rows and solves its rows beside it, and between them the rank-16 update of columns 16..31 as
// 16 x 16 tiles on the matrix unit.  Same contract as cbf_panel_factor: W holds the panel of L and the solved right-hand
// side row, sh.dinv_s the reciprocal pivots, sh.fail_s is set on a bad pivot, ends with a workgroup barrier.
#ifdef BCR_TIMING
#define BCP_TARG , long long* sh_tp
#define BCP_TPASS , tq2
#else
#define BCP_TARG
#define BCP_TPASS
#endif
template <int NTHR>
__device__ __forceinline__ void bcr_panel_factor_dpp(double (*W)[CH_NB + 1], int rows, CbfShared& sh BCP_TARG) {
  static_assert(CH_NB == 32, "two 16-column halves");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l16 = lane & 15;
#ifdef BCR_TIMING
  long long tt = __builtin_amdgcn_s_memtime();
#define BCP_STAMP(q) { const long long t2 = __builtin_amdgcn_s_memtime(); sh_tp[q] += t2 - tt; tt = t2; }
#else
#define BCP_STAMP(q)
#endif
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    const int c0 = 16 * half, r0 = c0 + 16, nrow = rows - r0;  // panel rows below the diagonal block, the right-hand side included
    if (wave * 64 < nrow) {  // wave-uniform
      const int prow = r0 + wave * 64 + lane;
      const bool valid = prow < rows;
      double d[16], p[16];
#pragma unroll
      for (int k = 0; k < 16; k++) {
        d[k] = W[c0 + l16][c0 + k];
        p[k] = valid ? W[prow][c0 + k] : 0.0;
      }
      bool good = true;
      BCP_STAMP(0)
      CsCol<0>::run(d, p, (wave == 0 && lane == 0) ? sh.dinv_s + c0 : (double*)nullptr, good);
      BCP_STAMP(1)
      if (wave == 0 && lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) W[c0 + lane][c0 + k] = k <= lane ? d[k] : 0.0;
      }
      if (valid) {
#pragma unroll
        for (int k = 0; k < 16; k++) W[prow][c0 + k] = p[k];
      }
      if (!good && lane == 0) sh.fail_s = 1;
      BCP_STAMP(2)
    }
    __syncthreads();
    BCP_STAMP(3)
    if (half == 0) {
      // columns 16..31 of the rows from 16 on lose their products with the first sixteen columns
      const int nt = (rows - 16 + 15) >> 4;
      for (int t = wave; t < nt; t += NTHR / 64) {
        const int R0 = 16 + 16 * t;
        const int ra = min(R0 + l16, rows - 1), rb = 16 + l16;
        v4d_t acc;
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = W[min(R0 + kq + 4 * q, rows - 1)][16 + l16];
#pragma unroll
        for (int m = 0; m < 4; m++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[ra][4 * m + kq], W[rb][4 * m + kq], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int r = R0 + kq + 4 * q;
          if (r < rows) W[r][16 + l16] = acc[q];
        }
      }
      BCP_STAMP(4)
      __syncthreads();
      BCP_STAMP(5)
    }
  }
}

#ifndef BCR_DPP_PANEL
#define BCR_DPP_PANEL 1  // 0: the round-2 panel factorisation (four 8-column sub-steps), kept for A/B builds
#endif
#define BCR_REG_THREADS 512  // 8 wavefronts = 2 per SIMD: 256 VGPRs each, room for the tiles beside the panel factorisation
template <int SLOTS>  // tiles per wavefront: 14 for B <= 224 (105 tiles), 17 for B <= 256 (136 tiles)
__device__ __forceinline__ bool bcr_chol_registers(int B, double* __restrict__ Dg, const double* __restrict__ bg, const double* __restrict__ pend0,
                                   const double* __restrict__ pend1, double* __restrict__ yg, double* __restrict__ dinvg,
                                   CbfShared& sh) {
  double(*W)[CH_NB + 1] = sh.W;
  double* bvec = sh.xw;  // B <= 256 entries of the (updated) right-hand side
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l16 = lane & 15;
  const int NT = B >> 4, ntiles = NT * (NT + 1) / 2;
  if (tid == 0) sh.fail_s = 0;
  for (int r = tid; r < B; r += BCR_REG_THREADS) bvec[r] = bg[r] - (pend0[r] + pend1[r]);
  // tile id = wave + 12 * slot -> (ti, tj), tj <= ti, row-major over the lower triangle; the table lives in LDS (two
  // more registers per slot would push the kernel into scratch)
  unsigned short* tile_of = (unsigned short*)sh.red;  // ti | tj << 8, 0xffff = no tile
  for (int id = tid; id < SLOTS * (BCR_REG_THREADS / 64); id += BCR_REG_THREADS) {
    int ti = (int)((__fsqrt_rn(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while (ti * (ti + 1) / 2 > id) ti--;
    while ((ti + 1) * (ti + 2) / 2 <= id) ti++;
    tile_of[id] = id < ntiles ? (unsigned short)(ti | ((id - ti * (ti + 1) / 2) << 8)) : (unsigned short)0xffff;
  }
  __syncthreads();
#ifdef BCR_TIMING
  long long tq[6] = {0, 0, 0, 0, 0, 0}, tq2[6] = {0, 0, 0, 0, 0, 0}, tt = __builtin_amdgcn_s_memtime();
#define BCR_STAMP(q) { const long long t2 = __builtin_amdgcn_s_memtime(); tq[q] += t2 - tt; tt = t2; }
#else
#define BCR_STAMP(q)
#endif
  double acc[SLOTS][4];  // (plain doubles: an array of ext-vectors stayed in scratch memory)
  // the wavefront's tile ids are wave-uniform: held in scalar registers (round 4; read from the LDS table in front of
  // every slot of every panel step they were 14 dependent LDS round trips per step in both tile loops)
  unsigned tsl[SLOTS];
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) tsl[sl] = (unsigned)__builtin_amdgcn_readfirstlane((int)tile_of[wave + (BCR_REG_THREADS / 64) * sl]);
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) {
    const unsigned t = tsl[sl];
    const int ti = t & 255, tj = t >> 8;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[sl][q] = t != 0xffffu ? Dg[(size_t)(16 * ti + kq + 4 * q) * B + 16 * tj + l16] : 0.0;
  }
  __syncthreads();
  BCR_STAMP(0)
  for (int k = 0; k < B; k += CH_NB) {
    const int m = B - k - CH_NB, rows = CH_NB + m + 1, tk = k >> 4;  // tk: first tile column of the panel
    // the panel into the window: the owners of tile columns tk, tk + 1 write their tiles (rows from matrix row k)
#pragma unroll
    for (int sl = 0; sl < SLOTS; sl++) {
      const unsigned t = tsl[sl];
      const int ti = t & 255, tj = t >> 8;
      if (t != 0xffffu && (tj == tk || tj == tk + 1)) {
#pragma unroll
        for (int q = 0; q < 4; q++) W[16 * ti - k + kq + 4 * q][16 * (tj - tk) + l16] = acc[sl][q];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < CH_NB) W[rows - 1][tid] = bvec[k + tid];
    __syncthreads();
    BCR_STAMP(1)
    if (BCR_DPP_PANEL)
      bcr_panel_factor_dpp<BCR_REG_THREADS>(W, rows, sh BCP_TPASS);
    else
      cbf_panel_factor<BCR_REG_THREADS>(W, rows, sh);
    BCR_STAMP(2)
    if (sh.fail_s) return false;  // workgroup-uniform (cbf_panel_factor ends with a barrier)
    // the factored panel, y and the reciprocal pivots to global memory; right-hand side of the rows below
    for (int idx = tid; idx < (CH_NB + m) * CH_NB; idx += BCR_REG_THREADS) {
      const int r = idx >> 5, c = idx & 31;
      if (r >= CH_NB || c <= r) Dg[(size_t)(k + r) * B + k + c] = W[r][c];
    }
    if (tid < CH_NB) {
      yg[k + tid] = W[rows - 1][tid];
      dinvg[k + tid] = sh.dinv_s[tid];
    }
    if (tid < m) {
      double dot = 0.0;
#pragma unroll
      for (int c = 0; c < CH_NB; c++) dot += W[CH_NB + tid][c] * W[rows - 1][c];
      bvec[k + CH_NB + tid] -= dot;
    }
    BCR_STAMP(3)
    // trailing tiles: acc -= P_i P_j^T, P = the solved panel rows (W row of matrix row r is r - k)
#pragma unroll
    for (int sl = 0; sl < SLOTS; sl++) {
      const unsigned t = tsl[sl];
      const int ti = t & 255, tj = t >> 8;
      if (t != 0xffffu && tj >= tk + 2) {
        const int ra = 16 * ti - k + l16, rb = 16 * tj - k + l16;
        v4d_t a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c0 = 0; c0 < CH_NB / 2; c0 += 4) {
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[ra][c0 + kq], W[rb][c0 + kq], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(W[ra][CH_NB / 2 + c0 + kq], W[rb][CH_NB / 2 + c0 + kq], a2, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc[sl][q] -= a1[q] + a2[q];
      }
      __builtin_amdgcn_sched_barrier(0);  // one tile at a time: hoisting every slot's 16 operand loads costs 400 registers
    }
    BCR_STAMP(4)
    __syncthreads();
    BCR_STAMP(5)
  }
#ifdef BCR_TIMING
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 3 || wave == 7))
    printf("bcr_chol wave %d: load %lld | tiles->W %lld | panel %lld | store+dot %lld | trailing %lld | barrier %lld  (x10 ns)\n", wave,
           tq[0], tq[1], tq[2], tq[3], tq[4], tq[5]);
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 3 || wave == 7))
    printf("   panel wave %d: lds->regs %lld | columns %lld | regs->lds %lld | barrier %lld | rank-16 %lld | barrier %lld\n", wave, tq2[0],
           tq2[1], tq2[2], tq2[3], tq2[4], tq2[5]);
#endif
  return true;
}

// dense Cholesky of the blocks jobs[].e of one level + z = L^-1 b (in y), by the single-workgroup band kernel
template <int SLOTS>
__global__ __launch_bounds__(BCR_REG_THREADS) void bcr_chol_kernel(const BcrJob* __restrict__ jobs, int B, double* __restrict__ D,
                                                               double* __restrict__ bb, double* __restrict__ yy,
                                                               double* __restrict__ dinv, const double* __restrict__ pend,
                                                               int* __restrict__ ok) {
  __shared__ CbfShared sh;
  const int e = jobs[blockIdx.x].e;
  // the right-hand-side updates this block received from eliminated neighbours (two slots: from the neighbour above and
  // from the one below, each written by one workgroup per level) are subtracted where the panel's entries are staged
  const bool good = bcr_chol_registers<SLOTS>(B, D + (size_t)e * B * B, bb + (size_t)e * B, pend + ((size_t)2 * e) * B,
                                              pend + ((size_t)2 * e + 1) * B, yy + (size_t)e * B, dinv + (size_t)e * B, sh);
  if (!good && threadIdx.x == 0) *ok = 0;
}

// the last remaining block: the whole solve
template <int SLOTS>
__global__ __launch_bounds__(BCR_REG_THREADS) void bcr_last_kernel(int e, int B, double* __restrict__ D, double* __restrict__ bb,
                                                               double* __restrict__ yy, double* __restrict__ dinv,
                                                               const double* __restrict__ pend, int* __restrict__ ok) {
  __shared__ CbfShared sh;
  if (!*ok) return;
  const bool good = bcr_chol_registers<SLOTS>(B, D + (size_t)e * B * B, bb + (size_t)e * B, pend + ((size_t)2 * e) * B,
                                              pend + ((size_t)2 * e + 1) * B, yy + (size_t)e * B, dinv + (size_t)e * B, sh);
  if (!good && threadIdx.x == 0) *ok = 0;
}

// ... and its backward substitution (a launch of its own: cbf_backward is written for CBF_THREADS threads)
__global__ __launch_bounds__(CBF_THREADS) void bcr_last_back_kernel(int e, int B, double* __restrict__ D, double* __restrict__ bb,
                                                                    double* __restrict__ yy, double* __restrict__ dinv,
                                                                    const int* __restrict__ ok) {
  __shared__ CbfShared sh;
  if (!*ok) return;
  CbfView v = {D + (size_t)e * B * B, B, B, B, B - 1, 0, 0, nullptr, nullptr, bb + (size_t)e * B, yy + (size_t)e * B, dinv + (size_t)e * B};
  cbf_backward<false>(v, B, sh);
}

// inverses of the 32 x 32 diagonal blocks of the factors of one level (one wavefront per block: lane = row of L in
// registers, column broadcasts by v_readlane, lane j solves L x = e_j): they turn the triangular part of the solves
// below into small dense products.  Linv[(job * B / 32 + blk) * 1024 + r * 32 + c]
__global__ __launch_bounds__(64) void bcr_dinv_kernel(const BcrJob* __restrict__ jobs, int B, const double* __restrict__ D,
                                                      const double* __restrict__ dinv, double* __restrict__ Linv,
                                                      const int* __restrict__ ok) {
  if (!*ok) return;
  const BcrJob jb = jobs[blockIdx.x];
  const int blk = blockIdx.y, lane = threadIdx.x & 31, k0 = 32 * blk;
  const double* L = D + (size_t)jb.e * B * B;
  double r[32], x[32];
#pragma unroll
  for (int c = 0; c < 32; c++) r[c] = c <= lane ? L[(size_t)(k0 + lane) * B + k0 + c] : 0.0;
  const double* di = dinv + (size_t)jb.e * B + k0;
#pragma unroll
  for (int rr = 0; rr < 32; rr++) {
    double t = (rr == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < rr; p++) t -= lane_bcast(r[p], rr) * x[p];
    x[rr] = (rr < lane) ? 0.0 : t * di[rr];
  }
  double* out = Linv + ((size_t)jb.u * (B / 32) + blk) * 1024;
  if (threadIdx.x < 32) {
#pragma unroll
    for (int rr = 0; rr < 32; rr++) out[rr * 32 + lane] = x[rr];
  }
}

// U = L_e^-1 R for 32 right-hand-side columns per workgroup: tiles 0 .. B/32 - 1 are columns of R1 = K(e, p), the next
// B/32 columns of R2 = K(q, e)^T.  Forward substitution by 32-row panels of L: the panel's product with the rows
// already solved by all 256 threads, its 32 x 32 diagonal part by one thread per column.
__global__ __launch_bounds__(256) void bcr_trsm_kernel(const BcrJob* __restrict__ jobs, int B, const double* __restrict__ D,
                                                       const double* __restrict__ Linv, const double* __restrict__ K,
                                                       double* __restrict__ U, const double* __restrict__ yy,
                                                       double* __restrict__ pend, const int* __restrict__ ok) {
  if (!*ok) return;
  __shared__ double Us[BCR_MAXB * 33];
  __shared__ double Lp[32 * (BCR_MAXB + 1)];
  const BcrJob jb = jobs[blockIdx.x];
  const int tiles = B / 32, tile = blockIdx.y;
  const bool second = tile >= tiles;
  if (second && jb.q < 0) return;
  const int c0 = 32 * (second ? tile - tiles : tile);
  const size_t BB = (size_t)B * B;
  const double* L = D + (size_t)jb.e * BB;
  const double* Li = Linv + (size_t)jb.u * (B / 32) * 1024;
  __shared__ double Ls[32 * 33];
  const double* R = K + (size_t)(second ? jb.kq : jb.kp) * BB;
  double* Uo = U + ((size_t)2 * jb.u + (second ? 1 : 0)) * BB;
  const int tid = threadIdx.x;
  // the right-hand-side tile into Us: R1[r][c0 + c], or R2[r][c0 + c] = K(q, e)[c0 + c][r]
  // (all global loads eight at a time: as one-element loops every load waited for the previous one -- 28 dependent
  // round trips per panel)
  if (!second) {
    for (int r0 = tid >> 5; r0 < B; r0 += 64) {  // 8 rows per thread per round, 32 contiguous doubles per row
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = r0 + 8 * u < B ? R[(size_t)(r0 + 8 * u) * B + c0 + (tid & 31)] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (r0 + 8 * u < B) Us[(r0 + 8 * u) * 33 + (tid & 31)] = v[u];
    }
  } else {
    for (int cc = 0; cc < 32; cc += 8) {  // R2[r][c0 + c] = K(q, e)[c0 + c][r]: a row of K per column, contiguous in r = tid
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = tid < B ? R[(size_t)(c0 + cc + u) * B + tid] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (tid < B) Us[tid * 33 + cc + u] = v[u];
    }
  }
  __syncthreads();
  const int col = tid & 31, grp = tid >> 5;  // 8 groups x 4 rows (the right-hand-side dot product below)
  const int wv = tid >> 6, lane = tid & 63, ti = wv >> 1, tj = wv & 1, kq = lane >> 4, l16 = lane & 15;
  // The panel's rows of L left of its diagonal block (column k = tid: B <= 256) and the inverse of that block are
  // fetched into registers TWO PANELS AHEAD (round 4): as loads at the top of a panel step they were two exposed global
  // round trips on a chain of seven dependent steps, and one step of two small matrix products is shorter than a round
  // trip.  Three barriers per step, both products on the matrix unit.
  // (thread = column pair c2 = tid & 127 of the rows 2 u + (tid >> 7): sixteen 16-byte loads, not thirty-two of 8)
  const int c2 = tid & 127, rh = tid >> 7;
  auto fetch = [&](int pn, double (&lv)[32], double (&li)[4]) {
    if (pn >= B) return;
    const double* src = L + (size_t)(pn + rh) * B + 2 * c2;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const double2 v = 2 * c2 < pn ? *(const double2*)(src + (size_t)(2 * u) * B) : make_double2(0.0, 0.0);
      lv[2 * u] = v.x;
      lv[2 * u + 1] = v.y;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) li[u] = Li[(size_t)(pn >> 5) * 1024 + tid + 256 * u];
  };
#ifdef BCR_TIMING
  long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tt = __builtin_amdgcn_s_memtime();
#define TRS_STAMP(q) { const long long t2 = __builtin_amdgcn_s_memtime(); tq[q] += t2 - tt; tt = t2; }
#else
#define TRS_STAMP(q)
#endif
  auto step = [&](int p0, double (&lv)[32], double (&li)[4]) {
    TRS_STAMP(0)
    if (2 * c2 < p0) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        Lp[(2 * u + rh) * (B + 1) + 2 * c2] = lv[2 * u];
        Lp[(2 * u + rh) * (B + 1) + 2 * c2 + 1] = lv[2 * u + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int idx = tid + 256 * u;
      Ls[(idx >> 5) * 33 + (idx & 31)] = li[u];
    }
    TRS_STAMP(1)
    __syncthreads();
    TRS_STAMP(2)
    fetch(p0 + 64, lv, li);
    TRS_STAMP(3)
    {  // T_p = R_p - L[p, 0:p0] U[0:p0]: one 16 x 16 tile per wavefront on the matrix unit (two chains); as four rows
       // per thread on the vector ALU this product was LDS-bound (five LDS reads per four multiply-adds)
      v4d_t acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
      const double* lrow = Lp + (16 * ti + l16) * (B + 1) + kq;
      const double* ucol = Us + kq * 33 + 16 * tj + l16;
      // (p0 is a multiple of 32: eight k-steps per round, their sixteen LDS operand reads issued together -- read
      // one step at a time, every matrix instruction waited for an LDS round trip: 28k of the kernel's 60k cycles)
      for (int k0 = 0; k0 < p0; k0 += 32) {
        double la[8], ub[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          la[u] = lrow[k0 + 4 * u];
          ub[u] = ucol[(k0 + 4 * u) * 33];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[u], ub[u], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[u + 1], ub[u + 1], acc2, 0, 0, 0);
        }
      }
#pragma unroll
      for (int qq = 0; qq < 4; qq++) Us[(p0 + 16 * ti + kq + 4 * qq) * 33 + 16 * tj + l16] -= acc[qq] + acc2[qq];
    }
    TRS_STAMP(4)
    __syncthreads();
    TRS_STAMP(5)
    {  // the 32 x 32 diagonal part: U_p = Linv_pp T_p, the same tile of the same wavefront
      v4d_t acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
      const double* arow = Ls + (16 * ti + l16) * 33 + kq;
      const double* tcol = Us + (p0 + kq) * 33 + 16 * tj + l16;
#pragma unroll
      for (int k0 = 0; k0 < 32; k0 += 8) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k0], tcol[k0 * 33], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k0 + 4], tcol[(k0 + 4) * 33], acc2, 0, 0, 0);
      }
      __syncthreads();
#pragma unroll
      for (int qq = 0; qq < 4; qq++) Us[(p0 + 16 * ti + kq + 4 * qq) * 33 + 16 * tj + l16] = acc[qq] + acc2[qq];
    }
    TRS_STAMP(6)
  };
  double lvA[32], liA[4], lvB[32], liB[4];
#pragma unroll
  for (int u = 0; u < 32; u++) lvA[u] = 0.0;  // (panel 0 has no rows to its left)
#pragma unroll
  for (int u = 0; u < 4; u++) liA[u] = Li[tid + 256 * u];
  fetch(32, lvB, liB);
  for (int p0 = 0; p0 < B; p0 += 64) {
    step(p0, lvA, liA);
    if (p0 + 32 < B) step(p0 + 32, lvB, liB);
  }
  __syncthreads();
  TRS_STAMP(7)
#ifdef BCR_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
    printf("bcr_trsm: stage R + prefetch %lld | lds stores %lld | barrier %lld | fetch issue %lld | product %lld | barrier %lld | diag %lld | tail %lld\n",
           tq[0], tq[1], tq[2], tq[3], tq[4], tq[5], tq[6], tq[7]);
#endif
  for (int r = tid >> 5; r < B; r += 8) Uo[(size_t)r * B + c0 + (tid & 31)] = Us[r * 33 + (tid & 31)];
  // right-hand side of the neighbour: (U^T z_e) for these 32 columns, accumulated in the neighbour's pending slot
  // (slot 1 of p: written by the eliminated block below it; slot 0 of q: by the one above; one writer per level)
  {
    __shared__ double part[8][33];
    const double* z = yy + (size_t)jb.e * B;
    double sacc = 0.0;
    for (int r = grp; r < B; r += 8) sacc += Us[r * 33 + col] * z[r];
    part[grp][col] = sacc;
    __syncthreads();
    if (tid < 32) {
      const int tgt = second ? jb.q : jb.p;
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < 8; g++) t += part[g][tid];
      pend[((size_t)2 * tgt + (second ? 0 : 1)) * B + c0 + tid] += t;
    }
  }
}

// C (B x B) = C - M^T N   (mode 0: D_p -= U1^T U1, lower tiles only; mode 1: D_q -= U2^T U2, lower tiles only) or
// K(q, p) = -U2^T U1 (mode 2, all tiles): one 32 x 32 tile per workgroup, the two 32-column slabs of M and N in LDS
__global__ __launch_bounds__(256) void bcr_gemm_kernel(const BcrJob* __restrict__ jobs, int B, int mode0, const double* __restrict__ U,
                                                       double* __restrict__ D, double* __restrict__ Knext,
                                                       const int* __restrict__ ok) {
  if (!*ok) return;
  const int mode = mode0 + 2 * (int)blockIdx.z;  // one launch does the D_p updates (z = 0) and the new couplings (z = 1)
  __shared__ double Ms[BCR_MAXB * 33];
  __shared__ double Ns[BCR_MAXB * 33];
  const BcrJob jb = jobs[blockIdx.x];
  const int tiles = B / 32, ti = blockIdx.y / tiles, tj = blockIdx.y - ti * tiles;
  if (mode != 2 && tj > ti) return;
  if (mode != 0 && jb.q < 0) return;
  if (mode == 2 && jb.knew < 0) return;
  const size_t BB = (size_t)B * B;
  const double* U1 = U + (size_t)2 * jb.u * BB;
  const double* U2 = U1 + BB;
  const double* M = mode == 0 ? U1 : U2;
  const double* N = mode == 1 ? U2 : U1;
  const int tid = threadIdx.x;
  for (int r0 = tid >> 5; r0 < B; r0 += 32) {  // 4 rows of both slabs per thread per round (eight loads in flight)
    double vm[4], vn[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + 8 * u;
      vm[u] = r < B ? M[(size_t)r * B + 32 * ti + (tid & 31)] : 0.0;
      vn[u] = r < B ? N[(size_t)r * B + 32 * tj + (tid & 31)] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = r0 + 8 * u;
      if (r < B) {
        Ms[r * 33 + (tid & 31)] = vm[u];
        Ns[r * 33 + (tid & 31)] = vn[u];
      }
    }
  }
  __syncthreads();
  const int i0 = (tid >> 4) * 2, j0 = (tid & 15) * 2;  // a 2 x 2 patch of the tile per thread
  double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
  for (int k = 0; k < B; k++) {
    const double m0 = Ms[k * 33 + i0], m1 = Ms[k * 33 + i0 + 1], n0 = Ns[k * 33 + j0], n1 = Ns[k * 33 + j0 + 1];
    a00 += m0 * n0;
    a01 += m0 * n1;
    a10 += m1 * n0;
    a11 += m1 * n1;
  }
  double* Cb = mode == 0 ? D + (size_t)jb.p * BB : (mode == 1 ? D + (size_t)jb.q * BB : Knext + (size_t)jb.knew * BB);
  const int gi = 32 * ti + i0, gj = 32 * tj + j0;
  if (mode == 2) {
    Cb[(size_t)gi * B + gj] = -a00;
    Cb[(size_t)gi * B + gj + 1] = -a01;
    Cb[(size_t)(gi + 1) * B + gj] = -a10;
    Cb[(size_t)(gi + 1) * B + gj + 1] = -a11;
  } else {
    Cb[(size_t)gi * B + gj] -= a00;
    Cb[(size_t)gi * B + gj + 1] -= a01;   // (entries above the diagonal of a diagonal tile are never read)
    Cb[(size_t)(gi + 1) * B + gj] -= a10;
    Cb[(size_t)(gi + 1) * B + gj + 1] -= a11;
  }
}

// All products of one level in ONE launch, target by target (round 4; the two launches of bcr_gemm_kernel above -- the
// D_p updates and new couplings, then the D_q updates, because D_p of one job is D_q of the job above it -- kept for A/B
// builds under BCR_GEMM_TARGETS 0).  A target is a remaining diagonal block with its one or two updates
// D_x -= U1(b)^T U1(b) + U2(a)^T U2(a) (lower 32 x 32 tiles only) or a new coupling K(q, p) = -U2^T U1 (all tiles); the sum
// over the two contributions runs in a fixed order.  16 x 16 tiles of the target on v_mfma_f64_16x16x4_f64, operands
// straight from the row-major U blocks (a k-row of 16 consecutive doubles per 16 lanes), no operand staging in LDS: the 32 x 32-per-workgroup form staged two 224 x 32 slabs
// (114 KB: one workgroup per compute unit) and spent five LDS reads per four multiply-adds of a 2 x 2 register patch.
struct BcrTarget {
  int c_is_k, c_idx;  // destination: Knext[c_idx] (assigned) or D[c_idx] (updated, lower tiles)
  int m1, n1;         // first product  M^T N: U block slots (2 u + 0 / 1)
  int m2, n2;         // second product (-1: none)
  int pad0, pad1;
};

#ifndef BCR_GEMM_TARGETS
#define BCR_GEMM_TARGETS 1
#endif
// workgroup = one 16 x 16 tile, the inner dimension split over its four wavefronts (a wavefront's share of one product
// is B / 16 <= 16 matrix instructions, all of its operand loads -- both products -- issued before the first of them: ONE
// exposed round trip; with the whole inner dimension per wavefront the loads of a batch took longer than its eight
// matrix instructions however far ahead they were issued, and the deepest levels -- one to three jobs -- ran as long as
// level 0); the four partial tiles are summed in wavefront order through LDS.
__global__ __launch_bounds__(256) void bcr_gemm_targets_kernel(const BcrTarget* __restrict__ targets, int B,
                                                               const double* __restrict__ U, double* __restrict__ D,
                                                               double* __restrict__ Knext, const int* __restrict__ ok) {
  if (!*ok) return;
  __shared__ double red[4][256];
  const BcrTarget tg = targets[blockIdx.x];
  const int tiles = B / 16, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ti = blockIdx.y / tiles, tj = blockIdx.y - ti * tiles;
  if (!tg.c_is_k && tj > ti) return;
  const size_t BB = (size_t)B * B;
  const int kq = lane >> 4, l16 = lane & 15;
  const int i0 = 16 * ti, j0 = 16 * tj;
  const int ks = B / 16;  // k-steps (of 4) per wavefront: 14 at B = 224, 16 at B = 256
  constexpr int KS_MAX = BCR_MAXB / 16;
  const size_t krow = (size_t)(wave * (B / 4) + kq) * B;
  double av[2][KS_MAX], bv[2][KS_MAX];
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int ms = pass ? tg.m2 : tg.m1, ns = pass ? tg.n2 : tg.n1;
    const double* Mp = U + (size_t)(ms < 0 ? 0 : ms) * BB + krow + i0 + l16;
    const double* Np = U + (size_t)(ns < 0 ? 0 : ns) * BB + krow + j0 + l16;
#pragma unroll
    for (int u = 0; u < KS_MAX; u++) {
      const bool in = ms >= 0 && u < ks;
      av[pass][u] = in ? Mp[(size_t)(4 * u) * B] : 0.0;
      bv[pass][u] = in ? Np[(size_t)(4 * u) * B] : 0.0;
    }
  }
  v4d_t acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    if (pass && tg.m2 < 0) break;
#pragma unroll
    for (int u = 0; u < KS_MAX; u += 2) {
      if (u < ks) {  // (ks is even)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pass][u], bv[pass][u], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[pass][u + 1], bv[pass][u + 1], acc2, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int qq = 0; qq < 4; qq++) red[wave][64 * qq + lane] = acc[qq] + acc2[qq];
  __syncthreads();
  {
    const int t = threadIdx.x, qq = t >> 6, l = t & 63;
    const double v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    double* C = tg.c_is_k ? Knext + (size_t)tg.c_idx * BB : D + (size_t)tg.c_idx * BB;
    double* c = C + (size_t)(i0 + (l >> 4) + 4 * qq) * B + j0 + (l & 15);
    if (tg.c_is_k)
      *c = -v;
    else
      *c -= v;  // (entries above the diagonal of a diagonal tile are never read)
  }
}

// t = z_e - U1 x_p - U2 x_q of the eliminated blocks of one level, rows spread over BCR_GV_CHUNKS workgroups per block
// (round 4: inside bcr_back_kernel this product -- 2 B^2 doubles read by ONE workgroup per block -- was ~15 us of its
// ~34 us in front of the seven dependent panel steps); t is left in the block's solution slot bb[e].
#define BCR_GV_CHUNKS 8
__global__ __launch_bounds__(256) void bcr_back_gemv_kernel(const BcrJob* __restrict__ jobs, int B, const double* __restrict__ U,
                                                            double* __restrict__ bb, const double* __restrict__ yy,
                                                            const int* __restrict__ ok) {
  if (!*ok) return;
  const BcrJob jb = jobs[blockIdx.x];
  const size_t BB = (size_t)B * B;
  const double* U1 = U + (size_t)2 * jb.u * BB;
  const double* U2 = U1 + BB;
  const double* xp = bb + (size_t)jb.p * B;
  const double* xq = jb.q >= 0 ? bb + (size_t)jb.q * B : nullptr;
  const double* z = yy + (size_t)jb.e * B;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (B + BCR_GV_CHUNKS - 1) / BCR_GV_CHUNKS, r_lo = per * blockIdx.y, r_hi = min(B, r_lo + per);
  double xpv[4], xqv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = lane + 64 * j;
    xpv[j] = c < B ? xp[c] : 0.0;
    xqv[j] = (xq && c < B) ? xq[c] : 0.0;
  }
  // one wavefront per row, lanes along the (contiguous) columns, fixed-order lane sum; four rows at a time so that their
  // loads are in flight together (B <= 256: four 64-column chunks per row)
  for (int r0 = r_lo + 4 * wave; r0 < r_hi; r0 += 16) {
    double a1[4][4], a2[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = r0 + u, c = lane + 64 * j;
        const bool in = r < r_hi && c < B;
        a1[u][j] = in ? U1[(size_t)r * B + c] : 0.0;
        a2[u][j] = (in && xq) ? U2[(size_t)r * B + c] : 0.0;
      }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double sacc = 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++) sacc += a1[u][j] * xpv[j] + a2[u][j] * xqv[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
      if (lane == 0 && r0 + u < r_hi) bb[(size_t)jb.e * B + r0 + u] = z[r0 + u] - sacc;
    }
  }
}

// solution of an eliminated block: t = z_e - U1 x_p - U2 x_q, then L_e^T x_e = t panel by panel from the bottom:
// x_p = Linv_pp^T (t_p - sum over the rows below of L[r][p]^T x_r) -- the 32 x 32 triangular part through the
// precomputed inverse of the diagonal block (no serial 32-step chain), the rest a column-sliced dot product.
__global__ __launch_bounds__(CBF_THREADS) void bcr_back_kernel(const BcrJob* __restrict__ jobs, int B, const double* __restrict__ D,
                                                               const double* __restrict__ Linv, double* __restrict__ bb,
                                                               const int* __restrict__ ok) {
  __shared__ double ts[BCR_MAXB];              // t, then x
  __shared__ double part[CBF_THREADS / 32][33];
  __shared__ double Lis[(BCR_MAXB / 32) * 1024];  // the inverses of the factor's diagonal blocks
  if (!*ok) return;
  const BcrJob jb = jobs[blockIdx.x];
  const size_t BB = (size_t)B * B;
  const double* L = D + (size_t)jb.e * BB;
  const double* Li = Linv + (size_t)jb.u * (B / 32) * 1024;
  const int tid = threadIdx.x;
  for (int r = tid; r < B; r += CBF_THREADS) ts[r] = bb[(size_t)jb.e * B + r];  // t (bcr_back_gemv_kernel)
  const int col = tid & 31, sl = tid >> 5;  // CBF_THREADS / 32 row slices x 32 columns
  constexpr int NS = CBF_THREADS / 32;
  // (round 4) nothing of the chain below waits for global memory: the diagonal-block inverses are staged once, the rows
  // of L below a panel are fetched one panel ahead, and the panel's solution comes from one wavefront without a barrier
  // between the partial sums and the product with the inverse (it was 7 steps x (2 exposed round trips + 3 barriers))
  auto fetch = [&](int p0, double* lv) {
#pragma unroll
    for (int u = 0; u < 11; u++) {  // the rows below the panel, NS apart per slice: at most ceil(224 / 24) = 10 per thread
      const int r = p0 + 32 + sl + NS * u;
      lv[u] = (p0 >= 0 && r < B) ? L[(size_t)r * B + p0 + col] : 0.0;
    }
  };
  double lv[11];
  fetch(B - 32, lv);
  for (int idx = tid; idx < (B / 32) * 1024; idx += CBF_THREADS) Lis[idx] = Li[idx];
  __syncthreads();
  for (int p0 = B - 32; p0 >= 0; p0 -= 32) {
    double sacc = 0.0;
#pragma unroll
    for (int u = 0; u < 11; u++) {
      const int r = p0 + 32 + sl + NS * u;
      if (r < B) sacc += lv[u] * ts[r];
    }
    part[sl][col] = sacc;
    fetch(p0 - 32, lv);
    __syncthreads();
    if (tid < 64) {  // lanes 32..63 mirror 0..31 (the broadcasts read lanes 0..31 only)
      double t = 0.0;
#pragma unroll
      for (int g = 0; g < NS; g++) t += part[g][tid & 31];
      const double w = ts[p0 + (tid & 31)] - t;
      double x = 0.0;  // x_c = sum_k Linv[k][c] w_k  (k >= c: the inverse is lower triangular)
      const double* Lb = Lis + (size_t)(p0 >> 5) * 1024;
#pragma unroll
      for (int k = 0; k < 32; k++) x += Lb[k * 32 + (tid & 31)] * lane_bcast(w, k);
      if (tid < 32) ts[p0 + tid] = x;
    }
    __syncthreads();
  }
  for (int r = tid; r < B; r += CBF_THREADS) bb[(size_t)jb.e * B + r] = ts[r];
}

__global__ void bcr_gather_kernel(const double* __restrict__ bb, int B, const int* __restrict__ off, double* __restrict__ b,
                                  double* __restrict__ neg_out, const int* __restrict__ ok) {
  if (!*ok) return;
  const int blk = blockIdx.y, o0 = off[blk], sz = off[blk + 1] - o0;
  for (int r = threadIdx.x; r < sz; r += blockDim.x) {
    const double v = bb[(size_t)blk * B + r];
    b[o0 + r] = v;
    if (neg_out) neg_out[o0 + r] = -v;
  }
}


// Block layout of the cyclic form: nblk blocks of floor / ceil (n / nblk) unknowns, every one >= bw + 1 (a block couples
// with its two ring neighbours only) and <= B (the kernels' block size, a multiple of 32 <= BCR_MAXB).  false: no such layout.
bool vsl_chol_bcr_cyclic_layout(int n, int bw, int* B_out, int* nblk_out) {
  const int most = n / (bw + 1);  // blocks of >= bw + 1 unknowns each
  for (int B = (bw + 1 + 31) / 32 * 32; B <= BCR_MAXB; B += 32) {
    const int nblk = std::max(8, (n + B - 1) / B);  // the fewest blocks of <= B unknowns (fewer blocks: fewer levels)
    if (nblk <= most) {
      *B_out = B;
      *nblk_out = nblk;
      return true;
    }
  }
  return false;
}

// S (band storage, n unknowns, half bandwidth bw) x = b by block cyclic reduction; b <- x.  cyclic: the band closes on
// itself (entries (i, j) with j - i >= n - bw live in the leading slots of row i, see bcr_extract_kernel): the blocks form
// a RING -- the neighbour below the last block is the first one -- and every level halves the ring: an even ring's last
// elimination creates the next ring's wrap coupling, an odd ring's wrap coupling moves on unchanged, a ring of two blocks
// is one coupling (the sum of both) and ends as a chain.
int vsl_chol_solve_bcr_dev(vsl_ctx* ctx, double* S, double* b, int n, int ld, int bw, int* ok_dev, int cyclic, double* neg_out) {
  int B = (bw + 1 + 31) / 32 * 32, nblk = (n + B - 1) / B;
  if (cyclic && !vsl_chol_bcr_cyclic_layout(n, bw, &B, &nblk))
    return vsl_fail(ctx, VSL_ERR_INVALID, "cyclic band of %d unknowns, half bandwidth %d: no block layout (the caller checks vsl_chol_bcr_cyclic_layout)", n, bw);
  const size_t BB = (size_t)B * B;
  std::vector<int> off(nblk + 1);
  for (int i = 0; i <= nblk; i++) off[i] = cyclic ? (int)((long long)i * n / nblk) : std::min(n, i * B);
  // levels on the host: active block lists, jobs
  struct LevelOp { int combine = 0, carry = 0; };  // combine: c[0] += c[1]^T before the solves; carry: c[m - 1] -> next c[m' - 1]
  std::vector<std::vector<BcrJob>> levels;
  std::vector<LevelOp> ops;
  std::vector<int> active(nblk), ncoup;
  for (int i = 0; i < nblk; i++) active[i] = i;
  int n_u = 0;
  while (active.size() > 1) {
    const int m = (int)active.size();
    std::vector<BcrJob> jobs;
    std::vector<int> next;
    LevelOp op;
    const bool ring = cyclic && m >= 3;  // a ring of two is handled as a chain with one (combined) coupling
    if (cyclic && m == 2) op.combine = 1;
    if (ring && (m & 1)) op.carry = 1;
    for (int j = 0; j < m; j++) {
      if (j & 1) {
        BcrJob jb;
        jb.e = active[j];
        jb.p = active[j - 1];
        jb.q = j + 1 < m ? active[j + 1] : (ring ? active[0] : -1);
        jb.kp = j - 1;
        jb.kq = j;
        jb.knew = jb.q >= 0 ? (j - 1) / 2 : -1;
        jb.u = n_u++;
        jobs.push_back(jb);
      } else {
        next.push_back(active[j]);
      }
    }
    ncoup.push_back(cyclic ? m : m - 1);
    levels.push_back(jobs);
    ops.push_back(op);
    active = next;
  }
  const int last = active[0];
  // product targets per level (bcr_gemm_targets_kernel): the remaining blocks with their one or two updates, the new couplings
  std::vector<std::vector<BcrTarget>> targets(levels.size());
  for (size_t l = 0; l < levels.size(); l++) {
    std::vector<BcrTarget>& tl = targets[l];
    std::vector<int> slot_of(nblk, -1);
    for (const BcrJob& jb : levels[l]) {   // jobs in block order: the update from the block above (as its q) comes first
      for (int side = 0; side < 2; side++) {
        const int x = side ? jb.q : jb.p, us = 2 * jb.u + side;
        if (x < 0) continue;
        if (slot_of[x] < 0) {
          slot_of[x] = (int)tl.size();
          tl.push_back(BcrTarget{0, x, us, us, -1, -1, 0, 0});
        } else {
          tl[slot_of[x]].m2 = tl[slot_of[x]].n2 = us;
        }
      }
      if (jb.knew >= 0 && l + 1 < levels.size()) tl.push_back(BcrTarget{1, jb.knew, 2 * jb.u + 1, 2 * jb.u, -1, -1, 0, 0});
    }
  }
  size_t n_k = 0, n_jobs = 0;
  for (size_t l = 0; l < levels.size(); l++) {
    n_k += (size_t)ncoup[l];
    n_jobs += levels[l].size();
  }
  // scratch: D | K (all levels) | U | bb | yy | dinv | jobs
  size_t max_nj = 1;
  for (auto& lv : levels) max_nj = std::max(max_nj, lv.size());
  const size_t linv_doubles = ((size_t)n_u + 1) * (size_t)(B / 32) * 1024;  // per eliminated block (+ the last one), kept for the back-substitution
  const size_t doubles = (size_t)nblk * BB + n_k * BB + (size_t)2 * n_u * BB + 5 * (size_t)nblk * B + linv_doubles + 16;
  size_t n_targets = 0;
  for (auto& tl : targets) n_targets += tl.size();
  const size_t job_only_bytes = ((n_jobs + 1) * sizeof(BcrJob) + 63) / 64 * 64;
  const size_t tgt_bytes = ((n_targets + 1) * sizeof(BcrTarget) + 63) / 64 * 64;
  const size_t job_bytes = job_only_bytes + tgt_bytes + sizeof(int) * ((size_t)nblk + 1);
  void* ws = nullptr;
  int rc = vsl_ctx_dscratch(ctx, sizeof(double) * doubles + 256, &ws);
  if (rc) return rc;
  double* D = (double*)ws;
  double* K0 = D + (size_t)nblk * BB;
  double* U = K0 + n_k * BB;
  double* bb = U + (size_t)2 * n_u * BB;
  double* yy = bb + (size_t)nblk * B;
  double* dinv = yy + (size_t)nblk * B;
  double* pend = dinv + (size_t)nblk * B;  // [nblk][2][B]
  double* Linv = pend + (size_t)2 * nblk * B;
  // the job lists depend on (n, bw) only: uploaded once and kept (an upload per solve needs a host synchronisation in
  // the middle of an otherwise asynchronous LM iteration)
  if (ctx->bcr_jobs_cap < job_bytes) {
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->bcr_jobs) (void)hipFree(ctx->bcr_jobs);
    ctx->bcr_jobs = nullptr;
    ctx->bcr_jobs_cap = 0;
    ctx->bcr_key_n = ctx->bcr_key_bw = -1;
    VSL_HIP(ctx, hipMalloc(&ctx->bcr_jobs, 2 * job_bytes));
    ctx->bcr_jobs_cap = 2 * job_bytes;
  }
  BcrJob* jobs_dev = (BcrJob*)ctx->bcr_jobs;
  std::vector<BcrJob> flat;
  std::vector<size_t> job_off, k_off;
  {
    size_t ko = 0;
    for (size_t l = 0; l < levels.size(); l++) {
      job_off.push_back(flat.size());
      k_off.push_back(ko);
      flat.insert(flat.end(), levels[l].begin(), levels[l].end());
      ko += (size_t)ncoup[l];
    }
  }
  BcrTarget* targets_dev = (BcrTarget*)((char*)ctx->bcr_jobs + job_only_bytes);
  int* off_dev = (int*)((char*)ctx->bcr_jobs + job_only_bytes + tgt_bytes);
  std::vector<BcrTarget> tflat;
  std::vector<size_t> tgt_off;
  for (auto& tl : targets) {
    tgt_off.push_back(tflat.size());
    tflat.insert(tflat.end(), tl.begin(), tl.end());
  }
  const int key_bw = cyclic ? -2 - bw : bw;  // (one cache slot: the cyclic form of the same (n, bw) is another plan)
  if (ctx->bcr_key_n != n || ctx->bcr_key_bw != key_bw) {
    VSL_HIP(ctx, hipMemcpyAsync(jobs_dev, flat.data(), flat.size() * sizeof(BcrJob), hipMemcpyHostToDevice, ctx->stream));
    if (!tflat.empty())
      VSL_HIP(ctx, hipMemcpyAsync(targets_dev, tflat.data(), tflat.size() * sizeof(BcrTarget), hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipMemcpyAsync(off_dev, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    VSL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `flat` / `tflat` / `off` are on this stack frame
    ctx->bcr_key_n = n;
    ctx->bcr_key_bw = key_bw;
  }
  hipStream_t q = ctx->stream;
  hipLaunchKernelGGL(bcr_extract_kernel, dim3(64, nblk), dim3(256), 0, q, S, ld, n, ld, b, B, nblk, off_dev, cyclic, D, K0, bb, pend, ok_dev);
  const int tiles = B / 32;
  for (size_t l = 0; l < levels.size(); l++) {
    const int nj = (int)levels[l].size();
    const BcrJob* jl = jobs_dev + job_off[l];
    double* Kl = K0 + k_off[l] * BB;
    double* Kn = l + 1 < levels.size() ? K0 + k_off[l + 1] * BB : nullptr;
    if (ops[l].combine)
      hipLaunchKernelGGL(bcr_add_transposed_kernel, dim3((B * B + 255) / 256), dim3(256), 0, q, B, Kl, Kl + BB, ok_dev);
    if (ops[l].carry && Kn)
      hipLaunchKernelGGL(bcr_copy_block_kernel, dim3((B * B + 255) / 256), dim3(256), 0, q, B, Kl + (size_t)(ncoup[l] - 1) * BB,
                         Kn + (size_t)(ncoup[l + 1] - 1) * BB, ok_dev);
    if (B <= 224)
      hipLaunchKernelGGL(bcr_chol_kernel<14>, dim3(nj), dim3(BCR_REG_THREADS), 0, q, jl, B, D, bb, yy, dinv, pend, ok_dev);
    else
      hipLaunchKernelGGL(bcr_chol_kernel<17>, dim3(nj), dim3(BCR_REG_THREADS), 0, q, jl, B, D, bb, yy, dinv, pend, ok_dev);
    hipLaunchKernelGGL(bcr_dinv_kernel, dim3(nj, tiles), dim3(64), 0, q, jl, B, D, dinv, Linv, ok_dev);
    hipLaunchKernelGGL(bcr_trsm_kernel, dim3(nj, 2 * tiles), dim3(256), 0, q, jl, B, D, Linv, Kl, U, yy, pend, ok_dev);
#if BCR_GEMM_TARGETS
    {
      const int t16 = B / 16;
      hipLaunchKernelGGL(bcr_gemm_targets_kernel, dim3((unsigned)targets[l].size(), t16 * t16), dim3(256), 0, q,
                         targets_dev + tgt_off[l], B, U, D, Kn, ok_dev);
    }
#else
    hipLaunchKernelGGL(bcr_gemm_kernel, dim3(nj, tiles * tiles, Kn ? 2 : 1), dim3(256), 0, q, jl, B, 0, U, D, Kn, ok_dev);
    hipLaunchKernelGGL(bcr_gemm_kernel, dim3(nj, tiles * tiles, 1), dim3(256), 0, q, jl, B, 1, U, D, Kn, ok_dev);
#endif
  }
  if (B <= 224)
    hipLaunchKernelGGL(bcr_last_kernel<14>, dim3(1), dim3(BCR_REG_THREADS), 0, q, last, B, D, bb, yy, dinv, pend, ok_dev);
  else
    hipLaunchKernelGGL(bcr_last_kernel<17>, dim3(1), dim3(BCR_REG_THREADS), 0, q, last, B, D, bb, yy, dinv, pend, ok_dev);
  hipLaunchKernelGGL(bcr_last_back_kernel, dim3(1), dim3(CBF_THREADS), 0, q, last, B, D, bb, yy, dinv, ok_dev);
  for (size_t l = levels.size(); l-- > 0;) {
    const int nj = (int)levels[l].size();
    hipLaunchKernelGGL(bcr_back_gemv_kernel, dim3(nj, BCR_GV_CHUNKS), dim3(256), 0, q, jobs_dev + job_off[l], B, U, bb, yy, ok_dev);
    hipLaunchKernelGGL(bcr_back_kernel, dim3(nj), dim3(CBF_THREADS), 0, q, jobs_dev + job_off[l], B, D, Linv, bb, ok_dev);
  }
  hipLaunchKernelGGL(bcr_gather_kernel, dim3(1, nblk), dim3(256), 0, q, bb, B, off_dev, b, neg_out, ok_dev);
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}
