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
// per panel and direction, no serial triangular solve.  Everything is deterministic
// (no atomics).  Written here rather than calling rocSOLVER: librocsolver.so is a 0.9 GB load that
// takes minutes to page in on a fresh machine.
#include "vsl_common.h"

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

__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ A, int n, int k, int nb, int* __restrict__ ok,
                                                       double* __restrict__ Linv) {
  if (!*ok) return;
  const int lane = threadIdx.x & 31;  // lanes 32..63 mirror 0..31 (their results are discarded)
  double r[CH_NB];
#pragma unroll
  for (int c = 0; c < CH_NB; c++)
    r[c] = (lane < nb && c < nb) ? A[(size_t)(k + lane) * n + k + c] : (lane == c ? 1.0 : 0.0);
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
      if (c <= lane && c < nb) A[(size_t)(k + lane) * n + k + c] = r[c];
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
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int n, int k, int nb, const int* __restrict__ ok,
                                                         const double* __restrict__ Linv) {
  __shared__ double Li[CH_NB][CH_NB + 1];
  __shared__ double R[64][CH_NB + 1];
  if (!*ok) return;
  const int i0 = k + nb + blockIdx.x * 64;
  for (int t = threadIdx.x; t < CH_NB * CH_NB; t += 256) Li[t / CH_NB][t % CH_NB] = Linv[t];
  for (int t = threadIdx.x; t < 64 * CH_NB; t += 256) {
    const int r = t / CH_NB, c = t % CH_NB;
    R[r][c] = (i0 + r < n && c < nb) ? A[(size_t)(i0 + r) * n + k + c] : 0.0;
  }
  __syncthreads();
  const int c = threadIdx.x % CH_NB;
  for (int r = threadIdx.x / CH_NB; r < 64; r += 8) {
    double s = 0;
#pragma unroll
    for (int p = 0; p < CH_NB; p++) s += R[r][p] * Li[c][p];  // Linv is zero above its diagonal
    if (i0 + r < n && c < nb) A[(size_t)(i0 + r) * n + k + c] = s;
  }
}

// (3) trailing update, lower triangle: C[i][j] -= sum_p P[i][p] P[j][p], i, j >= k+nb.
// grid.x enumerates tile pairs (ti >= tj) of 64x64 tiles; 256 threads, 4x4 outputs each.
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, int n, int k, int nb, int T,
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
    Pi[r][c] = (i0 + r < n && c < nb) ? A[(size_t)(i0 + r) * n + k + c] : 0.0;
    Pj[r][c] = (j0 + r < n && c < nb) ? A[(size_t)(j0 + r) * n + k + c] : 0.0;
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
      if (i < n && j < n && j <= i) A[(size_t)i * n + j] -= acc[a][b];
    }
}

// forward, panel k:  y_k = Linv_kk b_k  (every workgroup computes it, workgroup 0 stores it in y), then
// b[i] -= L[i, k:k+nb] . y_k for the rows i >= k+nb of this workgroup.  b_k itself is only read.
__global__ __launch_bounds__(256) void chol_fwd_kernel(const double* __restrict__ A, double* __restrict__ b, double* __restrict__ y,
                                                       const double* __restrict__ Linv, int n, int k, int nb) {
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
  const double* row = A + (size_t)i * n + k;
  double s = 0;
  for (int p = 0; p < nb; p++) s += row[p] * yk[p];
  b[i] -= s;
}

// backward, panel k:  x_k = Linv_kk^T y_k  (stored into b by workgroup 0), then y[c] -= sum_r L[k+r][c] x_k[r]
// for the columns c < k of this workgroup.
__global__ __launch_bounds__(256) void chol_bwd_kernel(const double* __restrict__ A, double* __restrict__ b, double* __restrict__ y,
                                                       const double* __restrict__ Linv, int n, int k, int nb) {
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
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= k) return;
  double s = 0;
  for (int r = 0; r < nb; r++) s += A[(size_t)(k + r) * n + c] * xk[r];
  y[c] -= s;
}

// Solves S x = b in place (S destroyed, b <- x).  *ok_dev = 1 on success, 0 if S is not SPD.
int vsl_chol_solve_dev(vsl_ctx* ctx, double* S, double* b, int n, int* ok_dev) {
  const int one = 1;
  const int n_panels = (n + CH_NB - 1) / CH_NB;
  void* ws = nullptr;  // inverse diagonal blocks + the intermediate vector y
  int rc = vsl_ctx_dscratch(ctx, sizeof(double) * ((size_t)n_panels * CH_NB * CH_NB + (size_t)n + 8), &ws);
  if (rc) return rc;
  double* Linv = (double*)ws;
  double* y = Linv + (size_t)n_panels * CH_NB * CH_NB;
  VSL_HIP(ctx, hipMemcpyAsync(ok_dev, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0, pi = 0; k < n; k += CH_NB, pi++) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    double* Li = Linv + (size_t)pi * CH_NB * CH_NB;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, ctx->stream, S, n, k, nb, ok_dev, Li);
    const int m = n - k - nb;
    if (m > 0) {
      hipLaunchKernelGGL(chol_panel_kernel, dim3((m + 63) / 64), dim3(256), 0, ctx->stream, S, n, k, nb, ok_dev, Li);
      const int T = (m + 63) / 64;
      hipLaunchKernelGGL(chol_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, ctx->stream, S, n, k, nb, T, ok_dev);
    }
  }
  VSL_CHECK_LAUNCH(ctx);
  // (a failed factorisation leaves Linv / y undefined; the caller looks at *ok_dev before using b)
  for (int k = 0, pi = 0; k < n; k += CH_NB, pi++) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    const int m = n - k - nb;
    hipLaunchKernelGGL(chol_fwd_kernel, dim3(m > 0 ? (m + 255) / 256 : 1), dim3(256), 0, ctx->stream, S, b, y,
                       Linv + (size_t)pi * CH_NB * CH_NB, n, k, nb);
  }
  for (int pi = n_panels - 1; pi >= 0; pi--) {
    const int k = pi * CH_NB;
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    hipLaunchKernelGGL(chol_bwd_kernel, dim3(k > 0 ? (k + 255) / 256 : 1), dim3(256), 0, ctx->stream, S, b, y,
                       Linv + (size_t)pi * CH_NB * CH_NB, n, k, nb);
  }
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}
