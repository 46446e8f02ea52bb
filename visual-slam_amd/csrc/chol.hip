// chol.hip -- dense fp64 Cholesky solve for the reduced camera system of large (global) bundle
// adjustment: S x = b with S = (6C) x (6C), C up to ~1000 cameras.
//
// Ceres hands this system to a sparse Cholesky (SPARSE_SCHUR, include/visnav/map_utils.h:408); on an
// MI355X the dense factorisation of a 6000^2 matrix is ~72 GFLOP of fp64 and the matrix (288 MB)
// stays resident in HBM, so a plain blocked right-looking factorisation is used:
//   for each 32-column panel k:  (1) factor the 32x32 diagonal block in LDS,
//                                (2) triangular-solve the panel below it (one row per thread),
//                                (3) rank-32 update of the trailing lower triangle, 64x64 tiles,
//                                    4x4 register blocking, operands staged through LDS.
// Forward / backward substitution reuse the same panel structure.  Everything is deterministic
// (no atomics).  Written here rather than calling rocSOLVER: librocsolver.so is a 0.9 GB load that
// takes minutes to page in on a fresh machine.
#include "vsl_common.h"

#define CH_NB 32

// (1) factor A[k:k+nb, k:k+nb] in place (lower); ok=0 if not positive definite
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ A, int n, int k, int nb, int* __restrict__ ok) {
  __shared__ double T[CH_NB][CH_NB + 1];
  __shared__ int good;
  const int tid = threadIdx.x;
  for (int t = tid; t < nb * nb; t += 256) T[t / nb][t % nb] = A[(size_t)(k + t / nb) * n + k + t % nb];
  if (tid == 0) good = *ok;
  __syncthreads();
  if (!good) return;
  for (int j = 0; j < nb; j++) {
    if (tid == 0) {
      const double d = T[j][j];
      if (!(d > 0.0) || !isfinite(d)) good = 0;
      T[j][j] = sqrt(d);
    }
    __syncthreads();
    if (!good) break;
    const double djj = T[j][j];
    if (tid > j && tid < nb) T[tid][j] /= djj;
    __syncthreads();
    const int m = nb - j - 1;
    for (int t = tid; t < m * m; t += 256) {
      const int i = j + 1 + t / m, c = j + 1 + t % m;
      if (c <= i) T[i][c] -= T[i][j] * T[c][j];
    }
    __syncthreads();
  }
  if (!good) {
    if (tid == 0) *ok = 0;
    return;
  }
  for (int t = tid; t < nb * nb; t += 256)
    if (t % nb <= t / nb) A[(size_t)(k + t / nb) * n + k + t % nb] = T[t / nb][t % nb];
}

// (2) rows i >= k+nb:  A[i, k:k+nb] <- A[i, k:k+nb] * L_kk^-T   (one row per thread)
__global__ __launch_bounds__(64) void chol_panel_kernel(double* __restrict__ A, int n, int k, int nb, const int* __restrict__ ok) {
  __shared__ double Lk[CH_NB][CH_NB + 1];
  if (!*ok) return;
  for (int t = threadIdx.x; t < nb * nb; t += 64) Lk[t / nb][t % nb] = A[(size_t)(k + t / nb) * n + k + t % nb];
  __syncthreads();
  const int i = k + nb + blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double a[CH_NB];
  double* row = A + (size_t)i * n + k;
#pragma unroll
  for (int j = 0; j < CH_NB; j++) a[j] = j < nb ? row[j] : 0.0;
#pragma unroll
  for (int j = 0; j < CH_NB; j++) {
    if (j < nb) {
      double s = a[j];
#pragma unroll
      for (int p = 0; p < CH_NB; p++)
        if (p < j) s -= a[p] * Lk[j][p];
      a[j] = s / Lk[j][j];
    }
  }
#pragma unroll
  for (int j = 0; j < CH_NB; j++)
    if (j < nb) row[j] = a[j];
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

// forward: y_k = L_kk^-1 b_k (one wave), then b[i] -= L[i, k:k+nb] . y_k for i >= k+nb
__global__ __launch_bounds__(64) void chol_fwd_diag_kernel(const double* __restrict__ A, double* __restrict__ b, int n, int k, int nb) {
  __shared__ double y[CH_NB];
  if (threadIdx.x == 0) {
    for (int j = 0; j < nb; j++) {
      double s = b[k + j];
      for (int p = 0; p < j; p++) s -= A[(size_t)(k + j) * n + k + p] * y[p];
      y[j] = s / A[(size_t)(k + j) * n + k + j];
    }
    for (int j = 0; j < nb; j++) b[k + j] = y[j];
  }
}

__global__ __launch_bounds__(256) void chol_fwd_update_kernel(const double* __restrict__ A, double* __restrict__ b, int n, int k, int nb) {
  const int i = k + nb + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int p = 0; p < nb; p++) s += A[(size_t)i * n + k + p] * b[k + p];
  b[i] -= s;
}

// backward: x_k = L_kk^-T y_k, then y[c] -= sum_r L[k+r][c] x[k+r] for every column c < k
__global__ __launch_bounds__(64) void chol_bwd_diag_kernel(const double* __restrict__ A, double* __restrict__ b, int n, int k, int nb) {
  __shared__ double x[CH_NB];
  if (threadIdx.x == 0) {
    for (int j = nb - 1; j >= 0; j--) {
      double s = b[k + j];
      for (int p = j + 1; p < nb; p++) s -= A[(size_t)(k + p) * n + k + j] * x[p];
      x[j] = s / A[(size_t)(k + j) * n + k + j];
    }
    for (int j = 0; j < nb; j++) b[k + j] = x[j];
  }
}

__global__ __launch_bounds__(256) void chol_bwd_update_kernel(const double* __restrict__ A, double* __restrict__ b, int n, int k, int nb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= k) return;
  double s = 0;
  for (int r = 0; r < nb; r++) s += A[(size_t)(k + r) * n + c] * b[k + r];
  b[c] -= s;
}

// Solves S x = b in place (S destroyed, b <- x).  *ok_dev = 1 on success, 0 if S is not SPD.
int vsl_chol_solve_dev(vsl_ctx* ctx, double* S, double* b, int n, int* ok_dev) {
  const int one = 1;
  VSL_HIP(ctx, hipMemcpyAsync(ok_dev, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < n; k += CH_NB) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, ctx->stream, S, n, k, nb, ok_dev);
    const int m = n - k - nb;
    if (m > 0) {
      hipLaunchKernelGGL(chol_panel_kernel, dim3((m + 63) / 64), dim3(64), 0, ctx->stream, S, n, k, nb, ok_dev);
      const int T = (m + 63) / 64;
      hipLaunchKernelGGL(chol_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, ctx->stream, S, n, k, nb, T, ok_dev);
    }
  }
  VSL_CHECK_LAUNCH(ctx);
  for (int k = 0; k < n; k += CH_NB) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    hipLaunchKernelGGL(chol_fwd_diag_kernel, dim3(1), dim3(64), 0, ctx->stream, S, b, n, k, nb);
    const int m = n - k - nb;
    if (m > 0) hipLaunchKernelGGL(chol_fwd_update_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, S, b, n, k, nb);
  }
  for (int k = ((n - 1) / CH_NB) * CH_NB; k >= 0; k -= CH_NB) {
    const int nb = n - k < CH_NB ? n - k : CH_NB;
    hipLaunchKernelGGL(chol_bwd_diag_kernel, dim3(1), dim3(64), 0, ctx->stream, S, b, n, k, nb);
    if (k > 0) hipLaunchKernelGGL(chol_bwd_update_kernel, dim3((k + 255) / 256), dim3(256), 0, ctx->stream, S, b, n, k, nb);
  }
  VSL_CHECK_LAUNCH(ctx);
  return VSL_OK;
}
