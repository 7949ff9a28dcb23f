// dev_linalg.h -- small dense SPD kernels that run inside one workgroup on an LDS-resident
// matrix (n <= ~100): right-looking Cholesky with the reference's failure test, triangular
// solves, and the smallest eigenvalue of a symmetric matrix (Householder tridiagonalisation +
// 64-way Sturm multisection).  These replace Eigen::LLT / SelfAdjointEigenSolver as used in
// Gradient_admm.h:38-53, Optimization3D_multi.h:697-722 and :423-446.  No MFMA: n is tiny and
// the work is latency bound; the matrix never leaves LDS.
#pragma once
#include "dev_common.h"

namespace tj {

// In-place lower Cholesky of the row-major n x n matrix A (only the lower triangle is read).
// Returns false as soon as a pivot is <= 0 (Eigen LLT.h:320-323).  Subtractions happen in
// column order, i.e. the same association as the left-looking scalar loop of the CPU oracle.
// Must be called by all `nth` threads of the block; contains barriers.
constexpr int CHOL_R = 19;  // max rows touched per pivot by the row-per-thread kernel (bw <= 18, dense n <= 20)
__device__ inline bool chol_arrow_lds(double* A, int n, int bw, int tid, int nth);

__device__ inline bool chol_lds(double* A, int n, int tid, int nth) {
  if (n <= CHOL_R + 1 && nth >= CHOL_R) return chol_arrow_lds(A, n, n, tid, nth);  // the 19x19 / 13x13 piece systems
  for (int k = 0; k < n; k++) {
    __syncthreads();
    const double x = A[k * n + k];
    if (x <= 0) return false;  // uniform: every thread reads the same LDS word
    const double sx = sqrt(x);
    __syncthreads();
    if (tid == 0) A[k * n + k] = sx;
    for (int i = k + 1 + tid; i < n; i += nth) A[i * n + k] = A[i * n + k] / sx;
    __syncthreads();
    for (int i = k + 1 + tid; i < n; i += nth) {  // one thread per row: no div/mod, two barriers per pivot
      const double lik = A[i * n + k];
      for (int j = k + 1; j <= i; j++) A[i * n + j] -= lik * A[j * n + k];
    }
  }
  __syncthreads();
  return true;
}

// Same factorisation for the x-update's reduced Hessian, whose sparsity is known: pieces couple
// control points at most 17 coordinates apart (half-bandwidth bw) and only the last row/column
// (piece time) is dense ("arrowhead").  Cholesky creates no fill outside that pattern, so every
// skipped update would subtract an exact 0: results are bit-identical to chol_lds at ~1/6 of the work.
__device__ inline bool chol_arrow_lds(double* A, int n, int bw, int tid, int nth) {
  const int last = n - 1;
  for (int k = 0; k < n; k++) {
    __syncthreads();
    const double x = A[k * n + k];
    if (x <= 0) return false;
    const double sx = sqrt(x);
    __syncthreads();
    const int mb = max(0, min(bw, last - 1 - k));            // band rows below k (arrow row excluded)
    const int nrows = mb + ((k < last) ? 1 : 0);              // + the arrow row; <= CHOL_R
    if (tid == 0) A[k * n + k] = sx;
    // one thread per affected row: no index arithmetic, two barriers per pivot; the row and the
    // scaled pivot column are pulled into registers with back-to-back LDS loads so the update is
    // not a chain of dependent LDS round trips
    if (tid < nrows) { const int i = tid < mb ? k + 1 + tid : last; A[i * n + k] = A[i * n + k] / sx; }
    __syncthreads();
    if (tid < nrows) {
      const int i = tid < mb ? k + 1 + tid : last;
      double lk[CHOL_R], ar[CHOL_R];
#pragma unroll
      for (int c = 0; c < CHOL_R; c++) {
        const int j = c < mb ? k + 1 + c : last;
        lk[c] = A[j * n + k];
        ar[c] = A[i * n + j];
      }
      const double lik = A[i * n + k];
#pragma unroll
      for (int c = 0; c < CHOL_R; c++) {
        const int j = c < mb ? k + 1 + c : last;
        if (c <= tid && c < nrows) A[i * n + j] = ar[c] - lik * lk[c];
      }
    }
  }
  __syncthreads();
  return true;
}
__device__ inline void chol_arrow_solve_lds(const double* L, int n, int bw, const double* b, double* y, int tid, int nth) {
  const int last = n - 1;
  for (int i = tid; i < n; i += nth) y[i] = b[i];
  __syncthreads();
  for (int j = 0; j < n; j++) {
    if (tid == 0) y[j] = y[j] / L[j * n + j];
    __syncthreads();
    const double yj = y[j];
    const int mb = max(0, min(bw, last - 1 - j));
    const int nrows = mb + ((j < last) ? 1 : 0);
    for (int r = tid; r < nrows; r += nth) { const int i = r < mb ? j + 1 + r : last; y[i] -= yj * L[i * n + j]; }
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; j--) {
    if (tid == 0) y[j] = y[j] / L[j * n + j];
    __syncthreads();
    const double yj = y[j];
    const int lo = (j == last) ? 0 : max(0, j - bw);          // row j of L is dense only for the arrow row
    for (int i = lo + tid; i < j; i += nth) y[i] -= yj * L[j * n + i];
    __syncthreads();
  }
}

// x = L^-T L^-1 b, column-oriented substitutions (same order as oracle chol_solve).  y is LDS scratch[n].
__device__ inline void chol_solve_lds(const double* L, int n, const double* b, double* y, int tid, int nth) {
  for (int i = tid; i < n; i += nth) y[i] = b[i];
  __syncthreads();
  for (int j = 0; j < n; j++) {
    if (tid == 0) y[j] = y[j] / L[j * n + j];
    __syncthreads();
    const double yj = y[j];
    for (int i = j + 1 + tid; i < n; i += nth) y[i] -= yj * L[i * n + j];
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; j--) {
    if (tid == 0) y[j] = y[j] / L[j * n + j];
    __syncthreads();
    const double yj = y[j];
    for (int i = tid; i < j; i += nth) y[i] -= yj * L[j * n + i];
    __syncthreads();
  }
}

// Smallest eigenvalue of the symmetric row-major n x n matrix A (lower triangle authoritative;
// A is destroyed).  d,e,v,p are LDS scratch of length n.  Called by the whole block.
__device__ inline double min_eig_lds(double* A, int n, double* d, double* e, double* v, double* p, int tid, int nth) {
  __shared__ double s_scal[4];
  // mirror lower -> upper so that rows can be read contiguously
  for (int idx = tid; idx < n * n; idx += nth) { int i = idx / n, j = idx % n; if (j > i) A[i * n + j] = A[j * n + i]; }
  __syncthreads();
  for (int k = 0; k + 2 < n; k++) {
    const int m = n - k - 1;  // size of trailing block, rows/cols k+1..n-1
    if (tid == 0) {
      double sig = 0;
      for (int i = 1; i < m; i++) sig += A[(k + 1 + i) * n + k] * A[(k + 1 + i) * n + k];
      const double x0 = A[(k + 1) * n + k];
      if (sig == 0) { s_scal[0] = 0; s_scal[1] = x0; }  // nothing to eliminate
      else {
        const double nrm = sqrt(x0 * x0 + sig);
        const double alpha = x0 > 0 ? -nrm : nrm;
        const double v0 = x0 - alpha;
        s_scal[0] = 2.0 / (v0 * v0 + sig);  // beta
        s_scal[1] = alpha;
        s_scal[2] = v0;
      }
    }
    __syncthreads();
    const double beta = s_scal[0];
    if (tid == 0) { d[k] = A[k * n + k]; e[k] = s_scal[1]; }
    if (beta != 0) {
      for (int i = tid; i < m; i += nth) v[i] = (i == 0) ? s_scal[2] : A[(k + 1 + i) * n + k];
      __syncthreads();
      for (int i = tid; i < m; i += nth) {
        double acc = 0;
        for (int j = 0; j < m; j++) acc += A[(k + 1 + i) * n + (k + 1 + j)] * v[j];
        p[i] = beta * acc;
      }
      __syncthreads();
      if (tid == 0) { double kk = 0; for (int i = 0; i < m; i++) kk += v[i] * p[i]; s_scal[3] = 0.5 * beta * kk; }
      __syncthreads();
      const double K = s_scal[3];
      for (int i = tid; i < m; i += nth) p[i] = p[i] - K * v[i];  // q
      __syncthreads();
      for (int idx = tid; idx < m * m; idx += nth) {
        const int i = idx / m, j = idx % m;
        A[(k + 1 + i) * n + (k + 1 + j)] -= v[i] * p[j] + p[i] * v[j];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (n >= 2) { d[n - 2] = A[(n - 2) * n + (n - 2)]; e[n - 2] = A[(n - 1) * n + (n - 2)]; }
    d[n - 1] = A[(n - 1) * n + (n - 1)];
    // Gershgorin interval
    double lo = d[0], hi = d[0];
    for (int i = 0; i < n; i++) {
      double r = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < n ? fabs(e[i]) : 0.0);
      lo = fmin(lo, d[i] - r); hi = fmax(hi, d[i] + r);
    }
    s_scal[0] = lo; s_scal[1] = hi;
  }
  __syncthreads();
  // multisection on the Sturm count "#eigenvalues < x >= 1"; lanes 0..63 of the first wave
  for (int round = 0; round < 13; round++) {
    const double lo = s_scal[0], hi = s_scal[1];
    __syncthreads();
    if (tid < 64) {
      const double x = lo + (hi - lo) * (double(tid + 1) / 65.0);
      double q = d[0] - x;
      int cnt = q < 0;
      for (int i = 1; i < n; i++) {
        if (q == 0) q = 1e-300;
        q = d[i] - x - e[i - 1] * e[i - 1] / q;
        cnt += q < 0;
      }
      const unsigned long long mask = __ballot(cnt >= 1);
      if (tid == 0) {
        if (mask == 0) { s_scal[0] = lo + (hi - lo) * (64.0 / 65.0); }
        else {
          const int f = __ffsll((long long)mask) - 1;  // first shift with an eigenvalue below it
          s_scal[1] = lo + (hi - lo) * (double(f + 1) / 65.0);
          if (f > 0) s_scal[0] = lo + (hi - lo) * (double(f) / 65.0);
        }
      }
    }
    __syncthreads();
  }
  const double r = 0.5 * (s_scal[0] + s_scal[1]);
  __syncthreads();
  return r;
}

}  // namespace tj
