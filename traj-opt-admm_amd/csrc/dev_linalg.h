// dev_linalg.h -- small dense SPD kernels that run inside one workgroup on an LDS-resident
// matrix (n <= ~100): right-looking Cholesky with the reference's failure test, triangular
// solves, and the smallest eigenvalue of a symmetric matrix (Householder tridiagonalisation +
// 64-way Sturm multisection).  These replace Eigen::LLT / SelfAdjointEigenSolver as used in
// Gradient_admm.h:38-53, Optimization3D_multi.h:697-722 and :423-446.  No MFMA: n is tiny and
// the work is latency bound; the matrix never leaves LDS.
//
// The systems on the hot path are all "arrowhead + band": unknowns are control-point coordinates
// that couple at most `bw` positions apart, plus one dense last row/column (piece time).  A
// factorisation is a chain of n dependent pivots executed by ONE wavefront, so what matters is the
// instruction count per pivot, not parallel width: each lane owns a fixed handful of positions of
// the (bw+1)^2/2 update window, the scaled pivot column is exchanged through a 19-word LDS vector,
// and the right-hand side rides along as one more row so that the forward substitution costs no
// extra dependent steps.  Every entry sees exactly the subtraction sequence of the scalar
// left-looking loop (CPU oracle / Eigen's unblocked LLT), i.e. results are bit-identical to it.
#pragma once
#include "dev_common.h"

namespace tj {

constexpr int CHOL_MB = 18;  // max band rows below a pivot handled by the wave kernel (bw <= 18, dense n <= 20)

// In-place lower Cholesky of the row-major n x n matrix A (lower triangle is read and written).
// Pattern: half-bandwidth bw (bw >= n-1 means dense) plus a dense last row.  Returns false as soon as
// a pivot is <= 0 (Eigen LLT.h:320-323; NaN pivots pass, like Eigen).  If y != nullptr, y <- L^-1 y.
// Must be called by all `nth` threads of the block (contains barriers); the first wave does the work.
__device__ inline bool chol_arrow_lds(double* A, int n, int bw, int tid, int nth, double* y = nullptr) {
  __shared__ double s_col[CHOL_MB + 2];  // scaled pivot column: [0..mb) band rows, [CHOL_MB] arrow row
  const int last = n - 1;
  // fixed ownership of the lower-triangular update window (r,c), c <= r < CHOL_MB: 171 positions over 64 lanes
  int er[3], ec[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int e = tid + 64 * t;
    er[t] = -1; ec[t] = 0;
    if (tid < 64 && e < CHOL_MB * (CHOL_MB + 1) / 2) {
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= e) r++;
      er[t] = r; ec[t] = e - r * (r + 1) / 2;
    }
  }
  for (int k = 0; k < n; k++) {
    __syncthreads();
    const double x = A[k * n + k];
    if (x <= 0) return false;  // uniform: every thread reads the same LDS word
    const double sx = sqrt(x);
    const int mb = max(0, min(min(bw, CHOL_MB), last - 1 - k));  // band rows below the pivot (arrow row excluded)
    const bool arrow = k < last;
    double yk = 0;
    if (y) yk = y[k] / sx;
    __syncthreads();
    if (tid == 0) A[k * n + k] = sx;
    if (tid < mb) { const double v = A[(k + 1 + tid) * n + k] / sx; A[(k + 1 + tid) * n + k] = v; s_col[tid] = v; }
    else if (tid == mb && arrow) { const double v = A[last * n + k] / sx; A[last * n + k] = v; s_col[CHOL_MB] = v; }
    __syncthreads();
    if (tid < 64) {
      double* base = A + (size_t)(k + 1) * n + (k + 1);
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int r = er[t], c = ec[t];
        if (r >= 0 && r < mb) base[r * n + c] = base[r * n + c] - s_col[r] * s_col[c];
      }
      if (arrow) {
        const double la = s_col[CHOL_MB];
        if (tid < mb) A[last * n + k + 1 + tid] = A[last * n + k + 1 + tid] - la * s_col[tid];
        if (tid == CHOL_MB + 1) A[last * n + last] = A[last * n + last] - la * la;
      }
      if (y) {
        if (tid < mb) y[k + 1 + tid] = y[k + 1 + tid] - yk * s_col[tid];
        else if (tid == mb && arrow) y[last] = y[last] - yk * s_col[CHOL_MB];
        if (tid == 63) y[k] = yk;
      }
    }
  }
  __syncthreads();
  return true;
}

// Generic dense variant for larger matrices (only the known-answer hook uses n > 20).
__device__ inline bool chol_lds(double* A, int n, int tid, int nth, double* y = nullptr) {
  if (n <= CHOL_MB + 2) return chol_arrow_lds(A, n, n, tid, nth, y);  // the 19x19 / 13x13 piece systems
  for (int k = 0; k < n; k++) {
    __syncthreads();
    const double x = A[k * n + k];
    if (x <= 0) return false;
    const double sx = sqrt(x);
    double yk = 0;
    if (y) yk = y[k] / sx;
    __syncthreads();
    if (tid == 0) { A[k * n + k] = sx; if (y) y[k] = yk; }
    for (int i = k + 1 + tid; i < n; i += nth) A[i * n + k] = A[i * n + k] / sx;
    __syncthreads();
    for (int i = k + 1 + tid; i < n; i += nth) {
      const double lik = A[i * n + k];
      for (int j = k + 1; j <= i; j++) A[i * n + j] -= lik * A[j * n + k];
      if (y) y[i] -= yk * lik;
    }
  }
  __syncthreads();
  return true;
}

// x = L^-T y in place (column oriented; row j of L is dense only for the arrow row)
__device__ inline void chol_arrow_backsolve_lds(const double* L, int n, int bw, double* y, int tid, int nth) {
  const int last = n - 1;
  for (int j = n - 1; j >= 0; j--) {
    __syncthreads();
    const double yj = y[j] / L[j * n + j];   // every thread computes it; one publishes it
    __syncthreads();
    if (tid == 0) y[j] = yj;
    const int lo = (j == last) ? 0 : max(0, j - bw);
    for (int i = lo + tid; i < j; i += nth) y[i] -= yj * L[j * n + i];
  }
  __syncthreads();
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
