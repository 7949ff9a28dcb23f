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
// extra dependent steps.  Two arithmetic forms (template flag FAST): the EXACT one (IEEE sqrt / division, unfused a - l*l)
// gives every entry the subtraction sequence of the scalar left-looking loop -- bit-identical to the CPU oracle / Eigen's
// unblocked LLT, kept wherever a pass/fail decision is pinned to it -- and the FAST one of the x-update's big systems
// (reciprocal root, fma; see pivot_rsqrt), whose register, LDS and band variants agree with each other bit for bit.
#pragma once
#include "dev_common.h"

namespace tj {

// Block barrier, or -- when the caller guarantees that ONE wave is all that is left of the block --
// just an ordering point: a wave's LDS operations complete in issue order, so lanes of the same
// wave only need the compiler (and the LDS counter) not to reorder across it.
template <bool ONE_WAVE>
__device__ __forceinline__ void blk_sync() {
  if constexpr (ONE_WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
  else __syncthreads();
}


// Barrier among SOME of a block's waves (the others are busy or gone, so s_barrier cannot be used): arrivals are counted in an
// LDS word that starts at 0; `target` is the caller's running total (nwaves more per barrier).  Every wave of the group must call it.
__device__ __forceinline__ void group_barrier(int* cnt, int& target, int nwaves) {
  target += nwaves;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int CHOL_MB = 18;  // max band rows below a pivot handled by the wave kernel (bw <= 18, dense n <= 20)

// FAST forms of the factorisations below (the x-update's per-robot Newton systems, which are one dependent chain of up to
// 9P-2 pivots on a single wave): the pivot's RECIPROCAL root comes from v_rsq_f64 + two Newton steps (rounding level) and
// multiplies the column, the right-hand side and -- stored on the factor's diagonal -- the back substitution, so the IEEE
// sqrt and the divisions (~60 dependent instructions per pivot) leave the chain; the trailing update is one fma per entry.
// Entries differ from the exact form by an ulp or so, which is what Eigen's blocked LLT differs by anyway (the reference
// factors these n >= 43 systems panel-wise); all FAST variants (registers, LDS, band) perform the same operations in the same
// order and agree bit for bit.  The exact forms stay where a pass/fail decision is pinned to Eigen's unblocked LLT (per-piece
// 19x19 check, slack system, known-answer hook).
__device__ __forceinline__ double pivot_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  double e = fma(-(x * r), r, 1.0);
  r = fma(0.5 * r, e, r);
  e = fma(-(x * r), r, 1.0);
  return fma(0.5 * r, e, r);
}

// In-place lower Cholesky of the row-major n x n matrix A (lower triangle is read and written).
// Pattern: half-bandwidth bw (bw >= n-1 means dense) plus a dense last row.  Returns false as soon as
// a pivot is <= 0 (Eigen LLT.h:320-323; NaN pivots pass, like Eigen).  If y != nullptr, y <- L^-1 y.
// Must be called by all `nth` threads of the block (contains barriers); the first wave does the work.
// npiv < n stops after npiv pivots: with npiv = n-1 the last diagonal entry is left as the Schur
// complement a_nn - sum l_nk^2 and y[n-1] as the matching reduced right-hand side (coupled mode:
// the shared-time corner is completed across robots before its pivot can be taken).
template <bool ONE_WAVE = false, bool FAST = false>
__device__ inline bool chol_arrow_lds(double* A, int n, int bw, int tid, int nth, double* y = nullptr, int npiv = -1) {
  if (npiv < 0) npiv = n;
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
  // Each pivot costs three LDS round trips, not one per operand: every lane first issues ALL the
  // loads it will need (pivot, its column entry, its window positions, arrow and rhs entries, with
  // clamped indices so that no load sits behind a branch), then the scaled column is exchanged
  // through s_col, then everything is stored.
  for (int k = 0; k < npiv; k++) {
    blk_sync<ONE_WAVE>();
    const int mb = max(0, min(min(bw, CHOL_MB), last - 1 - k));  // band rows below the pivot (arrow row excluded)
    const bool arrow = k < last;
    const int myrow = tid < mb ? k + 1 + tid : last;
    const double x = A[k * n + k];
    const double ci = A[myrow * n + k];
    double w[3]; int widx[3]; bool wact[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      wact[t] = er[t] >= 0 && er[t] < mb;
      widx[t] = wact[t] ? (k + 1 + er[t]) * n + (k + 1 + ec[t]) : k * n + k;
      w[t] = A[widx[t]];
    }
    const int aidx = last * n + min(k + 1 + tid, last);
    const double ae = A[aidx], add = A[last * n + last];
    double yk_raw = 0, yi = 0, ylast = 0;
    if (y) { yk_raw = y[k]; yi = y[min(k + 1 + tid, last)]; ylast = y[last]; }
    if (x <= 0) return false;  // uniform: every thread read the same LDS word
    const double sx = FAST ? pivot_rsqrt(x) : sqrt(x);   // FAST: the diagonal holds 1 / l_kk
    const double v = FAST ? ci * sx : ci / sx;
    const double yk = FAST ? yk_raw * sx : yk_raw / sx;
    blk_sync<ONE_WAVE>();
    if (tid == 0) A[k * n + k] = sx;
    if (tid < mb) { A[myrow * n + k] = v; s_col[tid] = v; }
    else if (tid == mb && arrow) { A[last * n + k] = v; s_col[CHOL_MB] = v; }
    blk_sync<ONE_WAVE>();
    if (tid < 64) {
      const double la = s_col[CHOL_MB];
      const double cm = s_col[min(tid, CHOL_MB - 1)];
      double cr[3], cc[3];
#pragma unroll
      for (int t = 0; t < 3; t++) { cr[t] = s_col[max(er[t], 0)]; cc[t] = s_col[ec[t]]; }
#pragma unroll
      for (int t = 0; t < 3; t++) if (wact[t]) A[widx[t]] = FAST ? fma(-cr[t], cc[t], w[t]) : w[t] - cr[t] * cc[t];
      if (arrow) {
        if (tid < mb) A[aidx] = FAST ? fma(-la, cm, ae) : ae - la * cm;
        if (tid == CHOL_MB + 1) A[last * n + last] = FAST ? fma(-la, la, add) : add - la * la;
      }
      if (y) {
        if (tid < mb) y[k + 1 + tid] = FAST ? fma(-yk, cm, yi) : yi - yk * cm;
        else if (tid == mb && arrow) y[last] = FAST ? fma(-yk, la, ylast) : ylast - yk * la;
        if (tid == 63) y[k] = yk;
      }
    }
  }
  blk_sync<ONE_WAVE>();
  return true;
}

// Dense N x N LLT success test in the registers of ONE wave: lane i holds row i (r[c] = A[i][c]).
// Same right-looking operation order as chol_arrow_lds (a_ij - l_ik * l_jk, k ascending), so the
// pivots are bit-identical; pivot and scaled column are broadcast with v_readlane -- no LDS round
// trip, no barrier.  Fully unrolled (N is the 19 of a piece block).  Lanes >= N carry don't-cares.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// esum() by one wave: the four running sums of Eigen's unrolled 2-wide reduction advance on four lanes at once (lane c adds
// e[c], e[c+4], ...) and are then combined exactly as esum combines them -- same association, a quarter of the dependent adds.
__device__ __forceinline__ double esum_wave(const double* e, int n, int lane) {
  if (n < 8) return esum(e, n);
  const int a2 = (n / 4) * 4, a1 = (n / 2) * 2, c = lane & 3;
  double acc = e[c];
  for (int i = 4 + c; i < a2; i += 4) acc += e[i];
  double r0a = readlane_f64(acc, 0), r0b = readlane_f64(acc, 1);
  r0a += readlane_f64(acc, 2); r0b += readlane_f64(acc, 3);
  if (a1 > a2) { r0a += e[a2]; r0b += e[a2 + 1]; }
  double r = r0a + r0b;
  for (int i = a1; i < n; i++) r += e[i];
  return r;
}

__device__ __forceinline__ bool uni_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
template <int N>
__device__ __forceinline__ bool chol_check_wave(double (&r)[N]) {
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double x = readlane_f64(r[k], k);
    if (x <= 0) return false;
    const double lik = r[k] / sqrt(x);
#pragma unroll
    for (int j = k + 1; j < N; j++) r[j] = r[j] - lik * readlane_f64(lik, j);
  }
  return true;
}

// Arrowhead-band Cholesky of an N x N system (N <= 64) entirely in the registers of ONE wave: lane i holds
// the full row i (r[j] = A[i][j], both triangles), the right-hand side rides along in y (lane i = y_i).
// Same operations in the same order as chol_arrow_lds<.., FAST = true> (fma(-l_ik, l_jk, a_ij) with k ascending; column
// scaled by the refined reciprocal root, which is what the diagonal keeps), so factor and forward-substituted rhs are
// bit-identical to it -- but a pivot is ~75 straight-line instructions (v_readlane broadcasts, no LDS round trip, no
// barrier) instead of three LDS round trips.  Pattern: half-bandwidth BW plus a dense last row.  npiv = N - 1 leaves the Schur
// complement of the last diagonal entry in lane N-1 (see chol_arrow_lds).  Returns false on a pivot <= 0.
// PIECES: the system is the x-update's (k_xsolve): band row i is reduced coordinate i + 6 of a chain of 18-coordinate piece
// blocks that overlap by 9, so column k meets rows up to 9 * ((k + 6) / 9) + 11 only -- 17 - (k + 6) % 9 of them, 13 on
// average, not BW = 17.  The entries beyond are structural zeros (never filled); their updates fma(-0 * x, ., a) are skipped,
// which leaves every stored value unchanged (up to the sign of an exact zero).
template <int N, int BW, bool PIECES = false, bool BATCH = true>   // BATCH = false: broadcasts and updates pairwise (few SGPRs: the form kept for out-of-line callers)
__device__ __forceinline__ bool chol_arrow_wave(double (&r)[N], double& y, int lane, int npiv) {
  // What one wave pays here (tools/micro/issue_probe.hip): ~6 cycles per fp64 instruction whether or not it depends on the one
  // before, ~35 cycles from a v_readlane to the first VALU use of the SGPR it wrote, ~25 for a branch on a fresh VALU
  // comparison.  So: (1) the next pivot's diagonal is formed by every lane from its OWN column entry (lane k+1 holds
  // l_{k+1,k} itself: same operands, same bits as the broadcast form) and read out first, so its broadcast travels while the
  // column updates issue; (2) all broadcasts of the column go out before the first update, each into its own SGPR pair.
  // (The early exit on a non-positive pivot stays a branch per pivot: without it the 43 pivots are one basic block of 3 000
  // instructions, and the scheduler's reordering inside it doubles the register count -- 180 VGPRs, 106 SGPRs, spills.)
  constexpr int LAST = N - 1;
  if constexpr (!BATCH) {   // the plain right-looking loop (round 2's form)
#pragma unroll
    for (int k = 0; k < N; k++) {
      if (k == N - 1 && npiv < N) break;
      const double x = readlane_f64(r[k], k);
      if (x <= 0) return false;
      const double rs = pivot_rsqrt(x);
      const double lik = r[k] * rs;
      const double yk = readlane_f64(y, k) * rs;
      r[k] = lane == k ? rs : lik;
      const int jband = (k + BW < LAST - 1) ? k + BW : LAST - 1;
      const int jreach = 9 * ((k + 6) / 9) + 11;
      const int jhi = (PIECES && jreach < jband) ? jreach : jband;
#pragma unroll
      for (int j = k + 1; j <= jhi; j++) r[j] = fma(-lik, readlane_f64(lik, j), r[j]);
      if (k < LAST) r[LAST] = fma(-lik, readlane_f64(lik, LAST), r[LAST]);
      y = lane == k ? yk : (lane > k ? fma(-yk, lik, y) : y);
    }
    return true;
  }
  double x = readlane_f64(r[0], 0);
#pragma unroll
  for (int k = 0; k < N; k++) {
    if (k == N - 1 && npiv < N) break;
    if (x <= 0) return false;
    const double rs = pivot_rsqrt(x);
    const double lik = r[k] * rs;
    const double ykb = readlane_f64(y, k);   // consumed after the column updates
    r[k] = lane == k ? rs : lik;
    const int jband = (k + BW < LAST - 1) ? k + BW : LAST - 1;
    const int jreach = 9 * ((k + 6) / 9) + 11;
    const int jhi = (PIECES && jreach < jband) ? jreach : jband;
    // next pivot: a_{k+1,k+1} - l_{k+1,k}^2, formed by lane k+1 from its own column entry (k + 1 == LAST: the arrow row's diagonal)
    if (k < LAST) { const double dn = fma(-lik, lik, r[k + 1]); x = readlane_f64(dn, k + 1); }
    double lj[BW + 1];
#pragma unroll
    for (int j = k + 1; j <= jhi; j++) lj[j - k - 1] = readlane_f64(lik, j);
    const double larrow = k < LAST ? readlane_f64(lik, LAST) : 0.0;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = k + 1; j <= jhi; j++) r[j] = fma(-lik, lj[j - k - 1], r[j]);
    if (k < LAST) r[LAST] = fma(-lik, larrow, r[LAST]);
    const double yk = ykb * rs;
    y = lane == k ? yk : (lane > k ? fma(-yk, lik, y) : y);
  }
  return true;
}

// ---- band storage (long trajectories: piece_num > 10, where a dense n x n copy no longer fits LDS) ----------------------
// The same arrowhead-band factorisation and solve on compact storage: Bd[m][BS] holds the band rows (m = n - 1 of them),
// Bd[i][c] = A[i][i - (BS-1) + c], c = BS-1 the diagonal; Ar[n] holds the arrow row A[n-1][0..n-1].  Same operation order as
// the FAST chol_arrow_lds / chol_arrow_backsolve_lds (fma(-l_ik, l_jk, a_ij), k ascending; reciprocal root on the diagonal), executed by
// ONE wave; y rides along (y <- L^-1 y).  Returns false on a pivot <= 0.
constexpr int BAND_BS = 18;   // half-bandwidth 17 + diagonal
// last_pivot = false (coupled mode): the arrow row's diagonal is left as the Schur complement a_nn - sum l_nk^2 and y[n-1] as the
// matching reduced right-hand side, like chol_arrow_lds with npiv = n - 1 (the shared-time corner is completed across robots)
__device__ inline bool chol_band_lds(double* Bd, double* Ar, int n, int tid, double* y, bool last_pivot = true) {
  __shared__ double s_colb[CHOL_MB + 2];
  const int m = n - 1, BS = BAND_BS;
  int er[3], ec[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int e = tid + 64 * t;
    er[t] = -1; ec[t] = 0;
    if (e < (BS - 1) * BS / 2) { int r = 0; while ((r + 1) * (r + 2) / 2 <= e) r++; er[t] = r; ec[t] = e - r * (r + 1) / 2; }
  }
  for (int k = 0; k < m; k++) {
    blk_sync<true>();
    const int mb = min(BS - 1, m - 1 - k);
    const double x = Bd[k * BS + BS - 1];
    const int myrow = k + 1 + min(tid, max(mb - 1, 0));
    const double ci = mb > 0 ? Bd[myrow * BS + (BS - 2 - min(tid, mb - 1))] : 0.0;
    double w[3]; int widx[3]; bool wact[3];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      wact[t] = er[t] >= 0 && er[t] < mb;
      widx[t] = wact[t] ? (k + 1 + er[t]) * BS + (ec[t] - er[t] + BS - 1) : k * BS + BS - 1;
      w[t] = Bd[widx[t]];
    }
    const double ak = Ar[k], ae = Ar[min(k + 1 + tid, m)], add = Ar[m];
    const double yk_raw = y[k], yi = y[min(k + 1 + tid, m)], ylast = y[m];
    if (x <= 0) return false;
    const double sx = pivot_rsqrt(x);   // FAST form: reciprocal root, kept on the diagonal
    const double v = ci * sx, la = ak * sx, yk = yk_raw * sx;
    blk_sync<true>();
    if (tid == 0) { Bd[k * BS + BS - 1] = sx; Ar[k] = la; y[k] = yk; }
    if (tid < mb) { Bd[myrow * BS + (BS - 2 - tid)] = v; s_colb[tid] = v; }
    blk_sync<true>();
    const double cm = s_colb[min(tid, CHOL_MB - 1)];
#pragma unroll
    for (int t = 0; t < 3; t++) if (wact[t]) Bd[widx[t]] = fma(-s_colb[er[t]], s_colb[ec[t]], w[t]);
    if (tid < mb) { Ar[k + 1 + tid] = fma(-la, cm, ae); y[k + 1 + tid] = fma(-yk, cm, yi); }
    if (tid == 63) { Ar[m] = fma(-la, la, add); y[m] = fma(-yk, la, ylast); }
  }
  blk_sync<true>();
  if (!last_pivot) return true;
  const double x = Ar[m];
  if (x <= 0) return false;
  blk_sync<true>();
  if (tid == 0) { const double sx = pivot_rsqrt(x); Ar[m] = sx; y[m] = y[m] * sx; }
  blk_sync<true>();
  return true;
}
__device__ inline void chol_band_backsolve_lds(const double* Bd, const double* Ar, int n, double* y, int tid) {
  const int m = n - 1, BS = BAND_BS;
  blk_sync<true>();
  const double xl = y[m] * Ar[m];
  blk_sync<true>();
  if (tid == 0) y[m] = xl;
  for (int i = tid; i < m; i += 64) y[i] = fma(-xl, Ar[i], y[i]);
  for (int j = m - 1; j >= 0; j--) {
    blk_sync<true>();
    const int lo = max(0, j - (BS - 1));
    const int i0 = min(lo + tid, j);
    const double yj = y[j] * Bd[j * BS + BS - 1], lji = Bd[j * BS + (i0 - j + BS - 1)], yi = y[i0];
    blk_sync<true>();
    if (tid == 0) y[j] = yj;
    if (lo + tid < j) y[i0] = fma(-yj, lji, yi);
  }
  blk_sync<true>();
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
template <bool ONE_WAVE = false, bool FAST = false>
__device__ inline void chol_arrow_backsolve_lds(const double* L, int n, int bw, double* y, int tid, int nth) {
  const int last = n - 1;
  for (int j = n - 1; j >= 0; j--) {
    blk_sync<ONE_WAVE>();
    const int lo = (j == last) ? 0 : max(0, j - bw);
    const int i0 = min(lo + tid, j);                 // first element of this lane (clamped: load is unconditional)
    const double yj_raw = y[j], ljj = L[j * n + j], lji = L[j * n + i0], yi = y[i0];
    const double yj = FAST ? yj_raw * ljj : yj_raw / ljj;   // every thread computes it; one publishes it (FAST: the diagonal holds 1 / l_jj)
    blk_sync<ONE_WAVE>();
    if (tid == 0) y[j] = yj;
    if (lo + tid < j) y[i0] = FAST ? fma(-yj, lji, yi) : yi - yj * lji;
    for (int i = lo + tid + nth; i < j; i += nth) y[i] = FAST ? fma(-yj, L[j * n + i], y[i]) : y[i] - yj * L[j * n + i];   // only the arrow row of large systems
  }
  blk_sync<ONE_WAVE>();
}

// Sturm test "does the symmetric tridiagonal T (diagonal d, squared off-diagonals e2) have an eigenvalue <= x": T - xI is positive
// definite iff all its leading principal minors p_i = (d_i - x) p_{i-1} - e2_i p_{i-2} are > 0.  The minors are carried in PRODUCT
// form: two multiplications and a subtraction on the dependent chain per row, where the quotient form q_i = p_i / p_{i-1}
// (LAPACK dstebz) costs a division -- ~100 cycles of a chain of 18 x 9 steps that sets k_grad's duration when a block is
// repaired.  Both forms evaluate the same recurrence with O(eps) relative error per step.  Entries are scaled by a power of
// two (exact) so that a step grows a minor at most 3x, and every third step the pair is renormalised by the exponent of the
// current minor (v_frexp_exp / v_ldexp, exact), so graded spectra cannot underflow to a false zero.
struct SturmScale { double is, is2; };   // 2^-k and 2^-2k with 2^k > max(|lo|, |hi|) of the Gershgorin interval
__device__ __forceinline__ SturmScale sturm_scale(double lo, double hi) {
  const double s = fmax(fabs(lo), fabs(hi));
  const int k = (s > 0 && s < 1e300) ? __builtin_amdgcn_frexp_exp(s) : 0;
  return SturmScale{__builtin_ldexp(1.0, -k), __builtin_ldexp(1.0, -2 * k)};
}
__device__ __forceinline__ void sturm_first(double d0, double x, const SturmScale sc, double& p0, double& p1, bool& below) {
  p0 = 1.0; p1 = (d0 - x) * sc.is; below = !(p1 > 0);
}
__device__ __forceinline__ void sturm_next(double di, double e2, double x, const SturmScale sc, int i, double& p0, double& p1, bool& below) {
  const double pn = ((di - x) * sc.is) * p1 - (e2 * sc.is2) * p0;
  p0 = p1; p1 = pn; below = below || !(pn > 0);
  if (i % 3 == 0) { const int ex = __builtin_amdgcn_frexp_exp(p1); p0 = __builtin_ldexp(p0, -ex); p1 = __builtin_ldexp(p1, -ex); }
}
constexpr int STURM_ROUNDS = 9;   // 64-way multisection: 65^9 > 2^54 subdivisions of the Gershgorin interval (its width is a few times the
                                  // matrix norm, so the result is good to ~1e-16 of the norm; the parity bar is 1e-12 of it)

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
  const SturmScale sc = sturm_scale(s_scal[0], s_scal[1]);
  for (int round = 0; round < STURM_ROUNDS; round++) {
    const double lo = s_scal[0], hi = s_scal[1];
    __syncthreads();
    if (tid < 64) {
      const double x = lo + (hi - lo) * (double(tid + 1) / 65.0);
      double p0, p1; bool below;
      sturm_first(d[0], x, sc, p0, p1, below);
      for (int i = 1; i < n; i++) sturm_next(d[i], e[i - 1] * e[i - 1], x, sc, i, p0, p1, below);
      const unsigned long long mask = __ballot(below);
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

// min_eig_lds for an N x N block held one FULL row per lane in the registers of one wave (both
// triangles): no LDS round trips, no barriers, nothing serialised on one thread.  This is the copy on k_grad's critical path
// (a repaired piece waits for it), shaped by what a single wave pays (tools/micro/issue_probe.hip: ~6 cycles per fp64
// instruction dependent or not, ~35 from a v_readlane to the first use of its SGPR):
//   * lane k holds row k = column k of the symmetric block, so the Householder scalars of step k (column norm, alpha, v0, beta)
//     are formed by every lane from its OWN row and five values are read out of lane k -- not one broadcast per column entry;
//   * the broadcasts of v and of q go out in a batch, each into its own SGPR pair, before the products that use them;
//   * v . p is a DPP tree over the two rows of 16 lanes instead of a chain of N readlane-adds;
//   * products are fused, norm and beta come from v_rsq / v_rcp + Newton steps, the Sturm recurrence is 3 instructions per row.
// Householder tridiagonalisation is backward stable under any of these roundings; the routine agrees with min_eig_lds to
// ~1e-15 of the norm (known-answer hook: 1e-13) and with Eigen's value to the 1e-12 the parity tests ask for.
// stop (optional, LDS): the caller no longer needs the result once *stop == 1 (checked between Householder steps; uniform)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false));
}
// sum over lanes 0..31 (two DPP rows), the same value in every lane of those rows' first lanes' broadcast; lanes >= 32 are ignored
__device__ __forceinline__ double sum32_wave(double v) {
  v += dpp_mov_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x124>(v);   // row_ror:4
  v += dpp_mov_f64<0x128>(v);   // row_ror:8
  return readlane_f64(v, 0) + readlane_f64(v, 16);
}
template <int N>
__device__ __forceinline__ double min_eig_wave(double (&r)[N], int lane, const volatile int* stop = nullptr) {
  static_assert(N <= 32, "rows live in lanes 0..31 (sum32_wave)");
  double d[N], e[N];  // wave-uniform
#pragma unroll
  for (int k = 0; k + 2 < N; k++) {
    if (stop && *stop == 1) return 0.0;
    // every lane: the scalars of "its" Householder step; lane k's are the ones of step k
    double sig = 0;
#pragma unroll
    for (int i = k + 2; i < N; i++) sig = fma(r[i], r[i], sig);
    const double x0l = r[k + 1];
    const double s2 = fma(x0l, x0l, sig);
    const double nrm = s2 * pivot_rsqrt(s2);
    const double alphal = x0l > 0 ? -nrm : nrm;
    const double v0l = x0l - alphal;
    const double den = fma(v0l, v0l, sig);
    double rc = __builtin_amdgcn_rcp(den);
    rc = fma(rc, fma(-den, rc, 1.0), rc);
    rc = fma(rc, fma(-den, rc, 1.0), rc);
    const double sigk = readlane_f64(sig, k), x0 = readlane_f64(x0l, k);
    d[k] = readlane_f64(r[k], k);
    const double alpha = readlane_f64(alphal, k), v0 = readlane_f64(v0l, k), beta = 2.0 * readlane_f64(rc, k);
    if (sigk == 0) { e[k] = x0; continue; }  // nothing to eliminate (uniform)
    e[k] = alpha;
    const double v = lane == k + 1 ? v0 : ((lane > k + 1 && lane < N) ? r[k] : 0.0);
    double vj[N];
#pragma unroll
    for (int j = k + 1; j < N; j++) vj[j] = readlane_f64(v, j);
    __builtin_amdgcn_sched_barrier(0);
    double acc = 0;
#pragma unroll
    for (int j = k + 1; j < N; j++) acc = fma(r[j], vj[j], acc);
    const double p = (lane > k && lane < N) ? beta * acc : 0.0;
    const double kk = sum32_wave(v * p);
    const double K = 0.5 * beta * kk;
    const double q = fma(-K, v, p);
    double qj[N];
#pragma unroll
    for (int j = k + 1; j < N; j++) qj[j] = readlane_f64(q, j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = k + 1; j < N; j++) r[j] = fma(-q, vj[j], fma(-v, qj[j], r[j]));
  }
  d[N - 2] = readlane_f64(r[N - 2], N - 2); e[N - 2] = readlane_f64(r[N - 2], N - 1);
  d[N - 1] = readlane_f64(r[N - 1], N - 1);
  double lo = d[0], hi = d[0];
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double rad = (i > 0 ? fabs(e[i - 1]) : 0.0) + (i + 1 < N ? fabs(e[i]) : 0.0);
    lo = fmin(lo, d[i] - rad); hi = fmax(hi, d[i] + rad);
  }
  // Sturm recurrence on pre-scaled entries: p_i = (d_i s - x s) p_{i-1} - (e_i^2 s^2) p_{i-2}, one subtraction, one product, one fma per row
  const SturmScale sc = sturm_scale(lo, hi);
  double ds[N], es[N];
#pragma unroll
  for (int i = 0; i < N; i++) { ds[i] = d[i] * sc.is; es[i] = i > 0 ? (e[i - 1] * e[i - 1]) * sc.is2 : 0.0; }
  for (int round = 0; round < STURM_ROUNDS; round++) {
    const double x = lo + (hi - lo) * (double(lane + 1) / 65.0);
    const double xs = x * sc.is;
    double p0 = 1.0, p1 = ds[0] - xs;
    bool below = !(p1 > 0);
#pragma unroll
    for (int i = 1; i < N; i++) {
      const double pn = fma(ds[i] - xs, p1, -(es[i] * p0));
      p0 = p1; p1 = pn; below = below | !(pn > 0);
      if (i % 3 == 0) { const int ex = __builtin_amdgcn_frexp_exp(p1); p0 = __builtin_ldexp(p0, -ex); p1 = __builtin_ldexp(p1, -ex); }
    }
    const unsigned long long mask = __ballot(below);
    if (mask == 0) lo = lo + (hi - lo) * (64.0 / 65.0);
    else {
      const int f = __ffsll((long long)mask) - 1;
      const double nhi = lo + (hi - lo) * (double(f + 1) / 65.0);
      if (f > 0) lo = lo + (hi - lo) * (double(f) / 65.0);
      hi = nhi;
    }
  }
  return 0.5 * (lo + hi);
}

}  // namespace tj
