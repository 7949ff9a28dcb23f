// kernels_newton.h -- the x-update's Newton system.
//
//   k_grad    one workgroup per (robot, piece): 19-vector gradient and 19x19 Hessian of the
//             augmented Lagrangian restricted to that piece, with the reference's per-piece PSD
//             repair.  Replaces Gradient_admm::local_spline_gradient (Gradient_admm.h:67-164),
//             local_plane_barrier_gradient (:331-407), local_bound_gradient (:409-572) and the
//             LLT / eigen-shift block of global_spline_gradient (:38-53).
//   k_xsolve  one workgroup per robot: overlap-add of the piece blocks (Gradient_admm.h:55-62),
//             removal of the fixed end control points, dense Cholesky in LDS with eigen-shift
//             fallback, solve, wolfe and |g|.  Replaces spline_descent_direction
//             (Optimization3D_multi.h:659-752, Optimization3D_admm.h:400-503).
//
// GPU shape: one thread owns one lower-triangle Hessian entry (or gradient entry); no atomics, no reduction trees, so a
// result is a function of the inputs alone (bitwise reproducible, and identical in both launch forms).  Barrier derivatives
// (the only transcendental work) are computed once per (plane, control point) by the whole block and staged through LDS; the
// planes of a segment then enter an entry only through the per-control-point sums M_j = sum_k e2 n n^T and v_j = sum_k e1 n
// (grad_plane_batch), the velocity / acceleration terms through one 3x3 matrix per record (grad_velacc_records) -- the
// reference's double sums with the inner sum taken first.  The Kronecker selector matrices A_list / A_vel_list / A_acc_list
// of the reference are never formed: A[tr][j] * c == basis(j,:)^T (x) c.
#pragma once
#include "dev_common.h"
#include "dev_linalg.h"
#include "kernels_pairs.h"

namespace tj {

constexpr int GRAD_THREADS = 192;
constexpr int GRAD_FOLD_THREADS = 512;   // FOLD: 8 waves compact the piece's segments (one each) before 3 of them go on

// dynamic LDS layout of k_grad, in doubles; npl = cap_obs + cap_self (capacity of one plane batch)
constexpr int GRAD_MAXRES = 16;  // segments per piece staged at once ("res" of 3D.json, shipped value 8)
__host__ __device__ inline size_t grad_lds_doubles(int npl, int res) {  // sized by the actual res: 2 blocks must fit one CU
  return (size_t)res * (18 + 36) + 16 * (size_t)npl + (size_t)res * 9 * 20 + (size_t)res * 54 + 361 + 19 + 4 * 19 + 2 * GRAD_MAXRES + 16;
}
// folded launch only: behind that layout, the planes the block's own compaction hands to the gradient through LDS -- per segment the
// first GRAD_PST obstacle planes and the first GRAD_PST robot-pair planes (a longer list is read back from the global list, as before)
constexpr int GRAD_PST = 16;
__host__ __device__ inline size_t grad_fold_extra_doubles(int res) { return (size_t)res * GRAD_PST * 4 * 2; }

// Velocity / acceleration barrier terms of a piece (Gradient_admm.h:107-129, :409-572).  grad_velacc_records: one thread per
// (segment, record) -- 5 velocity and 4 acceleration records per segment -- writes the GRAD_REC values the accumulation needs and
// a bitmask of the active records (most are inactive: the limits bind on few segments).  A record's contribution to Hessian
// entry ((a,q),(a',q')) is w_a w_a' (e2 dp_q dp_q' + e1 hp_qq') with w the record's 6-vector of basis differences: the 3x3 matrix
// N = e2 dp dp^T + e1 hp (symmetric, 6 values) is formed ONCE per record here instead of once per entry in the accumulation.
constexpr int GRAD_REC = 20;   // N[6] (00,10,11,20,21,22), e1*dp[3], e3*dp[3], w[6], time gradient, time Hessian
struct GradRole { int tid, hi, ai, qi, ak, qk, vr, av, qv, cq; bool scal; };   // cq: index of the symmetric pair (qi, qk) in {00,10,11,20,21,22}
__device__ __forceinline__ void grad_velacc_records(const Dev& D, int tid, int sp, int res, double m, double pt, const double* Pall, const double* Ball, const double* wseg, double* bt, unsigned long long* amask) {
  bool rec_act = false;
  if (tid < res * 9) {
    const int i = tid / 9, b = tid % 9;
    const double w = wseg[i];
    const double* P = Pall + i * 18; const double* Bs = Ball + i * 36;
    double* t = bt + tid * GRAD_REC;
    double Dv[3], len, d, coef = 0, e1 = 0, e2 = 0, e3 = 0, tg = 0, th = 0;
    bool act;
    if (b < 5) {
      const int j = b;
      for (int a = 0; a < 3; a++) Dv[a] = P[3 * (j + 1) + a] - P[3 * j + a];
      len = norm3(Dv[0], Dv[1], Dv[2]);
      const double v = 5 * len / w;
      d = D.vel_limit - v / pt;
      act = d < m;
      if (act) {
        barrier_d(w, d, m, e1, e2);
        tg = e1 * v / (pt * pt);
        th = -2 * e1 * v / pow3(pt) + e2 * v * v / pow4(pt);
        coef = -5 / (w * pt);
        e3 = -e1 / pt + e2 * (D.vel_limit - d) / pt;
        for (int a = 0; a < 6; a++) t[12 + a] = Bs[(j + 1) * 6 + a] - Bs[j * 6 + a];
      }
    } else {
      const int j = b - 5;
      for (int a = 0; a < 3; a++) Dv[a] = P[3 * (j + 2) + a] - 2 * P[3 * (j + 1) + a] + P[3 * j + a];
      len = norm3(Dv[0], Dv[1], Dv[2]);
      const double acc = 20 * len / (w * w);
      d = D.acc_limit - acc / (pt * pt);
      act = d < m;
      if (act) {
        barrier_d(w, d, m, e1, e2);
        tg = 2 * e1 * acc / pow3(pt);
        th = -6 * e1 * acc / pow4(pt) + 4 * e2 * acc * acc / pow6(pt);
        const double wp = w * pt;
        coef = -20 / (wp * wp);
        e3 = -2 * e1 / pt + 2 * e2 * (D.acc_limit - d) / pt;
        for (int a = 0; a < 6; a++) t[12 + a] = Bs[(j + 2) * 6 + a] - 2 * Bs[(j + 1) * 6 + a] + Bs[j * 6 + a];
      }
    }
    rec_act = act;
    if (act) {
      const double len3 = pow3(len);
      double dp[3];
      for (int q = 0; q < 3; q++) dp[q] = coef * Dv[q] / len;
      for (int q = 0, c = 0; q < 3; q++) for (int s = 0; s <= q; s++, c++) {
        const double hp = coef * ((q == s ? 1.0 : 0.0) / len - Dv[q] * Dv[s] / len3);
        t[c] = e2 * (dp[q] * dp[s]) + e1 * hp;
      }
      for (int q = 0; q < 3; q++) { t[6 + q] = e1 * dp[q]; t[9 + q] = e3 * dp[q]; }
      t[18] = tg; t[19] = th;
    }
  }
  const unsigned long long bal = __ballot(rec_act);
  if ((tid & 63) == 0 && tid < GRAD_THREADS) amask[tid >> 6] = bal;
}
// ... and their accumulation into this thread's entry: per segment partial sums, added in segment order; an all-inactive
// segment adds an exact +0.  a0: Hessian entry / gradient entry / time gradient, a1: time-column entry / time Hessian.
struct GradAcc { double a0, a1; };   // returned by value: accumulators handed around by reference ended up as an indexed array in scratch memory
__device__ __forceinline__ GradAcc grad_velacc_accumulate(const GradRole R, int res, const double* bt, const unsigned long long* amask) {
  double a0 = 0, a1 = 0;
  const unsigned long long am0 = amask[0], am1 = amask[1], am2 = amask[2];   // selected by compares: an indexed array would live in scratch memory
  const int hi_ = R.hi, ai = R.ai, ak = R.ak, vr = R.vr, av = R.av, qv = R.qv, cq = R.cq;
  for (int i = 0; i < res; i++) {
    const double* bts = bt + i * 9 * GRAD_REC;
    const int base = i * 9, aw = base >> 6, ao = base & 63;
    unsigned long long av9 = (aw == 0 ? am0 : (aw == 1 ? am1 : am2)) >> ao;
    if (ao > 55 && aw < 2) av9 |= (aw == 0 ? am1 : am2) << (64 - ao);
    const unsigned bits0 = (unsigned)av9 & 0x1ffu;
    if (!bits0) continue;
    if (__ballot(hi_ < 0) == 0ull) {   // a wave of Hessian entries only
      double seg = 0;   // hessian += e2*d_x*d_x^T + e1*A^T*h_p*A (Gradient_admm.h:504,553), with d_x = w (x) dp
      for (unsigned bits = bits0; bits; bits &= bits - 1) {
        const int b = __ffs(bits) - 1;
        const double* t = bts + b * GRAD_REC;
        seg += (t[12 + ai] * t[12 + ak]) * t[cq];
      }
      a0 += seg;
    } else if (hi_ >= 0 || vr >= 0 || R.scal) {
      // A wave that holds more than one kind of entry (the one-group launch's third wave: 43 Hessian entries, the 18 gradient / time-column entries, the
      // time scalar; the folded launch's fourth B wave: the latter two) walks the records ONCE -- a wave runs its branches one after the other, and three
      // walks per segment made that wave the block's last by 4 us.  One body: a0 += (x0 * x1) * x2, a1 += x3 * x1, with 1.0 where a kind of entry has
      // no factor -- a multiplication by 1.0 changes no bit, and a1 of a Hessian entry is never read.
      const bool hs = hi_ >= 0;
      const int i0 = hs ? 12 + ai : (R.scal ? 18 : 6 + qv), i1 = hs ? 12 + ak : 12 + av, i2 = cq, i3 = R.scal ? 19 : 9 + qv;
      double sg = 0, spp = 0;
      for (unsigned bits = bits0; bits; bits &= bits - 1) {
        const int b = __ffs(bits) - 1;
        const double* t = bts + b * GRAD_REC;
        const double x1 = R.scal ? 1.0 : t[i1], x2 = hs ? t[i2] : 1.0;
        sg += (t[i0] * x1) * x2; spp += t[i3] * x1;
      }
      a0 += sg; a1 += spp;
    }
  }
  return GradAcc{a0, a1};
}

// One batch of segments [sb, se) of a piece: planes -> staging buffer, barrier derivatives, accumulation into this
// thread's Hessian / gradient entry.  The staging buffer is LDS (the common case) or the block's HBM scratch (a segment
// with more planes than the LDS buffer holds); one instantiation per address space, same summation order.
// GSYNC: barrier among the three waves that do the plane work -- the whole block in the one-group launch, an LDS-counter
// barrier in the folded launch, whose second wave group works on the velocity / acceleration terms at its own pace.
struct GradSync { int* cnt; int target; int nwaves = GRAD_THREADS / 64; };
template <bool GSYNC>
__device__ __forceinline__ void grad_sync(GradSync& g) { if constexpr (GSYNC) group_barrier(g.cnt, g.target, g.nwaves); else __syncthreads(); }
template <bool GSYNC>
__device__ __forceinline__ double grad_plane_batch(const Dev& D, double* pcb, double* E1b, double* E2b, int sb, int se, int tot, int u, int sp, int res, double m,
                                                 const double* Pall, const double* Ball, const double* wseg, const int* segn, const int* segno, int* sego, double* Mv, const GradRole R, GradSync& gs, double run,
                                                 const double* pst = nullptr) {   // run: this thread's running sum (Hessian or gradient entry) in, updated sum out; pst: the planes are in LDS already (folded launch)
  const int tid = R.tid, hi_ = R.hi, ai = R.ai, ak = R.ak, vr = R.vr, av = R.av, qv = R.qv;
    // offsets of the batch's segments: every wave writes the same values itself (lanes over segments), so only wave-local
    // ordering is needed -- no barrier, no serial loop on one thread
    if (sb > 0) grad_sync<GSYNC>(gs);   // a previous batch's readers are done with sego
    if ((tid & 63) >= sb && (tid & 63) < se) { int o = 0; for (int i = sb; i < (tid & 63); i++) o += segn[i]; sego[tid & 63] = o; }
    blk_sync<true>();
    // planes of the batch: obstacle list first, then inter-robot list, per segment
    for (int it = tid; it < 4 * tot; it += GRAD_THREADS) {
      const int w = it >> 2, c = it & 3;
      int i = sb; while (i + 1 < se && sego[i + 1] <= w) i++;
      const int tr = sp * res + i, k = w - sego[i], no = segno[i];
      if (pst) pcb[it] = k < no ? pst[((size_t)(2 * i) * GRAD_PST + k) * 4 + c] : pst[((size_t)(2 * i + 1) * GRAD_PST + (k - no)) * 4 + c];
      else pcb[it] = k < no ? D.oplanes[(((size_t)u * D.S + tr) * D.cap_obs + k) * 4 + c]
                           : D.splanes[(((size_t)u * D.S + tr) * D.cap_self + (k - no)) * 4 + c];
    }
    grad_sync<GSYNC>(gs);
    TJ_TIC(D, K_SEP_SELF_COMPACT, 4);
    // barrier derivatives for every (plane, control point) of the batch, stored [segment][j][k]
    for (int it = tid; it < 6 * tot; it += GRAD_THREADS) {
      int i = sb; while (i + 1 < se && 6 * sego[i + 1] <= it) i++;
      const int n = segn[i], loc = it - 6 * sego[i], j = loc / n, k = loc % n;
      const double* P = Pall + i * 18; const double* pl = pcb + 4 * (sego[i] + k);
      const double d = P[3 * j] * pl[0] + P[3 * j + 1] * pl[1] + P[3 * j + 2] * pl[2] + pl[3];
      double e1 = 0, e2 = 0;  // inactive terms contribute an exact +0
      if (d < m) barrier_d(wseg[i], d, m, e1, e2);
      E1b[it] = e1; E2b[it] = e2;
    }
    grad_sync<GSYNC>(gs);
    TJ_TIC(D, K_SEP_SELF_COMPACT, 5);
    // The planes of a segment meet a control point j of its hull only through M_j = sum_k e2[j][k] n_k n_k^T (3x3 symmetric) and
    // v_j = sum_k e1[j][k] n_k: Hessian entry ((a,q),(a',q')) = sum_j B[j][a] B[j][a'] M_j[q][q'], gradient entry (a,q) =
    // sum_j B[j][a] v_j[q] (Gradient_admm.h:331-407 written out).  54 sums over the planes per segment (6 control points x
    // {6 + 3}) by as many threads, then 6 terms per entry -- instead of every one of the 190 entries walking all 6 n products
    // itself, which was LDS-bandwidth bound and up to 15 us for a piece next to an obstacle.
    for (int idx = tid; idx < (se - sb) * 54; idx += GRAD_THREADS) {
      const int i = sb + idx / 54, r = idx % 54, j = r / 9, c = r % 9, n = segn[i];
      const double* pls = pcb + 4 * sego[i];
      double acc = 0;
      if (c < 6) {
        const int q = c < 1 ? 0 : (c < 3 ? 1 : 2), q2 = c - q * (q + 1) / 2;
        const double* e2s = E2b + 6 * sego[i] + j * n;
#pragma unroll 4
        for (int k = 0; k < n; k++) acc += (pls[4 * k + q] * pls[4 * k + q2]) * e2s[k];
      } else {
        const double* e1s = E1b + 6 * sego[i] + j * n;
#pragma unroll 4
        for (int k = 0; k < n; k++) acc += e1s[k] * pls[4 * k + (c - 6)];
      }
      Mv[i * 54 + r] = acc;
    }
    grad_sync<GSYNC>(gs);
    TJ_TIC(D, K_SEP_SELF_COMPACT, 6);
    for (int i = sb; i < se; i++) {  // per segment: accumulate from zero, then add (reference's += of local matrices)
      if (segn[i] == 0) continue;
      const double* Bs = Ball + i * 36; const double* Ms = Mv + i * 54;
      if (hi_ >= 0 || vr >= 0) {   // one body for both kinds of entry (the wave that holds the gradient entries holds Hessian entries too): b * 1.0 == b
        const int c1 = hi_ >= 0 ? ai : av, cm = hi_ >= 0 ? R.cq : 6 + qv;
        double seg = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) { const double b2 = hi_ >= 0 ? Bs[j * 6 + ak] : 1.0; seg += (Bs[j * 6 + c1] * b2) * Ms[j * 9 + cm]; }
        run += seg;
      }
    }
    return run;
}

// Launch order of k_grad's blocks (Dev::grad_bal; run by a few extra blocks of the NEXT iteration's k_front): item i = (robot, piece) has rank r = the number of
// items that cost more last time (ties: lower index first).  n items on C compute units, L = n - C of them late:
//   C < n < 2C   positions [0, L)  <- ranks [n - 2L, n - L)   (cheap: the CU mates of the late blocks)
//                positions [L, C)  <- ranks [0, n - 2L)       (the expensive ones, a CU each)
//                positions [C, n)  <- ranks [n - L, n)        (the cheapest: late, and ~20 % slower for it)
//   otherwise    position = rank (longest first; the host only switches the order on in the case above, TJ_GRAD_BALANCE=1 forces it)
constexpr int GRAD_ORDER_CHUNK = 2048;   // costs staged per pass (ints; the one-wave block's LDS buffer holds at least that many)
template <bool FA = false>   // FA (asynchronous front): the permutation goes out written through
__device__ __forceinline__ void grad_order_body(const Dev& D, int blk, int* cs) {
  const int n = (D.u1 - D.u0) * D.P, C = D.num_cu, lane = (int)(threadIdx.x & 63), i = blk * 64 + lane;
  const int ci = i < n ? D.grad_cost[i] : 0;
  int r = 0;
  for (int j0 = 0; j0 < n; j0 += GRAD_ORDER_CHUNK) {   // the costs pass through LDS (a loop over global words was 320 dependent round trips: 10 us)
    const int m = min(GRAD_ORDER_CHUNK, n - j0);
    blk_sync<true>();
    for (int j = lane; j < m; j += 64) cs[j] = D.grad_cost[j0 + j];
    blk_sync<true>();
#pragma unroll 8
    for (int j = 0; j < m; j++) { const int cj = cs[j]; r += (cj > ci || (cj == ci && j0 + j < i)) ? 1 : 0; }
  }
  if (i >= n) return;
  int pos = r;
  if (n > C && n < 2 * C) { const int L = n - C; pos = r < n - 2 * L ? L + r : (r < n - L ? r - (n - 2 * L) : r - (n - L) + C); }
  if constexpr (FA) xf_store_i(D.grad_perm + pos, i); else D.grad_perm[pos] = i;
}

// FOLD: the block first turns the stamped candidate / partner slots of ITS OWN segments into plane lists (the work of
// k_sep_self_compact, one wave per segment), so that kernel -- and its boundary -- drops out of the single-GPU chain.
// The folded launch also keeps a SECOND group of three waves (B) alive: the plane terms (group A) and the velocity /
// acceleration terms (group B) of a piece are independent sums over the same 190 entries, each a serial walk per thread, and
// they were two thirds of the kernel's critical path one after the other.  Each group synchronises within itself through an
// LDS counter (group_barrier), so neither waits for the other before the hand-over: B leaves its sums in LDS and retires.
// Both launch forms add (sum over plane segments) + (sum over velocity/acceleration segments), so they agree bit for bit.
template <bool FOLD>
__global__ __launch_bounds__(FOLD ? GRAD_FOLD_THREADS : GRAD_THREADS) void k_grad(Dev D) {
  const int item = D.grad_bal ? D.grad_perm[blockIdx.x] : (int)blockIdx.x;   // (robot, piece) of this launch position (grad_order_body); read BEFORE the stop
                                                                             // test so that the two scalar loads share one round trip
  if (D.xs_seq > 0 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(D.xs_go(), D.xs_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // opens the gate in front of k_xsolve on the other queue (also when the run has converged)
  if (TJ_DONE(D)) return;
  TJ_TIC_ENTRY(D, K_GRAD);
  if (D.keep_async) keep_wait(D);   // "optimal_plane":1: the stored planes' refinement runs on a queue of its own since the start of the iteration; the compaction below reads its planes
  const long long t_entry = wall_clock64();
  extern __shared__ double sm[];
  __shared__ int s_cnt[GRAD_MAXRES][2];   // folded launch: {obstacle planes, robot-pair planes} of the block's segments, left by its own compaction
  __shared__ int s_fits;                  // ... and whether every list fits the LDS hand-over (else the gradient reads the global lists, as the one-group launch does)
  const int npl = D.grad_npl;
  double* pst = sm + grad_lds_doubles(npl, D.res);   // [res][2][GRAD_PST][4] (folded launch only: its dynamic LDS is that much longer)
  double pall_v = 0, ball_v = 0, ball_v2 = 0;   // ball_v2: entries [512, res * 36) of the bases (res = 15, 16: 540 / 576 entries for 512 threads)
  static_assert(GRAD_MAXRES * 36 <= 2 * GRAD_FOLD_THREADS && GRAD_MAXRES * 18 <= GRAD_FOLD_THREADS, "folded k_grad stages the bases with two registers per thread, the hulls with one");
  if constexpr (FOLD) {
    const int u_ = D.u0 + item / D.P, sp_ = item % D.P;
    // hulls and bases of the piece's segments: issued first, so that their round trip runs under the compaction's
    if ((int)threadIdx.x < D.res * 18) pall_v = hull_entry(D, D.spline + (size_t)u_ * 3 * D.T, sp_ * D.res + threadIdx.x / 18, (threadIdx.x % 18) / 3, threadIdx.x % 3);
    if ((int)threadIdx.x < D.res * 36) ball_v = D.basis[(size_t)sp_ * D.res * 36 + threadIdx.x];
    if ((int)threadIdx.x + GRAD_FOLD_THREADS < D.res * 36) ball_v2 = D.basis[(size_t)sp_ * D.res * 36 + GRAD_FOLD_THREADS + threadIdx.x];
    if (threadIdx.x == 0) s_fits = 1;
    __syncthreads();
    for (int i = threadIdx.x >> 6; i < D.res; i += GRAD_FOLD_THREADS / 64)
      compact_segment(D, u_, sp_ * D.res + i, threadIdx.x & 63, pst + (size_t)(2 * i) * GRAD_PST * 4, pst + (size_t)(2 * i + 1) * GRAD_PST * 4, s_cnt[i], GRAD_PST, &s_fits);
  }
  double* Pall = sm;                          // [res][18] hulls of the piece's segments, row-major [6][3]
  double* Ball = Pall + D.res * 18;           // [res][36] their bases
  double* pc = Ball + D.res * 36;             // [npl][4] planes of the current batch of segments
  double* E1 = pc + 4 * npl;                  // [6 * planes] barrier derivatives (0 when inactive)
  double* E2 = E1 + 6 * npl;
  double* bt = E2 + 6 * npl;                  // [res][9] velocity / acceleration records x GRAD_REC
  double* Mv = bt + D.res * 9 * GRAD_REC;     // [res][6][9] per segment and control point: M (6) and v (3) of the plane terms
  double* H = Mv + D.res * 54;                // [361]
  double* g = H + 361;                        // [19]
  double* scr = g + 19;                       // [4*19] d,e,v,p
  int* segn = (int*)(scr + 4 * 19);           // [res] planes per segment, [res+1] offsets inside the batch
  int* sego = segn + GRAD_MAXRES;
  unsigned long long* amask = (unsigned long long*)(sego + GRAD_MAXRES);  // [3] active vel/acc records

  // Group B is FOUR waves: its first three hold the 171 Hessian entries, the fourth (wave 6 of the block) the 18 gradient / time-column entries and the time
  // scalar.  With those twenty threads in the third wave (as in group A) that wave walked the active records three times per segment -- once per kind of
  // entry, the branches of a wave run one after the other -- and arrived at the hand-over 4 us after the other five (measured per wave, round 4).
  const bool grpB = FOLD && threadIdx.x >= GRAD_THREADS;
  const int tid = grpB ? threadIdx.x - GRAD_THREADS : threadIdx.x;   // position inside the wave group
  const int rt = !grpB ? tid : (tid < 171 ? tid : (tid >= GRAD_THREADS && tid < GRAD_THREADS + 19 ? tid - GRAD_THREADS + 171 : 190));   // entry this thread accumulates (190: none)
  constexpr int NTH = FOLD ? 2 * GRAD_THREADS : GRAD_THREADS;
  const int u = D.u0 + item / D.P, sp = item % D.P;
  const double* net = D.spline + (size_t)u * 3 * D.T;
  const double m = D.margin, pt = D.piece_time[u];
  const int res = D.res;

  // role of this thread
  int hi_ = -1, hk_ = -1;  // Hessian entry (row >= col) for rt < 171
  if (rt < 171) { int i = 0; while ((i + 1) * (i + 2) / 2 <= rt) i++; hi_ = i; hk_ = rt - i * (i + 1) / 2; }
  const int vr = (rt >= 171 && rt < 189) ? rt - 171 : -1;  // gradient / time-column entry
  const bool scal = rt == 189;
  const int ai = hi_ >= 0 ? hi_ / 3 : 0, qi = hi_ >= 0 ? hi_ % 3 : 0, ak = hk_ >= 0 ? hk_ / 3 : 0, qk = hk_ >= 0 ? hk_ % 3 : 0;
  const int av = vr >= 0 ? vr / 3 : 0, qv = vr >= 0 ? vr % 3 : 0;
  double pacc_ = 0, vb0 = 0, vb1 = 0;   // plane terms of this thread's entry; velocity / acceleration terms

  TJ_TIC(D, K_GRAD, 0);
  __shared__ int s_gsync[2];   // arrival counters of the two wave groups' private barriers (folded launch)
  if (threadIdx.x < 2) s_gsync[threadIdx.x] = 0;
  __shared__ double s_wseg[GRAD_MAXRES];   // seg_weight of the piece's segments (two divisions and a modulo per use otherwise)
  if (threadIdx.x >= 64 && threadIdx.x < 64 + res) s_wseg[threadIdx.x - 64] = seg_weight(D, sp * res + threadIdx.x - 64);
  __shared__ int s_no[GRAD_MAXRES];   // obstacle planes per segment (the plane lists are read without a second trip for the count)
  bool staged = false;
  if constexpr (FOLD) {
    // ---- folded launch: hulls and bases arrived under the compaction; counts and planes come from it through LDS ----
    if ((int)threadIdx.x < res * 18) Pall[threadIdx.x] = pall_v;
    if ((int)threadIdx.x < res * 36) Ball[threadIdx.x] = ball_v;
    if ((int)threadIdx.x + GRAD_FOLD_THREADS < res * 36) Ball[GRAD_FOLD_THREADS + threadIdx.x] = ball_v2;
    __threadfence_block();
    __syncthreads();
    staged = s_fits != 0;
    if (threadIdx.x >= 2 * GRAD_THREADS + 64) return;   // wave 7: the remaining barriers count surviving waves only
    if ((int)threadIdx.x < res) {
      if (staged) { s_no[threadIdx.x] = s_cnt[threadIdx.x][0]; segn[threadIdx.x] = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1]; }
      else {
        const int no = D.ocount[u * D.S + sp * res + threadIdx.x];
        s_no[threadIdx.x] = no;
        segn[threadIdx.x] = no + (D.multi() ? D.scount[u * D.S + sp * res + threadIdx.x] : 0);
      }
    }
    __syncthreads();
  } else {
    // ---- stage every segment of the piece once: hull, basis, plane counts ----
    for (int idx = threadIdx.x; idx < res * 18; idx += NTH) Pall[idx] = hull_entry(D, net, sp * res + idx / 18, (idx % 18) / 3, idx % 3);
    for (int idx = threadIdx.x; idx < res * 36; idx += NTH) Ball[idx] = D.basis[(size_t)sp * res * 36 + idx];
    if (threadIdx.x < res) {
      const int no = D.ocount[u * D.S + sp * res + threadIdx.x];
      s_no[threadIdx.x] = no;
      segn[threadIdx.x] = no + (D.multi() ? D.scount[u * D.S + sp * res + threadIdx.x] : 0);
    }
    __syncthreads();
  }

  TJ_TIC(D, K_GRAD, 1);
  const int qhi = max(qi, qk), qlo = min(qi, qk);
  const GradRole role{tid, hi_, ai, qi, ak, qk, vr, av, qv, qhi * (qhi + 1) / 2 + qlo, scal};
  if (grpB) {
    // ---- group B (folded launch): velocity / acceleration records, then their accumulation, at its own pace ----
    GradSync gb{&s_gsync[1], 0, GRAD_THREADS / 64 + 1};
    TJ_TICB(D, K_SEP_SELF_COMPACT, 0);
    grad_velacc_records(D, tid, sp, res, m, pt, Pall, Ball, s_wseg, bt, amask);
    TJ_TICB(D, K_SEP_SELF_COMPACT, 1);
    grad_sync<true>(gb);
    TJ_TICB(D, K_SEP_SELF_COMPACT, 2);
    const GradAcc vb = grad_velacc_accumulate(role, res, bt, amask);
    if (rt < 190) H[rt] = vb.a0;
    if (rt >= 171 && rt < 190) g[rt - 171] = vb.a1;   // hand-over: H / g are not in use yet
    TJ_TICB(D, K_SEP_SELF_COMPACT, 3);
  } else {
    if (!FOLD) grad_velacc_records(D, tid, sp, res, m, pt, Pall, Ball, s_wseg, bt, amask);
    // ---- plane barrier terms (Gradient_admm.h:85-105, :331-407), segments in batches that fit the LDS plane buffer ----
    // The buffer holds `npl` planes (16 doubles each: plane, e1[6], e2[6]) -- sized for what segments really carry, not for the
    // configured capacity, so that several blocks share a CU when there are hundreds of robots.  A segment with more planes
    // than that (a robot inside a dense obstacle slab) is staged through a per-block HBM scratch buffer instead: same code,
    // same summation order, instantiated once per address space.
    GradSync ga{&s_gsync[0], 0};
    for (int sb = 0; sb < res;) {
      int se = sb, tot = 0;
      while (se < res && (se == sb || tot + segn[se] <= npl)) { tot += segn[se]; se++; }  // uniform: same LDS words for all threads
      if (tot > 0) {
        if (tot <= npl) pacc_ = grad_plane_batch<FOLD>(D, pc, E1, E2, sb, se, tot, u, sp, res, m, Pall, Ball, s_wseg, segn, s_no, sego, Mv, role, ga, pacc_, staged ? pst : nullptr);
        else { double* gs = D.grad_scr + (size_t)blockIdx.x * 16 * (size_t)(D.cap_obs + D.cap_self); pacc_ = grad_plane_batch<FOLD>(D, gs, gs + 4 * (size_t)tot, gs + 10 * (size_t)tot, sb, se, tot, u, sp, res, m, Pall, Ball, s_wseg, segn, s_no, sego, Mv, role, ga, pacc_); }
      }
      sb = se;
    }
    TJ_ORDER(pacc_);
    TJ_TIC(D, K_GRAD, 2);
    if (!FOLD) { __syncthreads(); const GradAcc vb = grad_velacc_accumulate(role, res, bt, amask); vb0 = vb.a0; vb1 = vb.a1; }
  }
#ifdef TJ_PHASE_TIMING
  if ((threadIdx.x & 63) == 0 && blockIdx.x < TJ_TIC_BLOCKS) D.dbg[((size_t)K_CCD_PREP * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + (threadIdx.x >> 6)] = wall_clock64();   // arrival of every wave at the hand-over
#endif
  if constexpr (FOLD) {
    __syncthreads();
    if (grpB) return;
    vb0 = H[tid]; if (tid >= 171 && tid < 190) vb1 = g[tid - 171];
  }
  __syncthreads();
  const double Hacc = pacc_ + vb0, gacc = Hacc;   // (sum over plane segments) + (sum over velocity / acceleration segments)
  const double pacc = vb1, gt = vb0, ht = vb1;

  TJ_TIC(D, K_GRAD, 3);
  // ---- scale by lambda, add consensus + dual terms (Gradient_admm.h:132-163) ----
  // The Hessian first: it needs nothing from the slack / dual blocks, and the LLT check and the eigenvalue wave wait for it.
  // The gradient's consensus and dual terms (two global round trips for z and Lambda) are formed by wave 2 -- which holds all
  // gradient entries -- WHILE waves 0 and 1 run the check and the eigenvalue.
  const double* C = D.convert + (size_t)sp * 36;
  const int P6 = 6 * D.P;
  double* delta = scr;          // [18] col-major 6x3: C x - z
  double* lamb = scr + 18;      // [18]
  if (hi_ >= 0) {
    double h = Hacc * D.lambda;
    if (qi == qk) {
      double mab = 0;
      for (int j = 0; j < 6; j++) mab += C[j * 6 + ai] * C[j * 6 + ak];
      h += D.mu * mab;
    }
    H[hi_ * 19 + hk_] = h; H[hk_ * 19 + hi_] = h;
  } else if (vr >= 0) {
    const double pc_ = pacc * D.lambda;
    H[vr * 19 + 18] = pc_; H[18 * 19 + vr] = pc_;
  } else if (scal) {
    H[18 * 19 + 18] = ht * D.lambda + D.mu;
  }

  TJ_TIC(D, K_GRAD, 4);
  // ---- PSD repair: only if LLT fails and lambda_min < 0 (Gradient_admm.h:38-53) ----
  // Wave 0 runs the LLT check (~4 us) while wave 1 already works on the smallest eigenvalue of the same block, both in
  // registers, one row per lane.  A successful check stops the eigenvalue wave at its next Householder step; a failed one
  // finds the eigenvalue 4 us further along than if it had been started afterwards -- and it is the repaired blocks
  // (a third of them in the early iterations) that set the kernel's duration.
  __shared__ int s_llt;      // -1 unknown, 0 failed, 1 passed
  __shared__ double s_ev;
  if (tid == 0) s_llt = -1;
  __syncthreads();
  if (tid >= 128) {   // wave 2: gradient entries 171..189 live here
    const int t2 = tid - 128;
    if (t2 < 18) {
      const int j = t2 % 6, a = t2 / 6;
      double acc = 0;
      for (int k = 0; k < 6; k++) acc += C[j * 6 + k] * net[sp * 3 + k + D.T * a];
      delta[j + 6 * a] = acc - D.p_slack[(size_t)u * 3 * P6 + sp * 6 + j + P6 * a];
      lamb[j + 6 * a] = D.p_lambda[(size_t)u * 3 * P6 + sp * 6 + j + P6 * a];
    }
    blk_sync<true>();
    if (vr >= 0) {
      double x1 = 0, x2 = 0;
      for (int j = 0; j < 6; j++) { x1 += C[j * 6 + av] * delta[j + 6 * qv]; x2 += C[j * 6 + av] * lamb[j + 6 * qv]; }
      g[vr] = gacc * D.lambda + (D.mu * x1 + x2);
    } else if (scal) {
      g[18] = gt * D.lambda + (D.mu * (pt - D.t_slack[u * D.P + sp]) + D.t_lambda[u * D.P + sp]);
    }
  }
  if (tid < 128) {
    __builtin_amdgcn_s_setprio(3);   // the check and the eigenvalue set the block's length: ahead of the other block's waves on the CU
    double r[19];
    const int row = min(tid & 63, 18);
#pragma unroll
    for (int c = 0; c < 19; c++) r[c] = H[row * 19 + c];
    if (tid < 64) {
      const bool llt_ok = chol_check_wave<19>(r);
      if (tid == 0) s_llt = llt_ok ? 1 : 0;   // (the statistic is taken at the very end: an atomic here would be waited for by the next barrier, on the repair path)
      TJ_TIC(D, K_GRAD, 7);
    } else {
      const double ev = min_eig_wave<19>(r, tid & 63, &s_llt);
      if (tid == 64) s_ev = ev;
    }
  }
  __syncthreads();
  if (s_llt == 0) {
    const double ev = s_ev;
    if (ev < 0 && tid < 19) H[tid * 19 + tid] = H[tid * 19 + tid] - ev * 1.0 + 0.01 * 1.0;
  }
  __syncthreads();
  TJ_TIC(D, K_GRAD, 5);
  double* og = D.lg + ((size_t)u * D.P + sp) * 19;
  double* oh = D.lh + ((size_t)u * D.P + sp) * 361;
  const bool wt = D.xs_async != 0;   // the solve is waiting on the other queue: the block goes out written through, then the robot's ticket
  if (tid < 19) xs_out(wt, og + tid, g[tid]);
  for (int idx = tid; idx < 361; idx += GRAD_THREADS) xs_out(wt, oh + idx, H[idx]);
  if (tid == 0 && s_llt == 0) D.blk_stats[(size_t)u * D.P + sp] += 1ull;   // PSD repairs of this piece: only this block writes the word
  if (tid == 0 && D.grad_bal) D.grad_cost[item] = (int)(wall_clock64() - t_entry);   // 10 ns ticks; read by the next iteration's k_front
  if (wt) {
    sig_acked();                     // every store of this wave has been acknowledged ...
    __syncthreads();                 // ... and of the other two
    asm volatile("" ::: "memory");
    if (tid == 0) __hip_atomic_fetch_add(D.xs_ticket(u), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  }
  TJ_TIC(D, K_GRAD, 6);
}

// ---- per-robot reduced Newton solve -------------------------------------------------------------
constexpr int XS_THREADS = 64;   // the factorisation runs in one wave per robot: its sync points are wave-local
constexpr int XS_LOAD_THREADS = 512;  // the whole block streams the piece blocks in and assembles; waves 1..7 then retire, or (Dev::fuse) wait for the solve and share the swept-hull tail
constexpr int XS_BAND = 17;      // pieces couple reduced coordinates at most 17 apart
constexpr int XS_HELP = XS_LOAD_THREADS - XS_THREADS;   // helper threads (waves 1..7)
constexpr int XS_KPT = 6;        // (segment, axis) interval pairs a helper thread carries across the solve (S <= 54)
// n = 9P-2.  Layout: H[n*n] L[n*n] g0[n] x0[n] scr[6n] lhu[P*361] lgu[P*19] | tail: net[3T] dir[3T].  The swept-hull tail
// (Dev::fuse) reuses the front of the buffer for its S x 54 hull values, so the front part is at least that large.
__host__ __device__ inline size_t xsolve_front_doubles(int n) {
  const size_t P = ((size_t)n + 2) / 9, base = 2 * (size_t)n * n + 8 * (size_t)n + 16 + P * 380;
  const size_t hulls = 8 * P * 54 * 2;   // res <= 16 segments per piece
  return base > hulls ? base : hulls;
}
__host__ __device__ inline size_t xsolve_lds_doubles(int n) { return xsolve_front_doubles(n) + 6 * (((size_t)n + 2) / 3 + 4) + 8; }

// Factor the reduced system held in LDS (L, row-major n x n) with the register-resident wave kernel when
// n = 9P-2 fits one row per lane (P <= 7); the factor's band + arrow row and the forward-substituted
// right-hand side go back to LDS for the back substitution.  handled = false: caller uses chol_arrow_lds.
template <int N, bool BATCH = true>
__device__ __forceinline__ bool xs_factor_regs_body(double* L, double* x0, int tid, int npiv) {
  double r[N];
  const int row = min(tid, N - 1);
#pragma unroll
  for (int j = 0; j < N; j++) r[j] = tid < N ? L[row * N + j] : 0.0;
  double y = tid < N ? x0[row] : 0.0;
  if (!chol_arrow_wave<N, XS_BAND, true, BATCH>(r, y, tid, npiv)) return false;
#pragma unroll
  for (int j = 0; j < N; j++)
    if (tid < N && j <= tid && (j + XS_BAND >= tid || tid == N - 1)) L[tid * N + j] = r[j];
  if (tid < N) x0[tid] = y;
  blk_sync<true>();
  return true;
}
// Out of line for the generic k_xsolve<0>, in the pairwise form: it stays within the caller-saved registers (the batched form, as
// a called function, opens with 41 scratch stores at 43 rows).  k_xsolve<N>, N <= 52, inlines the batched body instead.
template <int N>
__device__ __noinline__ bool xs_factor_regs_n(double* L, double* x0, int tid, int npiv) { return xs_factor_regs_body<N, false>(L, x0, tid, npiv); }
__device__ __forceinline__ bool xs_factor_regs(double* L, double* x0, int n, int tid, int npiv, bool& handled) {
  handled = n == 61;   // the smaller sizes have their own kernel instantiations (k_xsolve<N>, chosen by the host)
  if (handled) return xs_factor_regs_n<61>(L, x0, tid, npiv);
  return false;
}
// x = L^-T y for the arrowhead-band factor in LDS (row-major n x n), by ONE wave with y in registers (lane i = y_i, n <= 64).
// Same operations as the FAST chol_arrow_backsolve_lds (x_j = y_j * (1/l_jj), then y_i = fma(-x_j, l_ji, y_i)), but the row of L needed by the next
// step is fetched while the current division runs, x_j travels by v_readlane, and there is no LDS round trip or barrier on
// the dependent chain: ~0.1 us per unknown instead of ~0.19.
__device__ __forceinline__ double backsolve_wave(const double* L, int n, int bw, double y, int lane) {
  const int last = n - 1;
  {  // the arrow row is dense
    const double lrow = L[last * n + min(lane, last)];
    const double xl = readlane_f64(y, last) * readlane_f64(lrow, last);   // the diagonal holds 1 / l_jj (FAST factor)
    y = lane == last ? xl : (lane < last ? fma(-xl, lrow, y) : y);
  }
  double lrow = L[max(last - 1, 0) * n + min(lane, last)];
#pragma unroll 1
  for (int j = last - 1; j >= 0; j--) {
    const double nxt = L[max(j - 1, 0) * n + min(lane, last)];   // next step's row: independent of the chain
    const double xj = readlane_f64(y, j) * readlane_f64(lrow, j);
    y = lane == j ? xj : ((lane < j && lane + bw >= j) ? fma(-xj, lrow, y) : y);
    lrow = nxt;
  }
  return y;
}

// The same back substitution with the size known at compile time: lane indices of the broadcasts are immediates and the rows of L
// are fetched ahead of the chain by the scheduler (the generic loop above pays the SGPR-lane-select hazards 4 x per unknown).
template <int N>
__device__ __forceinline__ double backsolve_wave_body(const double* L, double y, int lane) {
  constexpr int last = N - 1;
  const int col = min(lane, last);
  {
    const double lrow = L[last * N + col];
    const double xl = readlane_f64(y, last) * readlane_f64(lrow, last);
    y = lane == last ? xl : (lane < last ? fma(-xl, lrow, y) : y);
  }
#pragma unroll
  for (int j = last - 1; j >= 0; j--) {
    const double lrow = L[j * N + col];
    const double xj = readlane_f64(y, j) * readlane_f64(lrow, j);
    y = lane == j ? xj : ((lane < j && lane + XS_BAND >= j) ? fma(-xj, lrow, y) : y);
  }
  return y;
}
template <int N>
__device__ __noinline__ double backsolve_wave_n(const double* L, double y, int lane) { return backsolve_wave_body<N>(L, y, lane); }
__device__ __forceinline__ double xs_backsolve(const double* L, int n, double y, int lane) {
  switch (n) {
    case 16: return backsolve_wave_n<16>(L, y, lane);
    case 25: return backsolve_wave_n<25>(L, y, lane);
    case 34: return backsolve_wave_n<34>(L, y, lane);
    case 43: return backsolve_wave_n<43>(L, y, lane);
    case 52: return backsolve_wave_n<52>(L, y, lane);
    case 61: return backsolve_wave_n<61>(L, y, lane);
  }
  return backsolve_wave(L, n, XS_BAND, y, lane);
}
// one factorisation attempt: registers when the size allows, LDS otherwise.  NREG > 0: the size is the kernel's template argument
template <int NREG>
__device__ __forceinline__ bool xs_factor(double* L, double* x0, int n, int tid, int npiv) {
  if constexpr (NREG > 0) return xs_factor_regs_body<NREG>(L, x0, tid, npiv);
  bool handled;
  const bool ok = xs_factor_regs(L, x0, n, tid, npiv, handled);
  if (handled) return ok;
  return chol_arrow_lds<true, true>(L, n, XS_BAND, tid, XS_THREADS, x0, npiv);
}

// wave 0 of k_xsolve: factor, solve, direction record (all sync points are wave-local)
template <int NREG>
__device__ __forceinline__ void xs_wave0(const Dev& D, int u, int tid, int n, int m, double* H, double* L, double* g0, double* x0, double* scr) {
  const int T = D.T;
  TJ_TIC(D, K_XSOLVE, 2);
  if (D.coupled()) {
    // Optimization3D_multi::update_spline (Optimization3D_multi.h:519-557): this robot's block of the
    // arrowhead system.  Eliminate the m control-point unknowns; what is left of the last row is the
    // robot's contribution to the shared-time corner (Schur complement) -- k_xsolve_c2 completes it.
    if (!xs_factor<NREG>(L, x0, n, tid, n - 1)) {
      if (tid == 0) { atomicAdd(&D.ctl->llt_fail_robot, 1ull); atomicOr(&D.ctl->error, ERR_NOT_SPD); }
    }
    blk_sync<true>();
    if (D.c2_fold) {
      // One context, every robot's block resident at once (round 5): the corner terms go out written through, the block counts itself in, waits until all uav_num are there
      // and finishes the arrowhead solve itself -- k_xsolve_c2's operations on the factor that is still in LDS (that launch re-read 15 KB per robot): the corner summed in
      // robot order, its pivot, the back substitution, the direction record.  Same operations in the same order, hence the same bits (TJ_C2_FOLD=0: the separate launch).
      double* oc = D.xcorner + (size_t)u * 4;
      if (tid == 0) { xf_store(oc, L[m * n + m]); xf_store(oc + 1, x0[m]); xf_store(oc + 2, g0[m]); xf_store(oc + 3, 0.0); }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0);
      asm volatile("" ::: "memory");
      if (tid == 0) __hip_atomic_fetch_add(&D.ctl->c2_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      {
        const long long t_end = wall_clock64() + 500000;   // 5 ms: a logic error must not hang the device
        while (__hip_atomic_load(&D.ctl->c2_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < D.U) {
          if (wall_clock64() > t_end) { if (tid == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PASS_TIMEOUT); break; }
          __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
      }
      double* s_cstage = scr; double* s_red = scr + 64;   // (the scratch of the solve, >= 96 doubles and idle here: static arrays would push the 88-row system over the LDS limit)
      {
        double acc = 0;
        for (int r0 = 0; r0 < D.U; r0 += 16) {   // sixteen robots per pass; the sums run in robot order whatever the chunking
          const int nr = min(16, D.U - r0);
          blk_sync<true>();
          for (int i = tid; i < 4 * nr; i += XS_THREADS) s_cstage[i] = xf_load(D.xcorner + (size_t)r0 * 4 + i);
          blk_sync<true>();
          if (tid < 3) for (int r = 0; r < nr; r++) acc += s_cstage[4 * r + tid];
        }
        blk_sync<true>();
        if (tid < 3) s_red[tid] = acc;
      }
      blk_sync<true>();
      const double corner = s_red[0], rhs = s_red[1];
      blk_sync<true>();
      if (!(corner > 0) && tid == 0 && blockIdx.x == 0) atomicOr(&D.ctl->error, ERR_NOT_SPD);
      const double lc = pivot_rsqrt(corner);
      double* y = x0;
      if (tid == 0) { L[m * n + m] = lc; y[m] = rhs * lc; }
      blk_sync<true>();
      if (n <= 64) {
        double yv = y[min(tid, n - 1)];
        yv = xs_backsolve(L, n, yv, tid);
        blk_sync<true>();
        if (tid < n) y[tid] = yv;
        blk_sync<true>();
      } else chol_arrow_backsolve_lds<true, true>(L, n, XS_BAND, y, tid, XS_THREADS);
      for (int i = tid; i < n; i += XS_THREADS) y[i] = -y[i];
      blk_sync<true>();
      for (int i = tid; i < m; i += XS_THREADS) { scr[i] = y[i] * g0[i]; scr[n + i] = g0[i] * g0[i]; }
      blk_sync<true>();
      const bool wtc = D.xs_async != 0;   // asynchronous solve: k_ccd's units (other queue) read the record while this launch is still running
      double* dirc = D.dirp(u);
      for (int idx = tid; idx < 3 * T; idx += XS_THREADS) {
        const int row = idx % T, a = idx / T;
        xs_out(wtc, dirc + idx, (row >= 2 && row < T - 2) ? y[3 * (row - 2) + a] : 0.0);
      }
      if (tid == 0) {
        xs_out(wtc, &D.wolfe(u), esum(scr, m));
        xs_out(wtc, &D.gn(u), esum(scr + n, m));
        xs_out(wtc, &D.tdir(u), y[m]);
        xs_out(wtc, D.xdir + (size_t)u * D.xs + 3 * T + 3, g0[m]);
      }
      return;
    }
    double* oL = D.xL + (size_t)u * n * n; double* oy = D.xy + (size_t)u * n; double* og = D.xg + (size_t)u * n;
    for (int idx = tid; idx < n * n; idx += XS_THREADS) oL[idx] = L[idx];
    for (int i = tid; i < n; i += XS_THREADS) { oy[i] = x0[i]; og[i] = g0[i]; }
    if (tid == 0) { double* oc = D.xcorner + (size_t)u * 4; oc[0] = L[m * n + m]; oc[1] = x0[m]; oc[2] = g0[m]; oc[3] = 0; }
    return;
  }
  if (!xs_factor<NREG>(L, x0, n, tid, n)) {  // forward substitution fused: x0 <- L^-1 g0
    if (tid == 0) atomicAdd(&D.ctl->llt_fail_robot, 1ull);
    blk_sync<true>();
    if (D.mode == 1) {  // multi: eigen-shift fallback (Optimization3D_multi.h:703-719); single has none
      for (int idx = tid; idx < n * n; idx += XS_THREADS) L[idx] = H[idx];
      blk_sync<true>();
      const double ev = min_eig_lds(L, n, scr, scr + n, scr + 2 * n, scr + 3 * n, tid, XS_THREADS);
      if (ev < 0) for (int i = tid; i < n; i += XS_THREADS) H[i * n + i] = H[i * n + i] - ev * 1.0 + 0.01 * 1.0;
      blk_sync<true>();
    }
    for (int idx = tid; idx < n * n; idx += XS_THREADS) L[idx] = H[idx];
    for (int i = tid; i < n; i += XS_THREADS) x0[i] = g0[i];
    blk_sync<true>();
    xs_factor<NREG>(L, x0, n, tid, n);  // like the reference, the second factorisation is not re-checked
    blk_sync<true>();
  }
  TJ_TIC(D, K_XSOLVE, 3);
  if (n <= 64) {
    double yv = x0[min(tid, n - 1)];
    if constexpr (NREG > 0) yv = backsolve_wave_body<NREG>(L, yv, tid); else yv = xs_backsolve(L, n, yv, tid);
    blk_sync<true>();
    if (tid < n) x0[tid] = yv;
    blk_sync<true>();
  } else chol_arrow_backsolve_lds<true, true>(L, n, XS_BAND, x0, tid, XS_THREADS);
  TJ_TIC(D, K_XSOLVE, 4);
  for (int i = tid; i < n; i += XS_THREADS) { x0[i] = -x0[i]; scr[i] = 0; }
  blk_sync<true>();
  const bool wt = D.xs_async != 0;
  double* dir = D.dirp(u);
  for (int idx = tid; idx < 3 * T; idx += XS_THREADS) {
    const int row = idx % T, a = idx / T;
    xs_out(wt, dir + idx, (row >= 2 && row < T - 2) ? x0[3 * (row - 2) + a] : 0.0);
  }
  if (wt) {
    // asynchronous solve: k_ccd's units of this robot wait for the DIRECTION only -- its flag goes up as soon as those stores are acknowledged; wolfe, |g| and the
    // time direction (read by k_ccd's finisher and by k_linesearch) follow and are counted (xs_done) -- k_ccd does not end before the count is full: its finisher
    // waits for it, and so does every robot's unit of segment 0 when its walk is over (a single UAV's k_ccd has no finisher)
    sig_acked();
    if (tid == 0) __hip_atomic_store(D.xs_flag(u), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  }
  for (int i = tid; i < n; i += XS_THREADS) { scr[i] = x0[i] * g0[i]; scr[n + i] = g0[i] * g0[i]; }
  blk_sync<true>();
  const double w_ = esum_wave(scr, n, tid), g_ = esum_wave(scr + n, n, tid);
  if (tid == 0) {
    xs_out(wt, &D.wolfe(u), -w_);
    xs_out(wt, &D.gn(u), sqrt(g_));
    xs_out(wt, &D.tdir(u), x0[m]);
    if (!D.multi()) xs_out(wt, &D.ctl->gnorm, sqrt(g_));   // single UAV (Optimization3D_admm.h:499): what k_ccd_self_seq would copy; the chain skips that launch
  }
  if (D.xch) {   // direct exchange (sharded contexts): the record goes straight into every peer's receive buffer, under the swept-hull tail of this block
    blk_sync<true>();
    for (int idx = tid; idx < 3 * T; idx += XS_THREADS) { const int row = idx % T, a = idx / T; scr[idx] = (row >= 2 && row < T - 2) ? x0[3 * (row - 2) + a] : 0.0; }
    if (tid == 0) { scr[3 * T] = x0[m]; scr[3 * T + 1] = -w_; scr[3 * T + 2] = sqrt(g_); }
    blk_sync<true>();
    xch_push_robot<true>(D, 1, u, D.xs, scr, 3 * T + 3, tid, XS_THREADS);
  }
}

// asynchronous solve: the gate in front of k_xsolve on the second queue (one wave, no LDS: it may sit there through the rest of the previous iteration)
__global__ __launch_bounds__(64) void k_xs_gate(Dev D, int seq, int fault = 0) {
  if (fault) { if (threadIdx.x == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); return; }   // test hook (TJ_XS_FAULT): as if the wait below had run out
  const int* w = D.xs_go();
  const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;   // 2 s
  wait_begin();
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq < 0) {
    if (wall_clock64() > t_end) { if (threadIdx.x == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); break; }
    __builtin_amdgcn_s_sleep(16);
  }
  wait_end();
}

// NREG = n when the system fits one row per lane (n = 9P-2 <= 61, P <= 7: the register factorisation inlined, size known at
// compile time), 0 otherwise (size chosen at run time, LDS forms beyond 61)
template <int NREG>
__global__ __launch_bounds__(XS_LOAD_THREADS) void k_xsolve(Dev D) {
  if (TJ_DONE(D)) return;
  TJ_TIC_ENTRY(D, K_XSOLVE);
  extern __shared__ double sm[];
  const int tid = threadIdx.x;
  const int u = D.u0 + blockIdx.x;
  const int T = D.T, m = 3 * (T - 4), n = NREG > 0 ? NREG : m + 1;  // n = 9P-2
  double* H = sm;            // [n*n] reduced Hessian (symmetric)
  double* L = H + n * n;     // [n*n] factor / eigen scratch
  double* g0 = L + n * n;    // [n]
  double* x0 = g0 + n;       // [n]
  double* scr = x0 + n;      // [6n]

  // overlap-add of piece blocks; global index of local (sp, a) is 9*sp + a, time is 3T.
  // Reduced index r = global - 6 for 6 <= global < 3T-6, time -> m.
  // Gather form of the overlap-add: every reduced entry sums the (at most two, or P for the
  // time-time entry) piece blocks that cover it, in piece order like the reference's += sequence.
  // No read-modify-write, so all global loads of a pass are in flight together.
  // piece blocks first go to LDS with one streaming copy (independent loads, many in flight); the
  // scatter below then never waits on HBM/L2
  TJ_TIC(D, K_XSOLVE, 0);
  __shared__ int s_grp;               // arrival counter of the helper waves' private barrier
  if (tid == 0) s_grp = 0;
  double* lhu = scr + 6 * n;          // [P*361]
  double* lgu = lhu + D.P * 361;      // [P*19]
  const bool wt = D.xs_async != 0;   // asynchronous solve (Dev::xs_async): this launch started next to k_grad, on the other queue
  // the overlap-add's sources (Dev::xs_gather): fetched BEFORE the wait for the tickets -- the index arithmetic they replace (covering pieces of a row and a column, local
  // offsets: ~100 instructions per entry) was 1.8 us between the last ticket and the factorisation
  constexpr int XS_EPT = NREG > 0 ? (NREG * NREG + XS_LOAD_THREADS - 1) / XS_LOAD_THREADS : 1;   // entries of the reduced system per thread
  int gs0[XS_EPT], gs1[XS_EPT], gv0 = -1, gv1 = -1;
  if constexpr (NREG > 0) {
#pragma unroll
    for (int k = 0; k < XS_EPT; k++) {
      const int idx = tid + k * XS_LOAD_THREADS;
      gs0[k] = idx < n * n ? D.xs_gather[2 * idx] : -1; gs1[k] = idx < n * n ? D.xs_gather[2 * idx + 1] : -1;
    }
    if (tid < n) { gv0 = D.xs_gather[2 * n * n + 2 * tid]; gv1 = D.xs_gather[2 * n * n + 2 * tid + 1]; }
  }
  if (wt) {
    // the robot's P piece blocks are in (each took a ticket after its write-through stores were acknowledged): wave 0 sleeps on the word, the others at the barrier
    if (tid < 64) xs_wait(D, D.xs_ticket(u), D.P);
    __syncthreads();
#ifndef TJ_PHASE_LIGHT
    TJ_TIC(D, K_XSOLVE, 7);
#endif
  }
  {
    const double* gh = D.lh + (size_t)u * D.P * 361;
    const double* gg = D.lg + (size_t)u * D.P * 19;
    if (wt) {
      for (int i = tid; i < D.P * 361; i += XS_LOAD_THREADS) lhu[i] = xf_load(gh + i);
      for (int i = tid; i < D.P * 19; i += XS_LOAD_THREADS) lgu[i] = xf_load(gg + i);
    } else {
      for (int i = tid; i < D.P * 361; i += XS_LOAD_THREADS) lhu[i] = gh[i];
      for (int i = tid; i < D.P * 19; i += XS_LOAD_THREADS) lgu[i] = gg[i];
    }
  }
  __syncthreads();
  TJ_TIC(D, K_XSOLVE, 1);
  if constexpr (NREG > 0) {   // the same sums in the same order (0 + first covering piece + second), sources from the table
#pragma unroll
    for (int k = 0; k < XS_EPT; k++) {
      const int idx = tid + k * XS_LOAD_THREADS;
      if (idx < n * n) {
        double acc = 0;
        if (gs0[k] == -2) { for (int sp = 0; sp < D.P; sp++) acc += lhu[(size_t)sp * 361 + 18 * 19 + 18]; }
        else { if (gs0[k] >= 0) acc += lhu[gs0[k]]; if (gs1[k] >= 0) acc += lhu[gs1[k]]; }
        H[idx] = acc; L[idx] = acc;
      }
    }
    if (tid < n) {
      double acc = 0;
      if (gv0 == -2) { for (int sp = 0; sp < D.P; sp++) acc += lgu[sp * 19 + 18]; }
      else { if (gv0 >= 0) acc += lgu[gv0]; if (gv1 >= 0) acc += lgu[gv1]; }
      g0[tid] = acc; x0[tid] = acc;
    }
  } else {
    int ra = tid / n, rb = tid % n;  // (row, column) of entry idx, advanced incrementally
    const int dra = XS_LOAD_THREADS / n, drb = XS_LOAD_THREADS % n;
    for (int idx = tid; idx < n * n; idx += XS_LOAD_THREADS) {
      const int ga = ra == m ? -1 : ra + 6, gb = rb == m ? -1 : rb + 6;  // -1 = time
      // pieces covering a coordinate g: 9sp <= g <= 9sp+17
      int lo = 0, hi = D.P - 1;
      if (ga >= 0) { lo = max(lo, (ga - 17 + 8) / 9); hi = min(hi, ga / 9); }
      if (gb >= 0) { lo = max(lo, (gb - 17 + 8) / 9); hi = min(hi, gb / 9); }
      double acc = 0;
      for (int sp = max(lo, 0); sp <= hi; sp++) {
        const int a = ga < 0 ? 18 : ga - 9 * sp, b = gb < 0 ? 18 : gb - 9 * sp;
        acc += lhu[(size_t)sp * 361 + a * 19 + b];
      }
      H[idx] = acc; L[idx] = acc;
      ra += dra; rb += drb;
      if (rb >= n) { rb -= n; ra++; }
    }
  for (int ra = tid; ra < n; ra += XS_LOAD_THREADS) {
    const int ga = ra == m ? -1 : ra + 6;
    int lo = 0, hi = D.P - 1;
    if (ga >= 0) { lo = max(lo, (ga - 17 + 8) / 9); hi = min(hi, ga / 9); }
    double acc = 0;
    for (int sp = max(lo, 0); sp <= hi; sp++) acc += lgu[sp * 19 + (ga < 0 ? 18 : ga - 9 * sp)];
    g0[ra] = acc; x0[ra] = acc;
  }
  }
  __syncthreads();
  // From here on wave 0 works alone (its sync points are wave-local: blk_sync<true>).  Without the swept-hull tail the other
  // waves retire (s_barrier counts surviving waves only); with it (Dev::fuse) they wait at the barrier below.
  // While wave 0 factors and solves (10-14 us), the seven helper waves already prepare the part of the swept-hull tail that does
  // not depend on the direction: the control net in LDS, the hull of every segment, and -- carried in registers across the
  // wait -- the k-DOP intervals of the current hull for the (segment, axis) pairs each helper thread will finish afterwards.
  // The helpers synchronise among themselves through an LDS counter (wave 0 is busy and must not be waited for).
  const bool tail = D.fuse != 0 && !wt;   // (asynchronous solve: k_ccd's units build the swept-hull records from the direction -- the serial tail here would sit on the chain)
  if (tid >= XS_THREADS && !tail) return;
  const int S = D.S;
  double* netl = sm + xsolve_front_doubles(n);   // [3T] control net, [3T] direction (rows 0,1,T-2,T-1 are zero)
  double* dl = netl + 3 * T;
  double* php = lhu;                             // [S][18] hull of the current net (the piece blocks are assembled: lhu is free)
  const double* gnet = D.spline + (size_t)u * 3 * T;
  const bool pre = tail && S * 49 <= XS_HELP * XS_KPT && (size_t)S * 54 <= 2 * (size_t)n * n + 8 * (size_t)n && S * 18 <= D.P * 361;   // uniform
  double klo[XS_KPT], kup[XS_KPT];
  const int ht = tid - XS_THREADS;
  if (tid < XS_THREADS) {
    __builtin_amdgcn_s_setprio(3); xs_wave0<NREG>(D, u, tid, n, m, H, L, g0, x0, scr); __builtin_amdgcn_s_setprio(0);   // the factorisation: ahead of the helper wave on its SIMD
    if (wt) {
      // the direction record is out (write-through) and acknowledged: the robot's flag and the count -- k_ccd's units on the other queue are waiting for them
      sig_acked();
      if (tid == 0) {
        __hip_atomic_store(D.xs_flag(u), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (decoupled / single: raised earlier, right behind the direction)
        __hip_atomic_fetch_add(D.xs_done(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sig_sent();
      TJ_TIC(D, K_XSOLVE, 6);
      return;
    }
  }
  else if (pre) {
    int target = 0;
    for (int idx = ht; idx < 3 * T; idx += XS_HELP) netl[idx] = gnet[idx];
    group_barrier(&s_grp, target, XS_HELP / 64);
    for (int idx = ht; idx < S * 18; idx += XS_HELP) {
      const int tr = idx / 18, e = idx % 18, j = e / 3, a = e % 3;
      const double* B = D.basis + (size_t)tr * 36 + j * 6;
      const int r0 = div_small(tr, D.res) * 3 + T * a;
      double p = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) p += B[k] * netl[r0 + k];
      php[idx] = p;
    }
    group_barrier(&s_grp, target, XS_HELP / 64);
#pragma unroll
    for (int q_ = 0; q_ < XS_KPT; q_++) {
      const int idx = ht + q_ * XS_HELP;
      double up = -INFINITY, lo = INFINITY;
      if (idx < S * 49) {
        const int tr = idx / 49, k = idx % 49;
        const double* q = php + tr * 18;
        const double x = D.kdop[3 * k], y = D.kdop[3 * k + 1], z = D.kdop[3 * k + 2];
        for (int i = 0; i < 6; i++) { const double lv = x * q[3 * i] + y * q[3 * i + 1] + z * q[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
      }
      klo[q_] = lo; kup[q_] = up;
    }
  }
  if (!tail) return;
  __syncthreads();
  TJ_TIC(D, K_XSOLVE, 5);
  // ---- swept-hull cache of this robot for the two CCD stages (what k_ccd_prep computes; same expressions, same bits) ----
  // BVH::CCDCollision box (BVH.cpp:195-250), SelfCCDCollision box (:289-330), k-DOP intervals of {P, P + D} (CCD.h:416-533).
  // Flat over the whole block: 54 hull values per segment into LDS (the front of the buffer is free now), then boxes and the
  // 49 x S interval pairs.  Folding this into the solve kernel removes one kernel boundary from the chain.
  {
    for (int idx = tid; idx < 3 * T; idx += XS_LOAD_THREADS) {
      const int row = idx % T, a = idx / T;
      if (!pre) netl[idx] = gnet[idx];
      dl[idx] = (row >= 2 && row < T - 2) ? x0[3 * (row - 2) + a] : 0.0;
    }
    __syncthreads();
    double* Ph = sm;   // [S][54]: P[18], Dh[18], PD[18]; PS = P + Dh is formed on the fly
    for (int idx = tid; idx < S * 18; idx += XS_LOAD_THREADS) {
      const int tr = idx / 18, e = idx % 18, j = e / 3, a = e % 3;
      const double* B = D.basis + (size_t)tr * 36 + j * 6;
      const int r0 = div_small(tr, D.res) * 3 + T * a;
      double p = 0, dh = 0, pd = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) { p += B[k] * netl[r0 + k]; dh += B[k] * dl[r0 + k]; pd += B[k] * (netl[r0 + k] + dl[r0 + k]); }
      Ph[tr * 54 + e] = p; Ph[tr * 54 + 18 + e] = dh; Ph[tr * 54 + 36 + e] = pd;
    }
    __syncthreads();
    for (int idx = tid; idx < S * 36; idx += XS_LOAD_THREADS) {   // P, Dh
      const int tr = idx / 36, e = idx % 36;
      xs_out(wt, D.ccdinfo + ((size_t)u * S + tr) * CCD_STRIDE + e, Ph[tr * 54 + e]);
    }
    for (int idx = tid; idx < S * 3; idx += XS_LOAD_THREADS) {    // obstacle box over {P, PD}, pair box over {P, P + Dh}
      const int tr = idx / 3, a = idx % 3;
      const double* q = Ph + tr * 54;
      double lo = INFINITY, hi = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
      for (int j = 0; j < 6; j++) {
        double v = q[3 * j + a]; if (v < lo) lo = v; if (v > hi) hi = v; if (v < lo2) lo2 = v; if (v > hi2) hi2 = v;
        v = q[36 + 3 * j + a]; if (v < lo) lo = v; if (v > hi) hi = v;
        v = q[3 * j + a] + q[18 + 3 * j + a]; if (v < lo2) lo2 = v; if (v > hi2) hi2 = v;
      }
      double* o = D.ccdinfo + ((size_t)u * S + tr) * CCD_STRIDE;
      xs_out(wt, o + 36 + a, lo); xs_out(wt, o + 39 + a, hi); xs_out(wt, o + 42 + a, lo2); xs_out(wt, o + 45 + a, hi2);
      xs_out(wt, D.cbox + ((size_t)tr * 6 + a) * D.U + u, lo2); xs_out(wt, D.cbox + ((size_t)tr * 6 + 3 + a) * D.U + u, hi2);
    }
    // 49-axis intervals of the swept hull at step 1: min / max over the 6 hull points and the 6 points P + Dh.  With the
    // hull's part already known (helpers), only the second half is left -- min and max do not depend on the order.
    if (pre) {
      if (tid >= XS_THREADS) {
#pragma unroll
        for (int q_ = 0; q_ < XS_KPT; q_++) {
          const int idx = ht + q_ * XS_HELP;
          if (idx < S * 49) {
            const int tr = idx / 49, k = idx % 49;
            const double* q = Ph + tr * 54;
            const double x = D.kdop[3 * k], y = D.kdop[3 * k + 1], z = D.kdop[3 * k + 2];
            double up = kup[q_], lo = klo[q_];
            for (int i = 0; i < 6; i++) { const double lv = x * (q[3 * i] + q[18 + 3 * i]) + y * (q[3 * i + 1] + q[18 + 3 * i + 1]) + z * (q[3 * i + 2] + q[18 + 3 * i + 2]); if (lv < lo) lo = lv; if (lv > up) up = lv; }
            double* o = D.ccdinfo + ((size_t)u * S + tr) * CCD_STRIDE;
            xs_out(wt, o + 48 + k, lo); xs_out(wt, o + 97 + k, up);
          }
        }
      }
    } else for (int idx = tid; idx < S * 49; idx += XS_LOAD_THREADS) {
      const int tr = idx / 49, k = idx % 49;
      const double* q = Ph + tr * 54;
      const double x = D.kdop[3 * k], y = D.kdop[3 * k + 1], z = D.kdop[3 * k + 2];
      double up = -INFINITY, lo = INFINITY;
      for (int i = 0; i < 6; i++) { const double lv = x * q[3 * i] + y * q[3 * i + 1] + z * q[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
      for (int i = 0; i < 6; i++) { const double lv = x * (q[3 * i] + q[18 + 3 * i]) + y * (q[3 * i + 1] + q[18 + 3 * i + 1]) + z * (q[3 * i + 2] + q[18 + 3 * i + 2]); if (lv < lo) lo = lv; if (lv > up) up = lv; }
      double* o = D.ccdinfo + ((size_t)u * S + tr) * CCD_STRIDE;
      xs_out(wt, o + 48 + k, lo); xs_out(wt, o + 97 + k, up);
    }
  }
#ifdef TJ_PHASE_LIGHT
  __builtin_amdgcn_s_waitcnt(0);   // the stamp after the block's stores have been acknowledged
  TJ_STAMP_(D, K_XSOLVE, 4, XS_LOAD_THREADS - 1);   // ... and the last helper wave's: its interval pairs end ~2 us after wave 0's part (the tail is bound by the
                                                    // CU's fp64 issue rate: 1 960 interval pairs x 6 points x ~14 instructions on seven waves -- measured, round 4)
#endif
  TJ_TIC(D, K_XSOLVE, 6);
}

// ---- long trajectories (Dev::xs_band: piece_num > 10, or TJ_XS_BAND=1 for testing): the same solve on band storage ----------
// n = 9P-2 grows past what a dense n x n copy in LDS allows (P = 11 already needs 2 x 74 KB), while the system itself is a band
// of half-width 17 plus one arrow row: n x 18 + n doubles (26 KB at P = 20).  One workgroup per robot; the band entries are
// gathered straight from the per-piece blocks in HBM (each is the sum of at most two of them, in piece order like the
// reference's += sequence), wave 0 factors and solves with the band routines of dev_linalg.h -- the same operation order as
// the dense kernels, hence the same bits (cross-checked at P <= 10 by tests with TJ_XS_BAND=1).  The PSD fallback of the
// multi-UAV mode (LLT fails: shift by the smallest eigenvalue, Optimization3D_multi.h:703-719) needs the dense matrix; it is
// formed in a per-robot HBM scratch (rare path).  Decoupled and single-UAV modes only.
constexpr int XB_THREADS = 256;
__host__ __device__ inline size_t xsolve_band_lds_doubles(int n) { return (size_t)(n - 1) * BAND_BS + 3 * (size_t)n + 64; }
__device__ __forceinline__ double xb_entry(const double* gh, int P, int ga, int gb) {   // assembled Hessian entry; -1 = time
  int lo = 0, hi = P - 1;
  if (ga >= 0) { lo = max(lo, (ga - 17 + 8) / 9); hi = min(hi, ga / 9); }
  if (gb >= 0) { lo = max(lo, (gb - 17 + 8) / 9); hi = min(hi, gb / 9); }
  double acc = 0;
  for (int sp = max(lo, 0); sp <= hi; sp++) acc += gh[(size_t)sp * 361 + (ga < 0 ? 18 : ga - 9 * sp) * 19 + (gb < 0 ? 18 : gb - 9 * sp)];
  return acc;
}
__global__ __launch_bounds__(XB_THREADS) void k_xsolve_band(Dev D) {
  if (TJ_DONE(D)) return;
  extern __shared__ double sm[];
  const int tid = threadIdx.x, u = D.u0 + blockIdx.x;
  const int T = D.T, m = 3 * (T - 4), n = m + 1, BS = BAND_BS;
  double* Bd = sm; double* Ar = Bd + (size_t)m * BS; double* g0 = Ar + n; double* y = g0 + n;
  const double* gh = D.lh + (size_t)u * D.P * 361;
  const double* gg = D.lg + (size_t)u * D.P * 19;
  auto assemble = [&]() {
    for (int idx = tid; idx < m * BS; idx += XB_THREADS) {
      const int i = idx / BS, c = idx % BS, j = i - (BS - 1) + c;
      Bd[idx] = j >= 0 ? xb_entry(gh, D.P, i + 6, j + 6) : 0.0;
    }
    for (int j = tid; j < n; j += XB_THREADS) {
      Ar[j] = xb_entry(gh, D.P, -1, j == m ? -1 : j + 6);
      const int ga = j == m ? -1 : j + 6;
      int lo = 0, hi = D.P - 1;
      if (ga >= 0) { lo = max(lo, (ga - 17 + 8) / 9); hi = min(hi, ga / 9); }
      double acc = 0;
      for (int sp = max(lo, 0); sp <= hi; sp++) acc += gg[sp * 19 + (ga < 0 ? 18 : ga - 9 * sp)];
      g0[j] = acc; y[j] = acc;
    }
  };
  assemble();
  __syncthreads();
  __shared__ int s_ok;
  if (D.coupled()) {
    // Optimization3D_multi::update_spline (Optimization3D_multi.h:519-557) on band storage: eliminate the m control-point unknowns;
    // the band factor, the arrow row with the robot's Schur-complement contribution left on its diagonal, the forward-substituted
    // right-hand side and the raw gradient go to HBM for k_xsolve_c2_band (what k_xsolve leaves as a dense block up to piece_num 10)
    if (tid < 64) { const bool ok = chol_band_lds(Bd, Ar, n, tid, y, false); if (tid == 0) s_ok = ok; }
    __syncthreads();
    if (!s_ok && tid == 0) { atomicAdd(&D.ctl->llt_fail_robot, 1ull); atomicOr(&D.ctl->error, ERR_NOT_SPD); }
    const size_t per = (size_t)m * BS + n;
    double* oL = D.xL + (size_t)u * per; double* oy = D.xy + (size_t)u * n; double* og = D.xg + (size_t)u * n;
    for (size_t idx = tid; idx < per; idx += XB_THREADS) oL[idx] = Bd[idx];   // Bd and Ar are contiguous in LDS
    for (int i = tid; i < n; i += XB_THREADS) { oy[i] = y[i]; og[i] = g0[i]; }
    if (tid == 0) { double* oc = D.xcorner + (size_t)u * 4; oc[0] = Ar[m]; oc[1] = y[m]; oc[2] = g0[m]; oc[3] = 0; }
    return;
  }
  if (tid < 64) { const bool ok = chol_band_lds(Bd, Ar, n, tid, y); if (tid == 0) s_ok = ok; }
  __syncthreads();
  if (!s_ok) {
    if (tid == 0) atomicAdd(&D.ctl->llt_fail_robot, 1ull);
    double shift = 0;
    if (D.mode == 1) {   // dense copy in HBM scratch -> smallest eigenvalue -> shift the diagonal (whole block cooperates)
      double* Hd = D.xs_scr + (size_t)blockIdx.x * ((size_t)n * n + 4 * n);
      double* sc = Hd + (size_t)n * n;
      for (int idx = tid; idx < n * n; idx += XB_THREADS) { const int i = idx / n, j = idx % n; Hd[idx] = xb_entry(gh, D.P, i == m ? -1 : i + 6, j == m ? -1 : j + 6); }
      __threadfence_block();
      __syncthreads();
      const double ev = min_eig_lds(Hd, n, sc, sc + n, sc + 2 * n, sc + 3 * n, tid, XB_THREADS);
      if (ev < 0) shift = -ev * 1.0 + 0.01 * 1.0;
    }
    __syncthreads();
    assemble();
    __syncthreads();
    if (shift != 0) {
      for (int i = tid; i < m; i += XB_THREADS) Bd[i * BS + BS - 1] = Bd[i * BS + BS - 1] + shift;   // h0 - ev*I + 0.01*I, entry by entry like the dense kernel
      if (tid == 0) Ar[m] = Ar[m] + shift;
    }
    __syncthreads();
    if (tid < 64) chol_band_lds(Bd, Ar, n, tid, y);   // like the reference, the second factorisation is not re-checked
    __syncthreads();
  }
  if (tid >= 64) return;
  chol_band_backsolve_lds(Bd, Ar, n, y, tid);
  double* scr = Bd;   // the factor is no longer needed: [n] x.g products, [n] g.g products
  for (int i = tid; i < n; i += 64) y[i] = -y[i];
  blk_sync<true>();
  for (int i = tid; i < n; i += 64) { scr[i] = y[i] * g0[i]; scr[n + i] = g0[i] * g0[i]; }
  blk_sync<true>();
  double* dir = D.dirp(u);
  for (int idx = tid; idx < 3 * T; idx += 64) {
    const int row = idx % T, a = idx / T;
    dir[idx] = (row >= 2 && row < T - 2) ? y[3 * (row - 2) + a] : 0.0;
  }
  if (tid == 0) { D.wolfe(u) = -esum(scr, n); const double gnv = sqrt(esum(scr + n, n)); D.gn(u) = gnv; D.tdir(u) = y[m]; if (!D.multi()) D.ctl->gnorm = gnv; }
  if (D.xch) {   // direct exchange (sharded contexts), as in k_xsolve
    double* rec = scr + 2 * n;
    for (int idx = tid; idx < 3 * T; idx += 64) { const int row = idx % T, a = idx / T; rec[idx] = (row >= 2 && row < T - 2) ? y[3 * (row - 2) + a] : 0.0; }
    if (tid == 0) { rec[3 * T] = y[m]; rec[3 * T + 1] = -esum(scr, n); rec[3 * T + 2] = sqrt(esum(scr + n, n)); }
    blk_sync<true>();
    xch_push_robot<true>(D, 1, u, D.xs, rec, 3 * T + 3, tid, 64);
  }
}

// Coupled mode, second half of the arrowhead solve: one wave per robot.  The Schur corner
// sum_u (h_t,u - y_u.y_u) and its right-hand side are summed in robot order (every block forms the
// same bits), the corner pivot is taken, and the existing arrow back-substitution finishes the
// robot's block: x_u = L_u^-T (w_u - y_u t).  Per robot it leaves direction, t_direction and the
// partial sums of wolfe = -x0.G and |G|^2 (completed by k_ccd_self_seq in robot order).
__global__ __launch_bounds__(XS_THREADS) void k_xsolve_c2(Dev D) {
  if (TJ_DONE(D)) return;
  extern __shared__ double sm[];
  const int tid = threadIdx.x, u = D.u0 + blockIdx.x;
  const int T = D.T, m = 3 * (T - 4), n = m + 1;
  double* L = sm; double* y = L + n * n; double* g0 = y + n; double* scr = g0 + n;  // scr [2n]
  __shared__ double s_red[3];
  const double* gL = D.xL + (size_t)u * n * n;
  for (int idx = tid; idx < n * n; idx += XS_THREADS) L[idx] = gL[idx];
  for (int i = tid; i < n; i += XS_THREADS) { y[i] = D.xy[(size_t)u * n + i]; g0[i] = D.xg[(size_t)u * n + i]; }
  {  // lane 0: corner, lane 1: rhs, lane 2: G_t -- sequential sums in robot order; the terms of 64 robots come in with one
     // coalesced pass and are added out of LDS (one lane walking global memory was 64 dependent round trips)
    __shared__ double s_cstage[256];
    double acc = 0;
    for (int r0 = 0; r0 < D.U; r0 += 64) {
      const int nr = min(64, D.U - r0);
      blk_sync<true>();
      for (int i = tid; i < 4 * nr; i += XS_THREADS) s_cstage[i] = D.xcorner[(size_t)r0 * 4 + i];
      blk_sync<true>();
      if (tid < 3) for (int r = 0; r < nr; r++) acc += s_cstage[4 * r + tid];
    }
    if (tid < 3) s_red[tid] = acc;
  }
  blk_sync<true>();
  const double corner = s_red[0], rhs = s_red[1];
  if (!(corner > 0) && tid == 0 && blockIdx.x == 0) atomicOr(&D.ctl->error, ERR_NOT_SPD);
  const double lc = pivot_rsqrt(corner);   // reciprocal root, like the rest of the FAST factor's diagonal
  if (tid == 0) { L[m * n + m] = lc; y[m] = rhs * lc; }
  blk_sync<true>();
  if (n <= 64) {   // the unrolled register form of k_xsolve (same operations as the LDS form)
    double yv = y[min(tid, n - 1)];
    yv = xs_backsolve(L, n, yv, tid);
    blk_sync<true>();
    if (tid < n) y[tid] = yv;
    blk_sync<true>();
  } else chol_arrow_backsolve_lds<true, true>(L, n, XS_BAND, y, tid, XS_THREADS);
  for (int i = tid; i < n; i += XS_THREADS) y[i] = -y[i];
  blk_sync<true>();
  for (int i = tid; i < m; i += XS_THREADS) { scr[i] = y[i] * g0[i]; scr[n + i] = g0[i] * g0[i]; }
  blk_sync<true>();
  double* dir = D.dirp(u);
  for (int idx = tid; idx < 3 * T; idx += XS_THREADS) {
    const int row = idx % T, a = idx / T;
    dir[idx] = (row >= 2 && row < T - 2) ? y[3 * (row - 2) + a] : 0.0;
  }
  if (tid == 0) {
    D.wolfe(u) = esum(scr, m);          // sum_j x_j g_j over this robot's control-point unknowns
    D.gn(u) = esum(scr + n, m);         // sum_j g_j^2
    D.tdir(u) = y[m];                   // shared t_direction (same bits on every robot)
    D.xdir[(size_t)u * D.xs + 3 * T + 3] = g0[m];  // this robot's share of G_t
  }
}

// k_xsolve_c2 for long trajectories (piece_num > 10): the same corner sum, corner pivot and back substitution on the band
// factor k_xsolve_band left (Bd[m][18] | Ar[n]), by one wave per robot
__global__ __launch_bounds__(XS_THREADS) void k_xsolve_c2_band(Dev D) {
  if (TJ_DONE(D)) return;
  extern __shared__ double sm[];
  const int tid = threadIdx.x, u = D.u0 + blockIdx.x;
  const int T = D.T, m = 3 * (T - 4), n = m + 1, BS = BAND_BS;
  const size_t per = (size_t)m * BS + n;
  double* Bd = sm; double* Ar = Bd + (size_t)m * BS; double* y = Ar + n; double* g0 = y + n; double* scr = g0 + n;  // scr [2n]
  __shared__ double s_red[3];
  const double* gL = D.xL + (size_t)u * per;
  for (size_t idx = tid; idx < per; idx += XS_THREADS) Bd[idx] = gL[idx];
  for (int i = tid; i < n; i += XS_THREADS) { y[i] = D.xy[(size_t)u * n + i]; g0[i] = D.xg[(size_t)u * n + i]; }
  {
    __shared__ double s_cstage[256];
    double acc = 0;
    for (int r0 = 0; r0 < D.U; r0 += 64) {
      const int nr = min(64, D.U - r0);
      blk_sync<true>();
      for (int i = tid; i < 4 * nr; i += XS_THREADS) s_cstage[i] = D.xcorner[(size_t)r0 * 4 + i];
      blk_sync<true>();
      if (tid < 3) for (int r = 0; r < nr; r++) acc += s_cstage[4 * r + tid];
    }
    if (tid < 3) s_red[tid] = acc;
  }
  blk_sync<true>();
  const double corner = s_red[0], rhs = s_red[1];
  if (!(corner > 0) && tid == 0 && blockIdx.x == 0) atomicOr(&D.ctl->error, ERR_NOT_SPD);
  const double lc = pivot_rsqrt(corner);
  if (tid == 0) { Ar[m] = lc; y[m] = rhs * lc; }
  blk_sync<true>();
  chol_band_backsolve_lds(Bd, Ar, n, y, tid);
  for (int i = tid; i < n; i += XS_THREADS) y[i] = -y[i];
  blk_sync<true>();
  for (int i = tid; i < m; i += XS_THREADS) { scr[i] = y[i] * g0[i]; scr[n + i] = g0[i] * g0[i]; }
  blk_sync<true>();
  double* dir = D.dirp(u);
  for (int idx = tid; idx < 3 * T; idx += XS_THREADS) {
    const int row = idx % T, a = idx / T;
    dir[idx] = (row >= 2 && row < T - 2) ? y[3 * (row - 2) + a] : 0.0;
  }
  if (tid == 0) {
    D.wolfe(u) = esum(scr, m);
    D.gn(u) = esum(scr + n, m);
    D.tdir(u) = y[m];
    D.xdir[(size_t)u * D.xs + 3 * T + 3] = g0[m];
  }
}

}  // namespace tj
