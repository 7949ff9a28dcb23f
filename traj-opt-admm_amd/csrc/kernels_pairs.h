// kernels_pairs.h -- inter-robot separating planes.
//
// Replaces separate_self (Optimization3D_multi.h:237-342): BVH::SelfDCDCollision (BVH.cpp:252-287,
// a fresh dynamic AABB tree per segment), CCD::SelfKDOPDCD (CCD.h:535-587), Separate::selfgjk
// (Separate.h:165-304) and Optimal_plane::optimal_d (Optimal_plane.h:13-71).
//
//   k_hullinfo          per (robot, segment): hull, its AABB and its 49 k-DOP intervals, once.
//   k_sep_self_rows     one wavefront per (segment, lower robot p0): lanes over partners p1 > p0 for
//                       the box test; lanes over the 49 AXES for each box survivor (interval
//                       comparisons straight from the cache, no dot products, two coalesced loads
//                       per lane); pairs that pass go to a work list.
//   k_sep_self_solve    one wavefront per listed pair, solved cooperatively by its lanes (wave-cooperative
//                       GJK, the 12 barrier terms of the offset Newton on 12 lanes: plane_pair_wave).
//                       Each unordered pair is solved ONCE and
//                       the plane is stored for both robots, (c, d - off/2) and (-c, -d - off/2),
//                       in a dense [segment][robot][partner] slot table stamped with the current
//                       epoch (no clearing pass).
//   k_sep_self_compact  per (robot, segment): the obstacle planes from the stamped candidate slots (slot order), then
//                       the stamped partner slots in ascending partner order into the robot's plane lists --
//                       deterministic order, no atomics.
// The per-segment tree of the reference is replaced by the all-pairs box test (U <= a few hundred):
// the pair SET is what matters, and it is defined by the same inclusive overlap predicate
// (AABB.cc:131-148).
#pragma once
#include "dev_common.h"
#include "kernels_sep.h"

namespace tj {

constexpr int HULL_STRIDE = 18 + 6 + 98;  // P[6][3], lo[3], hi[3], kdop lo[49], kdop hi[49]

__global__ __launch_bounds__(64) void k_hullinfo(Dev D) {
  if (D.ctl->done) return;
  const int u = blockIdx.x / D.S, tr = blockIdx.x % D.S, lane = lane_id();
  __shared__ double P[18];
  if (lane < 18) P[lane] = hull_entry(D, D.spline + (size_t)u * 3 * D.T, tr, lane / 3, lane % 3);
  __syncthreads();
  double* o = D.hullinfo + ((size_t)u * D.S + tr) * HULL_STRIDE;
  if (lane < 18) o[lane] = P[lane];
  if (lane < 3) {
    double lo = INFINITY, hi = -INFINITY;
    for (int j = 0; j < 6; j++) { const double v = P[3 * j + lane]; if (v < lo) lo = v; if (v > hi) hi = v; }
    o[18 + lane] = lo; o[21 + lane] = hi;
    D.hbox[((size_t)tr * 6 + lane) * D.U + u] = lo; D.hbox[((size_t)tr * 6 + 3 + lane) * D.U + u] = hi;
  }
  if (lane < 49) {
    const double x = D.kdop[3 * lane], y = D.kdop[3 * lane + 1], z = D.kdop[3 * lane + 2];
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < 6; i++) { const double lv = x * P[3 * i] + y * P[3 * i + 1] + z * P[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    o[24 + lane] = lo; o[73 + lane] = up;
  }
}

// one (segment, lower robot p0, chunk of 64 partners p1 > p0) = one wavefront.  With hundreds of robots a whole row per wave
// left the rows of the middle segments (where everybody meets) as a long tail; only chunks that HAVE partners are launched:
// chunk c exists for p0 <= U - 2 - 64c, i.e. pair_units(U) = sum_c max(0, U - 1 - 64c) units per segment (636 instead of
// 1024 row-chunks at U = 256: workgroup dispatch is what bounds these kernels at that size).
__host__ __device__ inline int pair_units(int U) { int n = 0; for (int c = 0; U - 1 - 64 * c > 0; c++) n += U - 1 - 64 * c; return n; }
__host__ __device__ inline void pair_unit(int U, int bid, int& tr, int& p0, int& chunk) {
  const int per = pair_units(U);
  tr = bid / per;
  int r = bid % per;
  chunk = 0;
  while (r >= U - 1 - 64 * chunk) { r -= U - 1 - 64 * chunk; chunk++; }
  p0 = r;
}
__device__ __forceinline__ void sep_self_rows_body(const Dev& D, int bid) {
  int tr, p0, chunk;
  pair_unit(D.U, bid, tr, p0, chunk);
  const int lane = lane_id();
  const int U = D.U;
  __shared__ double A[HULL_STRIDE];   // hull, box and k-DOP intervals of robot p0
  __shared__ int todo[64];            // partner ids that passed box + k-DOP
  const double* a = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
  for (int i = lane; i < HULL_STRIDE; i += 64) A[i] = a[i];
  __syncthreads();
  const double dist = D.offset + 2 * D.margin, m = D.margin, off = D.offset;
  const int epoch = D.ctl->epoch;
  const bool own0 = p0 >= D.u0 && p0 < D.u1;
  {
    const int c0 = p0 + 1 + 64 * chunk;
    // 1. lanes over partners: AABB test (12 independent loads per lane)
    const int p1 = c0 + lane;
    bool hit = false;
    if (p1 < U && (own0 || (p1 >= D.u0 && p1 < D.u1))) {  // at least one robot of the pair belongs to this rank
      const double* b = D.hbox + (size_t)tr * 6 * U + p1;   // component k of partner p1: b[k * U], coalesced over lanes
      double bv[6];
#pragma unroll
      for (int k = 0; k < 6; k++) bv[k] = b[k * U];   // six independent loads, no branch between them
      hit = true;
#pragma unroll
      for (int k = 0; k < 3; k++) hit = hit & !((bv[3 + k] + dist < A[18 + k]) | (bv[k] > A[21 + k] + dist));
    }
    unsigned long long box = ballot(hit);
    // 2. per box survivor, wave-cooperative 49-axis interval test: lanes over AXES, two coalesced
    //    loads per lane instead of a 49-step dependent chain per pair
    int ntodo = 0;
    while (box) {
      const int l = __ffsll((long long)box) - 1;
      box &= box - 1;
      const int q = c0 + l;
      const double* b = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
      bool sep = false;
      if (lane < 49) { const double lob = b[24 + lane], hib = b[73 + lane]; sep = hib < A[24 + lane] - dist || A[73 + lane] < lob - dist; }
      if (ballot(sep) == 0ull) {
        if (lane == 0) todo[ntodo] = q;
        ntodo++;
      }
    }
    __syncthreads();
    // 3. hand the remaining pairs to k_sep_self_solve: one work item per pair.  Solving them here,
    //    one lane per pair, serialises up to 63 divergent GJK paths in one wave where many robots
    //    meet (measured: 87 us); one pair per wave keeps every path on its own program counter.
    if (ntodo > 0) {
      int base = 0;
      if (lane == 0) base = atomicAdd(D.pair_work_n, ntodo);
      base = __shfl(base, 0);
      if (lane < ntodo) {
        const int w = base + lane;
        if (w < D.cap_work) { D.pair_work[3 * w] = tr; D.pair_work[3 * w + 1] = p0; D.pair_work[3 * w + 2] = todo[lane]; }
        else atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void k_sep_self_rows(Dev D) {
  if (D.ctl->done) return;
  sep_self_rows_body(D, blockIdx.x);
}

// one wavefront per robot pair (stride over the work list: wave bid of nwaves), solved cooperatively by its lanes (plane_pair_wave)
__device__ __forceinline__ void sep_self_solve_body(const Dev& D, int bid, int nwaves) {
  const int lane = lane_id();
  __shared__ double A[18], B[18];
  const int n = min(*D.pair_work_n, D.cap_work), U = D.U;
  const double dist = D.offset + 2 * D.margin, m = D.margin, off = D.offset;
  const int epoch = D.ctl->epoch;
  // Long work list (hundreds of robots): a wave per pair is the lowest LATENCY but occupies 64 lanes for one chain of
  // dependent steps; with thousands of pairs THROUGHPUT decides, and one pair per LANE (per-lane GJK + Newton, the same
  // arithmetic: plane_pair == plane_pair_wave bit for bit) is ~20x cheaper per pair.  The switch is wave-uniform.
  if (n > 4 * nwaves) {
    unsigned long long nit_sum = 0, solved = 0; bool any_capped = false;
    for (int base = bid * 64; base < n; base += nwaves * 64) {
      const int w = base + lane;
      if (w < n) {
        const int tr = D.pair_work[3 * w], p0 = D.pair_work[3 * w + 1], q = D.pair_work[3 * w + 2];
        const double* Ag = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
        const double* Bg = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
        double e0, e1c, e2c, dpl; bool capped; int nit = 0;
        if (plane_pair(Ag, Bg, dist, m, off, true, e0, e1c, e2c, dpl, capped, &nit)) {
          nit_sum += (unsigned long long)nit; solved++; any_capped = any_capped || capped;
          const size_t s0 = ((size_t)tr * U + p0) * U + q, s1 = ((size_t)tr * U + q) * U + p0;
          double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
          q0[0] = e0; q0[1] = e1c; q0[2] = e2c; q0[3] = dpl - 0.5 * off;
          q1[0] = -e0; q1[1] = -e1c; q1[2] = -e2c; q1[3] = -dpl - 0.5 * off;
          D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch;
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) { nit_sum += __shfl_xor(nit_sum, o); solved += __shfl_xor(solved, o); }
    if (lane == 0 && solved) { atomicAdd(&D.ctl->newton_iters, nit_sum); atomicAdd(&D.ctl->pair_solves, solved); }
    if (any_capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
    return;
  }
  // The first `nwaves` items are taken statically (no atomic on the common path: SCN-C has fewer pairs than waves); a
  // wave that finishes early then draws further items from a shared cursor, so a long list (hundreds of robots) is
  // balanced dynamically instead of striding -- solve times vary 1 : 30.
  for (int w = bid; w < n;) {
    const bool first = w == bid;   // phase stamps (timing build only) describe a wave's first work item
    if (first) TJ_TIC(D, K_SEP_SELF_SOLVE, 0);
    const int tr = D.pair_work[3 * w], p0 = D.pair_work[3 * w + 1], q = D.pair_work[3 * w + 2];
    __syncthreads();
    if (lane < 18) { A[lane] = D.hullinfo[((size_t)p0 * D.S + tr) * HULL_STRIDE + lane]; B[lane] = D.hullinfo[((size_t)q * D.S + tr) * HULL_STRIDE + lane]; }
    __syncthreads();
    if (first) TJ_TIC(D, K_SEP_SELF_SOLVE, 1);
    {
      double e0, e1c, e2c, dpl; bool capped; int nit = 0, gk = 0;
      const bool okp = plane_pair_wave(A, B, dist, m, off, lane, e0, e1c, e2c, dpl, capped, &nit, &gk);  // whole wave, uniform result
#ifdef TJ_PHASE_TIMING
      if (lane == 0 && first && blockIdx.x < TJ_TIC_BLOCKS) { D.dbg[((size_t)K_SEP_SELF_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 6] = gk; D.dbg[((size_t)K_SEP_SELF_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 7] = okp ? nit : -1; }
#endif
      if (okp && lane == 0) {
        atomicAdd(&D.ctl->newton_iters, (unsigned long long)nit); atomicAdd(&D.ctl->pair_solves, 1ull);
        if (capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
        const size_t s0 = ((size_t)tr * U + p0) * U + q, s1 = ((size_t)tr * U + q) * U + p0;
        double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
        q0[0] = e0; q0[1] = e1c; q0[2] = e2c; q0[3] = dpl - 0.5 * off;
        q1[0] = -e0; q1[1] = -e1c; q1[2] = -e2c; q1[3] = -dpl - 0.5 * off;
        D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch;
      }
    }
    if (first) TJ_TIC(D, K_SEP_SELF_SOLVE, 2);
    int nxt = 0;
    if (lane == 0) nxt = nwaves + atomicAdd(D.pair_work_n + 1, 1);
    w = __shfl(nxt, 0);
  }
}
__global__ __launch_bounds__(64) void k_sep_self_solve(Dev D) {
  if (D.ctl->done) return;
  sep_self_solve_body(D, blockIdx.x, gridDim.x);
}

// per (owned robot, segment): obstacle planes from the stamped candidate slots (slot order), then -- multi-robot modes --
// the robot-pair planes from the stamped partner slots (ascending partner): deterministic lists, no atomics
__device__ __forceinline__ void compact_segment(const Dev& D, int u, int tr, int lane) {   // one wave
  const int U = D.U, epoch = D.ctl->epoch;
  const size_t seg = (size_t)u * D.S + tr;
  if (!(D.optimal_plane && !D.multi())) {  // single-UAV "optimal_plane":1 -- k_keep wrote the obstacle plane list itself
    const int n = D.ocand_n[seg];
    double* out = D.oplanes + seg * D.cap_obs * 4;
    int base = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
      const int sl = s0 + lane;
      const bool ok = sl < n && D.ostamp[seg * D.cap_obs + min(sl, n - 1)] == epoch;
      const unsigned long long mask = ballot(ok);
      const int idx = base + prefix_count(mask);
      if (ok) { const double* p = D.oraw + (seg * D.cap_obs + sl) * 4; out[4 * idx] = p[0]; out[4 * idx + 1] = p[1]; out[4 * idx + 2] = p[2]; out[4 * idx + 3] = p[3]; }
      base += __popcll(mask);
    }
    if (lane == 0) { D.ocount[seg] = base; D.seg_stats[seg * 6 + 4] += (unsigned long long)base; }
  }
  if (!D.multi()) return;
  double* out = D.splanes + seg * D.cap_self * 4;
  int base = 0;
  for (int q0 = 0; q0 < U; q0 += 64) {
    const int q = q0 + lane;
    const size_t slot = ((size_t)tr * U + u) * U + min(q, U - 1);
    const bool ok = q < U && q != u && D.pairstamp[slot] == epoch;
    const unsigned long long mask = ballot(ok);
    const int idx = base + prefix_count(mask);
    if (ok) {
      if (idx < D.cap_self) { const double* p = D.pairplane + 4 * slot; out[4 * idx] = p[0]; out[4 * idx + 1] = p[1]; out[4 * idx + 2] = p[2]; out[4 * idx + 3] = p[3]; }
      else atomicOr(&D.ctl->error, ERR_PLANE_OVERFLOW);
    }
    base += __popcll(mask);
  }
  if (lane == 0) {
    D.scount[seg] = min(base, D.cap_self);
    D.seg_stats[seg * 6 + 5] += (unsigned long long)base;
  }
}
__global__ __launch_bounds__(64) void k_sep_self_compact(Dev D) {
  if (D.ctl->done) return;
  compact_segment(D, D.u0 + blockIdx.x / D.S, blockIdx.x % D.S, lane_id());
}

}  // namespace tj
