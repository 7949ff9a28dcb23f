// kernels_pairs.h -- inter-robot separating planes.
//
// Replaces separate_self (Optimization3D_multi.h:237-342): BVH::SelfDCDCollision (BVH.cpp:252-287,
// a fresh dynamic AABB tree per segment), CCD::SelfKDOPDCD (CCD.h:535-587), Separate::selfgjk
// (Separate.h:165-304) and Optimal_plane::optimal_d (Optimal_plane.h:13-71).
//
//   k_hullinfo          per (robot, segment): hull, its AABB and its 49 k-DOP intervals, once.
//   k_sep_self_rows     one wavefront per (segment, lower robot p0), lanes over partners p1 > p0:
//                       box test and 49 interval comparisons straight from the cache (no dot
//                       products), then -- for the few survivors only -- hull-hull GJK and the
//                       Newton refinement of the offset.  Each unordered pair is solved ONCE and
//                       the plane is stored for both robots, (c, d - off/2) and (-c, -d - off/2),
//                       in a dense [segment][robot][partner] slot table stamped with the current
//                       epoch (no clearing pass).
//   k_sep_self_compact  per (robot, segment): gathers the stamped slots in ascending partner order
//                       into the robot's plane list -- deterministic order, no atomics.
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
  }
  if (lane < 49) {
    const double x = D.kdop[3 * lane], y = D.kdop[3 * lane + 1], z = D.kdop[3 * lane + 2];
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < 6; i++) { const double lv = x * P[3 * i] + y * P[3 * i + 1] + z * P[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    o[24 + lane] = lo; o[73 + lane] = up;
  }
}

__global__ __launch_bounds__(64) void k_sep_self_rows(Dev D) {
  if (D.ctl->done) return;
  const int tr = blockIdx.x / D.U, p0 = blockIdx.x % D.U, lane = lane_id();
  const int U = D.U;
  __shared__ double A[18];
  __shared__ double oth[64 * 18];
  const double* a = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
  if (lane < 18) A[lane] = a[lane];
  __syncthreads();
  const double dist = D.offset + 2 * D.margin, m = D.margin, off = D.offset;
  const int epoch = D.ctl->epoch;
  const bool own0 = p0 >= D.u0 && p0 < D.u1;
  for (int c0 = p0 + 1; c0 < U; c0 += 64) {
    const int p1 = c0 + lane;
    if (p1 >= U) continue;
    if (!own0 && !(p1 >= D.u0 && p1 < D.u1)) continue;  // neither robot belongs to this rank
    const double* b = D.hullinfo + ((size_t)p1 * D.S + tr) * HULL_STRIDE;
    bool hit = true;
    for (int k = 0; k < 3; k++) hit = hit && !(b[21 + k] + dist < a[18 + k] || b[18 + k] > a[21 + k] + dist);
    if (!hit) continue;
    bool pass = true;
    for (int k = 0; k < 49 && pass; k++) if (b[73 + k] < a[24 + k] - dist || a[73 + k] < b[24 + k] - dist) pass = false;
    if (!pass) continue;
    double* Bq = oth + lane * 18;
    for (int i = 0; i < 18; i++) Bq[i] = b[i];
    double e0, e1c, e2c, dpl; bool capped;
    if (!plane_pair(A, Bq, dist, m, off, true, e0, e1c, e2c, dpl, capped)) continue;
    if (capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
    const size_t s0 = ((size_t)tr * U + p0) * U + p1, s1 = ((size_t)tr * U + p1) * U + p0;
    double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
    q0[0] = e0; q0[1] = e1c; q0[2] = e2c; q0[3] = dpl - 0.5 * off;
    q1[0] = -e0; q1[1] = -e1c; q1[2] = -e2c; q1[3] = -dpl - 0.5 * off;
    D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch;
  }
}

__global__ __launch_bounds__(64) void k_sep_self_compact(Dev D) {
  if (D.ctl->done) return;
  const int u = D.u0 + blockIdx.x / D.S, tr = blockIdx.x % D.S, lane = lane_id();
  const int U = D.U, epoch = D.ctl->epoch;
  double* out = D.splanes + ((size_t)u * D.S + tr) * D.cap_self * 4;
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
    D.scount[u * D.S + tr] = min(base, D.cap_self);
    D.seg_stats[((size_t)u * D.S + tr) * 6 + 5] += (unsigned long long)base;
  }
}

}  // namespace tj
