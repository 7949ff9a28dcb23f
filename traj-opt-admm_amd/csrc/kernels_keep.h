// kernels_keep.h -- "optimal_plane":1 : the planes that persist across iterations.
//
// Reference: single UAV, separate_plane (Optimization3D_admm.h:120-192) -- the first plane Separate::opengjk finds for a
// (segment, obstacle) is stored in is_seperate / seperate_c / seperate_d and from then on EVERY stored plane of the
// segment is refined by Optimal_plane::optimal_cd each iteration and used, whether or not the obstacle is still a
// candidate.  Multi UAV, separate_self (Optimization3D_multi.h:276-338) -- the same for (segment, robot pair) with
// Separate::selfgjk and Optimal_plane::self_optimal_cd; the obstacle planes of the multi-UAV paths have no such branch.
//
//   mode 0     one wavefront per (robot, segment): stamped candidate slots of this iteration that are not yet in the
//              segment's list are appended (slot order, deterministic), then one lane per stored plane refines it and
//              writes the segment's plane list (the reference emits in ascending obstacle id; the order only matters
//              at rounding level, like the candidate order of the non-persistent path).
//   modes 1,2  lanes stride over the list of switched-on (segment, p0, p1) slots: refine, store, and publish the two
//              half-offset planes in the dense partner table that k_sep_self_compact reads (ascending partner = the
//              reference's p0 < p1 lexicographic emission order).
// The dense S x N bool table of the reference becomes a per-segment list here (a few hundred entries at most).
#pragma once
#include "dev_common.h"
#include "dev_optplane.h"
#include "kernels_pairs.h"

namespace tj {

// part: 0 = everything (the one-queue chain, the stage API), 1 = new pairs only, 2 = the planes stored before this iteration only (asynchronous refinement, Dev::keep_async:
// launched on a queue of its own at the start of the iteration; its planes go out written through, every wave counts itself done)
__global__ __launch_bounds__(64) void k_keep_gate(Dev D, int seq) {
  const int* w = D.keep_go();
  const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;   // 2 s
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq < 0) {
    if (wall_clock64() > t_end) { if (threadIdx.x == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); break; }
    __builtin_amdgcn_s_sleep(16);
  }
}
__global__ __launch_bounds__(64) void k_keep(Dev D, int part) {
  const bool wt = part == 2;
  if (TJ_DONE(D)) return;
  const int lane = lane_id();
  const int epoch = D.ctl->epoch;
  if (!D.multi()) {
    const int u = D.u0 + blockIdx.x / D.S, tr = blockIdx.x % D.S;
    const size_t seg = (size_t)u * D.S + tr, base = seg * D.cap_obs;
    __shared__ double P[18];
    if (lane < 18) P[lane] = D.ohull[seg * 18 + lane];
    __syncthreads();
    const int n = D.ocand_n[seg], nk0 = D.kobs_n[seg];
    int nk = nk0;
    for (int s0 = 0; s0 < n; s0 += 64) {
      const int sl = s0 + lane;
      bool fresh = sl < n && D.ostamp[base + min(sl, n - 1)] == epoch;
      const int pt = D.ocand[base + min(sl, n - 1)];
      for (int i = 0; i < nk0; i++) fresh = fresh & (D.kobs_id[base + i] != pt);   // uniform (broadcast) loads
      const unsigned long long mask = ballot(fresh);
      const int idx = nk + prefix_count(mask);
      if (fresh) {
        if (idx < D.cap_obs) {
          D.kobs_id[base + idx] = pt;
          const double* p = D.oraw + (base + sl) * 4;
          double* o = D.kobs_cd + (base + idx) * 4;
          o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
        } else atomicOr(&D.ctl->error, ERR_PLANE_OVERFLOW);
      }
      nk += __popcll(mask);
    }
    nk = min(nk, D.cap_obs);
    __syncthreads();   // entries appended by other lanes are read below
    bool capped = false;
    for (int i = lane; i < nk; i += 64) {
      const int pt = D.kobs_id[base + i];
      double* k = D.kobs_cd + (base + i) * 4;
      double cx = k[0], cy = k[1], cz = k[2], d = k[3];
      capped |= !opt_plane_obstacle(P, D.px[pt], D.py[pt], D.pz[pt], D.margin, D.offset, cx, cy, cz, d);
      k[0] = cx; k[1] = cy; k[2] = cz; k[3] = d;
      double* o = D.oplanes + (base + i) * 4;
      if (!(isfinite(cx) && isfinite(cy) && isfinite(cz) && isfinite(d))) { o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 1e300; }   // inert, see the pair planes below
      else { o[0] = cx; o[1] = cy; o[2] = cz; o[3] = d; }
    }
    if (capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PLANE_REFINE);
    if (lane == 0) { D.kobs_n[seg] = nk; D.ocount[seg] = nk; D.seg_stats[seg * 6 + 4] += (unsigned long long)nk; }
    return;
  }
  // multi UAV.  kpair_n[0] = switched-on slots, kpair_n[1] = its value when this iteration began (k_begin): slots
  // [0, kpair_n[1]) are refined by part 2; a slot switched on in part 1 is refined right there by the lane that found it.
  const int U = D.U;
  const double m = D.margin, off = D.offset, dist = D.offset + 2 * D.margin;
  bool capped = false;
  auto publish = [&](size_t s0, int tr, int p0, int q, double cx, double cy, double cz, double d) {
    double* k = D.kpair_cd + 4 * s0;
    k[0] = cx; k[1] = cy; k[2] = cz; k[3] = d;
    const size_t s1 = ((size_t)tr * U + q) * U + p0;
    double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
    if (wt) {   // read by k_grad's compaction on another queue, possibly before this launch has ended: written through (same values as below)
      const bool fin = isfinite(cx) && isfinite(cy) && isfinite(cz) && isfinite(d);
      xf_store(q0, fin ? cx : 0.0); xf_store(q0 + 1, fin ? cy : 0.0); xf_store(q0 + 2, fin ? cz : 0.0); xf_store(q0 + 3, fin ? d - 0.5 * off : 1e300);
      xf_store(q1, fin ? -cx : 0.0); xf_store(q1 + 1, fin ? -cy : 0.0); xf_store(q1 + 2, fin ? -cz : 0.0); xf_store(q1 + 3, fin ? -d - 0.5 * off : 1e300);
      __hip_atomic_store(&D.pairstamp[s0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&D.pairstamp[s1], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      D.pair_mark(tr, p0, q); D.pair_mark(tr, q, p0);
      return;
    }
    if (!(isfinite(cx) && isfinite(cy) && isfinite(cz) && isfinite(d))) {
      // The reference's refinement can return NaN (its gradient takes log(dist/m) of a point on the wrong side).  Every consumer
      // of the reference tests `dist < margin` / `dist <= 0` first, which a NaN fails: such a plane is inert but stays in the
      // lists.  The kernels here multiply inactive terms by an exact 0 instead of branching, so the plane is published in an
      // inert finite form (c = 0, d = huge); the table keeps the NaN, from which the next refinement returns at once.
      q0[0] = 0; q0[1] = 0; q0[2] = 0; q0[3] = 1e300;
      q1[0] = 0; q1[1] = 0; q1[2] = 0; q1[3] = 1e300;
    } else {
    q0[0] = cx; q0[1] = cy; q0[2] = cz; q0[3] = d - 0.5 * off;
    q1[0] = -cx; q1[1] = -cy; q1[2] = -cz; q1[3] = -d - 0.5 * off;
    }
    D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch; D.pair_mark(tr, p0, q); D.pair_mark(tr, q, p0);
  };
  // statistics of this mode (tj_stats): newton_iters / pair_solves = Newton rounds / planes refined (Optimal_plane::self_optimal_cd's outer loop),
  // gjk_max_sum = sum over the iterations of the longest refinement of a launch, in rounds -- the unit count of k_keep's critical path
  auto note_rounds = [&](int tr, int p0, int rounds) {
    unsigned long long* ps = D.pair_stats + 2 * ((size_t)p0 * D.S + tr);
    atomicAdd(ps, (unsigned long long)rounds); atomicAdd(ps + 1, 1ull);
    if (rounds >= 6) atomicMax(&D.ctl->gjk_max, rounds);
  };
  __shared__ int wpre[513];
  __shared__ double s_te[64];   // opt_plane_pair_wave: the energy terms of five Armijo candidates
  const int nwork = part == 2 ? 0 : pair_work_prefix(D, wpre, lane), nold = D.kpair_n[1];   // (part 2 alone runs while k_front is still writing this iteration's work list: it does not look at it)
  // Few planes (up to a handful per wave of the grid): one WAVE per plane -- the 12 barrier terms of a Newton round on 12
  // lanes (opt_plane_pair_wave), a quarter of the dependent chain; the kernel is as long as its slowest plane.  Many planes
  // (hundreds of robots): one plane per LANE, the same arithmetic bit for bit, for throughput.  The switch is grid-uniform.
  auto count_done = [&]() {   // asynchronous refinement: this wave's planes are out (acknowledged) -> one of sixteen counters, fire and forget
    if (!wt) return;
    sig_acked();
    if (lane == 0) __hip_atomic_fetch_add(D.keep_sync + (blockIdx.x & 15) * 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  };
  if (nwork + nold <= 4 * (int)gridDim.x) {
    if (part != 2)
    for (int w = blockIdx.x; w < nwork; w += gridDim.x) {    // part 1 (Optimization3D_multi.h:276-290)
      const size_t sl = (size_t)pair_work_slot(D, wpre, w);
      const int tr = D.pair_work[3 * sl], p0 = D.pair_work[3 * sl + 1], q = D.pair_work[3 * sl + 2];
      const size_t s0 = ((size_t)tr * U + p0) * U + q;
      if (D.kpair_on[s0]) continue;
      const double* A = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
      const double* B = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
      double cx, cy, cz, d; bool cp;
      if (!plane_pair(A, B, dist, m, off, false, cx, cy, cz, d, cp)) continue;   // every lane, same arguments: uniform
      if (lane == 0) { D.kpair_on[s0] = 1; D.kpair_list[atomicAdd(D.kpair_n, 1)] = (int)s0; }
      int rounds = 0;
      capped |= !opt_plane_pair_wave(A, B, m, off, lane, cx, cy, cz, d, s_te, &rounds, D.dbg ? D.dbg + ((size_t)K_KEEP * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS : nullptr);
      if (lane == 0) { publish(s0, tr, p0, q, cx, cy, cz, d); note_rounds(tr, p0, rounds); }
    }
    if (part != 1)
    for (int w = blockIdx.x; w < nold; w += gridDim.x) {     // part 2 (:310-338)
      const size_t s0 = (size_t)D.kpair_list[w];
      const int tr = (int)(s0 / ((size_t)U * U)), p0 = (int)((s0 / U) % U), q = (int)(s0 % U);
      const double* A = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
      const double* B = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
      const double* k = D.kpair_cd + 4 * s0;
      double cx = k[0], cy = k[1], cz = k[2], d = k[3];
      int rounds = 0;
      capped |= !opt_plane_pair_wave(A, B, m, off, lane, cx, cy, cz, d, s_te, &rounds, D.dbg ? D.dbg + ((size_t)K_KEEP * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS : nullptr);
      if (lane == 0) { publish(s0, tr, p0, q, cx, cy, cz, d); note_rounds(tr, p0, rounds); }
    }
    if (capped && lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PLANE_REFINE);
    count_done();
    return;
  }
  // part 1 (Optimization3D_multi.h:276-290): pairs that passed box + k-DOP this iteration and have no plane yet
  if (part != 2)
  for (int w = blockIdx.x * 64 + lane; w < nwork; w += gridDim.x * 64) {
    const size_t sl = (size_t)pair_work_slot(D, wpre, w);
    const int tr = D.pair_work[3 * sl], p0 = D.pair_work[3 * sl + 1], q = D.pair_work[3 * sl + 2];
    const size_t s0 = ((size_t)tr * U + p0) * U + q;
    if (D.kpair_on[s0]) continue;
    const double* A = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
    const double* B = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
    double cx, cy, cz, d; bool cp;
    if (!plane_pair(A, B, dist, m, off, false, cx, cy, cz, d, cp)) continue;
    D.kpair_on[s0] = 1;
    D.kpair_list[atomicAdd(D.kpair_n, 1)] = (int)s0;
    capped |= !opt_plane_pair(A, B, m, off, cx, cy, cz, d);
    publish(s0, tr, p0, q, cx, cy, cz, d);
  }
  // part 2 (:310-338): every plane stored before this iteration
  if (part != 1)
  for (int w = blockIdx.x * 64 + lane; w < nold; w += gridDim.x * 64) {
    const size_t s0 = (size_t)D.kpair_list[w];
    const int tr = (int)(s0 / ((size_t)U * U)), p0 = (int)((s0 / U) % U), q = (int)(s0 % U);
    const double* A = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
    const double* B = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
    const double* k = D.kpair_cd + 4 * s0;
    double cx = k[0], cy = k[1], cz = k[2], d = k[3];
    capped |= !opt_plane_pair(A, B, m, off, cx, cy, cz, d);
    publish(s0, tr, p0, q, cx, cy, cz, d);
  }
  if (capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PLANE_REFINE);
  count_done();
}

}  // namespace tj
