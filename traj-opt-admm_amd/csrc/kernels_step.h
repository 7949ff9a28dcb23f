// kernels_step.h -- CCD step clamps, the slack (z) + dual update, iteration bookkeeping, and the union kernels of the
// single-GPU iteration graph.  (The Armijo line search on the x-objective lives in kernels_ls.h.)
//
//   k_ccd_prep      per (robot, segment): hull P, direction hull D, query boxes and the 49-axis
//                   intervals of the swept hull at step 1, cached for the two CCD stages
//                   (BVH::CCDCollision BVH.cpp:195-250, SelfCCDCollision :289-330, CCD::KDOPCCD
//                   CCD.h:416-473, SelfKDOPCCD :475-533)
//   ccd_obs_body    Step::position_step (Step.h:21-110): per candidate cloud point the smallest
//                   exponent k with conv{P, P+0.8^k D} farther than `offset`; atomicMax per robot.
//                   The reference's running-step loop yields max_k over candidates because swept
//                   hulls are nested in k, so the result is order independent.
//   ccd_self_pairs_body / k_ccd_self_seq
//                   Step::self_step (Step.h:184-256).  Phase A (parallel, per segment) keeps the robot pairs whose
//                   swept boxes and k-DOPs overlap and whose swept hulls are within `offset` at FULL step -- no
//                   other pair can ever act, hulls being nested in the step.  Phase B (one wave, sequential)
//                   replays the reference's ORDER DEPENDENT joint back-off over those (rare) pairs, segment by
//                   segment; it also forms gnorm.
//   slack_body      update_slack_lambda (Optimization3D_multi.h:344-506): per piece Newton step on
//                   (z, t_z) with Armijo, then the dual ascent on (Lambda, tau).
//   k_front / k_mid / k_ccd   union kernels: independent one-wavefront stages sharing a launch (see below).
#pragma once
#include "dev_common.h"
#include "dev_linalg.h"
#include "kernels_sep.h"
#include "kernels_ls.h"
#include "kernels_pairs.h"
#include "dev_dyntree.h"

namespace tj {

// one (robot, segment) of the swept-hull cache, computed by ONE wave; net / dir are T x 3 column-major control nets
// (global or LDS), sh is 72 doubles of wave-private LDS
// XF (foreign-robot unit of a sharded context, xf_ccd_body): the record is read by the pair tiles of the SAME launch -- write-through stores; the direction
// comes from the receive buffer of the direct exchange when `sysdir` (system-scope loads).  Same expressions in every form, hence the same bits.
template <bool XF = false>
__device__ __forceinline__ void ccd_prep_segment(const Dev& D, const double* net, const double* dir, int u, int tr, int lane, double* sh, int sysdir = 0) {   // sysdir: 1 = system-scope loads of the
  // direction (receive buffer of the direct exchange), 2 = agent-scope loads (asynchronous solve: the records of two robots share cache lines, and a neighbour's line may have been fetched before its owner wrote it)
  double* P = sh; double* Dh = sh + 18; double* PD = sh + 36; double* PS = sh + 54;
  auto ldd = [&](const double* p) { return (XF && sysdir == 1) ? xch_load(p) : ((XF && sysdir == 2) ? xf_load(p) : *p); };
  auto st = [&](double* p, double v) { if constexpr (XF) xf_store(p, v); else *p = v; };
  if (lane < 18) P[lane] = hull_entry(D, net, tr, lane / 3, lane % 3);
  else if (lane < 36) {
    const int j = (lane - 18) / 3, a = (lane - 18) % 3;
    const double* B = D.basis + (size_t)tr * 36 + j * 6;
    const double* col = dir + div_small(tr, D.res) * 3 + D.T * a;
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) acc += B[k] * ldd(col + k);   // = hull_entry(D, dir, ...)
    Dh[lane - 18] = acc;
  }
  else if (lane < 54) {  // basis * (bz + bz_d), the box the reference uses for the cloud query
    const int j = (lane - 36) / 3, a = (lane - 36) % 3;
    const double* B = D.basis + (size_t)tr * 36 + j * 6;
    const int r0 = div_small(tr, D.res) * 3 + D.T * a;
    double acc = 0;
    for (int k = 0; k < 6; k++) acc += B[k] * (net[r0 + k] + ldd(dir + r0 + k));
    PD[lane - 36] = acc;
  }
  blk_sync<true>();
  if (lane < 18) PS[lane] = P[lane] + Dh[lane];  // (P + D) used by the pair box and by the k-DOP at step 1
  blk_sync<true>();
  double* o = D.ccdinfo + ((size_t)u * D.S + tr) * CCD_STRIDE;
  if (lane < 18) { st(o + lane, P[lane]); st(o + 18 + lane, Dh[lane]); }
  if (lane < 3) {
    double lo = INFINITY, hi = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
    for (int j = 0; j < 6; j++) {
      double v = P[3 * j + lane]; if (v < lo) lo = v; if (v > hi) hi = v; if (v < lo2) lo2 = v; if (v > hi2) hi2 = v;
      v = PD[3 * j + lane]; if (v < lo) lo = v; if (v > hi) hi = v;
      v = PS[3 * j + lane]; if (v < lo2) lo2 = v; if (v > hi2) hi2 = v;
    }
    st(o + 36 + lane, lo); st(o + 39 + lane, hi); st(o + 42 + lane, lo2); st(o + 45 + lane, hi2);
    st(D.cbox + ((size_t)tr * 6 + lane) * D.U + u, lo2); st(D.cbox + ((size_t)tr * 6 + 3 + lane) * D.U + u, hi2);
  }
  if (lane < 49) {
    const double x = D.kdop[3 * lane], y = D.kdop[3 * lane + 1], z = D.kdop[3 * lane + 2];
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < 6; i++) { const double lv = x * P[3 * i] + y * P[3 * i + 1] + z * P[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    for (int i = 0; i < 6; i++) { const double lv = x * PS[3 * i] + y * PS[3 * i + 1] + z * PS[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    st(o + 48 + lane, lo); st(o + 97 + lane, up);
  }
  blk_sync<true>();  // sh is reused by the caller's next segment
}

// Asynchronous Newton solve (Dev::xs_async): the obstacle unit of (robot, segment) in k_ccd builds the record itself -- and it was launched before the robot's solve has
// finished.  What does not depend on the direction (the segment's basis rows, the control points) is fetched BEFORE the wait for the robot's flag; afterwards one
// round trip for the direction entries (agent-scope loads: the records of two robots share cache lines), the sums of ccd_prep_segment in its order (same bits), the
// record into `info` (LDS: the unit's own walk reads it there, no read-back) and out to the cache written through for the pair tiles of the launch.  sh: 72 doubles.
__device__ __forceinline__ void ccd_record_async(const Dev& D, int u, int tr, int lane, double* info, double* sh) {
  double* P = sh; double* Dh = sh + 18; double* PD = sh + 36; double* PS = sh + 54;
  const double* net = D.spline + (size_t)u * 3 * D.T; const double* dir = D.dirp(u);
  double bk[6] = {0, 0, 0, 0, 0, 0}, nk[6] = {0, 0, 0, 0, 0, 0}, dk[6] = {0, 0, 0, 0, 0, 0};
  const int e = lane % 18, r0 = div_small(tr, D.res) * 3 + D.T * (e % 3);
  if (lane < 54) {
    const double* B = D.basis + (size_t)tr * 36 + (e / 3) * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) { bk[k] = B[k]; nk[k] = net[r0 + k]; }
  }
  const double kx = D.kdop[3 * min(lane, 48)], ky = D.kdop[3 * min(lane, 48) + 1], kz = D.kdop[3 * min(lane, 48) + 2];
  xs_wait<4>(D, D.xs_flag(u), 1);
  TJ_TIC(D, K_CCD, 3);
  if (lane >= 18 && lane < 54) {
#pragma unroll
    for (int k = 0; k < 6; k++) dk[k] = xf_load(dir + r0 + k);
  }
  {
    double acc = 0;
    if (lane < 18) {
#pragma unroll
      for (int k = 0; k < 6; k++) acc += bk[k] * nk[k];
      P[lane] = acc;
    } else if (lane < 36) {
#pragma unroll
      for (int k = 0; k < 6; k++) acc += bk[k] * dk[k];
      Dh[lane - 18] = acc;
    } else if (lane < 54) {
#pragma unroll
      for (int k = 0; k < 6; k++) acc += bk[k] * (nk[k] + dk[k]);
      PD[lane - 36] = acc;
    }
  }
  blk_sync<true>();
  if (lane < 18) { PS[lane] = P[lane] + Dh[lane]; info[lane] = P[lane]; info[18 + lane] = Dh[lane]; }
  blk_sync<true>();
  if (lane < 3) {
    double lo = INFINITY, hi = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
    for (int j = 0; j < 6; j++) {
      double v = P[3 * j + lane]; if (v < lo) lo = v; if (v > hi) hi = v; if (v < lo2) lo2 = v; if (v > hi2) hi2 = v;
      v = PD[3 * j + lane]; if (v < lo) lo = v; if (v > hi) hi = v;
      v = PS[3 * j + lane]; if (v < lo2) lo2 = v; if (v > hi2) hi2 = v;
    }
    info[36 + lane] = lo; info[39 + lane] = hi; info[42 + lane] = lo2; info[45 + lane] = hi2;
    xf_store(D.cbox + ((size_t)tr * 6 + lane) * D.U + u, lo2); xf_store(D.cbox + ((size_t)tr * 6 + 3 + lane) * D.U + u, hi2);
  }
  if (lane < 49) {
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < 6; i++) { const double lv = kx * P[3 * i] + ky * P[3 * i + 1] + kz * P[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    for (int i = 0; i < 6; i++) { const double lv = kx * PS[3 * i] + ky * PS[3 * i + 1] + kz * PS[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    info[48 + lane] = lo; info[97 + lane] = up;
  }
  blk_sync<true>();
  double* o = D.ccdinfo + ((size_t)u * D.S + tr) * CCD_STRIDE;
  for (int i = lane; i < CCD_REC; i += 64) xf_store(o + i, info[i]);
}

// u_first: first robot of the launch (all robots from 0; a sharded context whose foreign robots are handled inside k_ccd: its own from u0)
__global__ __launch_bounds__(64) void k_ccd_prep(Dev D, int u_first) {
  if (TJ_DONE(D)) return;
  const int u = u_first + blockIdx.x / D.S, tr = blockIdx.x % D.S;
  __shared__ double sh[18 * 3 + 18];
  ccd_prep_segment(D, D.spline + (size_t)u * 3 * D.T, D.dirp(u), u, tr, lane_id(), sh);
}

// ---- sharded contexts: the caches of the robots other ranks own ---------------------------------------------------------------------------------
// One wave per (foreign robot, segment), at the head of k_front (hull cache, from the owner's control points) and of k_ccd (swept-hull cache, from its
// direction record): the expressions of k_hullinfo / ccd_prep_segment -- which are those of ls_publish_hullinfo / k_xsolve's tail on the owner's side --
// so every rank holds the same bits for every robot.  Direct exchange (Dev::xch): the unit first waits for the owner's push (xch_wait_owner), reads the
// slice from the receive buffer, and the unit of segment 0 puts it in place (Dev::spline / Dev::xdir) for the kernels that follow.
__device__ __forceinline__ void xf_hull_body(const Dev& D, int f) {
  const int u = D.xf_robot(f / D.S), tr = f % D.S, lane = lane_id(), T = D.T;
  __shared__ double P[18];
  const double* net = D.spline + (size_t)u * 3 * T;
  if (D.xch) {
    if (D.xch_poll) xch_wait_owner(D, 0, D.owner_of(u));   // (after a timeout the error bit fails the batch; the unit still reports, so that no tile waits in vain)
    const double* rxn = D.rx[0] + (size_t)u * 3 * T;
    if (tr == 0) for (int i = lane; i < 3 * T; i += 64) D.spline[(size_t)u * 3 * T + i] = xch_load(rxn + i);   // read by later kernels only (k_ccd's units)
    if (lane < 18) {
      const int j = lane / 3, a = lane % 3;
      const double* B = D.basis + (size_t)tr * 36 + j * 6;
      const double* col = rxn + div_small(tr, D.res) * 3 + T * a;
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) acc += B[k] * xch_load(col + k);   // = hull_entry
      P[lane] = acc;
    }
  } else if (lane < 18) P[lane] = hull_entry(D, net, tr, lane / 3, lane % 3);
  blk_sync<true>();
  double* o = D.hullinfo + ((size_t)u * D.S + tr) * HULL_STRIDE;
  if (lane < 18) xf_store(o + lane, P[lane]);
  if (lane < 3) {
    double lo = INFINITY, hi = -INFINITY;
    for (int j = 0; j < 6; j++) { const double v = P[3 * j + lane]; if (v < lo) lo = v; if (v > hi) hi = v; }
    xf_store(o + 18 + lane, lo); xf_store(o + 21 + lane, hi);
    xf_store(D.hbox + ((size_t)tr * 6 + lane) * D.U + u, lo); xf_store(D.hbox + ((size_t)tr * 6 + 3 + lane) * D.U + u, hi);
  }
  if (lane < 49) {
    const double x = D.kdop[3 * lane], y = D.kdop[3 * lane + 1], z = D.kdop[3 * lane + 2];
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < 6; i++) { const double lv = x * P[3 * i] + y * P[3 * i + 1] + z * P[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    xf_store(o + 24 + lane, lo); xf_store(o + 73 + lane, up);
  }
  xf_signal(D, 0, tr);
}
__device__ __forceinline__ void xf_ccd_body(const Dev& D, int f, double* sh) {
  const int u = D.xf_robot(f / D.S), tr = f % D.S, lane = lane_id(), T = D.T;
  const double* dir = D.dirp(u);
  if (D.xch) {
    if (D.xch_poll) xch_wait_owner(D, 1, D.owner_of(u));
    dir = D.rx[1] + (size_t)u * D.xs;
    if (tr == 0) for (int i = lane; i < 3 * T + 3; i += 64) xf_store(D.xdir + (size_t)u * D.xs + i, xch_load(dir + i));   // |g| is read by this launch's finisher, the rest by k_linesearch
  }
  ccd_prep_segment<true>(D, D.spline + (size_t)u * 3 * T, dir, u, tr, lane, sh, D.xch != 0 ? 1 : 0);
  xf_signal(D, 1, tr);
}
// ranks that share a device (a test arrangement): the units' polling would hold LDS and wave slots the peer's producing kernel needs -- one
// one-wave launch in front of k_front / k_ccd waits for all peers instead (Dev::xch_poll = 0)
__global__ __launch_bounds__(64) void k_xch_wait(Dev D, int kind) {
  // lane r watches rank r's counter; the counter, the stop flag and this rank's own push count are fetched in ONE round trip (three dependent ones made this launch 5 - 6 us long)
  const int lane = lane_id();
  const bool mine = lane < D.world && lane != D.rank;
  const unsigned long long* w = D.xcnt + kind * XCH_MAX + (mine ? lane : 0);
  unsigned long long got = mine ? __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
  const int done = D.ctl->done, pushed = D.ctl->xpush[kind];
  if (done) return;
  const unsigned long long need = mine ? (unsigned long long)(pushed / (D.u1 - D.u0)) * (unsigned long long)D.owned_by(lane) : 0ull;
  if (ballot(got < need) == 0ull) return;
  const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;
  for (;;) {
    __builtin_amdgcn_s_sleep(4);
    if (mine) got = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (ballot(got < need) == 0ull) return;
    if (wall_clock64() > t_end) { if (lane == 0) atomicOr(&D.ctl->error, ERR_PEER_TIMEOUT); return; }
  }
}

constexpr int CCD_LDS_DOUBLES = 294 + (2 * FRONT_CAP + 128) / 2;   // info[146] kax[147] | fa fb cand
// the candidate a lane holds, handed to the whole wave
__device__ __forceinline__ V3 bcast_v3(const V3& v, int l) { return V3{readlane_f64(v.x, l), readlane_f64(v.y, l), readlane_f64(v.z, l)}; }
__device__ __forceinline__ BodyPoint bcast_body(const BodyPoint& b, int l) { return BodyPoint{bcast_v3(b.q, l)}; }
__device__ __forceinline__ BodyTri bcast_body(const BodyTri& b, int l) { return BodyTri{bcast_v3(b.a, l), bcast_v3(b.b, l), bcast_v3(b.c, l)}; }
// LEAN (k_ccd_lean, the build for scenes whose swept boxes meet next to no obstacle): the candidates that pass the k-DOP cull are taken ONE AFTER THE OTHER by the whole
// wave (wave-cooperative GJK, the form the robot-pair replay uses: same witness vectors bit for bit) instead of one per lane.  The per-lane swept-hull GJK needs ~240
// registers; compiled for three waves per SIMD (168) it spilled 216 of them -- 416 bytes of scratch per lane, 2.3 MB of spill traffic per launch on the headline scene,
// whose walks almost never reach it (rounds 2 - 4).  The cooperative form fits the budget without a spill; it is slower per candidate, which is why the host only
// picks this build while the candidates are few (tj_api.hip: choose_builds).
template <int PRIM, bool LEAN = false>
__device__ __forceinline__ void ccd_obs_body(const Dev& D, int bid, double* lds, bool publish = false) {
  const int u = D.u0 + bid / D.S, tr = bid % D.S;
  const int lane = lane_id();
  double* info = lds;
  int* fa = (int*)(lds + 294); int* fb = fa + FRONT_CAP; int* cand = fb + FRONT_CAP;
  bool kax_ready = false;
  V3 axv{0, 0, 0}; double lo_ax = 0, hi_ax = 0;
  const double* src = D.ccdinfo + ((size_t)u * D.S + tr) * CCD_STRIDE;
  const TopBox topb = bvh_top_box(D);   // travels with the segment's record
  if (D.xs_async) {   // asynchronous Newton solve: the record is built here, once this robot's block (on the other queue) has left its direction
    ccd_record_async(D, u, tr, lane, info, lds + 148);
    TJ_TIC(D, K_CCD, 4);
    xf_signal(D, 1, tr);   // (its write-through stores acknowledged) -> the pair tiles of the segment
    TJ_TIC(D, K_CCD, 5);
  } else
  if (publish) {   // coupled chain (Dev::xf_all), asynchronous solve: this unit builds its (robot, segment)'s swept-hull record itself -- what k_ccd_prep would --, written through for the pair tiles
    ccd_prep_segment<true>(D, D.spline + (size_t)u * 3 * D.T, D.dirp(u), u, tr, lane, lds);
    xf_signal(D, 1, tr);
    for (int i = lane; i < CCD_REC; i += 64) info[i] = xf_load(src + i);   // (its own write-through stores, acknowledged: read back past the L1)
  } else
  for (int i = lane; i < CCD_REC; i += 64) info[i] = src[i];
  __syncthreads();
  QBox q;
  for (int k = 0; k < 3; k++) { q.lo[k] = info[36 + k]; q.hi[k] = info[39 + k]; }
  const double off = D.offset;
  unsigned long long visits = 0;
  int kmax = 0;
  const int found = bvh_query<1, PRIM>(D, q, off, fa, fb, cand, &visits, [&](int pt) {
    if (!kax_ready) {   // wave-uniform: this lane's axis and the swept hull's interval on it
      const int ax = min(lane, 48);
      axv = V3{D.kdop[3 * ax], D.kdop[3 * ax + 1], D.kdop[3 * ax + 2]}; lo_ax = info[48 + ax]; hi_ax = info[97 + ax];
      kax_ready = true;
    }
    const int ncand = __popcll(ballot(pt >= 0));   // candidates sit in lanes [0, ncand)
    const typename PrimOf<PRIM>::Body qb = PrimOf<PRIM>::load(D, max(pt, 0));
    const bool pass = kdop_cull_wave(qb, ncand, axv, lo_ax, hi_ax, off, lane);
    if constexpr (LEAN) {
      unsigned long long pm = ballot(pass && pt >= 0);
      if (pm) {
        int k = max(kmax, __builtin_amdgcn_readfirstlane(atomicAdd(&D.k_obs[u], 0)));   // any earlier value is a valid lower bound (uniform)
        while (pm) {
          const int l = __ffsll((long long)pm) - 1;
          pm &= pm - 1ull;
          const typename PrimOf<PRIM>::Body qu = bcast_body(qb, l);
          while (k < STEP_CAP) {
            const V3 v = gjk_wave(BodySwept{info, info + 18, D.pow08[k]}, qu, lane);
            if (!(v.x * v.x + v.y * v.y + v.z * v.z <= off * off)) break;
            k++;
          }
        }
        if (k >= STEP_CAP && lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_CCD_STUCK);
        if (k > kmax) { kmax = k; if (lane == 0) atomicMax(&D.k_obs[u], k); }
      }
    } else
    if (pt >= 0) {
      if (pass) {
        int k = max(kmax, atomicAdd(&D.k_obs[u], 0));  // any earlier value is a valid lower bound
        while (k < STEP_CAP) {
          const V3 v = gjk(BodySwept{info, info + 18, D.pow08[k]}, qb);
          if (!(v.x * v.x + v.y * v.y + v.z * v.z <= off * off)) break;
          k++;
        }
        if (k >= STEP_CAP) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_CCD_STUCK);
        if (k > kmax) { kmax = k; atomicMax(&D.k_obs[u], k); }
      }
    }
  }, &topb);
  if (lane == 0) {   // fire-and-forget atomics: a read-modify-write would keep the wave alive for another memory round trip
    unsigned long long* st = D.seg_stats + ((size_t)u * D.S + tr) * 6;
    atomicAdd(&st[2], visits); atomicAdd(&st[3], (unsigned long long)found);
    if (found > 0) atomicAdd(&D.ccd_found[blockIdx.x & 63], found);
  }
  // asynchronous solve: a robot's flag goes up behind its direction, a few stores (wolfe, |g|, the time direction) BEFORE its block is through -- this launch must not end
  // before every block is: the unit of segment 0 looks at the count once its own work is done (it is full by then; the folded finisher waits for it as well)
  if (D.xs_async && tr == 0) xs_wait<8>(D, D.xs_done(), D.u1 - D.u0);
}

template <int PRIM>
__global__ __launch_bounds__(64) void k_ccd_obs(Dev D) {
  if (TJ_DONE(D)) return;
  __shared__ double lds[CCD_LDS_DOUBLES];
  ccd_obs_body<PRIM>(D, blockIdx.x, lds);
}

// Phase A: one wave per (segment, lower robot p0, chunk of 64 partners p1 > p0); lanes over the partners.  Acting pairs go to
// ONE global list as sortable keys; the replay kernel sorts it, so (segment, p0, p1) is a lexicographic, deterministic order.
constexpr int ACT_CAP = 4096;                                    // acting pairs of one iteration (all segments)
constexpr int ROBOT_BITS = 11;                                   // robot ids in packed pair keys: U <= 2048 (with S < 512 a key is 31 bits)
__device__ __forceinline__ int act_key(int tr, int p0, int p1) { return (tr << (2 * ROBOT_BITS)) | (p0 << ROBOT_BITS) | p1; }
template <bool LEAN = false>   // LEAN: the survivors of the box + k-DOP filter one after the other by the whole wave (see ccd_obs_body)
__device__ __forceinline__ int ccd_self_pairs_body(const Dev& D, int bid, double* lds, bool wait_xf = false) {   // returns the acting pairs this tile listed
  int tr, rb, cb;
  pair_unit(D.U, D.pair_rows, bid, tr, rb, cb);
  const int lane = lane_id();
  const int U = D.U;
  const double off = D.offset;
  double* rowbox = lds; int* list = (int*)(lds + PAIR_ROWS_MAX * 6);
  if (wait_xf || D.xs_async) { xf_wait_seg(D, 1, tr); TJ_TIC(D, K_CCD, 6); }   // (asynchronous Newton solve: the obstacle units of this launch build the records once their robot's direction is there)   // sharded contexts (union kernel): the swept-hull cache of the other ranks' robots is written by units at the head of this launch
  // swept boxes (lanes over partners), then swept 49-axis intervals (lanes over axes): BVH::SelfCCDCollision + CCD::SelfKDOPCCD
  const int m = pair_tile_filter(D.cbox + (size_t)tr * 6 * U, U, rb, D.pair_rows, cb, 0, U,
                                 [&](int q) { return D.ccdinfo + ((size_t)q * D.S + tr) * CCD_STRIDE; }, 48, 97, off, rowbox, list, lane);
  if (m == 0) return 0;
  int found = 0;
  // A pair can only ever ACT in the sequential replay if its swept hulls are within `offset` at FULL step: hulls are
  // nested in the step (conv{P, P+tD} shrinks with t), and the replay evaluates them at steps <= 1.  Deciding that
  // here, in parallel over all tiles (one surviving pair per lane), leaves the one-wave replay kernel with the (rare)
  // colliding pairs only.
  if constexpr (LEAN) {
    for (int i = 0; i < m; i++) {   // (uniform)
      const int p0 = list[i] >> 16, p1 = list[i] & 0xffff;
      const double* a = D.ccdinfo + ((size_t)p0 * D.S + tr) * CCD_STRIDE;
      const double* b = D.ccdinfo + ((size_t)p1 * D.S + tr) * CCD_STRIDE;
      const V3 v = gjk_wave(BodySwept{a, a + 18, D.pow08[0]}, BodySwept{b, b + 18, D.pow08[0]}, lane);
      if (v.x * v.x + v.y * v.y + v.z * v.z <= off * off) {
        found++;
        if (lane == 0) {
          const int w = atomicAdd(&D.ctl->any_pair, 1);
          if (w < ACT_CAP) __hip_atomic_store(&D.pair_list[w], act_key(tr, p0, p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW);
        }
      }
    }
    return found;
  }
  for (int i0 = 0; i0 < m; i0 += 64) {
    bool ok = false;
    int p0 = 0, p1 = 0;
    if (i0 + lane < m) {
      p0 = list[i0 + lane] >> 16; p1 = list[i0 + lane] & 0xffff;
      const double* a = D.ccdinfo + ((size_t)p0 * D.S + tr) * CCD_STRIDE;
      const double* b = D.ccdinfo + ((size_t)p1 * D.S + tr) * CCD_STRIDE;
      const V3 v = gjk(BodySwept{a, a + 18, D.pow08[0]}, BodySwept{b, b + 18, D.pow08[0]});
      ok = v.x * v.x + v.y * v.y + v.z * v.z <= off * off;
    }
    const unsigned long long mask = ballot(ok);
    found += __popcll(mask);
    if (mask) {   // rare
      int base = 0;
      if (lane == 0) base = atomicAdd(&D.ctl->any_pair, __popcll(mask));   // any_pair = number of acting pairs this iteration
      base = __shfl(base, 0);
      const int w = base + prefix_count(mask);
      if (ok) {
        if (w < ACT_CAP) __hip_atomic_store(&D.pair_list[w], act_key(tr, p0, p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (write-through: the folded replay reads it inside the same launch)
        else atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW);
      }
    }
  }
  return found;
}

__global__ __launch_bounds__(64) void k_ccd_self_pairs(Dev D) {
  if (TJ_DONE(D)) return;
  __shared__ double lds[PAIR_LDS_DOUBLES];
  ccd_self_pairs_body<false>(D, blockIdx.x, lds);
}

// Phase B + gnorm.  One workgroup of one wave; control flow is wave uniform.
// Per segment the acting pairs (phase A's survivors) are gathered in lexicographic order.  Pairs that share no robot
// commute; if two of them DO share a robot the result depends on the order in which the reference meets them, which is
// the emission order of its per-segment dynamic AABB tree: that tree is then rebuilt here (dev_dyntree.h, lane 0, LDS)
// and the segment's pairs are replayed in its order.  Ctl::order_ambiguous counts such segments; Ctl::order_unresolved
// those for which the order could not be established (tree does not fit LDS / more than SEQ_ACT_CAP acting pairs in one
// segment) -- tj_iterate reports that as TJ_ERR_UNSUPPORTED instead of silently continuing.
constexpr int SEQ_ACT_CAP = 256;   // acting pairs of ONE segment
constexpr int SEQ_STK_CAP = 1024;  // node-pair stack of the tree's self query
__host__ __device__ inline size_t seq_lds_bytes(int U, int S, bool with_tree) {
  size_t b = (with_tree ? dyntree_lds_bytes(U) + 6 * (size_t)U * sizeof(double) : 0);
  b += (size_t)U * sizeof(double);   // gnorm staging
  b += (2 * (size_t)U + ACT_CAP + 3 * SEQ_ACT_CAP + (with_tree ? 2 * SEQ_STK_CAP : 0)) * sizeof(int);
  (void)S;
  return b;
}
// where the replay keeps its arrays: the stand-alone kernel carves everything from its dynamic LDS; folded into the tail of
// k_ccd (below) the small arrays live in that kernel's static buffer and the tree -- touched by lane 0 only, and only when two
// acting pairs of a segment share a robot -- in global memory
struct SeqMem {
  double *tbox, *tarea, *bx, *gns;   // tree nodes [12U], [2U]; swept boxes of the segment [6U]; gnorm staging [U]
  int *ti, *ks, *seen, *keys, *act0, *act1, *ord, *stk;
  int key_cap;        // acting pairs the sort can hold (a power of two)
  bool lane0_stages;  // bx is global memory: lane 0 copies the boxes itself (no cross-lane traffic through global memory)
};
// doubles of k_ccd's static buffer the folded replay needs, and what the tree needs in global memory
__host__ __device__ inline size_t seq_fold_lds_bytes(int U, int key_cap) { return (size_t)U * sizeof(double) + (2 * (size_t)U + key_cap + 3 * SEQ_ACT_CAP) * sizeof(int); }
__host__ __device__ inline size_t seq_fold_gmem_doubles(int U) { return 20 * (size_t)U; }
__host__ __device__ inline size_t seq_fold_gmem_ints(int U) { return 10 * (size_t)U + 2 * SEQ_STK_CAP; }

// FOLD: called by the last block of k_ccd to finish (decoupled mode only); what other blocks of the SAME launch produced --
// the acting-pair keys and their count -- is read with agent-scope atomic loads (the producers store them that way and wait
// for the stores before they take their ticket).
template <bool FOLD>
__device__ __forceinline__ void ccd_self_seq_body(const Dev& D, const SeqMem& M, bool with_gnorm = true) {
  const int lane = lane_id();
  double* tbox = M.tbox; double* tarea = M.tarea; double* bx = M.bx; double* gns = M.gns;
  int* ti = M.ti; int* ks = M.ks; int* seen = M.seen; int* keys = M.keys;
  int* act0 = M.act0; int* act1 = M.act1; int* ord = M.ord; int* stk = M.stk;
  TJ_TIC(D, K_CCD_SELF_SEQ, 0);
  for (int i = lane; i < D.U; i += 64) { ks[i] = 0; seen[i] = -1; }
  // usually no pair is within `offset` at full step (any_pair counts the acting pairs the selection kernel listed)
  int n_act = 0;
  if (D.multi()) {
    const int listed = FOLD ? __hip_atomic_load(&D.ctl->any_pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : D.ctl->any_pair;
    n_act = min(listed, min(ACT_CAP, M.key_cap));
    if (listed > n_act && listed <= ACT_CAP && lane == 0) atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW);   // more acting pairs than the folded replay can sort (beyond ACT_CAP the selection has reported it)
  }
  if (n_act > 0) {
    // stage the keys and sort them ascending = (segment, p0, p1) lexicographic: bitonic network over the next power of two
    int npow = 1; while (npow < n_act) npow <<= 1;
    for (int i = lane; i < npow; i += 64) keys[i] = i < n_act ? (FOLD ? __hip_atomic_load(&D.pair_list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : D.pair_list[i]) : 0x7fffffff;
    blk_sync<true>();
    for (int k = 2; k <= npow; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < npow; i += 64) {
          const int l = i ^ j;
          if (l > i) {
            const int x = keys[i], y = keys[l];
            const bool up = (i & k) == 0;
            if ((x > y) == up) { keys[i] = y; keys[l] = x; }
          }
        }
        blk_sync<true>();
      }
  }
  __syncthreads();
  TJ_TIC(D, K_CCD_SELF_SEQ, 1);
  if (n_act > 0) {
    const bool shared = D.coupled();  // Step::couple_self_step (Step.h:112-182): one step for all robots, held in ks[0]
    const double off2 = D.offset * D.offset;
    int ambiguous = 0, unresolved = 0;
    for (int pos = 0; pos < n_act;) {
      // 1. this segment's acting pairs, lexicographic (p0, p1); does any robot appear twice?
      const int tr = keys[pos] >> (2 * ROBOT_BITS);
      int m = 0; bool share = false;
      while (pos < n_act && (keys[pos] >> (2 * ROBOT_BITS)) == tr) {
        const int p0 = (keys[pos] >> ROBOT_BITS) & ((1 << ROBOT_BITS) - 1), p1 = keys[pos] & ((1 << ROBOT_BITS) - 1);
        if (seen[p0] == tr || seen[p1] == tr) share = true;
        blk_sync<true>();
        if (lane == 0) { seen[p0] = tr; seen[p1] = tr; if (m < SEQ_ACT_CAP) { act0[m] = p0; act1[m] = p1; } }
        blk_sync<true>();
        m++; pos++;
      }
      if (m > SEQ_ACT_CAP) { if (lane == 0) atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW); m = SEQ_ACT_CAP; unresolved += share && !shared; share = false; }
      // 2. order: lexicographic unless two acting pairs share a robot, then the reference's tree order
      bool tree_order = false;
      if (share && !shared && m >= 2) {
        if (D.seq_tree) {
          if (M.lane0_stages) { if (lane == 0) for (int i = 0; i < 6 * D.U; i++) { const int k = i / D.U, u = i % D.U; bx[6 * u + k] = D.cbox[((size_t)tr * 6 + k) * D.U + u]; } }
          else for (int i = lane; i < 6 * D.U; i += 64) { const int k = i / D.U, u = i % D.U; bx[6 * u + k] = D.cbox[((size_t)tr * 6 + k) * D.U + u]; }  // swept boxes of ALL robots (BVH.cpp:301-325)
          blk_sync<true>();
          int found = 0;
          if (lane == 0) {
            DynTree t{tbox, tarea, ti, ti + 2 * D.U, ti + 4 * D.U, ti + 6 * D.U, ti + 8 * D.U, DT_NIL, 0};
            for (int u = 0; u < D.U; u++) dt_insert(t, u, bx + 6 * u);
            found = dt_pair_order(t, D.offset, act0, act1, m, ord, stk, SEQ_STK_CAP - 1);
          }
          blk_sync<true>();
          tree_order = __shfl(found, 0) == m;
        }
        if (tree_order) ambiguous++; else unresolved++;
      }
      // 3. the joint back-off, pair after pair (Step.h:213-251)
      for (int j = 0; j < m; j++) {
        const int i = tree_order ? ord[j] : j;
        const int p0 = act0[i], p1 = act1[i];
        const double* a = D.ccdinfo + ((size_t)p0 * D.S + tr) * CCD_STRIDE;
        const double* b = D.ccdinfo + ((size_t)p1 * D.S + tr) * CCD_STRIDE;
        int k0 = ks[shared ? 0 : p0], k1 = ks[shared ? 0 : p1];
        int guard = 0;
        while (guard++ < STEP_CAP) {
          const V3 v = gjk_wave(BodySwept{a, a + 18, D.pow08[min(k0, STEP_CAP)]}, BodySwept{b, b + 18, D.pow08[min(k1, STEP_CAP)]}, lane);  // the wave is uniform here: solve the pair cooperatively
          if (!(v.x * v.x + v.y * v.y + v.z * v.z <= off2)) break;
          k0++; k1++;
        }
        if (guard > STEP_CAP && lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_CCD_STUCK);
        blk_sync<true>();
        if (lane == 0) { if (shared) ks[0] = k0; else { ks[p0] = k0; ks[p1] = k1; } }
        blk_sync<true>();
      }
    }
    if (lane == 0 && ambiguous) atomicAdd(&D.ctl->order_ambiguous, ambiguous);
    if (lane == 0 && unresolved) atomicAdd(&D.ctl->order_unresolved, unresolved);
  }
  __syncthreads();
  TJ_TIC(D, K_CCD_SELF_SEQ, 2);
  {
    const int kshared = ks[0];
    __syncthreads();
    for (int i = lane; i < D.U; i += 64) { D.k_self[i] = D.coupled() ? kshared : ks[i]; seen[i] = 0; }
    if constexpr (!FOLD) if (D.coupled()) for (int i = D.u0 + lane; i < D.u1; i += 64) D.k_obs_f[i] = (double)D.k_obs[i];   // exchange buffer 3 (sharded contexts)
  }
  if (!with_gnorm) return;
  // gnorm exactly as the drivers form it (Optimization3D_multi.h:57,72,750; _admm.h:499): a
  // sequential sum in robot order; the values are first pulled into LDS by all lanes
  __syncthreads();
  for (int i = lane; i < D.U; i += 64) gns[i] = D.gn(i);
  __syncthreads();
  TJ_TIC(D, K_CCD_SELF_SEQ, 3);
  double gt = 0, xg = 0;
  if constexpr (!FOLD) if (D.coupled()) {   // per-robot terms of G_t and x0.G: fetched by the lanes (one round trip per 64 robots), added by lane 0 in robot order
    __shared__ double s_cp[2][64];
    for (int u0 = 0; u0 < D.U; u0 += 64) {
      const int nu = min(64, D.U - u0);
      __syncthreads();
      if (lane < nu) { s_cp[0][lane] = D.xdir[(size_t)(u0 + lane) * D.xs + 3 * D.T + 3]; s_cp[1][lane] = D.wolfe(u0 + lane); }
      __syncthreads();
      if (lane == 0) for (int j = 0; j < nu; j++) { gt += s_cp[0][j]; xg += s_cp[1][j]; }
    }
  }
  if (lane == 0) {
    double gsum = 0;
    for (int u = 0; u < D.U; u++) gsum += gns[u];
    if (!FOLD && D.coupled()) {
      // gnorm = |G| / uav_num and wolfe = -x0.G over the whole arrowhead system (Optimization3D_multi.h:558,580);
      // k_xsolve_c2 left per-robot partial sums, the shared-time entries are added here
      D.ctl->gnorm = sqrt(gsum + gt * gt) / double(D.U);
      D.ctl->wolfe_c = -(xg + D.tdir(0) * gt);
    } else D.ctl->gnorm = (D.mode == 1) ? gsum / double(D.U) : gns[0];
  }
  TJ_TIC(D, K_CCD_SELF_SEQ, 4);
}
__global__ __launch_bounds__(64) void k_ccd_self_seq(Dev D) {
  if (TJ_DONE(D)) return;
  extern __shared__ double seq_sm[];
  // doubles first (alignment): tree nodes + the segment's swept boxes + gnorm staging, then the int arrays
  SeqMem M;
  M.tbox = seq_sm; M.tarea = M.tbox + (D.seq_tree ? 12 * (size_t)D.U : 0); M.bx = M.tarea + (D.seq_tree ? 2 * (size_t)D.U : 0);
  M.gns = M.bx + (D.seq_tree ? 6 * (size_t)D.U : 0);               // [U]
  M.ti = (int*)(M.gns + D.U);                                        // [5][2U] parent, left, right, height, particle
  M.ks = M.ti + (D.seq_tree ? 10 * (size_t)D.U : 0);                // [U] exponents
  M.seen = M.ks + D.U;                                               // [U] last segment in which the robot appeared
  M.keys = M.seen + D.U;                                             // [ACT_CAP] acting pairs, sorted
  M.act0 = M.keys + ACT_CAP; M.act1 = M.act0 + SEQ_ACT_CAP; M.ord = M.act1 + SEQ_ACT_CAP;
  M.stk = M.ord + SEQ_ACT_CAP;
  M.key_cap = ACT_CAP; M.lane0_stages = false;
  ccd_self_seq_body<false>(D, M);
}

// ---- slack (z) and dual update -------------------------------------------------------------------
__device__ inline double z_energy(const Dev& D, const double* cx, double pt, const double* z, double t, const double* lam, double tl) {
  // Energy_admm::slack_energy / dynamic_energy (Energy_admm.h:172-215)
  double e = 0;
  const double s = D.ks / pow5(t) * 0.5;
  for (int a = 0; a < 3; a++) {
    double rrow[6], y[6];
    for (int k = 0; k < 6; k++) rrow[k] = s * z[k + 6 * a];
    for (int j = 0; j < 6; j++) { double acc = 0; for (int k = 0; k < 6; k++) acc += rrow[k] * D.mdyn[k * 6 + j]; y[j] = acc; }
    double q = 0; for (int j = 0; j < 6; j++) q += y[j] * z[j + 6 * a];
    e += q;
  }
  e = e + D.kt * pow(t, 1.1);
  double delta[18], prod[18];
  for (int i = 0; i < 18; i++) { delta[i] = cx[i] - z[i]; prod[i] = delta[i] * delta[i]; }
  e += D.mu / 2.0 * esum(prod, 18);
  e += D.mu / 2.0 * (pt - t) * (pt - t);
  for (int a = 0; a < 3; a++) { double pr[6]; for (int j = 0; j < 6; j++) pr[j] = lam[j + 6 * a] * delta[j + 6 * a]; e += esum(pr, 6); }
  e += tl * (pt - t);
  return e;
}

// ---- register forms of the slack system's factorisation (EXACT: IEEE sqrt / division, unfused a - l*l -- the same operations
// in the same order as chol_lds / chol_arrow_backsolve_lds, whose pass/fail decision is pinned to Eigen's unblocked LLT) ----
template <int N>
__device__ __forceinline__ bool slack_chol_wave(double (&r)[N], double& y, int lane) {   // lane i: row i of the dense N x N system, y_i
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double x = readlane_f64(r[k], k);
    if (x <= 0) return false;
    const double sx = sqrt(x);
    const double lik = r[k] / sx;
    const double yk = readlane_f64(y, k) / sx;
    r[k] = lane == k ? sx : lik;
#pragma unroll
    for (int j = k + 1; j < N; j++) r[j] = r[j] - lik * readlane_f64(lik, j);
    y = lane == k ? yk : (lane > k ? y - yk * lik : y);
  }
  return true;
}
// factor + forward substitution from the LDS copy (L untouched on failure); on success the factor's lower triangle and L^-1 g
// go back to LDS and the back substitution x = L^-T y runs with y in registers, rows of L fetched ahead of the chain
template <int N>
__device__ __noinline__ bool slack_solve_wave(double* L, double* x0, int lane) {
  double r[N];
  const int row = min(lane, N - 1);
#pragma unroll
  for (int j = 0; j < N; j++) r[j] = L[row * N + j];
  double y = x0[row];
  if (!slack_chol_wave<N>(r, y, lane)) return false;
  blk_sync<true>();
#pragma unroll
  for (int j = 0; j < N; j++) if (lane < N && j <= lane) L[lane * N + j] = r[j];
  blk_sync<true>();
#pragma unroll
  for (int j = N - 1; j >= 0; j--) {
    const double lrow = L[j * N + row];
    const double xj = readlane_f64(y, j) / readlane_f64(lrow, j);
    y = lane == j ? xj : (lane < j ? y - xj * lrow : y);
  }
  if (lane < N) x0[lane] = y;
  blk_sync<true>();
  return true;
}

// z_energy by one wave: the same sums in the same order, the independent pieces side by side on the lanes.  pw = pow(t, 1.1)
// (taken by the caller, several arguments per pass on different lanes); w18: LDS scratch [36].  Uniform result.
__device__ __forceinline__ double z_energy_wave(const Dev& D, const double* md, const double* cx, double pt, const double* z, double t, const double* lam, double tl,
                                                double pw, double* w18, int lane) {
  const double s = D.ks / pow5(t) * 0.5;
  const int a = min(lane, 17) / 6, j = min(lane, 17) % 6;
  double y = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) y += (s * z[k + 6 * a]) * md[k * 6 + j];      // lane (a, j): y_j of axis a
  const double delta = cx[min(lane, 17)] - z[min(lane, 17)];
  blk_sync<true>();
  if (lane < 18) { w18[lane] = delta * delta; w18[18 + lane] = lam[lane] * delta; }
  double e = 0;
#pragma unroll
  for (int ax = 0; ax < 3; ax++) {
    double q = 0;
#pragma unroll
    for (int jj = 0; jj < 6; jj++) q += readlane_f64(y, 6 * ax + jj) * z[jj + 6 * ax];
    e += q;
  }
  e = e + D.kt * pw;
  blk_sync<true>();
  e += D.mu / 2.0 * esum_wave(w18, 18, lane);
  e += D.mu / 2.0 * (pt - t) * (pt - t);
#pragma unroll
  for (int ax = 0; ax < 3; ax++) e += esum(w18 + 18 + 6 * ax, 6);
  e += tl * (pt - t);
  return e;
}

// deferred = 1: this launch belongs to the NEXT iteration's graph (or to a flush) and performs the
// update the previous iteration still owes, concurrently with the next iteration's plane kernels
// (they only read the control points).  deferred = 0: stage API, update of the current iteration.
// One wave per piece, and a chain of dependent steps: Newton system in LDS -> factorisation and solve in REGISTERS (one row
// per lane, v_readlane broadcasts: the LDS form pays three barriers per pivot) -> Armijo search with the objective evaluated by
// the whole wave (z_energy_wave) and every pow() of a pass taken on different lanes at once (the scalar loop on one lane was
// a third of the 27 us this body took; it is the whole of k_mid for one or a few robots).
constexpr int SLACK_LDS_DOUBLES = 5 * 18 + 19 + 2 * 361 + 2 * 19 + 4 * 19 + 36 + 36;   // 1017
__device__ __forceinline__ void slack_body(const Dev& D, int bid, int deferred, double* lds) {   // lds: SLACK_LDS_DOUBLES doubles the KERNEL owns (k_mid overlays them with the pair tile)
  const int tid = threadIdx.x;
  const int u = D.u0 + bid / D.P, sp = bid % D.P;
  const int P6 = 6 * D.P, T = D.T;
  double* cx = lds; double* z = cx + 18; double* lam = z + 18; double* zt = lam + 18; double* dirz = zt + 18; double* g = dirz + 18; double* H = g + 19; double* L = H + 361;
  double* g0 = L + 361; double* x0 = g0 + 19; double* scr = x0 + 19; double* md = scr + 4 * 19; double* w18 = md + 36;
  const double* net = D.spline + (size_t)u * 3 * T;
  const double* C = D.convert + (size_t)sp * 36;
  const double pt = D.piece_time[u];
  double t = D.t_slack[u * D.P + sp];
  const double tl = D.t_lambda[u * D.P + sp];
  if (tid < 18) {
    const int j = tid % 6, a = tid / 6;
    double acc = 0;
    for (int k = 0; k < 6; k++) acc += C[j * 6 + k] * net[sp * 3 + k + T * a];
    cx[tid] = acc;
    z[tid] = D.p_slack[(size_t)u * 3 * P6 + sp * 6 + j + P6 * a];
    lam[tid] = D.p_lambda[(size_t)u * 3 * P6 + sp * 6 + j + P6 * a];
  }
  if (tid < 36) md[tid] = D.mdyn[tid];
  for (int i = tid; i < 361; i += 64) H[i] = 0;
  // the three powers of t this update needs, one pass: lane 0 t^0.1, lane 1 t^-0.9, lane 2 t^1.1
  const double pwl = pow(t, tid == 0 ? 0.1 : (tid == 1 ? -0.9 : 1.1));
  const double pw01 = readlane_f64(pwl, 0), pwm09 = readlane_f64(pwl, 1), pw11 = readlane_f64(pwl, 2);
  __syncthreads();
  // Gradient_admm::slack_gradient / dynamic_gradient (Gradient_admm.h:574-671)
  const double sc = D.ks / pow5(t);
  if (tid < 18) {
    const int k = tid / 3, a = tid % 3;
    double mz = 0;
    for (int j = 0; j < 6; j++) mz += md[k * 6 + j] * z[j + 6 * a];
    const double g1 = sc * mz;
    const double g2 = D.mu * (z[k + 6 * a] - cx[k + 6 * a]) - lam[k + 6 * a];
    g[tid] = g1 + g2;
    const double pg = -5 * g1 / t;
    H[tid * 19 + 18] = pg; H[18 * 19 + tid] = pg;
    for (int b = 0; b < 6; b++) H[tid * 19 + 3 * b + a] = sc * md[k * 6 + b] + (k == b ? D.mu : 0.0);
  }
  {  // time entries: the quadratic form z^T M z per axis on lanes (a, j) like z_energy_wave, combined uniformly
    const double s = sc * 0.5;
    const int a = min(tid, 17) / 6, j = min(tid, 17) % 6;
    double y = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) y += (s * z[k + 6 * a]) * md[k * 6 + j];
    double dyn = 0;
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
      double q = 0;
#pragma unroll
      for (int jj = 0; jj < 6; jj++) q += readlane_f64(y, 6 * ax + jj) * z[jj + 6 * ax];
      dyn += q;
    }
    double g_t = -5 * dyn / t;
    g_t += D.kt * 1.1 * pw01;
    double h_t = 30 * dyn / (t * t);
    h_t += D.kt * 0.11 * pwm09;
    g_t += D.mu * (t - pt) - tl;
    h_t += D.mu;
    if (tid == 32) { g[18] = g_t; H[18 * 19 + 18] = h_t; }
  }
  __syncthreads();
  // boundary pieces keep two control points fixed (Optimization3D_multi.h:374-420)
  int lo = 0, tn = 6;
  if (sp == 0) { lo = 2; tn = 4; } else if (sp == D.P - 1) { lo = 0; tn = 4; }
  const int n = 3 * tn + 1;
  auto mapi = [&](int i) { return i < 3 * tn ? 3 * lo + i : 18; };
  for (int idx = tid; idx < n * n; idx += 64) { const int i = idx / n, j = idx % n; L[idx] = H[mapi(i) * 19 + mapi(j)]; }
  if (tid < n) { g0[tid] = g[mapi(tid)]; x0[tid] = g[mapi(tid)]; }
  __syncthreads();
  const bool solved = n == 19 ? slack_solve_wave<19>(L, x0, tid) : slack_solve_wave<13>(L, x0, tid);   // x0 <- (L L^T)^-1 g0
  if (!solved) {  // LLT failed (L and x0 untouched): eigen-shift like the reference, everything in LDS -- rare
    for (int idx = tid; idx < n * n; idx += 64) H[idx] = L[idx];  // H now holds the reduced system (n x n)
    __syncthreads();
    const double ev = min_eig_lds(L, n, scr, scr + 19, scr + 38, scr + 57, tid, 64);
    if (ev < 0 && tid < n) H[tid * n + tid] = H[tid * n + tid] - ev * 1.0 + 0.01 * 1.0;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 64) L[idx] = H[idx];
    if (tid < n) x0[tid] = g0[tid];
    __syncthreads();
    chol_lds(L, n, tid, 64, x0);
    __syncthreads();
    chol_arrow_backsolve_lds(L, n, n, x0, tid, 64);
  }
  if (tid < n) x0[tid] = -x0[tid];
  if (tid < 18) dirz[tid] = 0;
  __syncthreads();
  if (tid < 3 * tn) { const int i = tid / 3, a = tid % 3; dirz[(lo + i) + 6 * a] = x0[tid]; }
  if (tid < 19) scr[tid] = tid < n ? x0[tid] * g0[tid] : 0.0;
  __syncthreads();
  // Armijo loop (update_slack_lambda): uniform on the wave
  const double wolfe = -(n == 19 ? esum(scr, 19) : esum(scr, 13));
  const double t_dir = x0[3 * tn];
  double step = 1.0;
  if (t + step * t_dir <= 0) step = -0.95 * t / t_dir;
  const double t_init = t;
  double tt = t_init + step * t_dir;
  const double pwt = pow(tt, 1.1);
  const double e = z_energy_wave(D, md, cx, pt, z, t, lam, tl, pw11, w18, tid);
  double pwc = pwt;
  int guard = 0;
  for (;;) {
    blk_sync<true>();
    if (tid < 18) zt[tid] = z[tid] + step * dirz[tid];
    blk_sync<true>();
    const double en = z_energy_wave(D, md, cx, pt, zt, tt, lam, tl, pwc, w18, tid);
    if (!(e - 1e-4 * wolfe * step < en)) break;
    if (++guard >= STEP_CAP) { if (tid == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_SLACK_ARMIJO); break; }
    step *= 0.8;
    tt = t_init + step * t_dir;
    pwc = pow(tt, 1.1);
  }
  const double s_t = tt;
  __syncthreads();
  if (tid < 18) {
    const int j = tid % 6, a = tid / 6;
    const size_t o = (size_t)u * 3 * P6 + sp * 6 + j + P6 * a;
    D.p_slack[o] = zt[tid];
    D.p_lambda[o] += D.mu * (cx[tid] - zt[tid]);
  }
  if (tid == 0) {
    D.t_slack[u * D.P + sp] = s_t;
    D.t_lambda[u * D.P + sp] += D.mu * (pt - s_t);
    if (!deferred && bid == 0) D.ctl->slack_next = 0;  // paid
  }
}
__global__ __launch_bounds__(64) void k_slack(Dev D, int deferred) {
  if (deferred ? !D.ctl->slack_now : TJ_DONE(D)) return;
  __shared__ double lds[SLACK_LDS_DOUBLES];
  slack_body(D, blockIdx.x, deferred, lds);
}

// ---- union kernels of the single-GPU iteration graph ---------------------------------------------------------------
// Measured with rocprofv3: kernels that follow each other on ONE hardware queue start back to back, while every
// fork/join between the branches of a multi-stream graph costs 12-25 us of cross-queue signalling -- four of them
// sat on the critical path of a 260 us iteration.  Independent stages therefore share a launch instead of a
// stream: a block's index range selects the stage it works for (all of these stages are one-wavefront work items),
// and the whole iteration is a linear chain on one queue.
//   k_front  obstacle candidate query (owned * S)  |  robot-pair rows (S * U)
//   k_mid    slack + dual update the previous iteration still owes (owned * P)  |  robot-pair solves  |  obstacle-candidate solves
//   k_ccd    obstacle CCD clamp (owned * S)  |  robot-pair CCD selection (S * U)
// FA (asynchronous front, dev_common.h Dev::fa_seq): this launch runs on the second queue NEXT TO the previous iteration's k_linesearch (behind k_fa_gate).  Stop flag
// and epoch come from fa_early_begin's record -- the control block still belongs to the running iteration --, units wait for their robots' commit flags, every
// store a later kernel reads is written through, and each block counts itself done behind them (the last k_linesearch block waits for that count).
template <int PRIM, bool FA = false>
__global__ __launch_bounds__(64) void k_front(Dev D) {
  if (D.keep_seq > 0 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(D.keep_go(), D.keep_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // opens the gate of the asynchronous plane refinement
  int fa_epoch = 0;
  if constexpr (FA) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(D.fa_fstart(blockIdx.x), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // resident (the last k_linesearch block may wait for that)
    fa_epoch = xf_load_i(D.fa_rec() + 1);
    if (xf_load_i(D.fa_rec() + 2) | (D.ctl->error & ERR_XS_TIMEOUT)) { fa_count(D.fa_fdone(blockIdx.x)); return; }   // (the begun iteration's stop test has fired, or a cross-queue wait has run out: the batch is being abandoned)
  } else if (TJ_DONE(D)) return;
  const int n_obs = (D.u1 - D.u0) * D.S;
  TJ_TIC(D, K_FRONT, 0);
  __shared__ double lds[OBS_LDS_DOUBLES > PAIR_LDS_DOUBLES ? OBS_LDS_DOUBLES : PAIR_LDS_DOUBLES];   // one buffer for whichever body this block runs
  static_assert(sizeof(lds) >= GRAD_ORDER_CHUNK * sizeof(int), "grad_order_body stages its costs in this buffer");
  const int n_spec = D.spec ? SPEC_CAP : 0;   // GJK head starts of last iteration's slow pairs lead the grid: they are the longest blocks
  const int n_ord = D.grad_bal ? ((D.u1 - D.u0) * D.P + 63) / 64 : 0;   // the first blocks of the grid: launch order of this iteration's k_grad (kernels_newton.h; ~3 us each --
                                                                        // as the LAST blocks they started when the first query blocks retired and ended 1 us after everything else)
  const int n_xf = D.xf_units();   // sharded contexts: hull cache of the other ranks' robots (coupled chain: of every robot), AHEAD of everything that reads it (head starts, pair tiles)
  const int bx = (int)blockIdx.x;
  const int b = bx - n_ord - n_xf - n_spec;
  const bool pub = D.xf_all != 0 || (D.fa_units != 0 && D.multi());   // the query forms its hull itself and publishes the record (coupled chain; Dev::fa_units: the k_linesearch before this launch published none)
  if (bx < n_ord) grad_order_body<FA>(D, bx, (int*)lds);
  else if (bx < n_ord + n_xf) xf_hull_body(D, bx - n_ord);
  else if (b < 0) spec_pair_body<FA>(D, bx - n_ord - n_xf, lds, fa_epoch);
  else if (b < n_obs) obs_query_body<PRIM, FA>(D, b, lds, !pub, pub);
  else sep_self_rows_body<FA>(D, b - n_obs, lds, D.xf != 0 || D.fa_units != 0);
  TJ_TIC(D, K_FRONT, 1);
  if constexpr (FA) fa_count(D.fa_fdone(blockIdx.x));
}
// asynchronous front: the gate in front of k_front on the second queue (one wave, no LDS) -- it ends when every block of the k_linesearch launch of this pairing has
// started, i.e. when all of them are resident: the k_front blocks that then sleep on commit flags cannot keep a k_linesearch block off its compute unit
__global__ __launch_bounds__(64) void k_fa_gate(Dev D, int want) { fa_wait16(D, D.fa_res(0), want); }
// two waves per SIMD (<= 256 VGPRs; 244 used, no spills since the slack body was rewritten): 2 048 one-wave blocks are resident
// at once -- on SCN-C the 320 slack blocks, the 1 024 pair waves and the first 704 of the 1 024 obstacle-solve waves (its ~250
// candidates all fall to those); the rest follow as slack blocks retire after ~13 us.  At the natural 340 VGPRs of round 1 a
// third of the blocks started only when an earlier one had finished, 18-33 us into the kernel.
// FA (asynchronous front, Dev::fa_mid): this launch started while the iteration's k_front was still running on the other queue.  Block 0 is the WATCHER: it polls k_front's
// done counters (every block of it counts itself behind its acknowledged write-through stores) and then raises the 64 go words; a pair / obstacle solve wave sleeps on
// one of them before it touches anything k_front leaves, and reads that past the caches.  The slack blocks start at once.
template <int PRIM, bool FA = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mid(Dev D, int n_pair_waves, int n_obs_waves) {
  const int n_slack = (D.u1 - D.u0) * D.P;
  if constexpr (FA) {
    if (blockIdx.x == 0) {
      fa_wait16(D, D.fa_fdone(0), (int)((unsigned)D.fa_seq * (unsigned)D.fa_nfront));
      sig_acked();   // (nothing of its own to acknowledge: the watcher only relays k_front's count)
      xf_store_i(D.fa_go(threadIdx.x), D.fa_seq);
      sig_sent();
      return;
    }
  }
  const int b = (int)blockIdx.x - (FA ? 1 : 0);
  TJ_TIC(D, K_MID, 0);
#ifdef TJ_PHASE_TIMING
  if (threadIdx.x == 0 && blockIdx.x < TJ_TIC_BLOCKS) {   // where the block runs: HW_ID (wave, SIMD, CU, SE) and XCC_ID
    D.dbg[((size_t)K_MID * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    D.dbg[((size_t)K_MID * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
  __shared__ double mid_lds[SLACK_LDS_DOUBLES > PAIR_TILE_DOUBLES ? SLACK_LDS_DOUBLES : PAIR_TILE_DOUBLES];   // one buffer for whichever body this block runs: the slack system, or the pair tile
  if (D.mid_order) {
    // Hundreds of robots (TJ_MID_ORDER): pair waves | obstacle solves | slack.  Blocks b and b + 1024 share a SIMD (two waves per SIMD, 1 024 SIMDs); with the slack
    // blocks leading, every producer wave of the one-pair-per-lane path had a slack wave as its mate and both took 80 - 100 us where they take 12 + 30 alone (phase
    // stamps + HW_ID, round 5).  With the pair waves leading, a producer's mate is a consumer that spins until pairs are passed on, the obstacle solves fill the
    // remaining slots and the slack waves follow as those retire: config 5's k_mid 76.5 -> 61 us.  Small fleets keep slack first (their k_mid is its slowest pair).
    const int s0 = n_pair_waves + n_obs_waves;
    if (b >= s0) { if (D.ctl->slack_now) slack_body(D, b - s0, 1, mid_lds); TJ_TIC(D, K_MID, 1); return; }
    if (TJ_DONE(D)) return;
    if (b < n_pair_waves) sep_self_solve_body<FA>(D, b, n_pair_waves, D.spec != 0, mid_lds);   // (FA: waits for k_front's end itself -- a dedicated wave after its early solve)
    else { if constexpr (FA) fa_wait_flag(D, D.fa_go(b), D.fa_seq); obs_solve_body<PRIM, FA>(D, b - n_pair_waves, n_obs_waves); }
    TJ_TIC(D, K_MID, 1);
    return;
  }
  if (b < n_slack) { if (D.ctl->slack_now) slack_body(D, b, 1, mid_lds); TJ_TIC(D, K_MID, 1); return; }   // long single-wave tasks first
  else if (TJ_DONE(D)) return;
  if (b < n_slack + n_pair_waves) sep_self_solve_body<FA>(D, b - n_slack, n_pair_waves, D.spec != 0, mid_lds);   // (FA: waits for k_front's end itself -- a dedicated wave after its early solve)
  else { if constexpr (FA) { fa_wait_flag(D, D.fa_go(b), D.fa_seq); TJ_TIC(D, K_MID, 4); } obs_solve_body<PRIM, FA>(D, b - n_slack - n_pair_waves, n_obs_waves); }
  TJ_TIC(D, K_MID, 1);
}
template <int PRIM, bool LEAN>
__device__ __forceinline__ void ccd_union_body(const Dev& D) {
  const int n_obs = (D.u1 - D.u0) * D.S;
  __shared__ double lds[CCD_LDS_DOUBLES > PAIR_LDS_DOUBLES ? CCD_LDS_DOUBLES : PAIR_LDS_DOUBLES];
  int found = 0;
  TJ_TIC(D, K_CCD, 0);
  // with the replay folded in (below) the grid has one block more: block 0 is the finisher and has no other work
  const int fin = D.seq_fold ? 1 : 0;
  const int n_xf = D.xf_units();   // sharded contexts: swept-hull cache of the other ranks' robots (coupled chain: of every robot), ahead of what reads it
  const int b = (int)blockIdx.x - fin - n_xf;
  if ((int)blockIdx.x >= fin && b < 0) xf_ccd_body(D, (int)blockIdx.x - fin, lds);
  else if (b >= 0 && b < n_obs) ccd_obs_body<PRIM, LEAN>(D, b, lds, D.xf_all != 0);
  else if (b >= 0) found = ccd_self_pairs_body<LEAN>(D, b - n_obs, lds, D.xf != 0);
  TJ_TIC(D, K_CCD, 1);
  // The sequential replay of the acting pairs + gnorm (k_ccd_self_seq: one wave with ~1 us of work in the usual case of no acting
  // pair, 4.6 us as a launch of its own) is finished inside this launch.  A first version -- every block takes a ticket, the
  // last one finishes -- cost 35 us (2 720 returning atomics on one address, ~13 ns each), and with two-level tickets of the
  // pair-selection blocks only the finisher's two atomic round trips + the replay's own loads still hung 3.5 us behind the
  // kernel's natural end.  Now nobody waits for a ticket: a selection block adds 1 (+ 65 536 if it listed an acting pair) to one of sixteen
  // counters, fire and forget, after its list entries have been performed (write-through stores); block 0 -- an extra block
  // with no other work -- forms gnorm (which
  // needs nothing of this launch) and then polls the sixteen counters -- one load per lane -- until every selection block is
  // in.  Their sum also tells it whether any pair acts: none, almost always, and then it is done (k_begin has zeroed k_self).
  if (!D.seq_fold) return;
  const int lane = lane_id();
  const int n_t = (int)gridDim.x - 1 - n_xf - n_obs;   // selection blocks
  if (b >= n_obs) {
    sig_acked();
    if (lane == 0) atomicAdd(&D.ctl->ccd_sub[(b - n_obs) & 15], 1 + (found ? 0x10000 : 0));
    sig_sent();
  }
  if (blockIdx.x != 0) return;
  __syncthreads();
  SeqMem M;
  M.gns = lds;                                   // [U]
  M.ks = (int*)(lds + D.U); M.seen = M.ks + D.U;  // [U], [U]
  M.act0 = M.seen + D.U; M.act1 = M.act0 + SEQ_ACT_CAP; M.ord = M.act1 + SEQ_ACT_CAP;
  M.keys = M.ord + SEQ_ACT_CAP; M.key_cap = D.seq_fold;   // Dev::seq_fold = the key capacity that fits the buffer (a power of two)
  M.tbox = D.seq_gmem_d; M.tarea = M.tbox + 12 * (size_t)D.U; M.bx = M.tarea + 2 * (size_t)D.U;
  M.ti = D.seq_gmem_i; M.stk = M.ti + 10 * (size_t)D.U;
  M.lane0_stages = true;
  {   // gnorm: the sequential sum in robot order (Optimization3D_multi.h:57,72,750)
    if (D.xs_async) xs_wait(D, D.xs_done(), D.u1 - D.u0);   // asynchronous Newton solve: every robot's |g| is in
    if (D.xf) xf_wait_seg(D, 1, 0);   // sharded contexts: the other ranks' |g| values are put in place by this launch's foreign units of segment 0
    for (int i = lane; i < D.U; i += 64) M.gns[i] = (D.xf || D.xs_async) ? xf_load(&D.gn(i)) : D.gn(i);
    __syncthreads();
    if (!D.coupled()) { if (lane == 0) { double gsum = 0; for (int u = 0; u < D.U; u++) gsum += M.gns[u]; D.ctl->gnorm = gsum / double(D.U); } }
    else {
      // coupled mode (one context; round 5: the replay is folded here too): gnorm = |G| / uav_num and wolfe = -x0.G over the whole arrowhead system
      // (Optimization3D_multi.h:558,580) from k_xsolve_c2's per-robot partial sums, the shared-time entries added in robot order -- what k_ccd_self_seq forms
      __shared__ double s_cp[2][64];
      double gt = 0, xg = 0;
      for (int u0 = 0; u0 < D.U; u0 += 64) {
        const int nu = min(64, D.U - u0);
        __syncthreads();
        if (lane < nu) {
          const double* pg = D.xdir + (size_t)(u0 + lane) * D.xs + 3 * D.T + 3;
          s_cp[0][lane] = D.xs_async ? xf_load(pg) : *pg; s_cp[1][lane] = D.xs_async ? xf_load(&D.wolfe(u0 + lane)) : D.wolfe(u0 + lane);   // (asynchronous solve: records of two robots share cache lines)
        }
        __syncthreads();
        if (lane == 0) for (int j = 0; j < nu; j++) { gt += s_cp[0][j]; xg += s_cp[1][j]; }
      }
      if (lane == 0) {
        double gsum = 0;
        for (int u = 0; u < D.U; u++) gsum += M.gns[u];
        D.ctl->gnorm = sqrt(gsum + gt * gt) / double(D.U);
        D.ctl->wolfe_c = -(xg + (D.xs_async ? xf_load(&D.tdir(0)) : D.tdir(0)) * gt);
      }
    }
    __syncthreads();
  }
  int any_act = 0;
  {
    const long long t_end = wall_clock64() + (D.xs_async ? XCH_TIMEOUT_TICKS : 500000 + 100ll * gridDim.x);   // 5 ms + 1 us per block of the launch: a logic error must not hang the device
                                                                                                                // (asynchronous solve: the tiles themselves wait for units that wait for the other queue -- 2 s, like them)
    for (;;) {
      int v = lane < 16 ? __hip_atomic_load(&D.ctl->ccd_sub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
      v = __shfl(v, 0);
      if ((v & 0xffff) == n_t) {
        any_act = v >> 16;
        // every selection block of THIS launch is in: take exactly what was observed back out, so that a launch that is repeated without a begin in between
        // (tj_iterate_phase on the CCD phase twice, a re-run after an error) starts from zero instead of finding the counters full.  Not after a
        // time-out: late blocks may still be adding (the next begin_body zeroes the words anyway).
        if (lane < 16) { const int mine = __hip_atomic_load(&D.ctl->ccd_sub[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (mine) atomicSub(&D.ctl->ccd_sub[lane], mine); }
        break;
      }
      if (wall_clock64() > t_end) { if (lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PASS_TIMEOUT); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (any_act > 0) ccd_self_seq_body<true>(D, M, false);   // (reads the pairs' count and keys with agent-scope loads)
  TJ_TIC(D, K_CCD, 2);
}
// Two builds of the same code.  The per-lane swept-hull GJK needs ~240 VGPRs, which leaves 2 waves per SIMD -- fewer slots
// (2 048) than SCN-C has units (2 720), although almost none of them ever reaches the GJK there.  k_ccd_lean is compiled
// for 3 waves per SIMD (168 VGPRs, the GJK spills ~215 registers): 3 us faster where the swept boxes meet (next to) no
// obstacle, 3 % slower where thousands of candidates per iteration go through the GJK (SCN-E).  The host picks by the
// candidate rate the device counted during the previous batch (Dev::ccd_found); both give the same bits.
template <int PRIM>
__global__ __launch_bounds__(64) void k_ccd(Dev D) {
  if (TJ_DONE(D)) return;
  ccd_union_body<PRIM, false>(D);
}
template <int PRIM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ccd_lean(Dev D) {
  if (TJ_DONE(D)) return;
  ccd_union_body<PRIM, true>(D);
}

// ---- iteration bookkeeping ---------------------------------------------------------------------
// the first iteration of a batch; direct exchange (sharded contexts): it also feeds the peers with this rank's control points as they stand now (inside a
// batch k_linesearch does, robot by robot) -- so a tj_set_state between batches reaches every rank like it did through the all-gather
// (self-healing, below: a region of the state and its place in the snapshot arena)
struct SnapRegion { char* live; char* snap; unsigned long long bytes; };
__device__ __forceinline__ void snapshot_copy(const SnapRegion* tab, int n, int dir, unsigned long long first, unsigned long long stride) {
  for (int r = 0; r < n; r++) {
    const SnapRegion R = tab[r];
    const unsigned long long n16 = R.bytes / 16;
    const uint4* src = (const uint4*)(dir ? R.snap : R.live); uint4* dst = (uint4*)(dir ? R.live : R.snap);
    for (unsigned long long i = first; i < n16; i += stride) dst[i] = src[i];
    if (first < (R.bytes & 15)) (dir ? R.live : R.snap)[n16 * 16 + first] = (dir ? R.snap : R.live)[n16 * 16 + first];
  }
}
// snap_n > 0 (self-healing): the launch also takes the batch's snapshot -- blocks 1 .. gridDim.x - 1 copy the state regions (nothing of block 0's begin work touches them),
// block 0 the control block before it begins the iteration; one launch instead of two at the head of every batch
__global__ void k_begin(Dev D, const SnapRegion* snap_tab, int snap_n, Ctl* ctl_snap) {
  if (blockIdx.x > 0) { snapshot_copy(snap_tab, snap_n, 0, (unsigned long long)(blockIdx.x - 1) * blockDim.x + threadIdx.x, (unsigned long long)(gridDim.x - 1) * blockDim.x); return; }
  if (snap_n > 0) { if (threadIdx.x == 0) *ctl_snap = *D.ctl; __syncthreads(); }
  const bool done = begin_body(D);
  if (!D.xch || done) return;
  const XchPeers* xp = D.xp;
  const int np = xp->n, T = D.T, own = D.u1 - D.u0;
  const size_t off = (size_t)D.u0 * 3 * T, cnt = (size_t)own * 3 * T;
  for (int q = 0; q < np; q++) for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) xch_store(xp->rx[q][0] + off + i, D.spline[off + i]);
  sig_acked();
  __syncthreads();
  asm volatile("" ::: "memory");
  if ((int)threadIdx.x < np) __hip_atomic_fetch_add(xp->cnt[threadIdx.x] + D.rank, (unsigned long long)own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  sig_sent();
  if (threadIdx.x == 0) atomicAdd(&D.ctl->xpush[0], own);
}
// ---- self-healing of the cross-queue schedules (tj_api.hip: heal_check) ----
// One launch copies every region of a table (dir = 0: state -> snapshot arena at the start of a batch; 1: back).  The control block is handled apart: on a restore
// the epoch stays where the abandoned batch left it (stamps of its iterations must never look current again) and the error bits of the incident are cleared.
__global__ void k_snapshot(const SnapRegion* tab, int n, int dir, Ctl* ctl, Ctl* ctl_snap) {
  for (int r = blockIdx.y; r < n; r += gridDim.y) snapshot_copy(tab + r, 1, dir, (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, (unsigned long long)gridDim.x * blockDim.x);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    if (!dir) *ctl_snap = *ctl;
    else {
      const int epoch = ctl->epoch, err = ctl->error & ~(ERR_XS_TIMEOUT | ERR_LOOP_CAP | ERR_PASS_TIMEOUT | ERR_LS_RANGE), give = ctl->ls_giveups, hto = ctl->ls_helper_timeouts;
      *ctl = *ctl_snap;
      ctl->epoch = epoch; ctl->error = ctl_snap->error | err; ctl->ls_giveups = give; ctl->ls_helper_timeouts = hto;
    }
  }
}
// test hook (TJ_XS_FAULT=<n>): the n-th gate launch of the asynchronous solve behaves as if its wait had run out
// only used by the stage API: commit the iteration counter explicitly
// hand a still-owed slack/dual update to the next k_slack(deferred) launch without starting an iteration
// cancel = 1: the last k_linesearch has already begun an iteration (begin_next) that the host then did not enqueue -- take that back: the update owed is the
// finished iteration's (already in slack_now), nothing is pending
__global__ void k_flush(Dev D, int cancel) {
  if (cancel) { D.ctl->pending = 0; D.ctl->slack_next = 0; return; }
  D.ctl->slack_now = D.ctl->slack_next; D.ctl->slack_next = 0;
}
__global__ void k_end(Dev D) {
  if (D.ctl->pending) { D.ctl->iter++; D.ctl->pending = 0; }
}

}  // namespace tj
