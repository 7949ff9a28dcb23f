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
#include "dev_linalg.h"
#include "kernels_sep.h"

namespace tj {

constexpr int HULL_STRIDE = HULL_INFO_STRIDE;  // P[6][3], lo[3], hi[3], kdop lo[49], kdop hi[49]

__global__ __launch_bounds__(64) void k_hullinfo(Dev D) {
  if (TJ_DONE(D)) return;
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

// One unit of the robot-pair broad phase = one wavefront = a TILE of the (lower robot p0, partner p1 > p0) triangle of one
// segment: R consecutive rows p0 (R = Dev::pair_rows: 2, 4, 8 or 16) against one aligned block of 64 partners.  A row per wave
// (round 1) and then a 64-partner chunk of a row per wave (35 680 one-wave workgroups per launch at 256 robots) were bound by
// workgroup dispatch and by one memory latency per handful of box tests; a tile loads the 64 partner boxes ONCE for R rows,
// and 25 440 units become 1 600.  Tiles of a segment: row blocks grouped by the column block g that holds their diagonal
// (rows [64g, 64g+64)); a row block of group g meets column blocks g .. nc-1.
__host__ __device__ inline int pair_units(int U, int R) {
  const int nc = (U + 63) / 64;
  int n = 0;
  for (int g = 0; 64 * g < U - 1; g++) { const int rows = min(64 * g + 64, U - 1) - 64 * g; n += ((rows + R - 1) / R) * (nc - g); }
  return n;
}
__host__ __device__ inline void pair_unit(int U, int R, int bid, int& tr, int& rb, int& cb) {
  const int per = pair_units(U, R), nc = (U + 63) / 64;
  tr = bid / per;
  int r = bid % per;
  rb = 0; cb = 0;
  for (int g = 0; 64 * g < U - 1; g++) {
    const int rows = min(64 * g + 64, U - 1) - 64 * g;
    const int cnt = ((rows + R - 1) / R) * (nc - g);
    if (r < cnt) { rb = 64 * g + (r / (nc - g)) * R; cb = 64 * (g + r % (nc - g)); return; }
    r -= cnt;
  }
}
constexpr int PAIR_CONSUMERS_MAX = 4096;            // bound on the waves that wait for passed-on pairs (all idle pair waves do: with 512 of them k_mid took 74 us on SCN-D, with 864 65)
constexpr unsigned long long PAIR_OVF_STOP = 1ull << 31;   // entry flag: the list ends here
constexpr int PAIR_LANE_GJK_CAP = 3;   // GJK iterations a lane spends on its pair before passing it on (large fleets, see sep_self_solve_body; 2 .. 4 measured alike, 8 slower)
constexpr int PT_FLIGHT = 8;   // pairs whose interval records a tile fetches together
constexpr int PAIR_ROWS_MAX = 16;
constexpr int PAIR_TILE_CAP = PAIR_ROWS_MAX * 64;

// The pair work list is S sub-lists, one per segment: sub-list tr occupies slots [tr * cap_seg, (tr + 1) * cap_seg) of pair_work
// and has its own counter pair_work_n[tr] (pair_work_n[S] is the cursor of the solve waves).  One global counter was hit by
// ~8 000 producer waves per iteration at 256 robots, and same-address atomics serialise (each waits for its return value):
// the plane stage of k_front was bound by that, not by its work.  Consumers index the concatenation: wprefix() leaves the
// exclusive prefix of the (clamped) counts in LDS, wslot() maps an item number to its slot.
__device__ __forceinline__ int pair_work_cap_seg(const Dev& D) { return D.cap_work / D.S; }
template <bool FA = false>   // FA: the counts were written while this launch was running (asynchronous front, Dev::fa_mid): read past the caches
__device__ __forceinline__ int pair_work_prefix(const Dev& D, int* pre, int lane) {   // one wave; pre[0..S]; returns the total
  const int S = D.S, cap = pair_work_cap_seg(D);
  int run = 0;
  for (int base = 0; base < S; base += 64) {
    const int tr = base + lane;
    const int c = tr < S ? min(FA ? xf_load_i(D.pair_work_n + tr) : D.pair_work_n[tr], cap) : 0;
    int x = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (tr < S) pre[tr] = run + x - c;
    run += __shfl(x, 63);
  }
  if (lane == 0) pre[S] = run;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
  return run;
}
__device__ __forceinline__ int pair_work_slot(const Dev& D, const int* pre, int w) {
  int lo = 0, hi = D.S;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= w) lo = mid; else hi = mid; }
  return lo * pair_work_cap_seg(D) + (w - pre[lo]);
}

// Broad phase of one tile, shared by the plane stage (hull boxes / hull intervals) and the CCD selection (swept boxes / swept
// intervals).  Box test: lanes over the 64 partners (their boxes in registers, loaded once), loop over the R rows (boxes
// broadcast from LDS); hits are appended to list[] as (p0 << 16 | p1) in (p0, p1) order.  Then, per box survivor, lanes over the
// 49 axes against both robots' interval records; the records of FOUR pairs are fetched together -- the loop is a chain of
// memory latencies, not of arithmetic -- and the list is compacted in place.  Same comparisons as the one-pair-at-a-time
// forms (CCD.h:535-587), hence the same decisions.  Returns the number of pairs left in list[].
//   segbox  [6][U] lo.xyz, hi.xyz of every robot for this segment      rec(q) -> robot q's record, intervals lo at LO, hi at HI
//   [own0, own1): robots of this rank -- a pair is skipped unless it touches one (sharded contexts; pass 0, U otherwise)
template <bool FA = false, class RecOf>   // FA (asynchronous front): boxes and records were written (through) by blocks of a kernel that runs at the same time -- read past the caches
__device__ __forceinline__ int pair_tile_filter(const double* segbox, int U, int rb, int R, int cb, int own0, int own1, RecOf rec, int LO, int HI, double d,
                                                double* rowbox, int* list, int lane) {
  auto ldd = [&](const double* p) { if constexpr (FA) return xf_load(p); else return *p; };
  for (int i = lane; i < R * 6; i += 64) rowbox[i] = ldd(segbox + (size_t)(i % 6) * U + min(rb + i / 6, U - 1));
  const int p1 = cb + lane;
  const int pc = min(p1, U - 1);
  double bv[6];
#pragma unroll
  for (int k = 0; k < 6; k++) bv[k] = ldd(segbox + (size_t)k * U + pc);   // independent loads, no branch between them
  const bool own1p = p1 >= own0 && p1 < own1;
  blk_sync<true>();
  int n = 0;
  for (int r = 0; r < R; r++) {
    const int p0 = rb + r;
    if (p0 >= U - 1) break;   // uniform
    bool hit = p1 < U && p1 > p0 && (own1p || (p0 >= own0 && p0 < own1));
#pragma unroll
    for (int k = 0; k < 3; k++) hit = hit & !((bv[3 + k] + d < rowbox[6 * r + k]) | (bv[k] > rowbox[6 * r + 3 + k] + d));
    const unsigned long long mask = ballot(hit);
    if (hit) list[n + prefix_count(mask)] = (p0 << 16) | p1;
    n += __popcll(mask);
  }
  blk_sync<true>();
  if (n == 0) return 0;
  const int ax = min(lane, 48);
  int m = 0;
  for (int base = 0; base < n; base += PT_FLIGHT) {
    int pr[PT_FLIGHT]; double alo[PT_FLIGHT], ahi[PT_FLIGHT], blo[PT_FLIGHT], bhi[PT_FLIGHT];
#pragma unroll
    for (int c = 0; c < PT_FLIGHT; c++) pr[c] = list[min(base + c, n - 1)];
#pragma unroll
    for (int c = 0; c < PT_FLIGHT; c++) {
      const double* a = rec(pr[c] >> 16); const double* b = rec(pr[c] & 0xffff);
      alo[c] = ldd(a + LO + ax); ahi[c] = ldd(a + HI + ax); blo[c] = ldd(b + LO + ax); bhi[c] = ldd(b + HI + ax);
    }
    blk_sync<true>();   // every lane has read this batch's list entries before the compaction overwrites earlier slots
#pragma unroll
    for (int c = 0; c < PT_FLIGHT; c++) {
      if (base + c >= n) break;
      const bool sep = lane < 49 && (bhi[c] < alo[c] - d || ahi[c] < blo[c] - d);
      if (ballot(sep) == 0ull) {
        if (lane == 0) list[m] = pr[c];
        m++;
      }
    }
  }
  blk_sync<true>();
  return m;
}

constexpr int PAIR_LDS_DOUBLES = PAIR_ROWS_MAX * 6 + PAIR_TILE_CAP / 2;   // rowbox | list
template <bool FA = false>   // FA (asynchronous front): the work items go out written through
__device__ __forceinline__ void sep_self_rows_body(const Dev& D, int bid, double* lds, bool wait_xf = false) {
  int tr, rb, cb;
  pair_unit(D.U, D.pair_rows, bid, tr, rb, cb);
  const int lane = lane_id();
  const int U = D.U;
  double* rowbox = lds; int* list = (int*)(lds + PAIR_ROWS_MAX * 6);
  const double dist = D.offset + 2 * D.margin;
  if (wait_xf) xf_wait_seg(D, 0, tr);   // sharded contexts (union kernel): the hull cache of the other ranks' robots is written by units at the head of this launch
  const int m = pair_tile_filter<FA>(D.hbox + (size_t)tr * 6 * U, U, rb, D.pair_rows, cb, D.u0, D.u1,   // (FA: boxes and interval records read past the caches -- a box line holds sixteen robots' entries, written by sixteen units)
                                 [&](int q) { return D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE; }, 24, 73, dist, rowbox, list, lane);
  if (m == 0) return;
  // hand the remaining pairs to the solve stage (k_mid / k_sep_self_solve): one work item per pair.  Solving them here,
  // one lane per pair, serialises up to 63 divergent GJK paths in one wave where many robots meet (measured: 87 us).
  int base = 0;
  if (lane == 0) base = atomicAdd(D.pair_work_n + tr, m);   // the segment's own counter
  base = __shfl(base, 0);
  const int cap = pair_work_cap_seg(D);
  for (int i = lane; i < m; i += 64) {
    const int w = base + i;
    if (w < cap) {
      const size_t sl = (size_t)tr * cap + w;
      if constexpr (FA) { xf_store_i(D.pair_work + 3 * sl, tr); xf_store_i(D.pair_work + 3 * sl + 1, list[i] >> 16); xf_store_i(D.pair_work + 3 * sl + 2, list[i] & 0xffff); }
      else { D.pair_work[3 * sl] = tr; D.pair_work[3 * sl + 1] = list[i] >> 16; D.pair_work[3 * sl + 2] = list[i] & 0xffff; }
    }
    else atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW);
  }
}

// ---- GJK head start ------------------------------------------------------------------------------------------------------
// Once the work list is spread evenly (sep_self_solve_body), k_mid is as long as its slowest robot pair: a pair of stacked hulls
// whose GJK takes 10 - 16 iterations of ~1.3 us while the machine idles; k_front in front of it lasts ~10 us whatever the pairs
// do.  In the steady phase of a run the slow pairs are the same from one iteration to the next (tests/devtools/
// gjk_persistence.py: 17 of 18 on SCN-C; in the first twenty iterations one to two thirds), and a GJK query is a pure function
// of the two hulls, which are final when k_front starts (hull cache).  So: k_mid lists the pairs whose query took at least
// max(SPEC_GJK_MIN, longest of the previous launch - SPEC_GJK_WINDOW) iterations; the next k_front carries SPEC_CAP extra
// one-wave blocks at the head of its grid, block b runs the first SPEC_GJK_BUDGET iterations of entry b's query (three: the
// blocks then end with the obstacle queries) and stores the loop state (GjkState, 20 doubles + 9 ints) under a tag (epoch, pair);
// wave b of k_mid is dedicated to entry b and continues the same loop at t = 0 of that kernel.  Same instructions on the same
// operands in the same order: the witness vector and the iteration count are those of an uninterrupted query (test: head start
// on / off, bitwise).  A listed pair that left the broad phase costs two waves that publish nothing; the list's order is free.
constexpr int SPEC_CAP = 128, SPEC_GJK_MIN = 5, SPEC_GJK_WINDOW = 10, SPEC_GJK_BUDGET = 3;
constexpr int SPEC_STATE_DOUBLES = 20, SPEC_STATE_INTS = 16;
constexpr unsigned SPEC_KEY_NONE = 0xffffffffu;   // asynchronous front: "this entry's block has run for the epoch in the upper half and has no pair" (a plain 0 could be a tag not written yet)
__device__ __forceinline__ unsigned pair_key(int tr, int p0, int q) { return (unsigned)(tr | (p0 << 9) | (q << 20)); }   // 9 + 11 + 11 bits
// FA (asynchronous front): the launch runs next to the k_linesearch that commits the control nets -- the epoch is the begun iteration's (fa_early_begin's record, the
// control block still shows the running one), the block waits for its two robots' commit flags and forms the hulls from the nets read past the caches, and what
// it leaves for k_mid goes out written through.
template <bool FA = false>
__device__ __forceinline__ void spec_pair_body(const Dev& D, int b, double* lds, int fa_epoch = 0) {
  __builtin_amdgcn_s_setprio(3);   // the longest blocks of the launch, on SIMDs they share with four or five query waves
  const int lane = lane_id();
  const int epoch = FA ? fa_epoch : D.ctl->epoch, par = epoch & 1;
  auto tag_out = [&](unsigned long long v) { if constexpr (FA) __hip_atomic_store(D.spec_tag + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else D.spec_tag[b] = v; };
  if (b == 0 && lane == 0) { if constexpr (FA) xf_store_i(D.spec_n + par, 0); else D.spec_n[par] = 0; }   // the list this iteration's k_mid fills
  const int n = min(D.spec_n[par ^ 1], SPEC_CAP);
  // every block leaves a tag, valid or not: the tags k_mid sees are always those of the k_front in front of it
  if (b >= n) { if (lane == 0) tag_out(FA ? (((unsigned long long)(unsigned)epoch << 32) | SPEC_KEY_NONE) : 0ull); return; }
  const unsigned key = (unsigned)D.spec_list[(par ^ 1) * SPEC_CAP + b];
  const int tr = (int)(key & 0x1ff), p0 = (int)((key >> 9) & 0x7ff), q = (int)((key >> 20) & 0x7ff);
  if (tr >= D.S || p0 >= D.U || q >= D.U) { if (lane == 0) tag_out(FA ? (((unsigned long long)(unsigned)epoch << 32) | SPEC_KEY_NONE) : 0ull); return; }   // (cannot happen: the list is this context's own)
  const double* A = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_INFO_STRIDE;
  const double* B = D.hullinfo + ((size_t)q * D.S + tr) * HULL_INFO_STRIDE;
  if constexpr (FA) {   // the records are written by this launch's obstacle units (later in the grid: not waited for): the two hulls from the nets, behind the robots' commit flags, read past the caches
    const int r = lane < 18 ? p0 : q, e = min(lane < 18 ? lane : lane - 18, 17);
    double bk[6];
    const double* Bs = D.basis + (size_t)tr * 36 + (e / 3) * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) bk[k] = Bs[k];
    const double* col = D.spline + (size_t)r * 3 * D.T + div_small(tr, D.res) * 3 + D.T * (e % 3);
    fa_wait_flag(D, D.fa_commit(p0), D.fa_seq);
    fa_wait_flag(D, D.fa_commit(q), D.fa_seq);
    if (lane < 36) {
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) acc += bk[k] * xf_load(col + k);   // = hull_entry
      lds[lane] = acc;
    }
    blk_sync<true>();
    A = lds; B = lds + 18;
  } else
  if (D.xf_all || D.fa_units) {   // coupled chain: the records are being written by this launch's obstacle units (LATER in the grid: not waited for) -- the two hulls straight from the control nets
    if (lane < 36) { const int r = lane < 18 ? p0 : q, e = lane < 18 ? lane : lane - 18; lds[lane] = hull_entry(D, D.spline + (size_t)r * 3 * D.T, tr, e / 3, e % 3); }
    blk_sync<true>();
    A = lds; B = lds + 18;
  } else if (D.xf && (p0 < D.u0 || p0 >= D.u1 || q < D.u0 || q >= D.u1)) xf_wait_seg(D, 0, tr);   // sharded contexts: a hull of another rank's robot comes from this launch's foreign units
  GjkState st; bool fin;
  gjk_wave_run(BodyHull{A}, BodyHull{B}, lane, st, true, D.spec_budget, fin);
  if (lane == 0) {
    double* o = D.spec_state + (size_t)b * SPEC_STATE_DOUBLES;
    const double ov[SPEC_STATE_DOUBLES] = {st.v.x, st.v.y, st.v.z, st.s.v0.x, st.s.v0.y, st.s.v0.z, st.s.v1.x, st.s.v1.y, st.s.v1.z, st.s.v2.x, st.s.v2.y, st.s.v2.z,
                                           st.s.v3.x, st.s.v3.y, st.s.v3.z, st.s.l0, st.s.l1, st.s.l2, st.s.l3, st.wmax2};
    int* oi = D.spec_sti + (size_t)b * SPEC_STATE_INTS;
    const int iv[9] = {st.s.n, st.s.w0, st.s.w1, st.s.w2, st.s.w3, st.c1, st.c2, st.k, fin ? 1 : 0};
#pragma unroll
    for (int i = 0; i < SPEC_STATE_DOUBLES; i++) { if constexpr (FA) xf_store(o + i, ov[i]); else o[i] = ov[i]; }
#pragma unroll
    for (int i = 0; i < 9; i++) { if constexpr (FA) xf_store_i(oi + i, iv[i]); else oi[i] = iv[i]; }
  }
  if constexpr (FA) sig_acked();   // asynchronous front: the dedicated wave of k_mid polls the TAG and then reads the state -- the state is in memory before the tag goes out
  if (lane == 0) tag_out(((unsigned long long)(unsigned)epoch << 32) | key);
  if constexpr (FA) sig_sent();
}
// the state entry b holds, fetched by the whole wave (one load per lane, then broadcasts).  The values go straight back into
// vector registers: left in scalar ones (50 of them, live across the merge with the fresh query's start) they cost k_mid a stack frame.
__device__ __forceinline__ double spec_bcast(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  int vlo, vhi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(vlo), "=v"(vhi) : "s"(lo), "s"(hi));
  return __hiloint2double(vhi, vlo);
}
template <bool FA = false>
__device__ __forceinline__ void spec_state_load(const Dev& D, int b, int lane, GjkState& st, bool& fin) {
  const double* px_ = D.spec_state + (size_t)b * SPEC_STATE_DOUBLES + min(lane, SPEC_STATE_DOUBLES - 1);
  const int* py_ = D.spec_sti + (size_t)b * SPEC_STATE_INTS + (lane & (SPEC_STATE_INTS - 1));
  const double x = FA ? xf_load(px_) : *px_;
  const int y = FA ? xf_load_i(py_) : *py_;
  st.v = V3{spec_bcast(x, 0), spec_bcast(x, 1), spec_bcast(x, 2)};
  st.s.v0 = V3{spec_bcast(x, 3), spec_bcast(x, 4), spec_bcast(x, 5)}; st.s.v1 = V3{spec_bcast(x, 6), spec_bcast(x, 7), spec_bcast(x, 8)};
  st.s.v2 = V3{spec_bcast(x, 9), spec_bcast(x, 10), spec_bcast(x, 11)}; st.s.v3 = V3{spec_bcast(x, 12), spec_bcast(x, 13), spec_bcast(x, 14)};
  st.s.l0 = spec_bcast(x, 15); st.s.l1 = spec_bcast(x, 16); st.s.l2 = spec_bcast(x, 17); st.s.l3 = spec_bcast(x, 18); st.wmax2 = spec_bcast(x, 19);
  st.s.n = __builtin_amdgcn_readlane(y, 0); st.s.w0 = __builtin_amdgcn_readlane(y, 1); st.s.w1 = __builtin_amdgcn_readlane(y, 2);
  st.s.w2 = __builtin_amdgcn_readlane(y, 3); st.s.w3 = __builtin_amdgcn_readlane(y, 4);
  st.c1 = __builtin_amdgcn_readlane(y, 5); st.c2 = __builtin_amdgcn_readlane(y, 6); st.k = __builtin_amdgcn_readlane(y, 7);
  fin = __builtin_amdgcn_readlane(y, 8) != 0;
}

__global__ __launch_bounds__(64) void k_sep_self_rows(Dev D) {
  if (TJ_DONE(D)) return;
  __shared__ double lds[PAIR_LDS_DOUBLES];
  sep_self_rows_body(D, blockIdx.x, lds);
}

// one wavefront per robot pair (stride over the work list: wave bid of nwaves), solved cooperatively by its lanes (plane_pair_wave)
// tile (large fleets, one pair per lane): PAIR_TILE_DOUBLES doubles of LDS in which every lane keeps its pair's two hulls, transposed (entry e of lane l at
// tile[e * lpw + l]: conflict-free).  The per-lane GJK and the offset Newton read a hull entry dozens of times; from global memory that was 36 scattered
// 8-byte loads per lane and GJK iteration -- the producers' GJK phase took 4 ... 58 us depending on what else the memory pipeline was doing (phase stamps,
// round 5), from LDS it does not depend on it.
constexpr int PAIR_TILE_LANES = 32, PAIR_TILE_DOUBLES = 36 * PAIR_TILE_LANES;
// FA (asynchronous front, Dev::fa_mid): the k_front of the same iteration ended while THIS launch was already running -- its work lists, head-start states and hull records are read past the caches
template <bool FA = false>
__device__ __forceinline__ void sep_self_solve_body(const Dev& D, int bid, int nwaves, bool head_start, double* tile) {   // head_start: k_mid only -- the k_front of the same iteration ran in front of it
  const int lane = lane_id();
  __shared__ double A[18], B[18];
  __shared__ int wpre[513];
  auto ldi = [&](const int* p) { if constexpr (FA) return xf_load_i(p); else return *p; };
  auto ldd = [&](const double* p) { if constexpr (FA) return xf_load(p); else return *p; };
  auto ldt = [&](const unsigned long long* p) { if constexpr (FA) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p; };
  // (SPEC_CAP = 128: two tags per lane) issued first, used after the work item has arrived; the list takes the pairs within
  // SPEC_GJK_WINDOW iterations of the previous launch's longest query, so that it holds the tail and not the first to report
  // Asynchronous front (FA): this launch started while k_front still runs.  A wave that may be DEDICATED to a head-start entry does not wait for k_front's end to begin: it
  // polls its entry's tag (the entry's k_front block leaves it behind its acknowledged state -- a few microseconds after the pair's two robots have committed), and runs the
  // rest of the pair's GJK and the offset Newton at once, from the saved state and the two hulls formed from the committed control nets (hull_entry's sums: the records' bits).
  // Only then does it wait -- like every other solve wave -- for k_front's end, to see whether the broad phase listed the pair again, and publishes.  The slow pairs of the
  // previous iteration, which set this kernel's length, thus run under k_front's tail.
  bool early = false, e_okp = false, e_capped = false; double e_e0 = 0, e_e1 = 0, e_e2 = 0, e_dpl = 0; int e_nit = 0, e_gk = 0;
  if constexpr (FA) {
    const int epoch_now = D.ctl->epoch;
    if (head_start && bid < min(nwaves / 2, SPEC_CAP)) {
      unsigned long long tg = 0;
      wait_begin();
      {
        const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;
        for (;;) {
          tg = __hip_atomic_load(D.spec_tag + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(tg >> 32) == epoch_now) break;
          if (wall_clock64() > t_end) { if (lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); tg = ((unsigned long long)(unsigned)epoch_now << 32) | SPEC_KEY_NONE; break; }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      wait_end();
      const unsigned key = (unsigned)tg;
      const int tr = (int)(key & 0x1ff), p0 = (int)((key >> 9) & 0x7ff), q = (int)((key >> 20) & 0x7ff);
      if (key != SPEC_KEY_NONE && tr < D.S && p0 < D.U && q < D.U) {
        __builtin_amdgcn_s_setprio(3);
        if (lane < 36) {
          const int r = lane < 18 ? p0 : q, e = lane < 18 ? lane : lane - 18;
          const double* Bs = D.basis + (size_t)tr * 36 + (e / 3) * 6;
          const double* col = D.spline + (size_t)r * 3 * D.T + div_small(tr, D.res) * 3 + D.T * (e % 3);
          double acc = 0;
#pragma unroll
          for (int k = 0; k < 6; k++) acc += Bs[k] * xf_load(col + k);   // = hull_entry
          if (lane < 18) A[lane] = acc; else B[lane - 18] = acc;
        }
        GjkState hs; bool hs_fin = false;
        spec_state_load<true>(D, bid, lane, hs, hs_fin);
        __syncthreads();
        e_okp = plane_pair_wave(A, B, D.offset + 2 * D.margin, D.margin, D.offset, lane, e_e0, e_e1, e_e2, e_dpl, e_capped, &e_nit, &e_gk, nullptr, hs, true, hs_fin);
        early = true;
        __builtin_amdgcn_s_setprio(0);
      }
    }
    fa_wait_flag(D, D.fa_go(bid), D.fa_seq);   // k_front is through: its work lists, tags and records are complete (read past the caches below)
  }
  const unsigned long long spec_tag0 = head_start ? ldt(D.spec_tag + lane) : 0ull, spec_tag1 = head_start ? ldt(D.spec_tag + 64 + lane) : 0ull;
  const int spec_thr = head_start ? max(D.spec_min, D.ctl->gjk_prev - SPEC_GJK_WINDOW) : 0;
  const int n = pair_work_prefix<FA>(D, wpre, lane), U = D.U;
  const double dist = D.offset + 2 * D.margin, m = D.margin, off = D.offset;
  const int epoch = D.ctl->epoch;
  // Long work list (hundreds of robots): a wave per pair is the lowest LATENCY but occupies 64 lanes for one chain of
  // dependent steps; with thousands of pairs THROUGHPUT decides, and one pair per LANE (per-lane GJK + Newton, the same
  // arithmetic: plane_pair == plane_pair_wave bit for bit) is ~20x cheaper per pair.  The switch is wave-uniform.
  if (n > 4 * nwaves) {
    // Almost every pair is through after one or two GJK iterations, a few dozen per iteration need 10-16 (measured on the
    // 256-robot scene: 98 % <= 2, 0.2 % >= 10) -- and a wave is as slow as its slowest lane, at ~2.8 us per per-lane iteration:
    // 45 us of a lone lane with the rest of the machine idle.  When the list leaves at least half of the launched waves
    // without a chunk, the lanes stop after PAIR_LANE_GJK_CAP iterations and pass an unfinished pair on (one 64-bit word
    // tagged with the iteration's epoch, appended to a device-wide list); the idle waves take these pairs one each and solve
    // them from the start with the wave-cooperative form, which reaches the same bits (plane_pair == plane_pair_wave).
    // Cross-wave traffic is agent-scope atomics only, and every waiting wave polls a word of its OWN (consumer c takes the
    // entries c, c + nc, ...): ~900 waves polling shared counters serialise at the memory side (~13 ns per access to one
    // address) and starve the producers -- measured.  The producer that counts itself done last writes nc STOP entries behind
    // the list, so every consumer's next word turns valid.  An append's returning add has been performed before its wave
    // counts itself done; no fence (= no L2 write-back) is needed anywhere.
    const int lpw = min(D.pair_lpw, PAIR_TILE_LANES);   // pairs (= active lanes) per producer wave (TJ_PAIR_LPW; 64 .. 8 lanes measured alike without the tile)
    const int np = min(nwaves, (n + lpw - 1) / lpw);      // waves that own a chunk of the list ("producers")
    const int nc = min(nwaves - np, PAIR_CONSUMERS_MAX);
    const bool pass_on = D.pair_pass_on && 2 * np <= nwaves;
    const int kcap = pass_on ? PAIR_LANE_GJK_CAP : 50;
    if (pass_on && bid >= np) {   // ---- consumer: the long solves, one per wave ----
      if (bid - np >= nc) return;
      TJ_TIC(D, K_SEP_SELF_SOLVE, 0);
      const long long t_end = wall_clock64() + 500000 + 100ll * nwaves;   // 5 ms + 1 us per launched wave (a profiler or a shared GPU stretches a large launch): a logic error must not hang the device
      for (int i = bid - np;; i += nc) {
        unsigned long long e = 0;
        bool have = false;
        if (i >= D.cap_work + PAIR_CONSUMERS_MAX) return;
        for (;;) {
          e = __hip_atomic_load(&D.pair_ovf_list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(e >> 32) == epoch) { have = true; break; }
          if (wall_clock64() > t_end) { if (lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PASS_TIMEOUT); break; }
          __builtin_amdgcn_s_sleep(16);
        }
        if (!have || (e & PAIR_OVF_STOP)) { TJ_TIC(D, K_SEP_SELF_SOLVE, 2); return; }
        if (i == bid - np) TJ_TIC(D, K_SEP_SELF_SOLVE, 1);
        const int tr = (int)(e & 0x1ff), p0 = (int)((e >> 9) & 0x7ff), q = (int)((e >> 20) & 0x7ff);   // 9 + 11 + 11 bits, bit 31 = PAIR_OVF_STOP
        __syncthreads();
        if (lane < 18) { A[lane] = ldd(D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE + lane); B[lane] = ldd(D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE + lane); }
        __syncthreads();
        double e0, e1c, e2c, dpl; bool capped; int nit = 0, gkc = 0;
        const bool okp = plane_pair_wave(A, B, dist, m, off, lane, e0, e1c, e2c, dpl, capped, &nit, &gkc);
        if (lane == 0 && gkc >= 6) atomicMax(&D.ctl->gjk_max, gkc);
        if (okp && lane == 0) {
          unsigned long long* ps = D.pair_stats + 2 * ((size_t)p0 * D.S + tr);
          atomicAdd(ps, (unsigned long long)nit); atomicAdd(ps + 1, 1ull);
          if (capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
          const size_t s0 = ((size_t)tr * U + p0) * U + q, s1 = ((size_t)tr * U + q) * U + p0;
          double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
          q0[0] = e0; q0[1] = e1c; q0[2] = e2c; q0[3] = dpl - 0.5 * off;
          q1[0] = -e0; q1[1] = -e1c; q1[2] = -e2c; q1[3] = -dpl - 0.5 * off;
          D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch; D.pair_mark(tr, p0, q); D.pair_mark(tr, q, p0);
        }
      }
    }
    unsigned long long nit_sum = 0, solved = 0; bool any_capped = false;
    TJ_TIC(D, K_SEP_SELF_SOLVE, 0);
    if (D.pair_prio) __builtin_amdgcn_s_setprio(3);   // the ~160 producer waves set k_mid's length at hundreds of robots: ahead of whatever shares their SIMD (TJ_PAIR_PRIO)
    for (int base = bid * lpw; base < n; base += nwaves * lpw) {
      const int w = base + lane;
      if (w < n && lane < lpw) {
        const size_t sl = (size_t)pair_work_slot(D, wpre, w);
        const int tr = ldi(D.pair_work + 3 * sl), p0 = ldi(D.pair_work + 3 * sl + 1), q = ldi(D.pair_work + 3 * sl + 2);
        const double* Ag = D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE;
        const double* Bg = D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE;
        constexpr int hst = PAIR_TILE_LANES;
        {   // this lane's column of the tile (private to the lane: no synchronisation)
          double* col = tile + lane;
#pragma unroll 6
          for (int e = 0; e < 18; e++) { col[e * hst] = ldd(Ag + e); col[(18 + e) * hst] = ldd(Bg + e); }
          Ag = col; Bg = col + 18 * hst;
        }
        double e0, e1c, e2c, dpl; bool capped; int nit = 0;
        int gkl = 0; bool cut = false;
        const V3 vw = gjk(BodyHullT<hst>{Ag}, BodyHullT<hst>{Bg}, &gkl, kcap, &cut);
#ifdef TJ_PHASE_TIMING
        if (!cut) atomicAdd((unsigned long long*)&D.dbg[((size_t)K_SEP_SELF_ROWS * TJ_TIC_BLOCKS + min(gkl, 63)) * TJ_TIC_SLOTS], 1ull);   // histogram of GJK iterations per pair (lane path)
#endif
        TJ_ORDER(vw.x); TJ_TIC(D, K_SEP_SELF_SOLVE, 1);
        if (!cut && gkl >= 6) atomicMax(&D.ctl->gjk_max, gkl);
        if (cut) {   // pass the pair on NOW (the consumers start while this wave's other lanes refine their offsets): slot from a returning add, then the tagged entry
          const int slot = atomicAdd(&D.pair_ovf[0], 1);
          if (slot < D.cap_work) __hip_atomic_store(&D.pair_ovf_list[slot], ((unsigned long long)(unsigned)epoch << 32) | (unsigned long long)(tr | (p0 << 9) | (q << 20)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW);   // cannot happen (the list holds cap_work entries and there are at most that many pairs); never drop a pair silently
        }
        if (!cut && plane_pair_finish(vw, Ag, Bg, dist, m, off, true, e0, e1c, e2c, dpl, capped, &nit, hst)) {
          nit_sum += (unsigned long long)nit; solved++; any_capped = any_capped || capped;
          const size_t s0 = ((size_t)tr * U + p0) * U + q, s1 = ((size_t)tr * U + q) * U + p0;
          double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
          q0[0] = e0; q0[1] = e1c; q0[2] = e2c; q0[3] = dpl - 0.5 * off;
          q1[0] = -e0; q1[1] = -e1c; q1[2] = -e2c; q1[3] = -dpl - 0.5 * off;
          D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch; D.pair_mark(tr, p0, q); D.pair_mark(tr, q, p0);
        }
      }
    }
    TJ_TIC(D, K_SEP_SELF_SOLVE, 3);
    if (pass_on) {   // this producer's appends have all been performed (their adds returned); the last producer closes the list
      int last = 0;
      if (lane == 0) last = atomicAdd(&D.pair_ovf[1], 1) == np - 1;
      if (__shfl(last, 0)) {
        const int cnt = min(__hip_atomic_load(&D.pair_ovf[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), D.cap_work);
        for (int j = lane; j < nc; j += 64) __hip_atomic_store(&D.pair_ovf_list[cnt + j], ((unsigned long long)(unsigned)epoch << 32) | PAIR_OVF_STOP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (int o = 32; o > 0; o >>= 1) { nit_sum += __shfl_xor(nit_sum, o); solved += __shfl_xor(solved, o); }
    if (lane == 0 && solved) { unsigned long long* ps = D.pair_stats + 2 * (size_t)(bid % (D.U * D.S)); atomicAdd(ps, nit_sum); atomicAdd(ps + 1, solved); }
    if (any_capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
    return;
  }
  // ---- one wave per pair -------------------------------------------------------------------------------------------------
  // SCN-C lists ~3 000 pairs per iteration for 1 024 - 1 728 waves: a wave solves two or three, and what it pays per pair besides
  // the arithmetic (2 - 6 us) were three dependent memory round trips (cursor -> work item -> hulls, ~0.9 us each: the lines were
  // written by another XCD a kernel earlier).  So the assignment is STATIC (wave r of the Wg generic waves takes the items r,
  // r + Wg, ...; at most five, n <= 4 nwaves here): lane j fetches the work item of the wave's j-th pair, all of them in one
  // round trip, and the hulls of the next pair travel while the current one is solved.
  // The pairs that were slow in the previous iteration do not wait in that queue: wave b of the first SPEC_CAP waves is
  // DEDICATED to head-start entry b (kernels_pairs.h: spec_pair_body) -- it needs the tag only to know its pair, starts at t = 0
  // from the saved GJK state, and publishes the plane if the broad phase listed the pair again (its segment's part of the work
  // list is scanned while the GJK runs); the generic wave that meets the pair in the list skips it.
  // At most half of the launched waves are dedicated (an entry beyond that stays with the generic wave that meets it in the list), so
  // the generic waves are never fewer than nwaves / 2 and a wave's share of the list stays <= 8 items whatever TJ_N_SOLVE / TJ_HS_MIN say.
  const int n_ded = min(nwaves / 2, SPEC_CAP);
  const bool tv0 = head_start && (int)(spec_tag0 >> 32) == epoch && (unsigned)spec_tag0 != SPEC_KEY_NONE && lane < n_ded, tv1 = head_start && (int)(spec_tag1 >> 32) == epoch && (unsigned)spec_tag1 != SPEC_KEY_NONE && 64 + lane < n_ded;
  const unsigned long long vm0 = ballot(tv0), vm1 = ballot(tv1);
  const int nd = __popcll(vm0) + __popcll(vm1);
  const bool dedicated = bid < SPEC_CAP && (((bid < 64 ? vm0 >> bid : vm1 >> (bid - 64)) & 1ull) != 0);
  auto publish = [&](bool okp, int tr, int p0, int q, double e0, double e1c, double e2c, double dpl, bool capped, int nit, int gk) {
    if (lane == 0 && head_start && gk >= spec_thr) {   // a few dozen pairs per launch: candidates for the next iteration's head start
      const int slot = atomicAdd(&D.spec_n[epoch & 1], 1);
      if (slot < SPEC_CAP) D.spec_list[(epoch & 1) * SPEC_CAP + slot] = (int)pair_key(tr, p0, q);
    }
#ifdef TJ_PHASE_TIMING
    if (lane == 0 && gk >= 4) {   // (key, iterations) of the slower queries of this launch: tests/devtools/gjk_persistence.py
      long long* base = D.dbg + (size_t)K_SEP_SELF_ROWS * TJ_TIC_BLOCKS * TJ_TIC_SLOTS;
      const unsigned long long i = atomicAdd((unsigned long long*)base, 1ull);
      if (i < 4000) { base[8 + 2 * i] = (long long)pair_key(tr, p0, q); base[9 + 2 * i] = gk; }
    }
#endif
    if (lane == 0 && gk >= 6) atomicMax(&D.ctl->gjk_max, gk);   // a handful of pairs per launch
    if (okp && lane == 0) {
      // statistics per (robot, segment): ~900 waves adding to ONE word of the control block serialise there (~13 ns each) and
      // the stores below wait for it
      unsigned long long* ps = D.pair_stats + 2 * ((size_t)p0 * D.S + tr);
      atomicAdd(ps, (unsigned long long)nit); atomicAdd(ps + 1, 1ull);
      if (capped) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
      const size_t s0 = ((size_t)tr * U + p0) * U + q, s1 = ((size_t)tr * U + q) * U + p0;
      double* q0 = D.pairplane + 4 * s0; double* q1 = D.pairplane + 4 * s1;
      q0[0] = e0; q0[1] = e1c; q0[2] = e2c; q0[3] = dpl - 0.5 * off;
      q1[0] = -e0; q1[1] = -e1c; q1[2] = -e2c; q1[3] = -dpl - 0.5 * off;
      D.pairstamp[s0] = epoch; D.pairstamp[s1] = epoch; D.pair_mark(tr, p0, q); D.pair_mark(tr, q, p0);
    }
  };
  if (dedicated) {
    __builtin_amdgcn_s_setprio(3);   // k_mid's tail: ahead of whatever shares the SIMD
    TJ_TIC(D, K_SEP_SELF_SOLVE, 0);
    const unsigned long long tg = bid < 64 ? __shfl(spec_tag0, bid) : __shfl(spec_tag1, bid - 64);
    const unsigned key = (unsigned)tg;
    const int tr = (int)(key & 0x1ff), p0 = (int)((key >> 9) & 0x7ff), q = (int)((key >> 20) & 0x7ff);
    if (tr >= D.S || p0 >= U || q >= U) return;   // (cannot happen)
    // in flight together: both hulls, the saved state, this segment's part of the work list
    GjkState hs; bool hs_fin = false;
    if (!early) {
      if (lane < 18) { A[lane] = ldd(D.hullinfo + ((size_t)p0 * D.S + tr) * HULL_STRIDE + lane); B[lane] = ldd(D.hullinfo + ((size_t)q * D.S + tr) * HULL_STRIDE + lane); }
      spec_state_load<FA>(D, bid, lane, hs, hs_fin);
    }
    const int cnt = wpre[tr + 1] - wpre[tr];
    const int* wl = D.pair_work + 3 * (size_t)tr * pair_work_cap_seg(D);
    bool member = false;
    for (int i = lane; i < cnt; i += 64) member = member || (ldi(wl + 3 * i + 1) == p0 && ldi(wl + 3 * i + 2) == q);
    __syncthreads();
    TJ_TIC(D, K_SEP_SELF_SOLVE, 1);
    double e0 = e_e0, e1c = e_e1, e2c = e_e2, dpl = e_dpl; bool capped = e_capped; int nit = e_nit, gk = e_gk;
    const bool okp = early ? e_okp : plane_pair_wave(A, B, dist, m, off, lane, e0, e1c, e2c, dpl, capped, &nit, &gk, nullptr, hs, true, hs_fin);   // (early: solved under k_front's tail, above)
#ifdef TJ_PHASE_TIMING
    if (lane == 0 && blockIdx.x < TJ_TIC_BLOCKS) { D.dbg[((size_t)K_SEP_SELF_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 6] = gk + 1000; D.dbg[((size_t)K_SEP_SELF_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 7] = okp ? nit : -1; }
#endif
    if (ballot(member)) {   // the broad phase listed the pair in this iteration too: the plane counts
      if (lane == 0) atomicAdd(&D.ctl->spec_taken, 1);
      publish(okp, tr, p0, q, e0, e1c, e2c, dpl, capped, nit, gk);
    }
    TJ_TIC(D, K_SEP_SELF_SOLVE, 2);
    return;
  }
  const int below = bid < 64 ? __popcll(vm0 & ((1ull << bid) - 1ull)) : (bid < 128 ? __popcll(vm0) + __popcll(vm1 & ((1ull << (bid - 64)) - 1ull)) : nd);
  const int r = bid - below, Wg = nwaves - nd;
  if (r >= n) return;
  TJ_TIC(D, K_SEP_SELF_SOLVE, 0);
  // lane j: the wave's j-th work item (all of them in one round trip)
  int it_tr = 0, it_p0 = 0, it_q = 0;
  const int n_mine = (n - r + Wg - 1) / Wg;   // <= 8 (n <= 4 nwaves, Wg >= nwaves / 2)
  if (n_mine > 64) { if (lane == 0) atomicOr(&D.ctl->error, ERR_PAIR_OVERFLOW); return; }   // (cannot happen: a lane holds one item)
  if (lane < n_mine) {
    const size_t sl = (size_t)pair_work_slot(D, wpre, r + lane * Wg);
    it_tr = ldi(D.pair_work + 3 * sl); it_p0 = ldi(D.pair_work + 3 * sl + 1); it_q = ldi(D.pair_work + 3 * sl + 2);
  }
  int tr = __builtin_amdgcn_readlane(it_tr, 0), p0 = __builtin_amdgcn_readlane(it_p0, 0), q = __builtin_amdgcn_readlane(it_q, 0);
  // this lane's entry of the pair's two hulls: lanes 0..17 A, 18..35 B
  auto hull_entry_of = [&](int tr_, int p0_, int q_) -> double {
    const int robot = lane < 18 ? p0_ : q_, e = lane < 18 ? lane : lane - 18;
    return lane < 36 ? ldd(D.hullinfo + ((size_t)robot * D.S + tr_) * HULL_STRIDE + e) : 0.0;
  };
  double hv = hull_entry_of(tr, p0, q);
  for (int j = 0; j < n_mine; j++) {
    const bool first = j == 0;   // phase stamps (timing build only) describe a wave's first work item
    __syncthreads();
    if (lane < 18) A[lane] = hv; else if (lane < 36) B[lane - 18] = hv;
    __syncthreads();
    const int ctr = tr, cp0 = p0, cq = q;
    if (j + 1 < n_mine) {   // the next pair's hulls travel while this one is solved
      tr = __builtin_amdgcn_readlane(it_tr, j + 1); p0 = __builtin_amdgcn_readlane(it_p0, j + 1); q = __builtin_amdgcn_readlane(it_q, j + 1);
      hv = hull_entry_of(tr, p0, q);
    }
    if (first) TJ_TIC(D, K_SEP_SELF_SOLVE, 1);
    if (head_start) {   // a dedicated wave has this pair
      const unsigned long long want = ((unsigned long long)(unsigned)epoch << 32) | pair_key(ctr, cp0, cq);
      if (ballot((tv0 && spec_tag0 == want) || (tv1 && spec_tag1 == want))) continue;
    }
    double e0, e1c, e2c, dpl; bool capped; int nit = 0, gk = 0;
    GjkState gs;
    const bool okp = plane_pair_wave(A, B, dist, m, off, lane, e0, e1c, e2c, dpl, capped, &nit, &gk,
#ifdef TJ_PHASE_LIGHT
                                     nullptr,
#else
                                     first ? &D : nullptr,
#endif
                                     gs, false, false);  // whole wave, uniform result
#ifdef TJ_PHASE_TIMING
    if (lane == 0 && first && blockIdx.x < TJ_TIC_BLOCKS) { D.dbg[((size_t)K_SEP_SELF_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 6] = gk; D.dbg[((size_t)K_SEP_SELF_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + 7] = okp ? nit : -1; }
#endif
    publish(okp, ctr, cp0, cq, e0, e1c, e2c, dpl, capped, nit, gk);
    if (first) TJ_TIC(D, K_SEP_SELF_SOLVE, 2);
  }
}
__global__ __launch_bounds__(64) void k_sep_self_solve(Dev D) {
  if (TJ_DONE(D)) return;
  __shared__ double tile[PAIR_TILE_DOUBLES];
  sep_self_solve_body(D, blockIdx.x, gridDim.x, false, tile);
}

// per (owned robot, segment): obstacle planes from the stamped candidate slots (slot order), then -- multi-robot modes --
// the robot-pair planes from the stamped partner slots (ascending partner): deterministic lists, no atomics
// The loads are ordered so that the chain is three memory latencies long, not six: epoch, candidate count and the first 256
// partner stamps go out together; then the candidate stamps and the stamped partners' planes; then the candidates' planes.
// hand-over (k_grad's folded launch): besides the global lists the wave leaves its segment's planes -- the first `pst_cap` of each
// list -- and the two counts in LDS (pst_o / pst_s: [pst_cap][4], cnt: {obstacle planes, all planes}), so that the gradient that
// follows in the same block reads neither the counts nor the lists back from global memory; *fits is cleared if a list is longer.
__device__ __forceinline__ void compact_segment(const Dev& D, int u, int tr, int lane, double* pst_o = nullptr, double* pst_s = nullptr, int* cnt = nullptr, int pst_cap = 0, int* fits = nullptr) {   // one wave
  const int U = D.U, epoch = D.ctl->epoch;
  const size_t seg = (size_t)u * D.S + tr;
  const bool obs_part = !(D.optimal_plane && !D.multi());  // single-UAV "optimal_plane":1 -- k_keep wrote the obstacle plane list itself
  const bool pair_part = D.multi();
  const int n = obs_part ? D.ocand_n[seg] : 0;
  // the row's index words (Dev::pairbits): one per 64 partners, fetched with the candidate count -- the stamps and planes of the SET bits follow in the second trip
  const int W = (U + 63) >> 6;
  unsigned long long* bw = D.pairbits + ((size_t)tr * U + u) * W;
  unsigned long long wbits = 0;   // lane w: word w (W <= 32)
  if (pair_part && lane < W) wbits = bw[lane];
  double* outp = D.splanes + seg * D.cap_self * 4;
  int pbase = 0;
  if (pair_part) {
    for (int q0 = 0; q0 < U; q0 += 64) {
      const unsigned long long word = __shfl(wbits, q0 >> 6);
      if (word == 0ull) continue;   // (uniform) nobody stamped a slot of this chunk
      const int q = q0 + lane;
      const bool bit = (word >> lane) & 1ull;
      const size_t slot = ((size_t)tr * U + u) * U + min(q, U - 1);
      int st = 0; double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
      if (bit && q < U) { st = D.pairstamp[slot]; const double* p = D.pairplane + 4 * slot; v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3]; }   // stamp and plane in one trip
      const bool ok = bit && q < U && q != u && st == epoch;
      const unsigned long long mask = ballot(ok);
      const int idx = pbase + prefix_count(mask);
      if (ok) {
        if (idx < D.cap_self) {
          outp[4 * idx] = v0; outp[4 * idx + 1] = v1; outp[4 * idx + 2] = v2; outp[4 * idx + 3] = v3;
          if (pst_s && idx < pst_cap) { pst_s[4 * idx] = v0; pst_s[4 * idx + 1] = v1; pst_s[4 * idx + 2] = v2; pst_s[4 * idx + 3] = v3; }
        }
        else atomicOr(&D.ctl->error, ERR_PLANE_OVERFLOW);
      }
      pbase += __popcll(mask);
    }
    if (lane < W && wbits != 0ull) bw[lane] = 0ull;   // read: cleared for the next iteration (its writers run in a later kernel; this wave is the row's only reader)
  }
  if (obs_part) {
    double* out = D.oplanes + seg * D.cap_obs * 4;
    int base = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
      const int sl = s0 + lane, slc = min(sl, n - 1);
      const int stamp = D.ostamp[seg * D.cap_obs + slc];
      const double* p = D.oraw + (seg * D.cap_obs + slc) * 4;
      const double o0 = p[0], o1 = p[1], o2 = p[2], o3 = p[3];   // (issued with the stamp; the 64 candidate slots fetched with the first trip, whatever the count, measured slower)
      const bool ok = sl < n && stamp == epoch;
      const unsigned long long mask = ballot(ok);
      const int idx = base + prefix_count(mask);
      if (ok) {
        out[4 * idx] = o0; out[4 * idx + 1] = o1; out[4 * idx + 2] = o2; out[4 * idx + 3] = o3;
        if (pst_o && idx < pst_cap) { pst_o[4 * idx] = o0; pst_o[4 * idx + 1] = o1; pst_o[4 * idx + 2] = o2; pst_o[4 * idx + 3] = o3; }
      }
      base += __popcll(mask);
    }
    if (lane == 0) { D.ocount[seg] = base; atomicAdd(&D.seg_stats[seg * 6 + 4], (unsigned long long)base); }   // fire-and-forget
    if (cnt && lane == 0) { cnt[0] = base; if (base > pst_cap) *fits = 0; }
  } else if (cnt && lane == 0) *fits = 0;   // ("optimal_plane":1, single UAV: k_keep wrote the obstacle list itself -- the gradient reads it from global memory)
  if (pair_part && lane == 0) {
    D.scount[seg] = min(pbase, D.cap_self);
    atomicAdd(&D.seg_stats[seg * 6 + 5], (unsigned long long)pbase);
  }
  if (cnt && lane == 0) { const int ns = pair_part ? min(pbase, D.cap_self) : 0; cnt[1] = ns; if (ns > pst_cap) *fits = 0; }
}
__global__ __launch_bounds__(64) void k_sep_self_compact(Dev D) {
  if (TJ_DONE(D)) return;
  if (D.keep_async) keep_wait(D);   // (see k_grad)
  compact_segment(D, D.u0 + blockIdx.x / D.S, blockIdx.x % D.S, lane_id());
}

}  // namespace tj
