// kernels_sep.h -- separating-plane construction ("z-update: GJK separating-plane projection"
// in BASELINE.json's vocabulary).
//
//   obstacle planes (obs_query_body / obs_solve_body, compaction in kernels_pairs.h): per (robot, Bezier segment)
//               hull -> static-BVH query -> 49-axis k-DOP cull; per candidate GJK -> plane (c,d).
//               Replaces BVH::DCDCollision (BVH.cpp:149-193),
//               aabb::Tree::query (AABB.cc:608-667), CCD::KDOPDCD (CCD.h:354-413) and
//               Separate::opengjk (Separate.h:18-163) as sequenced by separate_plane
//               (Optimization3D_multi.h:176-235 / Optimization3D_admm.h:69-197).
//   plane_pair  device function for one robot pair (hull-hull GJK + 1-D Newton on the offset),
//               used by kernels_pairs.h which replaces separate_self (Optimization3D_multi.h:237-342).
//
// The BVH is an implicit 8-ary box hierarchy over Morton-sorted primitives (points, or triangles for BASELINE config 5).
// A wave walks it level by level: 64 lanes test 8 frontier nodes x 8 children per step (coalesced 24-B fp32 boxes,
// rounded outward; 192 B per parent), survivors are compacted into the next frontier with ballot + popcount.  The tree
// shape is free: the candidate SET is defined by the reference's leaf predicate (AABB.cc:141, touching counts), which
// is evaluated here in fp64 on the primitives themselves (a point, or the exact box of a triangle's three vertices).
#pragma once
#include "dev_common.h"
#include "dev_linalg.h"
#include "dev_crmath.h"

namespace tj {

struct QBox { double lo[3], hi[3]; };
constexpr int HULL_INFO_STRIDE = 128;   // record of Dev::hullinfo (kernels_pairs.h HULL_STRIDE): hull 18, box lo/hi 6, 49 intervals lo/hi 98 = 122 values in a record of 128 doubles (eight 128-byte lines of its own, see CCD_STRIDE)

__device__ __forceinline__ bool box_hit(const float* b, const QBox& q, double m) {
  // query.overlaps(node): reject if node.hi + m < q.lo or node.lo > q.hi + m on any axis
  bool hit = true;
#pragma unroll
  for (int k = 0; k < 3; k++) hit = hit && !((double)b[3 + k] + m < q.lo[k] || (double)b[k] > q.hi[k] + m);
  return hit;
}

// ---- obstacle primitives as GJK bodies ----
template <int PRIM> struct PrimOf;
template <> struct PrimOf<1> {
  using Body = BodyPoint;
  static __device__ __forceinline__ Body load(const Dev& D, int i) { return BodyPoint{V3{D.px[i], D.py[i], D.pz[i]}}; }
};
template <> struct PrimOf<3> {
  using Body = BodyTri;
  static __device__ __forceinline__ Body load(const Dev& D, int i) {
    const double* t = D.tri + (size_t)i * 9;
    return BodyTri{V3{t[0], t[1], t[2]}, V3{t[3], t[4], t[5]}, V3{t[6], t[7], t[8]}};
  }
};

// Wave-cooperative query.  `process(pt)` is called by all 64 lanes with a candidate point index
// (or -1) and may use wave collectives.  Returns candidates found; adds visited boxes to *visits.
// BQ_UNROLL: chunks of 8 frontier nodes whose box loads are in flight together (4 in the plane query, whose kernel has
// registers to spare; 1 in the CCD query, where the per-lane GJK already fills the register file)
// The top level's boxes do not depend on the query: a caller can fetch its lane's box (bvh_top_box) TOGETHER with the record that
// holds the query box and hand it in -- one dependent round trip less at the head of the walk.
struct TopBox { float b[6]; };
constexpr int BVH_SKIP_MAX = 8;   // frontier nodes up to which a level pair is taken in one step (64 boxes per node)
__device__ __forceinline__ TopBox bvh_top_box(const Dev& D) {
  TopBox t{{0, 0, 0, 0, 0, 0}};
  if (D.N == 0) return t;
  const int top = D.nlevels - 1, lane = lane_id();
  const float* p = D.boxes + (size_t)(D.lvl_off[top] + min(lane, D.lvl_n[top] - 1)) * 6;
#pragma unroll
  for (int k = 0; k < 6; k++) t.b[k] = p[k];
  return t;
}
template <int BQ_UNROLL, int PRIM, class F>
__device__ int bvh_query(const Dev& D, const QBox& q, double m, int* fa, int* fb, int* cand, unsigned long long* visits, F&& process, const TopBox* pre = nullptr) {
  const int lane = lane_id();
  if (D.N == 0) return 0;
  int top = D.nlevels - 1;
  int count = 0;
  unsigned long long nv = 0;
  {
    const int n = D.lvl_n[top];
    bool hit = false;
    if (lane < n) hit = pre ? box_hit(pre->b, q, m) : box_hit(D.boxes + (size_t)(D.lvl_off[top] + lane) * 6, q, m);
    const unsigned long long mask = ballot(hit);
    if (hit) fa[prefix_count(mask)] = lane;
    count = __popcll(mask);
    nv += n;
  }
  if (count == 0) { if (visits) *visits += nv; return 0; }   // wave-uniform: nothing of the obstacle set is near this box (the common case)
  __syncthreads();
  int* cur = fa; int* nxt = fb;
  // DOUBLE STEPS while the frontier is small (the top of a deep pyramid: 1 M primitives are five levels): the 64 GRANDchildren of a frontier node are tested by
  // the 64 lanes at once -- one dependent round trip per two levels instead of two.  A box contains its children's boxes (unions, rounded outward monotonically),
  // so a grandchild that is hit has a hit parent: the survivors at level lv - 1 are the same nodes in the same (ascending) order as after two single steps.
  while (top >= 2 && count <= BVH_SKIP_MAX && D.bvh_skip) {
    const int lv = top - 2, nl = D.lvl_n[lv];
    const float* lvl = D.boxes + (size_t)D.lvl_off[lv] * 6;
    int ncount = 0; bool overflow = false;
    for (int base = 0; base < count && !overflow; base += BQ_UNROLL) {
      int gc[BQ_UNROLL]; bool live[BQ_UNROLL]; float bx[BQ_UNROLL][6];
#pragma unroll
      for (int c = 0; c < BQ_UNROLL; c++) {
        const int node = cur[min(base + c, count - 1)];
        gc[c] = node * 64 + lane;
        live[c] = base + c < count && gc[c] < nl;
        const float* b = lvl + (size_t)min(gc[c], nl - 1) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) bx[c][k] = b[k];
      }
#pragma unroll
      for (int c = 0; c < BQ_UNROLL; c++) {
        if (base + c >= count) break;
        bool hit = live[c];
#pragma unroll
        for (int k = 0; k < 3; k++) hit = hit & !(((double)bx[c][3 + k] + m < q.lo[k]) | ((double)bx[c][k] > q.hi[k] + m));
        const unsigned long long mask = ballot(hit);
        const int tot = __popcll(mask);
        if (ncount + tot > FRONT_CAP) { if (lane == 0) atomicOr(&D.ctl->error, ERR_FRONT_OVERFLOW); overflow = true; break; }
        if (hit) nxt[ncount + prefix_count(mask)] = gc[c];
        ncount += tot;
        nv += 64;
      }
    }
    __syncthreads();
    int* t = cur; cur = nxt; nxt = t;
    count = ncount;
    top -= 2;
    if (count == 0) break;
  }
  // Each step of the walk is a dependent global load (~0.7 us).  BQ_UNROLL chunks of 8 frontier nodes are therefore
  // fetched together: all box loads of the group are issued first (branch-free, clamped addresses), then the chunks are
  // tested and compacted one by one in frontier order (the order of the survivors, and with it of the candidate list,
  // is unchanged).
  for (int lv = top - 1; lv >= 0; lv--) {
    int ncount = 0;
    const int nl = D.lvl_n[lv];
    const float* lvl = D.boxes + (size_t)D.lvl_off[lv] * 6;
    bool overflow = false;
    for (int base = 0; base < count && !overflow; base += 8 * BQ_UNROLL) {
      int child[BQ_UNROLL]; bool live[BQ_UNROLL]; float bx[BQ_UNROLL][6];
#pragma unroll
      for (int c = 0; c < BQ_UNROLL; c++) {
        const int slot = base + 8 * c + (lane >> 3);
        const int node = cur[min(slot, count - 1)];
        child[c] = node * 8 + (lane & 7);
        live[c] = slot < count && child[c] < nl;
        const float* b = lvl + (size_t)min(child[c], nl - 1) * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) bx[c][k] = b[k];
      }
#pragma unroll
      for (int c = 0; c < BQ_UNROLL; c++) {
        if (base + 8 * c >= count) break;
        bool hit = live[c];
#pragma unroll
        for (int k = 0; k < 3; k++) hit = hit & !(((double)bx[c][3 + k] + m < q.lo[k]) | ((double)bx[c][k] > q.hi[k] + m));
        const unsigned long long mask = ballot(hit);
        const int tot = __popcll(mask);
        if (ncount + tot > FRONT_CAP) { if (lane == 0) atomicOr(&D.ctl->error, ERR_FRONT_OVERFLOW); overflow = true; break; }
        if (hit) nxt[ncount + prefix_count(mask)] = child[c];
        ncount += tot;
        nv += 8 * min(8, count - (base + 8 * c));
      }
    }
    __syncthreads();
    int* t = cur; cur = nxt; nxt = t;
    count = ncount;
  }
  if constexpr (BQ_UNROLL == 4) TJ_TIC(D, K_SEP_OBS, 2);   // timing build: the plane query's walk is done, leaves next
  int nc = 0, found = 0;
  for (int base = 0; base < count; base += 8 * BQ_UNROLL) {
    int pts[BQ_UNROLL]; bool live[BQ_UNROLL]; double px[BQ_UNROLL], py[BQ_UNROLL], pz[BQ_UNROLL]; float lb[BQ_UNROLL][6];
#pragma unroll
    for (int c = 0; c < BQ_UNROLL; c++) {  // leaf boxes -> primitives: again all loads of the group first
      const int slot = base + 8 * c + (lane >> 3);
      const int pt = cur[min(slot, count - 1)] * 8 + (lane & 7);
      pts[c] = pt; live[c] = slot < count && pt < D.N;
      const int pc = min(pt, D.N - 1);
      if constexpr (PRIM == 1) { px[c] = D.px[pc]; py[c] = D.py[pc]; pz[c] = D.pz[pc]; }
      else {
#pragma unroll
        for (int k = 0; k < 6; k++) lb[c][k] = D.leafbox[(size_t)pc * 6 + k];
      }
    }
#pragma unroll
    for (int c = 0; c < BQ_UNROLL; c++) {
      if (base + 8 * c >= count) break;
      bool hit;
      if constexpr (PRIM == 1) {
        const double x = px[c], y = py[c], z = pz[c];
        hit = live[c] & !((x + m < q.lo[0]) | (x > q.hi[0] + m)) & !((y + m < q.lo[1]) | (y > q.hi[1] + m)) & !((z + m < q.lo[2]) | (z > q.hi[2] + m));
      } else {
        // fp32 pre-filter (conservative), then the exact box of the three vertices in fp64 (BVH::InitObstacle, BVH.cpp:26-46)
        bool pre = live[c];
#pragma unroll
        for (int k = 0; k < 3; k++) pre = pre & !(((double)lb[c][3 + k] + m < q.lo[k]) | ((double)lb[c][k] > q.hi[k] + m));
        hit = false;
        if (pre) {
          const double* t = D.tri + (size_t)pts[c] * 9;
          hit = true;
#pragma unroll
          for (int k = 0; k < 3; k++) {
            double lo = INFINITY, hi = -INFINITY;
#pragma unroll
            for (int j = 0; j < 3; j++) { const double lv = t[3 * j + k]; if (lv < lo) lo = lv; if (lv > hi) hi = lv; }
            hit = hit & !((hi + m < q.lo[k]) | (lo > q.hi[k] + m));
          }
        }
      }
      const unsigned long long mask = ballot(hit);
      if (hit) cand[nc + prefix_count(mask)] = pts[c];
      nc += __popcll(mask);
      found += __popcll(mask);
      __syncthreads();
      if (nc >= 64) {
        process(cand[lane]);
        const int left = nc - 64;
        const int keep = lane < left ? cand[64 + lane] : 0;
        __syncthreads();
        if (lane < left) cand[lane] = keep;
        nc = left;
        __syncthreads();
      }
    }
  }
  if constexpr (BQ_UNROLL == 4) TJ_TIC(D, K_SEP_OBS, 3);
  if (nc > 0) process(lane < nc ? cand[lane] : -1);
  if (visits) *visits += nv;
  return found;
}

// 49-axis intervals of n points stored row-major [n][3] (CCD.h:373-390); lanes 0..48
__device__ __forceinline__ void kdop_intervals(const Dev& D, const double* pts, int n, double* klo, double* khi) {
  const int lane = lane_id();
  if (lane < 49) {
    const double x = D.kdop[3 * lane], y = D.kdop[3 * lane + 1], z = D.kdop[3 * lane + 2];
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < n; i++) {
      const double lv = x * pts[3 * i] + y * pts[3 * i + 1] + z * pts[3 * i + 2];
      if (lv < lo) lo = lv;
      if (lv > up) up = lv;
    }
    klo[lane] = lo; khi[lane] = up;
  }
}
// point vs cached hull intervals
__device__ __forceinline__ bool kdop_point_pass(const Dev& D, const double* klo, const double* khi, const V3& q, double d) {
  for (int k = 0; k < 49; k++) {
    const double lv = D.kdop[3 * k] * q.x + D.kdop[3 * k + 1] * q.y + D.kdop[3 * k + 2] * q.z;
    if (lv < klo[k] - d || khi[k] < lv - d) return false;
  }
  return true;
}

// obstacle primitive (1 or 3 vertices) vs cached hull intervals: CCD::KDOPDCD / KDOPCCD with the body-2 loop over
// _position.rows() (CCD.h:391-400; for one vertex it is kdop_point_pass)
// One candidate per LANE.  The axes come from LDS (kax, staged by stage_kdop_axes) and all 49 are evaluated without an early
// exit: with the axes in global memory and a `return false` per axis the loop was a chain of up to 49 dependent memory
// round trips -- 118 us for a segment under an obstacle slab (40 candidates), the whole tail of k_front at 256 robots.
// Same comparisons, hence the same decision.
__device__ __forceinline__ void stage_kdop_axes(const Dev& D, double* kax, int lane) {
  for (int i = lane; i < 147; i += 64) kax[i] = D.kdop[i];
  __syncthreads();
}
template <class B>
__device__ __forceinline__ bool kdop_body_pass(const double* kax, const double* klo, const double* khi, const B& body, double d) {
  bool sep = false;
  for (int k = 0; k < 49; k++) {
    const double x = kax[3 * k], y = kax[3 * k + 1], z = kax[3 * k + 2];
    double up = -INFINITY, lo = INFINITY;
#pragma unroll
    for (int i = 0; i < B::N; i++) {
      const V3 p = body.get(i);
      const double lv = x * p.x + y * p.y + z * p.z;
      if (lv < lo) lo = lv;
      if (lv > up) up = lv;
    }
    sep = sep | (up < klo[k] - d) | (khi[k] < lo - d);
  }
  return !sep;
}

// The same cull for up to 64 candidates held one per LANE (lanes [0, n)), done by the whole wave: lanes over the 49 AXES
// (axis and the hull's interval in registers), candidates one after the other with their vertices broadcast by v_readlane.
// ~20 instructions per point candidate instead of a 49-step loop per lane; returns this lane's verdict.  Same products and
// sums as kdop_body_pass, hence the same decision.  axv = this lane's axis (x, y, z), lo_ax / hi_ax = hull interval on it.
template <class B>
__device__ __forceinline__ bool kdop_cull_wave(const B& body, int n, const V3& axv, double lo_ax, double hi_ax, double d, int lane) {
  unsigned long long pass = 0ull;
  for (int i = 0; i < n; i++) {   // wave-uniform
    double up = -INFINITY, lo = INFINITY;
#pragma unroll
    for (int v = 0; v < B::N; v++) {
      const V3 p = body.get(v);
      const double lv = axv.x * readlane_f64(p.x, i) + axv.y * readlane_f64(p.y, i) + axv.z * readlane_f64(p.z, i);
      if (lv < lo) lo = lv;
      if (lv > up) up = lv;
    }
    const bool sep = lane < 49 && ((up < lo_ax - d) | (hi_ax < lo - d));
    if (ballot(sep) == 0ull) pass |= 1ull << i;
  }
  return (pass >> lane) & 1ull;
}

// Separate::opengjk (Separate.h:18-163): plane (c,d) between a 6-point hull and one cloud point
// plane (c,d) from the GJK witness vector v of hull - point (Separate.h:107-151): rejected if |v| > dist
__device__ __forceinline__ bool plane_from_witness(const V3& v, const V3& qp, double dist, double offset, double& c0, double& c1, double& c2, double& dd) {
  const double cn = norm3(v.x, v.y, v.z);
  if (cn > dist) return false;
  c0 = v.x / cn; c1 = v.y / cn; c2 = v.z / cn;
  const double d0 = -c0 * qp.x - c1 * qp.y - c2 * qp.z;
  dd = d0 - offset;
  return true;
}
// the same for a body of B::N vertices: d0 = min_i(-c . B_i), the loop the reference keeps commented out at Separate.h:123-131
template <class B>
__device__ __forceinline__ bool plane_from_witness_body(const V3& v, const B& body, double dist, double offset, double& c0, double& c1, double& c2, double& dd) {
  const double cn = norm3(v.x, v.y, v.z);
  if (cn > dist) return false;
  c0 = v.x / cn; c1 = v.y / cn; c2 = v.z / cn;
  const V3 q0 = body.get(0);
  double d0 = -c0 * q0.x - c1 * q0.y - c2 * q0.z;
#pragma unroll
  for (int i = 1; i < B::N; i++) { const V3 q = body.get(i); const double d_ = -c0 * q.x - c1 * q.y - c2 * q.z; if (d0 > d_) d0 = d_; }
  dd = d0 - offset;
  return true;
}
__device__ __forceinline__ bool plane_obstacle(const double* P, const V3& qp, double dist, double offset, double& c0, double& c1, double& c2, double& dd) {
  return plane_from_witness(gjk(BodyHull{P}, BodyPoint{qp}), qp, dist, offset, c0, c1, c2, dd);
}

__device__ __forceinline__ double dot_fixed3(double c0, double c1, double c2, const double* r) { return c0 * r[0] + (c1 * r[1] + c2 * r[2]); }  // Eigen unrolled 3-term order (Separate.h:268,276)

// 49-axis test between two 6-point hulls (CCD::SelfKDOPDCD, CCD.h:535-587)
__device__ inline bool kdop_hulls_pass(const Dev& D, const double* A, const double* Bq, double dist) {
  for (int k = 0; k < 49; k++) {
    const double x = D.kdop[3 * k], y = D.kdop[3 * k + 1], z = D.kdop[3 * k + 2];
    double upA = -INFINITY, loA = INFINITY, upB = -INFINITY, loB = INFINITY;
    for (int i = 0; i < 6; i++) {
      const double la = x * A[3 * i] + y * A[3 * i + 1] + z * A[3 * i + 2];
      if (la < loA) loA = la; if (la > upA) upA = la;
      const double lb = x * Bq[3 * i] + y * Bq[3 * i + 1] + z * Bq[3 * i + 2];
      if (lb < loB) loB = lb; if (lb > upB) upB = lb;
    }
    if (upB < loA - dist || upA < loB - dist) return false;
  }
  return true;
}

// Separate::selfgjk (Separate.h:165-304) + Optimal_plane::optimal_d (Optimal_plane.h:13-71).
// A is the hull of the lower robot index.  Returns false if the hulls are farther than dist.
// capped = true when the Newton loop hit NEWTON_CAP.
// second half of plane_pair: from the GJK witness vector to the plane (separate so that a caller can act between the halves)
// cr_log out of line: the per-lane pair path lives in k_mid at its 256-register cap, where the inlined double-double pieces spill
__device__ __noinline__ double cr_log_call(double x) { return cr_log(x); }
__device__ inline bool plane_pair_finish(const V3& v, const double* A, const double* Bq, double dist, double m, double off, bool refine, double& e0, double& e1c, double& e2c, double& dpl, bool& capped, int* newton_iters = nullptr, int st = 1);
__device__ inline bool plane_pair(const double* A, const double* Bq, double dist, double m, double off, bool refine, double& e0, double& e1c, double& e2c, double& dpl, bool& capped, int* newton_iters = nullptr, int* gjk_iters = nullptr) {
  const V3 v = gjk(BodyHull{A}, BodyHull{Bq}, gjk_iters);
  return plane_pair_finish(v, A, Bq, dist, m, off, refine, e0, e1c, e2c, dpl, capped, newton_iters);
}
// st: stride between the entries of A / Bq (1: row-major [6][3] in global memory or LDS; the lane-per-pair path of large fleets keeps a transposed tile, BodyHullT)
__device__ inline bool plane_pair_finish(const V3& v, const double* A, const double* Bq, double dist, double m, double off, bool refine, double& e0, double& e1c, double& e2c, double& dpl, bool& capped, int* newton_iters, int st) {
  capped = false;
  const double cn = norm3(v.x, v.y, v.z);
  if (cn > dist) return false;
  e0 = v.x / cn; e1c = v.y / cn; e2c = v.z / cn;
  double d0 = INFINITY, d1 = -INFINITY;
  for (int i = 0; i < 6; i++) { const double t = -(e0 * Bq[(3 * i) * st] + (e1c * Bq[(3 * i + 1) * st] + e2c * Bq[(3 * i + 2) * st])); if (d0 > t) d0 = t; }   // dot_fixed3's order (Eigen unrolled 3-term, Separate.h:268,276)
  for (int i = 0; i < 6; i++) { const double t = -(e0 * A[(3 * i) * st] + (e1c * A[(3 * i + 1) * st] + e2c * A[(3 * i + 2) * st])); if (d1 < t) d1 = t; }
  dpl = 0.5 * (d0 + d1);
  if (!refine) return true;
  int it = 0;
  for (; it < NEWTON_CAP; it++) {  // Newton on the offset until |grad| < 1e-2
    double grad = 0, hess = 0;
    for (int j = 0; j < 6; j++) {
      const double ds = (A[(3 * j) * st] * e0 + A[(3 * j + 1) * st] * e1c + A[(3 * j + 2) * st] * e2c) + dpl - 0.5 * off;
      if (ds < m) {
        const double lg = cr_log_call(ds / m);   // rounds like glibc's log (dev_crmath.h): the offset is then the reference's bit for bit
        const double g1 = -(2 * (ds - m) * lg + (ds - m) * (ds - m) / ds);
        const double g2 = -(2 * lg + 4 * (ds - m) / ds - (ds - m) * (ds - m) / (ds * ds));
        grad += g1; hess += g2;
      }
    }
    for (int j = 0; j < 6; j++) {
      const double ds = -(Bq[(3 * j) * st] * e0 + Bq[(3 * j + 1) * st] * e1c + Bq[(3 * j + 2) * st] * e2c) - dpl - 0.5 * off;
      if (ds < m) {
        const double lg = cr_log_call(ds / m);
        const double g1 = -(2 * (ds - m) * lg + (ds - m) * (ds - m) / ds);
        const double g2 = -(2 * lg + 4 * (ds - m) / ds - (ds - m) * (ds - m) / (ds * ds));
        grad += -g1; hess += g2;
      }
    }
    const double dir = -grad / hess;
    dpl = dpl + 1.0 * dir;
    if (fabs(grad) < 1e-2) break;
  }
  capped = it == NEWTON_CAP;
  if (newton_iters) *newton_iters = it + 1;
  return true;
}

// plane_pair computed by a whole wave for ONE robot pair (all 64 lanes call it with the same arguments, A and Bq in
// LDS): wave-cooperative GJK, then the Newton refinement of the offset with its 12 barrier terms (the only
// transcendental work) evaluated by 12 lanes and summed in the reference's order.  Same expressions, same
// summation order as plane_pair => identical results.
__device__ __forceinline__ bool plane_pair_wave(const double* A, const double* Bq, double dist, double m, double off, int lane, double& e0, double& e1c, double& e2c, double& dpl,
                                       bool& capped, int* newton_iters, int* gjk_iters, const Dev* tic,
                                       GjkState& gst, bool resume, bool resume_finished) {   // resume: the query's first iterations were run elsewhere (spec_pair_body) and gst holds their state
  capped = false;
  bool gfin;
  V3 v;
#ifdef TJ_PHASE_TIMING
  long long prof[7] = {0, 0, 0, 0, 0, 0, 0};
  if (resume && resume_finished) v = gst.v; else v = gjk_wave_run(BodyHull{A}, BodyHull{Bq}, lane, gst, !resume, 50, gfin, tic ? prof : nullptr);
  if (tic && threadIdx.x == 0 && blockIdx.x < TJ_TIC_BLOCKS) for (int i = 0; i < 7; i++) tic->dbg[((size_t)K_OBS_SOLVE * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + i] = prof[i];
#else
  if (resume && resume_finished) v = gst.v; else v = gjk_wave_run(BodyHull{A}, BodyHull{Bq}, lane, gst, !resume, 50, gfin);
#endif
  if (gjk_iters) *gjk_iters = gst.k;
  TJ_ORDER(v.x);
  if (tic) TJ_TIC(*tic, K_SEP_SELF_SOLVE, 3);
  const double cn = norm3(v.x, v.y, v.z);
  if (cn > dist) return false;
  e0 = v.x / cn; e1c = v.y / cn; e2c = v.z / cn;
  double d0 = INFINITY, d1 = -INFINITY;
  for (int i = 0; i < 6; i++) { const double t = -dot_fixed3(e0, e1c, e2c, Bq + 3 * i); if (d0 > t) d0 = t; }
  for (int i = 0; i < 6; i++) { const double t = -dot_fixed3(e0, e1c, e2c, A + 3 * i); if (d1 < t) d1 = t; }
  dpl = 0.5 * (d0 + d1);
  TJ_ORDER(dpl);
  if (tic) TJ_TIC(*tic, K_SEP_SELF_SOLVE, 4);
  const int j = lane < 6 ? lane : (lane < 12 ? lane - 6 : 0);
  const double* pt = lane < 6 ? A + 3 * j : Bq + 3 * j;
  const double px = pt[0], py = pt[1], pz = pt[2];
  int it = 0;
  for (; it < NEWTON_CAP; it++) {  // Newton on the offset until |grad| < 1e-2 (Optimal_plane.h:13-71)
    const double dp = px * e0 + py * e1c + pz * e2c;
    const double ds = lane < 6 ? dp + dpl - 0.5 * off : -dp - dpl - 0.5 * off;
    const bool act = lane < 12 && ds < m;
    double g1 = 0, g2 = 0;
    if (act) {
      const double lg = cr_log(ds / m);
      g1 = -(2 * (ds - m) * lg + (ds - m) * (ds - m) / ds);
      g2 = -(2 * lg + 4 * (ds - m) / ds - (ds - m) * (ds - m) / (ds * ds));
    }
    const unsigned mask = (unsigned)__ballot(act);
    double grad = 0, hess = 0;
#pragma unroll
    for (int q = 0; q < 12; q++)
      if (mask & (1u << q)) {  // uniform
        const double a1 = gjk_rl(g1, q), a2 = gjk_rl(g2, q);
        if (q < 6) grad += a1; else grad += -a1;
        hess += a2;
      }
    const double dir = -grad / hess;
    dpl = dpl + 1.0 * dir;
    if (fabs(grad) < 1e-2) break;
  }
  capped = it == NEWTON_CAP;
  if (newton_iters) *newton_iters = it + 1;
  TJ_ORDER(dpl);
  if (tic) TJ_TIC(*tic, K_SEP_SELF_SOLVE, 5);
  return true;
}

__device__ inline bool plane_pair_wave(const double* A, const double* Bq, double dist, double m, double off, int lane, double& e0, double& e1c, double& e2c, double& dpl,
                                       bool& capped, int* newton_iters = nullptr, int* gjk_iters = nullptr, const Dev* tic = nullptr) {
  GjkState gst;
  return plane_pair_wave(A, Bq, dist, m, off, lane, e0, e1c, e2c, dpl, capped, newton_iters, gjk_iters, tic, gst, false, false);
}

// ---- obstacle planes in three steps --------------------------------------------------------------------------------
// (1) obs_query_body  one wave per (owned robot, segment): hull, BVH walk, k-DOP cull; the surviving points go to the
//                     segment's candidate list (traversal order, deterministic) and, one work item each, to a global list.
// (2) obs_solve_body  one wave per candidate (stride over the work list): wave-cooperative GJK hull-vs-point and the plane
//                     (c,d), written to the candidate's own slot with an epoch stamp.
// (3) compaction      kernels_pairs.h: stamped slots of a segment -> its plane list, in slot order.
// A segment next to an obstacle slab has > 100 candidates; solved inside the segment's own wave (64 divergent GJK paths
// per round) they made a 45 us tail on a 5 us kernel.  One wave per candidate runs them all at once.
// Plane order = candidate order = what the fused version produced, so downstream sums see the same sequence.
constexpr int OBS_SINGLE_MAX = 16;  // candidates of a segment that still get a wave each
// LDS of one unit, carved from a buffer the KERNEL owns: the union kernels run a different body per block, and function-local
// __shared__ arrays of exclusive branches are not overlaid by the compiler (their sizes add up, and with them go residency).
constexpr int OBS_LDS_DOUBLES = 264 + (2 * FRONT_CAP + 128) / 2;   // P[18] klo[49] khi[49] kax[147] | fa fb cand
// use_cache: the hull cache (hullinfo: hull, box, 49 intervals -- the same expressions, written by k_linesearch / k_hullinfo) is
// valid for this robot; taking the record from there is ONE memory latency where recomputing costs two plus the projections.
// True in the iteration chains of the multi-robot modes, false in the stage API (the cache is rebuilt after this stage there).
// publish (coupled chain, Dev::xf_all; implies !use_cache): the unit also leaves the hull-cache record of its (robot, segment) -- what k_hullinfo would -- written through,
// and counts itself done on the segment's counter: the pair tiles of this launch wait for it
// FA (asynchronous front, Dev::fa_seq; implies publish): this launch runs next to the k_linesearch that commits the control net -- the unit fetches what does not depend
// on the net (its basis row, the top BVH boxes), waits for the robot's commit flag, reads the net past the caches (two robots' nets share cache lines) and forms the
// hull with hull_entry's sums in their order (same bits); everything it leaves for later kernels is written through (k_front's block then counts itself done).
template <int PRIM, bool FA = false>
__device__ __forceinline__ void obs_query_body(const Dev& D, int bid, double* lds, bool use_cache, bool publish = false) {
  const int u = D.u0 + bid / D.S, tr = bid % D.S;
  const int lane = lane_id();
  auto sti = [&](int* p, int v) { if constexpr (FA) xf_store_i(p, v); else *p = v; };
  double* P = lds; double* klo = P + 18; double* khi = klo + 49;
  int* fa = (int*)(lds + 264); int* fb = fa + FRONT_CAP; int* cand = fb + FRONT_CAP;
  bool kax_ready = false;
  V3 axv{0, 0, 0}; double lo_ax = 0, hi_ax = 0;
  const double* net = D.spline + (size_t)u * 3 * D.T;
  TJ_TIC(D, K_SEP_OBS, 0);
  QBox q;
  const TopBox topb = bvh_top_box(D);   // travels with the segment's record
  if (use_cache && D.multi()) {
    const double* h = D.hullinfo + ((size_t)u * D.S + tr) * HULL_INFO_STRIDE;
    if (lane < 18) P[lane] = h[lane];
    if (lane < 49) { klo[lane] = h[24 + lane]; khi[lane] = h[73 + lane]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { q.lo[k] = h[18 + k]; q.hi[k] = h[21 + k]; }
  } else {
    if constexpr (FA) {
      double bk[6] = {0, 0, 0, 0, 0, 0};
      const int e = min(lane, 17);
      const double* B = D.basis + (size_t)tr * 36 + (e / 3) * 6;
#pragma unroll
      for (int k = 0; k < 6; k++) bk[k] = B[k];
      const double* col = net + div_small(tr, D.res) * 3 + D.T * (e % 3);
      fa_wait_flag(D, D.fa_commit(u), D.fa_seq);
      TJ_TIC(D, K_SEP_OBS, 6);
      if (lane < 18) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) acc += bk[k] * xf_load(col + k);   // = hull_entry
        P[lane] = acc;
      }
    } else
    if (lane < 18) P[lane] = hull_entry(D, net, tr, lane / 3, lane % 3);
    __syncthreads();
    kdop_intervals(D, P, 6, klo, khi);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double lo = INFINITY, hi = -INFINITY;
      for (int j = 0; j < 6; j++) { const double v = P[3 * j + k]; if (v < lo) lo = v; if (v > hi) hi = v; }
      q.lo[k] = lo; q.hi[k] = hi;
    }
  }
  __syncthreads();
  if (publish) {
    double* o = D.hullinfo + ((size_t)u * D.S + tr) * HULL_INFO_STRIDE;
    if (lane < 18) xf_store(o + lane, P[lane]);
    if (lane < 3) {
      const double lo = lane == 0 ? q.lo[0] : (lane == 1 ? q.lo[1] : q.lo[2]), hi = lane == 0 ? q.hi[0] : (lane == 1 ? q.hi[1] : q.hi[2]);
      xf_store(o + 18 + lane, lo); xf_store(o + 21 + lane, hi);
      xf_store(D.hbox + ((size_t)tr * 6 + lane) * D.U + u, lo); xf_store(D.hbox + ((size_t)tr * 6 + 3 + lane) * D.U + u, hi);
    }
    if (lane < 49) { xf_store(o + 24 + lane, klo[lane]); xf_store(o + 73 + lane, khi[lane]); }
    xf_signal(D, 0, tr);
  }
  TJ_TIC(D, K_SEP_OBS, 1);
  const double dist = D.offset + D.margin;
  const size_t seg = (size_t)u * D.S + tr;
  int* list = D.ocand + seg * D.cap_obs;
  int base = 0;
  unsigned long long visits = 0;
  const int found = bvh_query<4, PRIM>(D, q, dist, fa, fb, cand, &visits, [&](int pt) {
    if (!kax_ready) {   // wave-uniform: this lane's axis and the hull's interval on it, kept in registers from the first candidate on
      const int ax = min(lane, 48);
      axv = V3{D.kdop[3 * ax], D.kdop[3 * ax + 1], D.kdop[3 * ax + 2]}; lo_ax = klo[ax]; hi_ax = khi[ax];
      kax_ready = true;
    }
    const int ncand = __popcll(ballot(pt >= 0));   // candidates sit in lanes [0, ncand)
    const bool ok = kdop_cull_wave(PrimOf<PRIM>::load(D, max(pt, 0)), ncand, axv, lo_ax, hi_ax, dist, lane) && pt >= 0;
    const unsigned long long mask = ballot(ok);
    const int idx = base + prefix_count(mask);
    if (ok) {
      if (idx < D.cap_obs) sti(list + idx, pt);
      else atomicOr(&D.ctl->error, ERR_PLANE_OVERFLOW);
    }
    base += __popcll(mask);
  }, &topb);
  TJ_TIC(D, K_SEP_OBS, 4);
  const int cnt = min(base, D.cap_obs);
  if (cnt > 0 && lane < 18) { if constexpr (FA) xf_store(D.ohull + seg * 18 + lane, P[lane]); else D.ohull[seg * 18 + lane] = P[lane]; }   // read by the solve waves of this segment's candidates only
  // Work items: a segment with few candidates hands each one to its own wave (cooperative GJK, lowest latency); a
  // segment in a dense part of the cloud hands them over in batches of up to 64, one candidate per lane (per-lane GJK,
  // highest throughput).  slot >= 0: single candidate; slot < 0: batch starting at candidate -(slot + 1).
  const bool batched = cnt > OBS_SINGLE_MAX;
  const int items = batched ? (cnt + 63) / 64 : cnt;
  int w0 = 0;
  if (lane == 0) {
    sti(D.ocand_n + seg, cnt);
    if (items > 0) w0 = atomicAdd(D.obs_work_n, items);
    unsigned long long* st = D.seg_stats + seg * 6;   // fire-and-forget atomics instead of a read-modify-write round trip
    atomicAdd(&st[0], visits); atomicAdd(&st[1], (unsigned long long)found);
  }
  w0 = __shfl(w0, 0);
  for (int i = lane; i < items; i += 64) { sti(D.obs_work + 2 * (size_t)(w0 + i), (int)seg); sti(D.obs_work + 2 * (size_t)(w0 + i) + 1, batched ? -(i * 64) - 1 : i); }
  TJ_TIC(D, K_SEP_OBS, 5);
}

// Separate::opengjk (Separate.h:18-163) for one candidate, by one wave
// FA (asynchronous front, Dev::fa_mid): the k_front whose lists this body reads ended while THIS launch was already running -- everything of it is read past the caches
template <int PRIM, bool FA = false>
__device__ __forceinline__ void obs_solve_body(const Dev& D, int bid, int nwaves) {
  const int lane = lane_id();
  __shared__ double P[18];
  auto ldi = [&](const int* p) { if constexpr (FA) return xf_load_i(p); else return *p; };
  const int n = ldi(D.obs_work_n);
  const double dist = D.offset + D.margin;
  const int epoch = D.ctl->epoch;
  for (int w = bid; w < n; w += nwaves) {
    const int seg = ldi(D.obs_work + 2 * (size_t)w), slot = ldi(D.obs_work + 2 * (size_t)w + 1);
    __syncthreads();
    if (lane < 18) { if constexpr (FA) P[lane] = xf_load(D.ohull + (size_t)seg * 18 + lane); else P[lane] = D.ohull[(size_t)seg * 18 + lane]; }
    __syncthreads();
    const bool batch = slot < 0;
    const int first = batch ? -(slot + 1) : slot;
    const int mine = batch ? first + lane : first;                     // candidate of this lane
    const bool live = !batch || mine < ldi(D.ocand_n + seg);
    const int pt = ldi(D.ocand + (size_t)seg * D.cap_obs + (live ? mine : first));
    const typename PrimOf<PRIM>::Body qb = PrimOf<PRIM>::load(D, pt);
    V3 v;
    if (batch) v = gjk(BodyHull{P}, qb);                               // one candidate per lane
    else v = gjk_wave(BodyHull{P}, qb, lane);                          // one candidate for the whole wave
    double c0, c1, c2, dd;
    if (plane_from_witness_body(v, qb, dist, D.offset, c0, c1, c2, dd) && live && (batch || lane == 0)) {
      double* o = D.oraw + ((size_t)seg * D.cap_obs + mine) * 4;
      o[0] = c0; o[1] = c1; o[2] = c2; o[3] = dd;
      D.ostamp[(size_t)seg * D.cap_obs + mine] = epoch;
    }
  }
}

template <int PRIM>
__global__ __launch_bounds__(64) void k_obs_query(Dev D) {
  if (TJ_DONE(D)) return;
  __shared__ double lds[OBS_LDS_DOUBLES];
  obs_query_body<PRIM>(D, blockIdx.x, lds, false);
}
template <int PRIM>
__global__ __launch_bounds__(64) void k_obs_solve(Dev D) {
  if (TJ_DONE(D)) return;
  obs_solve_body<PRIM>(D, blockIdx.x, gridDim.x);
}

}  // namespace tj
