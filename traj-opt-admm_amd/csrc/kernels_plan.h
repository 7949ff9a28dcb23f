// kernels_plan.h -- initial-trajectory planner: the motion-validity predicate of the reference's OMPL
// set-up, batched on the device.
//
// Reference: myMotionValidator::checkMotion (HighOrderCCD/OMPL/OMPL.cpp:36-98) and edge_collision
// (Main/multiPathPlanning3D.cpp:123-160): a straight edge is invalid if BVH::EdgeCollision (BVH.cpp:95-134, tree query with
// the edge's box inflated by d = offset + margin/2) returns a cloud point with CCD::GJKDCD(edge, point, d)
// (CCD.h:17-113: GJK distance^2 <= d^2), or if GJKDCD(edge, e', d) holds for an edge e' of an already planned robot.
//
//   k_edge_hit   one wavefront per edge piece: lanes over the prior edges (edge-edge GJK), then the static-BVH walk of
//                kernels_sep.h with the piece's box and one GJK edge-vs-point per surviving point and lane.
// Long edges are cut into collinear pieces by the host (plan_host in tj_api.hip) so that a piece's box stays small
// against the cloud -- dist(edge, p) is the minimum over its pieces, the decision is the OR.
#pragma once
#include "dev_common.h"
#include "kernels_sep.h"

namespace tj {

struct BodyEdge {  // a straight edge: 2 vertices held in registers
  V3 a, b;
  static constexpr int N = 2;
  __device__ __forceinline__ V3 get(int i) const { return i == 0 ? a : b; }
};

// pieces[n][6] = (a, b); owner[n] = edge the piece belongs to; hit[edge] is OR-ed (must be zeroed by the caller).
// PRIM = 3: triangle obstacles -- GJKDCD takes the body sizes from its arguments (CCD.h:31-35), so an edge against a
// triangle is the same call with a 3-row _position.
template <int PRIM>
__global__ __launch_bounds__(64) void k_edge_hit(Dev D, int n, const double* pieces, const int* owner, int n_prior, const double* prior, double d, int* hit) {
  const int e = blockIdx.x, lane = lane_id();
  if (e >= n) return;
  __shared__ int fa[FRONT_CAP], fb[FRONT_CAP], cand[128];
  const double* E = pieces + 6 * (size_t)e;
  const BodyEdge eb{V3{E[0], E[1], E[2]}, V3{E[3], E[4], E[5]}};
  bool h = false;
  for (int j = lane; j < n_prior; j += 64) {
    const double* Q = prior + 6 * (size_t)j;
    const V3 v = gjk(eb, BodyEdge{V3{Q[0], Q[1], Q[2]}, V3{Q[3], Q[4], Q[5]}});
    h |= (v.x * v.x + v.y * v.y + v.z * v.z <= d * d);
  }
  QBox q;
#pragma unroll
  for (int k = 0; k < 3; k++) { q.lo[k] = fmin(E[k], E[3 + k]); q.hi[k] = fmax(E[k], E[3 + k]); }
  unsigned long long visits = 0;
  bvh_query<1, PRIM>(D, q, d, fa, fb, cand, &visits, [&](int pt) {
    if (pt >= 0) {
      const V3 v = gjk(eb, PrimOf<PRIM>::load(D, pt));
      h |= (v.x * v.x + v.y * v.y + v.z * v.z <= d * d);
    }
  });
  if (ballot(h) != 0ull && lane == 0) atomicOr(hit + owner[e], 1);
}

}  // namespace tj
