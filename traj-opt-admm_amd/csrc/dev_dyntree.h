// dev_dyntree.h -- the reference's incrementally balanced AABB tree, on ONE lane with its nodes in LDS.
//
// Step::self_step (Step.h:184-256) shrinks the steps of BOTH robots of a colliding pair, pair after pair, in the
// order in which aabb::Tree::query(margin) (AABB.cc:669-734) emits the pairs of a tree that BVH::SelfCCDCollision
// (BVH.cpp:289-330) rebuilds per segment by inserting the robots' swept boxes one by one (insertLeaf AABB.cc:846-967,
// balance :1016-1138).  Pairs that share no robot commute, so the order only matters for a segment in which two acting
// pairs share a robot -- rare (it never happened in any benchmark scene), but then the result depends on it.  For such a
// segment k_ccd_self_seq builds this tree and sorts the segment's acting pairs by their emission index.
//
// GPU shape: deliberately none.  Insertion with rotations is a pointer-chasing, strictly sequential algorithm over <= 2U
// nodes; it runs on lane 0 against LDS (a few hundred ns per node visit) on a path that is cold by construction.  The
// wave-parallel part of the clamp (the GJK back-off of every acting pair) stays wave-cooperative in the caller.
#pragma once
#include "dev_common.h"

namespace tj {

struct DynTree {          // all arrays in LDS, capacity 2 * U nodes
  double* box;            // [cap][6] lo.xyz hi.xyz
  double* area;           // [cap] cached surface area
  int *parent, *left, *right, *height, *particle;
  int root, count;
};
constexpr int DT_NIL = -1;
__host__ __device__ inline size_t dyntree_lds_bytes(int U) { return (size_t)2 * U * (7 * sizeof(double) + 5 * sizeof(int)); }

__device__ __forceinline__ double dt_area(const double* b) {  // AABB::computeSurfaceArea (AABB.cc:86-110): 2 * sum of face products
  double sum = 0;
  for (int d1 = 0; d1 < 3; d1++) {
    double prod = 1;
    for (int d2 = 0; d2 < 3; d2++) { if (d1 == d2) continue; prod *= b[3 + d2] - b[d2]; }
    sum += prod;
  }
  return 2.0 * sum;
}
__device__ __forceinline__ void dt_merge(double* out, double& out_area, const double* a, const double* b) {
  double r[6];
  for (int i = 0; i < 3; i++) { r[i] = (b[i] < a[i]) ? b[i] : a[i]; r[3 + i] = (a[3 + i] < b[3 + i]) ? b[3 + i] : a[3 + i]; }  // std::min / std::max (AABB.cc:163-176)
  for (int i = 0; i < 6; i++) out[i] = r[i];
  out_area = dt_area(r);
}
__device__ __forceinline__ bool dt_leaf(const DynTree& t, int n) { return t.left[n] == DT_NIL; }
__device__ __forceinline__ int dt_alloc(DynTree& t) {
  const int n = t.count++;
  t.parent[n] = t.left[n] = t.right[n] = DT_NIL; t.height[n] = 0; t.particle[n] = DT_NIL; t.area[n] = 0;
  return n;
}

// AVL-like rotation (AABB.cc:1016-1138); returns the node now at a's place
__device__ __noinline__ int dt_balance(DynTree& t, int a) {
  if (dt_leaf(t, a) || t.height[a] < 2) return a;
  const int b = t.left[a], c = t.right[a];
  const int bal = t.height[c] - t.height[b];
  if (bal > 1 || bal < -1) {
    const bool up_right = bal > 1;
    const int up = up_right ? c : b, other = up_right ? b : c;     // `up` moves above a, `other` stays a's child
    const int f = t.left[up], g = t.right[up];
    t.left[up] = a; t.parent[up] = t.parent[a]; t.parent[a] = up;
    const int p = t.parent[up];
    if (p != DT_NIL) { if (t.left[p] == a) t.left[p] = up; else t.right[p] = up; } else t.root = up;
    const int stay = (t.height[f] > t.height[g]) ? f : g, move = (stay == f) ? g : f;
    t.right[up] = stay;
    if (up_right) t.right[a] = move; else t.left[a] = move;
    t.parent[move] = a;
    dt_merge(t.box + 6 * a, t.area[a], t.box + 6 * other, t.box + 6 * move);
    dt_merge(t.box + 6 * up, t.area[up], t.box + 6 * a, t.box + 6 * stay);
    t.height[a] = 1 + max(t.height[other], t.height[move]);
    t.height[up] = 1 + max(t.height[a], t.height[stay]);
    return up;
  }
  return a;
}

// insertParticle + insertLeaf (AABB.cc:389-443, :846-967): surface-area descent, new parent, rebalance on the way up
__device__ __noinline__ void dt_insert(DynTree& t, int particle, const double* bx) {
  const int lf = dt_alloc(t);
  for (int i = 0; i < 6; i++) t.box[6 * lf + i] = bx[i];
  t.area[lf] = dt_area(bx);
  t.particle[lf] = particle;
  if (t.root == DT_NIL) { t.root = lf; return; }
  const double* lb = t.box + 6 * lf;
  int idx = t.root;
  double tmp[6], ta;
  while (!dt_leaf(t, idx)) {
    const int l = t.left[idx], r = t.right[idx];
    const double cur = t.area[idx];
    dt_merge(tmp, ta, t.box + 6 * idx, lb);
    const double comb = ta, cost = 2.0 * comb, inherit = 2.0 * (comb - cur);
    dt_merge(tmp, ta, lb, t.box + 6 * l);
    const double cl = dt_leaf(t, l) ? ta + inherit : (ta - t.area[l]) + inherit;
    dt_merge(tmp, ta, lb, t.box + 6 * r);
    const double cr = dt_leaf(t, r) ? ta + inherit : (ta - t.area[r]) + inherit;
    if ((cost < cl) && (cost < cr)) break;
    idx = (cl < cr) ? l : r;
  }
  const int sib = idx, old_parent = t.parent[sib], np = dt_alloc(t);
  t.parent[np] = old_parent;
  dt_merge(t.box + 6 * np, t.area[np], lb, t.box + 6 * sib);
  t.height[np] = t.height[sib] + 1;
  if (old_parent != DT_NIL) { if (t.left[old_parent] == sib) t.left[old_parent] = np; else t.right[old_parent] = np; } else t.root = np;
  t.left[np] = sib; t.right[np] = lf; t.parent[sib] = np; t.parent[lf] = np;
  idx = t.parent[lf];
  while (idx != DT_NIL) {
    idx = dt_balance(t, idx);
    const int l = t.left[idx], r = t.right[idx];
    t.height[idx] = 1 + max(t.height[l], t.height[r]);
    dt_merge(t.box + 6 * idx, t.area[idx], t.box + 6 * l, t.box + 6 * r);
    idx = t.parent[idx];
  }
}

// `self`.overlaps(`other`, touchIsOverlap = true, margin)  (AABB.cc:131-148)
__device__ __forceinline__ bool dt_overlaps(const double* self, const double* other, double margin) {
  for (int i = 0; i < 3; ++i) if (other[3 + i] + margin < self[i] || other[i] > self[3 + i] + margin) return false;
  return true;
}

// Tree::query(margin) (AABB.cc:669-734) restricted to what the caller needs: the emission ORDER of the m acting pairs
// (a0[i] < a1[i]).  ord[j] = index of the j-th emitted acting pair.  Returns how many acting pairs were emitted (== m unless
// the stack overflowed).  stk holds 2 * stk_cap ints.
__device__ __noinline__ int dt_pair_order(const DynTree& t, double margin, const int* a0, const int* a1, int m, int* ord, int* stk, int stk_cap) {
  int top = 0, found = 0;
  stk[0] = t.root; stk[1] = t.root; top = 1;
  while (top > 0 && found < m) {
    top--;
    const int n = stk[2 * top], q = stk[2 * top + 1];
    if (n == DT_NIL || q == DT_NIL) continue;
    if (!dt_overlaps(t.box + 6 * q, t.box + 6 * n, margin)) continue;
    const bool ln = dt_leaf(t, n), lq = dt_leaf(t, q);
    if (ln && lq) {
      const int p0 = t.particle[n], p1 = t.particle[q];
      if (p0 < p1) for (int i = 0; i < m; i++) if (a0[i] == p0 && a1[i] == p1) { ord[found++] = i; break; }
      continue;
    }
    if (top + 4 > stk_cap) return -1;
    if (ln) { stk[2 * top] = n; stk[2 * top + 1] = t.left[q]; top++; stk[2 * top] = n; stk[2 * top + 1] = t.right[q]; top++; }
    else if (lq) { stk[2 * top] = t.left[n]; stk[2 * top + 1] = q; top++; stk[2 * top] = t.right[n]; stk[2 * top + 1] = q; top++; }
    else {
      stk[2 * top] = t.left[n]; stk[2 * top + 1] = t.left[q]; top++;
      stk[2 * top] = t.right[n]; stk[2 * top + 1] = t.right[q]; top++;
      stk[2 * top] = t.left[n]; stk[2 * top + 1] = t.right[q]; top++;
      stk[2 * top] = t.right[n]; stk[2 * top + 1] = t.left[q]; top++;
    }
  }
  return found;
}

}  // namespace tj
