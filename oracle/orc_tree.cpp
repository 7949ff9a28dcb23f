// oracle/orc_tree.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the incrementally built, height-balanced AABB tree that the reference
// uses as broad phase (HighOrderCCD/BVH/src/AABB.cc, a Box2D-style dynamic tree):
//   * leaf insertion with the surface-area descent heuristic      AABB.cc:846-967
//   * AVL-like rotation on the way back up                         AABB.cc:1016-1138
//   * box query (DFS, children pushed left then right)             AABB.cc:608-667
//   * self query over ordered node pairs, emits (p,q) with p<q     AABB.cc:669-734
//   * overlap predicate with margin, touching counts               AABB.cc:131-161
// The tree *shape* does not change which candidates are found, only their ORDER; the
// reference's inter-robot step clamp (Step.h:184-256) is order dependent, which is why the
// oracle reproduces the shape exactly.  Dimension is fixed to 3 and the skin thickness to
// 0.0 (BVH.cpp:63), periodic boxes are not used on this path.
#include "orc.h"
#include <algorithm>

namespace orc {

DynTree::DynTree(uint32_t capacity_hint) : root_(NIL), free_(NIL), count_(0), cap_(0) { nodes_.reserve(2 * capacity_hint); }

double DynTree::area(const Box3& b) {
  double sum = 0;
  for (int d1 = 0; d1 < 3; d1++) {
    double prod = 1;
    for (int d2 = 0; d2 < 3; d2++) {
      if (d1 == d2) continue;
      prod *= b.hi[d2] - b.lo[d2];
    }
    sum += prod;
  }
  return 2.0 * sum;
}

void DynTree::merge(Box3& out, double& out_area, const Box3& a, const Box3& b) {
  Box3 r;
  for (int i = 0; i < 3; i++) { r.lo[i] = std::min(a.lo[i], b.lo[i]); r.hi[i] = std::max(a.hi[i], b.hi[i]); }
  out = r;
  out_area = area(r);
}

uint32_t DynTree::alloc() {
  Node n;
  n.parent = n.left = n.right = n.next = NIL;
  n.height = 0; n.particle = NIL; n.area = 0;
  nodes_.push_back(n);
  return (uint32_t)nodes_.size() - 1;
}

void DynTree::insert(uint32_t particle, const Box3& box) {
  uint32_t n = alloc();
  nodes_[n].box = box;
  nodes_[n].area = area(box);
  nodes_[n].height = 0;
  insert_leaf(n);
  nodes_[n].particle = particle;
  count_++;
}

void DynTree::insert_leaf(uint32_t lf) {
  if (root_ == NIL) { root_ = lf; nodes_[lf].parent = NIL; return; }
  const Box3 lb = nodes_[lf].box;
  uint32_t idx = root_;
  Box3 tmp; double tmp_area;
  while (!leaf(idx)) {
    uint32_t l = nodes_[idx].left, r = nodes_[idx].right;
    double cur = nodes_[idx].area;
    merge(tmp, tmp_area, nodes_[idx].box, lb);
    double comb = tmp_area;
    double cost = 2.0 * comb;
    double inherit = 2.0 * (comb - cur);
    double cl, cr;
    merge(tmp, tmp_area, lb, nodes_[l].box);
    cl = leaf(l) ? tmp_area + inherit : (tmp_area - nodes_[l].area) + inherit;
    merge(tmp, tmp_area, lb, nodes_[r].box);
    cr = leaf(r) ? tmp_area + inherit : (tmp_area - nodes_[r].area) + inherit;
    if ((cost < cl) && (cost < cr)) break;
    idx = (cl < cr) ? l : r;
  }
  uint32_t sib = idx;
  uint32_t old_parent = nodes_[sib].parent;
  uint32_t np = alloc();
  nodes_[np].parent = old_parent;
  merge(nodes_[np].box, nodes_[np].area, lb, nodes_[sib].box);
  nodes_[np].height = nodes_[sib].height + 1;
  if (old_parent != NIL) {
    if (nodes_[old_parent].left == sib) nodes_[old_parent].left = np; else nodes_[old_parent].right = np;
  } else {
    root_ = np;
  }
  nodes_[np].left = sib; nodes_[np].right = lf;
  nodes_[sib].parent = np; nodes_[lf].parent = np;

  idx = nodes_[lf].parent;
  while (idx != NIL) {
    idx = balance(idx);
    uint32_t l = nodes_[idx].left, r = nodes_[idx].right;
    nodes_[idx].height = 1 + std::max(nodes_[l].height, nodes_[r].height);
    merge(nodes_[idx].box, nodes_[idx].area, nodes_[l].box, nodes_[r].box);
    idx = nodes_[idx].parent;
  }
}

uint32_t DynTree::balance(uint32_t a) {
  if (leaf(a) || nodes_[a].height < 2) return a;
  uint32_t b = nodes_[a].left, c = nodes_[a].right;
  int bal = nodes_[c].height - nodes_[b].height;
  auto relink_parent = [&](uint32_t up) {
    uint32_t p = nodes_[up].parent;
    if (p != NIL) { if (nodes_[p].left == a) nodes_[p].left = up; else nodes_[p].right = up; }
    else root_ = up;
  };
  if (bal > 1) {  // right child moves up
    uint32_t f = nodes_[c].left, g = nodes_[c].right;
    nodes_[c].left = a; nodes_[c].parent = nodes_[a].parent; nodes_[a].parent = c;
    relink_parent(c);
    uint32_t stay = (nodes_[f].height > nodes_[g].height) ? f : g;   // stays under c
    uint32_t move = (stay == f) ? g : f;                             // goes under a
    nodes_[c].right = stay; nodes_[a].right = move; nodes_[move].parent = a;
    merge(nodes_[a].box, nodes_[a].area, nodes_[b].box, nodes_[move].box);
    merge(nodes_[c].box, nodes_[c].area, nodes_[a].box, nodes_[stay].box);
    nodes_[a].height = 1 + std::max(nodes_[b].height, nodes_[move].height);
    nodes_[c].height = 1 + std::max(nodes_[a].height, nodes_[stay].height);
    return c;
  }
  if (bal < -1) {  // left child moves up
    uint32_t d = nodes_[b].left, e = nodes_[b].right;
    nodes_[b].left = a; nodes_[b].parent = nodes_[a].parent; nodes_[a].parent = b;
    relink_parent(b);
    uint32_t stay = (nodes_[d].height > nodes_[e].height) ? d : e;
    uint32_t move = (stay == d) ? e : d;
    nodes_[b].right = stay; nodes_[a].left = move; nodes_[move].parent = a;
    merge(nodes_[a].box, nodes_[a].area, nodes_[c].box, nodes_[move].box);
    merge(nodes_[b].box, nodes_[b].area, nodes_[a].box, nodes_[stay].box);
    nodes_[a].height = 1 + std::max(nodes_[c].height, nodes_[move].height);
    nodes_[b].height = 1 + std::max(nodes_[a].height, nodes_[stay].height);
    return b;
  }
  return a;
}

// `self` is the box whose overlaps() is called, `other` its argument (AABB.cc:131-148,
// touchIsOverlap = true).
static inline bool overlaps(const Box3& self, const Box3& other, double margin) {
  for (int i = 0; i < 3; ++i)
    if (other.hi[i] + margin < self.lo[i] || other.lo[i] > self.hi[i] + margin) return false;
  return true;
}

void DynTree::query(const Box3& q, double margin, std::vector<uint32_t>& out) const {
  out.clear();
  if (count_ == 0) return;
  std::vector<uint32_t> stack;
  stack.reserve(256);
  stack.push_back(root_);
  while (!stack.empty()) {
    uint32_t n = stack.back(); stack.pop_back();
    if (n == NIL) continue;
    if (overlaps(q, nodes_[n].box, margin)) {
      if (leaf(n)) out.push_back(nodes_[n].particle);
      else { stack.push_back(nodes_[n].left); stack.push_back(nodes_[n].right); }
    }
  }
}

void DynTree::self_query(double margin, std::vector<std::pair<uint32_t, uint32_t>>& out) const {
  out.clear();
  std::vector<std::pair<uint32_t, uint32_t>> stack;
  stack.reserve(256);
  stack.push_back({root_, root_});
  while (!stack.empty()) {
    auto pr = stack.back(); stack.pop_back();
    uint32_t n = pr.first, m = pr.second;
    if (n == NIL || m == NIL) continue;
    if (!overlaps(nodes_[m].box, nodes_[n].box, margin)) continue;
    bool ln = leaf(n), lm = leaf(m);
    if (ln && lm) {
      if (nodes_[n].particle < nodes_[m].particle) out.push_back({nodes_[n].particle, nodes_[m].particle});
    } else if (ln) {
      stack.push_back({n, nodes_[m].left}); stack.push_back({n, nodes_[m].right});
    } else if (lm) {
      stack.push_back({nodes_[n].left, m}); stack.push_back({nodes_[n].right, m});
    } else {
      stack.push_back({nodes_[n].left, nodes_[m].left}); stack.push_back({nodes_[n].right, nodes_[m].right});
      stack.push_back({nodes_[n].left, nodes_[m].right}); stack.push_back({nodes_[n].right, nodes_[m].left});
    }
  }
}

}  // namespace orc
