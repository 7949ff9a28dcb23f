// oracle/orc_amd.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The single-UAV Newton system is solved by the reference with Eigen::SimplicialLLT on `h0.sparseView()`
// (Optimization3D_admm.h:470-475): approximate-minimum-degree ordering of the pattern of exact non-zeros, then an up-looking
// sparse Cholesky in that order.  With ks = 1e-8 the system is conditioned ~1e8, so a different elimination ORDER shows up at
// the 1e-8 parity bar; this file restates what Eigen 3.3.7 does, step for step, so that the oracle's solve is the
// reference's solve:
//   amd_order          Eigen::internal::minimum_degree_ordering (lib/eigen3/Eigen/src/OrderingMethods/Amd.h:92-437; the
//                      CSparse / Davis quotient-graph AMD: element absorption, approximate external degrees, hashed
//                      supervariable detection, mass elimination, dense-row deferral, assembly-tree postorder) on the pattern
//                      A + A^T (Ordering.h:24-45, :60-77).  Ties are broken by the order in which nodes sit in the degree
//                      lists and in the adjacency storage, so the data layout below (one index pool with elbow room and a
//                      compacting garbage collector) is part of the specification, not an implementation choice.
//   sparse_llt_solve   SimplicialCholeskyBase::ordering / analyzePattern_preordered / factorize_preordered<false>
//                      (SparseCholesky/SimplicialCholesky.h:634-657, SimplicialCholesky_impl.h:50-189), the symmetric
//                      permutation that fixes the ENTRY ORDER inside each column (SparseCore/SparseSelfAdjointView.h:526-581)
//                      -- that order decides the sequence of subtractions in the up-looking row solve -- and the two
//                      triangular solves of _solve_impl (SimplicialCholesky.h:556-583, TriangularSolver.h).
// Pinned bit for bit against Eigen itself through oracle/_ref (tests/golden/amd_kat.npz).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "orc.h"

namespace orc {

namespace {
inline int flip(int i) { return -i - 2; }

// reset the mark array when the running mark would overflow or is non-positive
int clear_marks(int mark, int lemax, int* w, int n) {
  if (mark < 2 || (mark + lemax < 0)) {
    for (int k = 0; k < n; k++) if (w[k] != 0) w[k] = 1;
    mark = 2;
  }
  return mark;
}

// depth-first numbering of one assembly tree (children lists are LIFO: the youngest child first)
int tree_postorder(int root, int k, int* head, const int* next, int* post, int* stack) {
  int top = 0;
  stack[0] = root;
  while (top >= 0) {
    const int p = stack[top], child = head[p];
    if (child == -1) { top--; post[k++] = p; }
    else { head[p] = next[child]; stack[++top] = child; }
  }
  return k;
}
}  // namespace

// cp[n+1], ci[cp[n]]: full symmetric pattern incl. the diagonal, row indices ascending inside a column.
// order[k] = the node eliminated k-th.
void amd_order(int n, const std::vector<int>& cp_in, const std::vector<int>& ci_in, std::vector<int>& order) {
  int dense = std::max(16, (int)(10 * std::sqrt((double)n)));
  dense = std::min(n - 2, dense);
  int cnz = cp_in[n];
  const int pool = cnz + cnz / 5 + 2 * n;   // adjacency pool with elbow room
  std::vector<int> Cp(cp_in.begin(), cp_in.end()), Ci(pool, 0);
  std::copy(ci_in.begin(), ci_in.begin() + cnz, Ci.begin());
  std::vector<int> last(n + 1), len(n + 1), nv(n + 1), next(n + 1), head(n + 1), elen(n + 1), degree(n + 1), w(n + 1), hhead(n + 1);
  for (int k = 0; k < n; k++) len[k] = Cp[k + 1] - Cp[k];
  len[n] = 0;
  for (int i = 0; i <= n; i++) { head[i] = last[i] = next[i] = hhead[i] = -1; nv[i] = 1; w[i] = 1; elen[i] = 0; degree[i] = len[i]; }
  int mark = clear_marks(0, 0, w.data(), n);
  int nel = 0, mindeg = 0, lemax = 0;

  // degree lists; empty rows are eliminated at once, rows denser than the threshold are set aside (ordered last)
  for (int i = 0; i < n; i++) {
    bool has_diag = false;
    for (int p = Cp[i]; p < Cp[i + 1]; ++p) if (Ci[p] == i) { has_diag = true; break; }
    const int d = degree[i];
    if (d == 1 && has_diag) { elen[i] = -2; nel++; Cp[i] = -1; w[i] = 0; }
    else if (d > dense || !has_diag) { nv[i] = 0; elen[i] = -1; nel++; Cp[i] = flip(n); nv[n]++; }
    else { if (head[d] != -1) last[head[d]] = i; next[i] = head[d]; head[d] = i; }
  }
  elen[n] = -2; Cp[n] = -1; w[n] = 0;

  while (nel < n) {
    // pivot: head of the lowest non-empty degree list
    int k = -1;
    for (; mindeg < n && (k = head[mindeg]) == -1; mindeg++) {}
    if (next[k] != -1) last[next[k]] = -1;
    head[mindeg] = next[k];
    const int elenk = elen[k];
    int nvk = nv[k];
    nel += nvk;

    // compact the pool when the new element might not fit
    if (elenk > 0 && cnz + mindeg >= pool) {
      for (int j = 0; j < n; j++) { const int p = Cp[j]; if (p >= 0) { Cp[j] = Ci[p]; Ci[p] = flip(j); } }
      int q = 0;
      for (int p = 0; p < cnz;) {
        const int j = flip(Ci[p++]);
        if (j >= 0) { Ci[q] = Cp[j]; Cp[j] = q++; for (int t = 0; t < len[j] - 1; t++) Ci[q++] = Ci[p++]; }
      }
      cnz = q;
    }

    // new element Lk = union of k's variables and of the variables of k's elements (k's own list last)
    int dk = 0;
    nv[k] = -nvk;
    int p = Cp[k];
    const int pk1 = (elenk == 0) ? p : cnz;
    int pk2 = pk1;
    for (int k1 = 1; k1 <= elenk + 1; k1++) {
      int e, pj, ln;
      if (k1 > elenk) { e = k; pj = p; ln = len[k] - elenk; }
      else { e = Ci[p++]; pj = Cp[e]; ln = len[e]; }
      for (int k2 = 1; k2 <= ln; k2++) {
        const int i = Ci[pj++];
        const int nvi = nv[i];
        if (nvi <= 0) continue;
        dk += nvi; nv[i] = -nvi; Ci[pk2++] = i;
        if (next[i] != -1) last[next[i]] = last[i];
        if (last[i] != -1) next[last[i]] = next[i]; else head[degree[i]] = next[i];
      }
      if (e != k) { Cp[e] = flip(k); w[e] = 0; }
    }
    if (elenk != 0) cnz = pk2;
    degree[k] = dk; Cp[k] = pk1; len[k] = pk2 - pk1; elen[k] = -2;

    // |Le \ Lk| for every element adjacent to a variable of Lk
    mark = clear_marks(mark, lemax, w.data(), n);
    for (int pk = pk1; pk < pk2; pk++) {
      const int i = Ci[pk], eln = elen[i];
      if (eln <= 0) continue;
      const int nvi = -nv[i], wnvi = mark - nvi;
      for (int q = Cp[i]; q <= Cp[i] + eln - 1; q++) {
        const int e = Ci[q];
        if (w[e] >= mark) w[e] -= nvi; else if (w[e] != 0) w[e] = degree[e] + wnvi;
      }
    }

    // approximate degrees; absorb elements that became subsets of Lk; hash the variables
    for (int pk = pk1; pk < pk2; pk++) {
      const int i = Ci[pk];
      const int p1 = Cp[i], p2 = p1 + elen[i] - 1;
      int pn = p1, h = 0, d = 0;
      for (int q = p1; q <= p2; q++) {
        const int e = Ci[q];
        if (w[e] != 0) {
          const int dext = w[e] - mark;
          if (dext > 0) { d += dext; Ci[pn++] = e; h += e; }
          else { Cp[e] = flip(k); w[e] = 0; }
        }
      }
      elen[i] = pn - p1 + 1;
      const int p3 = pn, p4 = p1 + len[i];
      for (int q = p2 + 1; q < p4; q++) {
        const int j = Ci[q], nvj = nv[j];
        if (nvj <= 0) continue;
        d += nvj; Ci[pn++] = j; h += j;
      }
      if (d == 0) {   // mass elimination: i has no neighbour outside Lk
        Cp[i] = flip(k);
        const int nvi = -nv[i];
        dk -= nvi; nvk += nvi; nel += nvi; nv[i] = 0; elen[i] = -1;
      } else {
        degree[i] = std::min(degree[i], d);
        Ci[pn] = Ci[p3]; Ci[p3] = Ci[p1]; Ci[p1] = k;
        len[i] = pn - p1 + 1;
        h %= n;
        next[i] = hhead[h]; hhead[h] = i; last[i] = h;
      }
    }
    degree[k] = dk;
    lemax = std::max(lemax, dk);
    mark = clear_marks(mark + lemax, lemax, w.data(), n);

    // indistinguishable variables (same hash, same adjacency) merge into one supervariable
    for (int pk = pk1; pk < pk2; pk++) {
      int i = Ci[pk];
      if (nv[i] >= 0) continue;
      const int h = last[i];
      i = hhead[h]; hhead[h] = -1;
      for (; i != -1 && next[i] != -1; i = next[i], mark++) {
        const int ln = len[i], eln = elen[i];
        for (int q = Cp[i] + 1; q <= Cp[i] + ln - 1; q++) w[Ci[q]] = mark;
        int jlast = i;
        for (int j = next[i]; j != -1;) {
          bool same = (len[j] == ln) && (elen[j] == eln);
          for (int q = Cp[j] + 1; same && q <= Cp[j] + ln - 1; q++) if (w[Ci[q]] != mark) same = false;
          if (same) { Cp[j] = flip(i); nv[i] += nv[j]; nv[j] = 0; elen[j] = -1; j = next[j]; next[jlast] = j; }
          else { jlast = j; j = next[j]; }
        }
      }
    }

    // final Lk; its variables return to the degree lists
    int q = pk1;
    for (int pk = pk1; pk < pk2; pk++) {
      const int i = Ci[pk];
      const int nvi = -nv[i];
      if (nvi <= 0) continue;
      nv[i] = nvi;
      int d = degree[i] + dk - nvi;
      d = std::min(d, n - nel - nvi);
      if (head[d] != -1) last[head[d]] = i;
      next[i] = head[d]; last[i] = -1; head[d] = i;
      mindeg = std::min(mindeg, d);
      degree[i] = d;
      Ci[q++] = i;
    }
    nv[k] = nvk;
    if ((len[k] = q - pk1) == 0) { Cp[k] = -1; w[k] = 0; }
    if (elenk != 0) cnz = q;
  }

  // postorder of the assembly tree: absorbed variables first (highest index first into the child lists), then elements
  for (int i = 0; i < n; i++) Cp[i] = flip(Cp[i]);
  for (int j = 0; j <= n; j++) head[j] = -1;
  for (int j = n; j >= 0; j--) { if (nv[j] > 0) continue; next[j] = head[Cp[j]]; head[Cp[j]] = j; }
  for (int e = n; e >= 0; e--) { if (nv[e] <= 0) continue; if (Cp[e] != -1) { next[e] = head[Cp[e]]; head[Cp[e]] = e; } }
  std::vector<int> post(n + 1);
  for (int k = 0, i = 0; i <= n; i++) if (Cp[i] == -1) k = tree_postorder(i, k, head.data(), next.data(), post.data(), w.data());
  order.assign(post.begin(), post.begin() + n);
}

// H: dense symmetric n x n (column- or row-major alike), b: right-hand side.  x = H^-1 b exactly as
// `SimplicialLLT<SparseMatrix<double>> s; s.compute(H.sparseView()); x = s.solve(b)` forms it.  Returns false where Eigen
// reports NumericalIssue (a pivot <= 0); x is then left as Eigen's solve would produce from the partial factor -- not
// reproduced: callers treat it as failure.  order_out (optional) = permutationPinv().indices().
bool sparse_llt_solve(int n, const double* H, const double* b, double* x, int* order_out) {
  // pattern of exact non-zeros (sparseView: |v| <= 0 is dropped), full symmetric: C = A + A^T of the lower part mirrored
  std::vector<int> cp(n + 1, 0), ci;
  for (int j = 0; j < n; j++) {
    for (int i = 0; i < n; i++) {
      const double lo = i >= j ? H[i + (size_t)n * j] : H[j + (size_t)n * i];   // entry of the lower triangle that represents (i,j)
      if (!(std::fabs(lo) <= 0.0)) ci.push_back(i);
    }
    cp[j + 1] = (int)ci.size();
  }
  std::vector<int> order;
  amd_order(n, cp, ci, order);                     // order[k] = original index of the k-th pivot  (= m_Pinv.indices())
  if (order_out) std::memcpy(order_out, order.data(), sizeof(int) * n);
  std::vector<int> newidx(n);                      // m_P.indices(): original -> position
  for (int k = 0; k < n; k++) newidx[order[k]] = k;

  // ap = upper triangle of P A P^T, columns filled in the traversal order of the source's lower triangle
  // (permute_symm_to_symm<Lower, Upper>): source columns j ascending, rows i >= j ascending
  std::vector<int> ap(n + 1, 0);
  for (int j = 0; j < n; j++) for (int i = j; i < n; i++) if (!(std::fabs(H[i + (size_t)n * j]) <= 0.0)) ap[std::max(newidx[i], newidx[j]) + 1]++;
  for (int j = 0; j < n; j++) ap[j + 1] += ap[j];
  std::vector<int> ai(ap[n]), fill(ap.begin(), ap.end() - 1);
  std::vector<double> ax(ap[n]);
  for (int j = 0; j < n; j++)
    for (int i = j; i < n; i++) {
      const double v = H[i + (size_t)n * j];
      if (std::fabs(v) <= 0.0) continue;
      const int ip = newidx[i], jp = newidx[j];
      const int q = fill[std::max(ip, jp)]++;
      ai[q] = std::min(ip, jp); ax[q] = v;
    }

  // elimination tree and column counts (analyzePattern_preordered)
  std::vector<int> parent(n), nzcol(n), tags(n), Lp(n + 1);
  for (int k = 0; k < n; k++) {
    parent[k] = -1; tags[k] = k; nzcol[k] = 0;
    for (int q = ap[k]; q < ap[k + 1]; q++) {
      int i = ai[q];
      if (i < k) for (; tags[i] != k; i = parent[i]) { if (parent[i] == -1) parent[i] = k; nzcol[i]++; tags[i] = k; }
    }
  }
  Lp[0] = 0;
  for (int k = 0; k < n; k++) Lp[k + 1] = Lp[k] + nzcol[k] + 1;   // LLT keeps the diagonal in the column
  std::vector<int> Li(Lp[n]);
  std::vector<double> Lx(Lp[n]), y(n, 0.0);
  std::vector<int> pattern(n);
  bool ok = true;

  // up-looking factorisation, row k of L at a time (factorize_preordered<false>)
  for (int k = 0; k < n; k++) {
    y[k] = 0.0;
    int top = n;
    tags[k] = k; nzcol[k] = 0;
    for (int q = ap[k]; q < ap[k + 1]; q++) {
      int i = ai[q];
      if (i <= k) {
        y[i] += ax[q];
        int l = 0;
        for (; tags[i] != k; i = parent[i]) { pattern[l++] = i; tags[i] = k; }
        while (l > 0) pattern[--top] = pattern[--l];
      }
    }
    double d = y[k] * 1.0 + 0.0;   // shiftScale 1, shiftOffset 0
    y[k] = 0.0;
    for (; top < n; ++top) {
      const int i = pattern[top];
      double yi = y[i];
      y[i] = 0.0;
      const double l_ki = yi = yi / Lx[Lp[i]];
      const int p2 = Lp[i] + nzcol[i];
      int q;
      for (q = Lp[i] + 1; q < p2; ++q) y[Li[q]] -= Lx[q] * yi;
      d -= l_ki * yi;
      Li[q] = k; Lx[q] = l_ki; ++nzcol[i];
    }
    const int q = Lp[k] + nzcol[k]++;
    Li[q] = k;
    if (d <= 0.0) { ok = false; break; }
    Lx[q] = std::sqrt(d);
  }
  if (!ok) return false;

  // dest = P b;  L dest = dest (column-major lower, forward);  L^T dest = dest (backward);  x = P^-1 dest
  std::vector<double> t(n);
  for (int i = 0; i < n; i++) t[newidx[i]] = b[i];
  for (int i = 0; i < n; i++) {
    const double ti = t[i];
    if (ti != 0.0) {   // sparse_solve_triangular_selector<Lower, ColMajor>: skips zero entries
      const double v = t[i] = ti / Lx[Lp[i]];
      for (int q = Lp[i] + 1; q < Lp[i + 1]; q++) t[Li[q]] -= v * Lx[q];
    }
  }
  for (int i = n - 1; i >= 0; --i) {   // Upper view of the transpose = row-major upper: dot with the column, then divide
    double v = t[i];
    for (int q = Lp[i] + 1; q < Lp[i + 1]; q++) v -= Lx[q] * t[Li[q]];
    t[i] = v / Lx[Lp[i]];
  }
  for (int k = 0; k < n; k++) x[order[k]] = t[k];
  return true;
}

}  // namespace orc

extern "C" {
// order[n] = Eigen's permutationPinv().indices(); returns 1 on success, 0 on a non-positive pivot
int orc_sparse_llt_solve(int n, const double* H, const double* b, double* x, int* order) { return orc::sparse_llt_solve(n, H, b, x, order) ? 1 : 0; }
}
