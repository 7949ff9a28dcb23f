// dev_gjk.h -- device GJK (signed-volumes sub-algorithm) for gfx950, fp64, one query per lane.
//
// Replaces the reference's patched openGJK (lib/opengjk/src/openGJK.c:754-852, "Fast" build,
// returns the witness vector) on the hot path.  It is used by the separating-plane kernels
// (Separate.h:18-304) and by the CCD step kernels (CCD.h:116-352).
//
// GPU shape: bodies are never materialised as pointer tables.  A body is a small functor that
// yields vertex i on demand from LDS-staged hulls (wave-uniform, broadcast reads) or from
// registers (an obstacle point), and a swept hull {P, P + t*D} is generated on the fly so a
// CCD back-off loop does not rewrite 12 points per trial.  The simplex lives in registers; all
// vertex moves are expressed as whole-struct selects so nothing is indexed dynamically.
//
// The witness vector is only eps_rel = 1e-5 accurate, therefore the *decision path* must be the
// reference's: each floating-point expression keeps the reference's association order, the TU
// is compiled with -ffp-contract=off, and fp64 divide/sqrt are IEEE on gfx950, so results are
// bit-identical to the CPU reference (pinned by tests/golden/gjk_kat.npz).
#pragma once
#include <hip/hip_runtime.h>

namespace tj {

struct V3 { double x, y, z; };
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double sq(const V3& a) { double n = 0; n += a.x * a.x; n += a.y * a.y; n += a.z * a.z; return n; }
__device__ __forceinline__ double comp(const V3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__device__ __forceinline__ bool same_sign(double a, double b) { return (a > 0) == (b > 0); }

// ---- bodies ---------------------------------------------------------------------------------
struct BodyPoint {  // one obstacle point held in registers
  V3 q;
  static constexpr int N = 1;
  __device__ __forceinline__ V3 get(int) const { return q; }
};
struct BodyTri {  // one obstacle triangle held in registers (body 2 of the reference's dormant triangle path, Step.h:390-404)
  V3 a, b, c;
  static constexpr int N = 3;
  __device__ __forceinline__ V3 get(int i) const { return i == 0 ? a : (i == 1 ? b : c); }
  // the same for an index that differs from lane to lane (gjk_wave_run: lane -> vertex): component-wise selects on values pinned in registers.  Left to itself the
  // compiler turns the selects of a triangle that sits in memory into ONE load through a selected address -- a private array, i.e. scratch memory (k_mid<3>: 80 bytes
  // per lane and three scratch round trips at the head of every wave query).
  __device__ __forceinline__ V3 get_dyn(int i) const {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z, cx = c.x, cy = c.y, cz = c.z;
    asm volatile("" : "+v"(ax), "+v"(ay), "+v"(az), "+v"(bx), "+v"(by), "+v"(bz), "+v"(cx), "+v"(cy), "+v"(cz));
    const bool i0 = i == 0, i1 = i == 1;
    return V3{i0 ? ax : (i1 ? bx : cx), i0 ? ay : (i1 ? by : cy), i0 ? az : (i1 ? bz : cz)};
  }
};
struct BodyHull {  // 6 control points of one Bezier segment, row-major [6][3] (LDS or global)
  const double* p;
  static constexpr int N = 6;
  __device__ __forceinline__ V3 get(int i) const { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
};
template <int STRIDE>
struct BodyHullT {  // the same, one hull per LANE in a transposed LDS tile: entry e of this lane's hull at p[e * STRIDE] (p already points at the lane's column)
  const double* p;
  static constexpr int N = 6;
  __device__ __forceinline__ V3 get(int i) const { return V3{p[(3 * i) * STRIDE], p[(3 * i + 1) * STRIDE], p[(3 * i + 2) * STRIDE]}; }
};
struct BodySwept {  // conv{P, P + t*D}: 12 points (CCD.h:119-120), never stored
  const double* p; const double* d; double t;
  static constexpr int N = 12;
  __device__ __forceinline__ V3 get(int i) const {
    int j = i < 6 ? i : i - 6;
    double s = i < 6 ? 0.0 : t;
    return V3{p[3 * j] + s * d[3 * j], p[3 * j + 1] + s * d[3 * j + 1], p[3 * j + 2] + s * d[3 * j + 2]};
  }
};

// ---- simplex in registers --------------------------------------------------------------------
struct Simplex {
  int n;
  V3 v0, v1, v2, v3;
  int w0, w1, w2, w3;
  double l0, l1, l2, l3;
};
__device__ __forceinline__ V3 sx_v(const Simplex& s, int i) { return i == 0 ? s.v0 : (i == 1 ? s.v1 : (i == 2 ? s.v2 : s.v3)); }
__device__ __forceinline__ V3 sel3(bool c, const V3& a, const V3& b) { return V3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
// writes are value selects on every slot (never a store through a selected address), so the
// simplex can stay in registers
__device__ __forceinline__ void sx_set_v(Simplex& s, int i, const V3& p) { s.v0 = sel3(i == 0, p, s.v0); s.v1 = sel3(i == 1, p, s.v1); s.v2 = sel3(i == 2, p, s.v2); s.v3 = sel3(i == 3, p, s.v3); }
__device__ __forceinline__ int sx_w(const Simplex& s, int i) { return i == 0 ? s.w0 : (i == 1 ? s.w1 : (i == 2 ? s.w2 : s.w3)); }
__device__ __forceinline__ void sx_set_w(Simplex& s, int i, int w) { s.w0 = i == 0 ? w : s.w0; s.w1 = i == 1 ? w : s.w1; s.w2 = i == 2 ? w : s.w2; s.w3 = i == 3 ? w : s.w3; }
__device__ __forceinline__ double sx_l(const Simplex& s, int i) { return i == 0 ? s.l0 : (i == 1 ? s.l1 : (i == 2 ? s.l2 : s.l3)); }
__device__ __forceinline__ void sx_set_l(Simplex& s, int i, double l) { s.l0 = i == 0 ? l : s.l0; s.l1 = i == 1 ? l : s.l1; s.l2 = i == 2 ? l : s.l2; s.l3 = i == 3 ? l : s.l3; }

// sum_i lambda_i v_i, accumulated from zero in vertex order (openGJK.c:157-162)
__device__ __forceinline__ V3 sx_point(const Simplex& s) {
  V3 r{0, 0, 0};
  r.x += s.l0 * s.v0.x; r.y += s.l0 * s.v0.y; r.z += s.l0 * s.v0.z;
  if (s.n > 1) { r.x += s.l1 * s.v1.x; r.y += s.l1 * s.v1.y; r.z += s.l1 * s.v1.z; }
  if (s.n > 2) { r.x += s.l2 * s.v2.x; r.y += s.l2 * s.v2.y; r.z += s.l2 * s.v2.z; }
  if (s.n > 3) { r.x += s.l3 * s.v3.x; r.y += s.l3 * s.v3.y; r.z += s.l3 * s.v3.z; }
  return r;
}

// segment {v0 = B, v1 = A} (openGJK.c:82-163)
__device__ __forceinline__ void gjk_seg(Simplex& s) {
  const V3 b = s.v0, a = s.v1;
  const V3 t{b.x - a.x, b.y - a.y, b.z - a.z};
  const double f0 = fabs(t.x), f1 = fabs(t.y), f2 = fabs(t.z);
  int I = 1;
  if (f0 > f1) I = (f0 > f2) ? 0 : 2;
  else if (f0 < f1) I = (f1 > f2) ? 1 : 2;
  else if (f0 < f2) I = 2;
  else if (f1 < f2) I = 2;
  const double aI = comp(a, I), bI = comp(b, I), tI = comp(t, I);
  const double pt = dot(b, t) / dot(t, t) * (aI - bI) + bI;
  const double det_ap = aI - pt, det_pb = pt - bI;
  const int F0 = same_sign(tI, -1 * det_ap), F1 = same_sign(tI, -1 * det_pb);
  if (F0 + F1 == 2) {
    s.l0 = det_ap * -1.0 / tI; s.l1 = 1 - s.l0; s.w0 = 0; s.w1 = 1; s.n = 2;
  } else if (F0 == 0) {
    s.l0 = 1; s.w0 = 0; s.n = 1; s.v0 = s.v1;
  } else {
    s.l0 = 1; s.w0 = 1; s.n = 1;
  }
}

// triangle {v0 = C, v1 = B, v2 = A} (openGJK.c:168-393).
// Code-size note: the reference's S2D/S3D call the lower-dimensional routine from many branches.  Inlined
// naively that is ~50 copies of the segment routine and ~10 of the triangle routine per GJK instance
// (76 KB of ISA -- more than the instruction cache, and every divergent lane walks its own copy).  Here each
// routine first DECIDES which sub-simplices to visit, then visits them in a loop with ONE call site, then
// applies the branch's relabelling: same arithmetic and same results, a tenth of the code, and lanes on
// different branches reconverge on the shared call.
__device__ __forceinline__ void gjk_tri(Simplex& s) {
  const V3 c = s.v0, b = s.v1, a = s.v2;
  const V3 s21{b.x - a.x, b.y - a.y, b.z - a.z}, s31{c.x - a.x, c.y - a.y, c.z - a.z};
  // cofactors of the projected triangle, cyclic (k,l) = (1,2),(2,0),(0,1); sign (-1)^i
  const double nu0 = 1.0 * (b.y * c.z + a.y * b.z + c.y * a.z - b.y * a.z - c.y * b.z - a.y * c.z);
  const double nu1 = -1.0 * (b.z * c.x + a.z * b.x + c.z * a.x - b.z * a.x - c.z * b.x - a.z * c.x);
  const double nu2 = 1.0 * (b.x * c.y + a.x * b.y + c.x * a.y - b.x * a.y - c.x * b.y - a.x * c.y);
  const double f0 = fabs(nu0), f1 = fabs(nu1), f2 = fabs(nu2);
  int I = 1, J0 = -1, J1 = 0;  // J0 = -1: the reference reads out of bounds here; see oracle/orc_gjk.cpp
  if (f0 > f1) { if (f0 > f2) { I = 0; J0 = 1; J1 = 2; } else { I = 2; J0 = 0; J1 = 1; } }
  else if (f0 < f1) { if (f1 > f2) { I = 1; J0 = 0; J1 = 2; } else { I = 2; J0 = 0; J1 = 1; } }
  else if (f0 < f2) { I = 2; J0 = 0; J1 = 1; }
  const double nu_max = I == 0 ? nu0 : (I == 1 ? nu1 : nu2);
  V3 n;
  double nn = 0;
  n.x = s21.y * s31.z - s21.z * s31.y; nn += n.x * n.x;
  n.y = s21.z * s31.x - s21.x * s31.z; nn += n.y * n.y;
  n.z = s21.x * s31.y - s21.y * s31.x; nn += n.z * n.z;
  const double inv_len = 1 / sqrt(nn);
  n.x = n.x * inv_len; n.y = n.y * inv_len; n.z = n.z * inv_len;
  const double dna = dot(n, a);
  auto at = [](const V3& p, int j) { return j < 0 ? 0.0 : comp(p, j); };
  const double pp0 = dna * at(n, J0), pp1 = dna * at(n, J1);
  const double sa0 = at(a, J0), sa1 = at(a, J1), sb0 = at(b, J0), sb1 = at(b, J1), sc0 = at(c, J0), sc1 = at(c, J1);
  // (k,l) = (b,c), (c,a), (a,b)
  const double B0 = pp0 * sb1 + pp1 * sc0 + sb0 * sc1 - pp0 * sc1 - pp1 * sb0 - sc0 * sb1;
  const double B1 = pp0 * sc1 + pp1 * sa0 + sc0 * sa1 - pp0 * sa1 - pp1 * sc0 - sa0 * sc1;
  const double B2 = pp0 * sa1 + pp1 * sb0 + sa0 * sb1 - pp0 * sb1 - pp1 * sa0 - sb0 * sa1;
  const int F0 = same_sign(nu_max, B0), F1 = same_sign(nu_max, B1), F2 = same_sign(nu_max, B2);

  const bool both = F1 + F2 == 0 || isnan(n.x);  // origin outside two edges: try {B,A} and {C,A}, keep the closer
  if (!both && F0 + F1 + F2 == 3) {
    const double inv = 1 / nu_max;
    s.l0 = B2 * inv; s.l1 = B1 * inv; s.l2 = 1 - s.l0 - s.l1;
    s.w0 = 0; s.w1 = 1; s.w2 = 2; s.n = 3;
    return;
  }
  // edge visited by job j: kind 0 = {B,A} then {C,A}; 1 = {B,A} (F2 == 0); 2 = {C,A} (F1 == 0); 3 = {C,B}
  const int kind = both ? 0 : (F2 == 0 ? 1 : (F1 == 0 ? 2 : 3));
  const int njobs = both ? 2 : 1;
  Simplex aux, cur;
  aux.n = 1; aux.l0 = aux.l1 = 0; aux.w0 = aux.w1 = 0; aux.v0 = aux.v1 = c;
#pragma unroll 1
  for (int j = 0; j < njobs; ++j) {
    const bool edgeBA = (kind == 0 && j == 0) || kind == 1;
    const bool edgeCA = (kind == 0 && j == 1) || kind == 2;
    cur.n = 2;
    cur.v0 = edgeBA ? b : c;
    cur.v1 = (edgeBA || edgeCA) ? a : b;
    gjk_seg(cur);
    if (kind == 0 && j == 0) { aux.n = cur.n; aux.v0 = cur.v0; aux.v1 = cur.v1; aux.l0 = cur.l0; aux.l1 = cur.l1; aux.w0 = cur.w0; aux.w1 = cur.w1; }
  }
  s.n = cur.n; s.v0 = cur.v0; s.v1 = cur.v1; s.l0 = cur.l0; s.l1 = cur.l1; s.w0 = cur.w0; s.w1 = cur.w1;
  if (kind == 0) {
    const V3 vt = sx_point(aux), v = sx_point(s);
    if (dot(v, v) < dot(vt, vt)) {
      if (s.n > 1) s.w1 = s.w1 + 1;
    } else {  // labels and weights from the auxiliary segment, vertices stay (reference quirk)
      s.n = aux.n; s.l0 = aux.l0; s.w0 = aux.w0;
      if (s.n > 1) { s.l1 = aux.l1; s.w1 = aux.w1; }
    }
  } else if (kind == 2) {
    if (s.n > 1) s.w1 = s.w1 + 1;
  }
}

// vertex i of {p0,p1,p2,p3} as a chain of value selects on explicit operands: a by-index read of a copied struct is
// turned into a private-memory (scratch) array by the compiler, i.e. a ~0.5 us round trip per tetrahedron step
__device__ __forceinline__ V3 pick4(int i, const V3& p0, const V3& p1, const V3& p2, const V3& p3) {
  return sel3(i == 0, p0, sel3(i == 1, p1, sel3(i == 2, p2, p3)));
}
__device__ __forceinline__ int tri_lut(int i) {  // {3,3,3, 1,2,2, 0,0,1}
  return (0x100221333 >> (4 * i)) & 0xF;
}
__device__ __forceinline__ double det3x(const V3& p, const V3& q, const V3& r) {
  return p.x * q.y * r.z + p.y * q.z * r.x + p.z * q.x * r.y - p.z * q.y * r.x - p.y * q.x * r.z - p.x * q.z * r.y;
}

// tetrahedron {v0 = D, v1 = C, v2 = B, v3 = A} (openGJK.c:398-711)
__device__ __forceinline__ void gjk_tet(Simplex& s) {
  const V3 d = s.v0, c = s.v1, b = s.v2, a = s.v3;
  const double B0 = -1 * det3x(b, c, d);
  const double B1 = +1 * det3x(a, c, d);
  const double B2 = -1 * det3x(a, b, d);
  const double B3 = +1 * det3x(a, b, c);
  const double detM = B0 + B1 + B2 + B3;
  int F0 = 1, F1 = 1, F2 = 1, F3 = 1;
  const double eps = 1e-13;
  if (fabs(detM) < eps) {
    if (fabs(B2) < eps && fabs(B3) < eps) F1 = 0;
    else if (fabs(B1) < eps && fabs(B3) < eps) F2 = 0;
    else if (fabs(B1) < eps && fabs(B2) < eps) F3 = 0;
    else if (fabs(B0) < eps && fabs(B3) < eps) F1 = 0;
    else if (fabs(B0) < eps && fabs(B2) < eps) F1 = 0;
    else if (fabs(B0) < eps && fabs(B1) < eps) F2 = 0;
    else { F0 = F1 = F2 = F3 = 0; }
  } else {
    F0 = same_sign(detM, B0); F1 = same_sign(detM, B1); F2 = same_sign(detM, B2); F3 = same_sign(detM, B3);
  }
  const int facing = F1 + F2 + F3;
  if (F0 + facing == 4) {
    const double inv = 1 / detM;
    s.l3 = B0 * inv; s.l2 = B1 * inv; s.l1 = B2 * inv; s.l0 = 1 - s.l1 - s.l2 - s.l3;
    s.w0 = 0; s.w1 = 1; s.w2 = 2; s.w3 = 3; s.n = 4;
    return;
  }
  // Faces visited, as vertex triples (v0,v1,v2) of the sub-triangle taken from {D,C,B,A} = indices {0,1,2,3}:
  //   t0 = (D,C,A) "ACD", t1 = (D,B,A) "ABD", t2 = (C,B,A) "ABC", t3 = (D,C,B)
  //   facing 0: t0, t1, t2, keep the closest          (openGJK.c:470-520)
  //   facing 1: two faces; first -> aux, second -> s   (:521-640)
  //   facing 2: one face in place                      (:641-680)
  //   facing 3: t3, labels shifted by one              (:681-700)
  const int njobs = facing == 0 ? 3 : (facing == 1 ? 2 : 1);
  const int first = F1 == 0 ? 0 : 1;                          // facing 1: face of job 0
  const int second = (F1 == 0 && F2 == 0) ? 1 : 2;            // facing 1: face of job 1
  const int single = facing == 3 ? 3 : (F1 == 0 ? 0 : (F2 == 0 ? 1 : 2));  // facing 2 / 3
  Simplex aux, cur;
  aux = s;
  int id0 = 0, id1 = 0, id2 = 0, nbest = 0;
  double lb0 = 0, lb1 = 0, lb2 = 0, best = 0;
#pragma unroll 1
  for (int j = 0; j < njobs; ++j) {
    const int t = facing == 0 ? j : (facing == 1 ? (j == 0 ? first : second) : single);
    cur.n = 3;
    cur.v0 = sel3(t == 2, c, d); cur.v1 = sel3(t == 0 || t == 3, c, b); cur.v2 = sel3(t == 3, b, a);   // vertices (ia, ib, ic) of {d,c,b,a}
    gjk_tri(cur);
    if (facing == 0) {
      const V3 vt = sx_point(cur);
      const double dd = dot(vt, vt);
      if (j == 0 || dd < best) {
        best = dd; nbest = cur.n;
        id0 = tri_lut(j + cur.w0 * 3); lb0 = cur.l0;
        if (nbest > 1) { id1 = tri_lut(j + cur.w1 * 3); lb1 = cur.l1; }
        if (nbest > 2) { id2 = tri_lut(j + cur.w2 * 3); lb2 = cur.l2; }
      }
    } else if (facing == 1 && j == 0) {
      aux.n = cur.n; aux.v0 = cur.v0; aux.v1 = cur.v1; aux.v2 = cur.v2;
      aux.l0 = cur.l0; aux.l1 = cur.l1; aux.l2 = cur.l2; aux.w0 = cur.w0; aux.w1 = cur.w1; aux.w2 = cur.w2;
      const V3 vt = sx_point(cur); best = dot(vt, vt);
    }
  }
  if (facing == 0) {
    s.n = nbest;
    sx_set_v(s, nbest - 1, pick4(id0, d, c, b, a)); s.l0 = lb0; sx_set_w(s, nbest - 1, id0);
    if (nbest > 1) { sx_set_v(s, nbest - 2, pick4(id1, d, c, b, a)); s.l1 = lb1; sx_set_w(s, nbest - 2, id1); }
    if (nbest > 2) { sx_set_v(s, nbest - 3, pick4(id2, d, c, b, a)); s.l2 = lb2; sx_set_w(s, nbest - 3, id2); }
    return;
  }
  // the (last) visited face becomes the simplex
  s.n = cur.n; s.v0 = cur.v0; s.v1 = cur.v1; s.v2 = cur.v2;
  s.l0 = cur.l0; s.l1 = cur.l1; s.l2 = cur.l2; s.w0 = cur.w0; s.w1 = cur.w1; s.w2 = cur.w2;
  if (facing == 1) {
    const V3 v = sx_point(s);
    if (dot(v, v) < best) {
      for (int i = 0; i < s.n; ++i) sx_set_w(s, s.n - 1 - i, tri_lut(second + sx_w(s, i) * 3));  // in place, as the reference
    } else {
      s.n = aux.n; s.v0 = aux.v0; s.v1 = aux.v1; s.v2 = aux.v2;
      s.l0 = aux.l0; s.l1 = aux.l1; s.l2 = aux.l2;
      for (int i = 0; i < s.n; ++i) sx_set_w(s, aux.n - 1 - i, tri_lut(first + sx_w(aux, i) * 3));
    }
  } else if (facing == 2) {
    if (single == 1 && s.n > 2) s.w2 = s.w2 + 1;
  } else {
    s.w0 = s.w0 + 1;
    if (s.n > 1) s.w1 = s.w1 + 1;
    if (s.n > 2) s.w2 = s.w2 + 1;
  }
}

// "sticky" support: keep the previous support unless some vertex is strictly better, first
// maximum wins (openGJK.c:714-737)
template <class Body>
__device__ __forceinline__ void support(const Body& body, const V3& dir, V3& cur) {
  double best = dot(cur, dir);
  V3 pick = cur;
  if constexpr (Body::N <= 3) {   // a point or a triangle held in registers: written out -- a loop over its vertices indexes them at run time, i.e. through a private array (scratch memory)
#pragma unroll
    for (int i = 0; i < Body::N; ++i) {
      const V3 p = body.get(i);
      const double sd = dot(p, dir);
      if (sd > best) { best = sd; pick = p; }
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < Body::N; ++i) {
      const V3 p = body.get(i);
      const double sd = dot(p, dir);
      if (sd > best) { best = sd; pick = p; }
    }
  }
  cur = pick;
}

// witness vector of conv(b1) - conv(b2) (openGJK.c:754-852)
// kmax < 50 / cut: stop after kmax iterations; *cut tells whether the loop was cut short (the witness vector is then not final --
// the caller hands the query to a wave-cooperative solve that starts over and reaches the same bits as the uncut loop)
template <class B1, class B2>
__device__ __forceinline__ V3 gjk(const B1& b1, const B2& b2, int* iters_out = nullptr, int kmax = 50, bool* cut = nullptr) {
  const double eps_rel2 = 1e-5 * 1e-5, eps_tot = 1e-15;
  Simplex s;
  V3 s1 = b1.get(0), s2 = b2.get(0);
  V3 v{s1.x - s2.x, s1.y - s2.y, s1.z - s2.z};
  s.n = 1; s.v0 = v;
  s.v1 = s.v2 = s.v3 = V3{0, 0, 0};
  s.w0 = s.w1 = s.w2 = s.w3 = 0; s.l0 = s.l1 = s.l2 = s.l3 = 0;
  double wmax2 = 0;
  int k = 0;
  bool fin = false;
  do {
    k++;
    const V3 vm{-v.x, -v.y, -v.z};
    support(b1, vm, s1);
    support(b2, v, s2);
    const V3 w{s1.x - s2.x, s1.y - s2.y, s1.z - s2.z};
    if ((sq(v) - dot(v, w)) <= eps_rel2 * sq(v)) { fin = true; break; }
    if (sq(v) < eps_rel2) { fin = true; break; }
    sx_set_v(s, s.n, w);
    s.n++;
    if (s.n == 4) gjk_tet(s); else if (s.n == 3) gjk_tri(s); else gjk_seg(s);
    v = sx_point(s);
    { double t = sq(s.v0); if (t > wmax2) wmax2 = t; }
    if (s.n > 1) { double t = sq(s.v1); if (t > wmax2) wmax2 = t; }
    if (s.n > 2) { double t = sq(s.v2); if (t > wmax2) wmax2 = t; }
    if (s.n > 3) { double t = sq(s.v3); if (t > wmax2) wmax2 = t; }
    if (sq(v) <= (eps_tot * eps_tot * wmax2)) { fin = true; break; }
  } while ((s.n != 4) && (k != kmax));
  if (iters_out) *iters_out = k;
  if (cut) *cut = !fin && s.n != 4 && kmax < 50;
  return v;
}

// ---- wave-cooperative variant -----------------------------------------------------------------
// One query per WAVEFRONT (the inter-robot kernels: a robot pair per wave; the per-candidate obstacle solve).  With 64 lanes
// available the independent pieces run side by side and the results are broadcast back with v_readlane, so every decision is
// taken on wave-uniform values:
//   * both support searches at once: lanes 0..15 hold the vertices of body 1, lanes 16..31 those of body 2 (loaded once per
//     query); a 4-step DPP butterfly gives each row its maximum, a ballot picks the FIRST lane that attains it (openGJK's
//     "first maximum wins"), and the previous support -- tracked as a lane index -- is kept unless it is not among them;
//   * the three edges a triangle step may fall back to are solved speculatively on three lanes (seg_core, gjk_tri_uni);
//   * the faces a tetrahedron step has to visit (up to three) are solved by lanes 0..2 in parallel (gjk_tet_wave).
// Every floating-point expression is the one the per-lane version evaluates, on the same operands, so the
// witness vector is bit-identical (pinned by the same golden vectors, tests/test_gpu_parity.py).
__device__ __forceinline__ double gjk_rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ V3 gjk_rl3(const V3& v, int lane) { return V3{gjk_rl(v.x, lane), gjk_rl(v.y, lane), gjk_rl(v.z, lane)}; }
// maximum over each row of 16 lanes, left in every lane of the row (NaNs are ignored like the `>` test does)
// Every lane of a row is a valid DPP source for these controls, so the moves need no `old` operand (no copy in front of them),
// and the maximum is the bare instruction: fmax() would first quiet a signalling NaN of the shuffled operand -- one more
// v_max_f64 per step on a chain of four.
template <int CTRL>
__device__ __forceinline__ double gjk_dpp_mov(double v) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ double gjk_vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double gjk_row_max(double v) {
  v = gjk_vmax(v, gjk_dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
  v = gjk_vmax(v, gjk_dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
  v = gjk_vmax(v, gjk_dpp_mov<0x124>(v));  // row_ror:4
  v = gjk_vmax(v, gjk_dpp_mov<0x128>(v));  // row_ror:8
  return v;
}

// gjk_tet with the visited faces solved by lanes 0..2 side by side (see gjk_tet for the case analysis)
__device__ __forceinline__ void gjk_tet_wave(Simplex& s, int lane) {
  const V3 d = s.v0, c = s.v1, b = s.v2, a = s.v3;
  const double B0 = -1 * det3x(b, c, d);
  const double B1 = +1 * det3x(a, c, d);
  const double B2 = -1 * det3x(a, b, d);
  const double B3 = +1 * det3x(a, b, c);
  const double detM = B0 + B1 + B2 + B3;
  int F0 = 1, F1 = 1, F2 = 1, F3 = 1;
  const double eps = 1e-13;
  if (fabs(detM) < eps) {
    if (fabs(B2) < eps && fabs(B3) < eps) F1 = 0;
    else if (fabs(B1) < eps && fabs(B3) < eps) F2 = 0;
    else if (fabs(B1) < eps && fabs(B2) < eps) F3 = 0;
    else if (fabs(B0) < eps && fabs(B3) < eps) F1 = 0;
    else if (fabs(B0) < eps && fabs(B2) < eps) F1 = 0;
    else if (fabs(B0) < eps && fabs(B1) < eps) F2 = 0;
    else { F0 = F1 = F2 = F3 = 0; }
  } else {
    F0 = same_sign(detM, B0); F1 = same_sign(detM, B1); F2 = same_sign(detM, B2); F3 = same_sign(detM, B3);
  }
  const int facing = F1 + F2 + F3;
  if (F0 + facing == 4) {
    const double inv = 1 / detM;
    s.l3 = B0 * inv; s.l2 = B1 * inv; s.l1 = B2 * inv; s.l0 = 1 - s.l1 - s.l2 - s.l3;
    s.w0 = 0; s.w1 = 1; s.w2 = 2; s.w3 = 3; s.n = 4;
    return;
  }
  const int njobs = facing == 0 ? 3 : (facing == 1 ? 2 : 1);
  const int first = F1 == 0 ? 0 : 1;
  const int second = (F1 == 0 && F2 == 0) ? 1 : 2;
  const int single = facing == 3 ? 3 : (F1 == 0 ? 0 : (F2 == 0 ? 1 : 2));
  // this lane's face
  const int j = lane < njobs ? lane : njobs - 1;
  const int t = facing == 0 ? j : (facing == 1 ? (j == 0 ? first : second) : single);
  Simplex mine;
  mine.n = 3;
  mine.v0 = sel3(t == 2, c, d); mine.v1 = sel3(t == 0 || t == 3, c, b); mine.v2 = sel3(t == 3, b, a);   // vertices (ia, ib, ic) of {d,c,b,a}
  mine.v3 = V3{0, 0, 0}; mine.l0 = mine.l1 = mine.l2 = mine.l3 = 0; mine.w0 = mine.w1 = mine.w2 = mine.w3 = 0;
  gjk_tri(mine);
  // result of job jj, broadcast to the whole wave
  auto fetch = [&](int jj, Simplex& r) {
    r.n = __builtin_amdgcn_readlane(mine.n, jj);
    r.v0 = gjk_rl3(mine.v0, jj); r.v1 = gjk_rl3(mine.v1, jj); r.v2 = gjk_rl3(mine.v2, jj);
    r.l0 = gjk_rl(mine.l0, jj); r.l1 = gjk_rl(mine.l1, jj); r.l2 = gjk_rl(mine.l2, jj);
    r.w0 = __builtin_amdgcn_readlane(mine.w0, jj); r.w1 = __builtin_amdgcn_readlane(mine.w1, jj); r.w2 = __builtin_amdgcn_readlane(mine.w2, jj);
  };
  // every job's lane forms the squared length of ITS candidate point itself; only the three numbers travel, the comparisons
  // run in the reference's order on wave-uniform values, and only the winner's simplex is fetched (a fetch is 28 v_readlane)
  const V3 vm_ = sx_point(mine);
  const double dd_m = dot(vm_, vm_);
  Simplex cur;
  cur.v3 = V3{0, 0, 0}; cur.l3 = 0; cur.w3 = 0;
  if (facing == 0) {
    const double dd0 = gjk_rl(dd_m, 0), dd1 = gjk_rl(dd_m, 1), dd2 = gjk_rl(dd_m, 2);
    int jb = 0; double best = dd0;
    if (dd1 < best) { best = dd1; jb = 1; }
    if (dd2 < best) { best = dd2; jb = 2; }
    const int nbest = __builtin_amdgcn_readlane(mine.n, jb);
    const int cw0 = __builtin_amdgcn_readlane(mine.w0, jb), cw1 = __builtin_amdgcn_readlane(mine.w1, jb), cw2 = __builtin_amdgcn_readlane(mine.w2, jb);
    const double lb0 = gjk_rl(mine.l0, jb), lb1 = gjk_rl(mine.l1, jb), lb2 = gjk_rl(mine.l2, jb);
    const int id0 = tri_lut(jb + cw0 * 3), id1 = tri_lut(jb + cw1 * 3), id2 = tri_lut(jb + cw2 * 3);
    s.n = nbest;
    sx_set_v(s, nbest - 1, pick4(id0, d, c, b, a)); s.l0 = lb0; sx_set_w(s, nbest - 1, id0);
    if (nbest > 1) { sx_set_v(s, nbest - 2, pick4(id1, d, c, b, a)); s.l1 = lb1; sx_set_w(s, nbest - 2, id1); }
    if (nbest > 2) { sx_set_v(s, nbest - 3, pick4(id2, d, c, b, a)); s.l2 = lb2; sx_set_w(s, nbest - 3, id2); }
    return;
  }
  if (facing == 1) {   // two faces: job 0 is the reference's auxiliary simplex, job 1 the one left in place; the closer one stays
    const double best = gjk_rl(dd_m, 0), dcur = gjk_rl(dd_m, 1);
    if (dcur < best) {
      fetch(1, cur);
      s.n = cur.n; s.v0 = cur.v0; s.v1 = cur.v1; s.v2 = cur.v2;
      s.l0 = cur.l0; s.l1 = cur.l1; s.l2 = cur.l2; s.w0 = cur.w0; s.w1 = cur.w1; s.w2 = cur.w2;
      for (int i = 0; i < s.n; ++i) sx_set_w(s, s.n - 1 - i, tri_lut(second + sx_w(s, i) * 3));  // in place, as the reference
    } else {
      Simplex aux;
      aux.v3 = V3{0, 0, 0}; aux.l3 = 0; aux.w3 = 0;
      fetch(0, aux);
      s.n = aux.n; s.v0 = aux.v0; s.v1 = aux.v1; s.v2 = aux.v2;
      s.l0 = aux.l0; s.l1 = aux.l1; s.l2 = aux.l2;
      // the labels the reference leaves behind are those of the in-place simplex beyond aux.n ... it rewrites aux.n of them
      const int pw0 = __builtin_amdgcn_readlane(mine.w0, 1), pw1 = __builtin_amdgcn_readlane(mine.w1, 1), pw2 = __builtin_amdgcn_readlane(mine.w2, 1);
      s.w0 = pw0; s.w1 = pw1; s.w2 = pw2;
      for (int i = 0; i < s.n; ++i) sx_set_w(s, aux.n - 1 - i, tri_lut(first + sx_w(aux, i) * 3));
    }
    return;
  }
  fetch(njobs - 1, cur);  // the visited face becomes the simplex
  s.n = cur.n; s.v0 = cur.v0; s.v1 = cur.v1; s.v2 = cur.v2;
  s.l0 = cur.l0; s.l1 = cur.l1; s.l2 = cur.l2; s.w0 = cur.w0; s.w1 = cur.w1; s.w2 = cur.w2;
  if (facing == 2) {
    if (single == 1 && s.n > 2) s.w2 = s.w2 + 1;
  } else {
    s.w0 = s.w0 + 1;
    if (s.n > 1) s.w1 = s.w1 + 1;
    if (s.n > 2) s.w2 = s.w2 + 1;
  }
}

// ---- wave-UNIFORM control -----------------------------------------------------------------------------------------------
// One wave issues one instruction every 4-5 cycles whatever its lanes do, so a query that owns a wave is bound by its
// instruction COUNT.  The per-lane routines above, called with wave-uniform operands, still pay for divergence the compiler has
// to assume: exec-mask bookkeeping around every branch, whole-simplex copies at every merge, select chains instead of moves
// (a triangle step that ends on an edge was ~700 issued instructions, 1.5 us).  The routines below take every decision on a
// scalar condition (`uni`: ballot != 0), specialise the projection plane of a triangle step by a scalar switch, move vertices
// only on the path that needs it, track the sticky supports as LANE INDICES (no base dot products, no value selects), and
// solve the three edges a triangle step may fall back to speculatively on three lanes of ONE straight-line instruction stream
// (seg_core) next to the cofactor / normal chain.  Every floating-point expression is the one gjk_seg / gjk_tri evaluate, on
// the same operands, in the same association -- the witness vector is bit-identical (tests/golden/gjk_kat.npz, tri_kat.npz).
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }   // c is wave-uniform: make it a scalar condition
__device__ __forceinline__ double flip_sign(double x, unsigned sgn) { return __hiloint2double(__double2hiint(x) ^ (int)sgn, __double2loint(x)); }

// gjk_seg (openGJK.c:82-163) as straight-line code on per-lane operands {v0 = B = X, v1 = A = Y}:
//   ec = 0: both vertices stay, weights (l0, l1);  ec = 1: only A supports (v0 = v1, weight 1);  ec = 2: only B (weight 1)
__device__ __forceinline__ void seg_core(const V3& b, const V3& a, int& ec, double& l0, double& l1) {
  const V3 t{b.x - a.x, b.y - a.y, b.z - a.z};
  const double f0 = fabs(t.x), f1 = fabs(t.y), f2 = fabs(t.z);
  const bool g01 = f0 > f1, l01 = f0 < f1, g02 = f0 > f2, g12 = f1 > f2, l02 = f0 < f2, l12 = f1 < f2;   // mask arithmetic, no branches
  const bool is0 = g01 & g02;
  const bool is2 = (g01 & !g02) | (!g01 & l01 & !g12) | (!g01 & !l01 & (l02 | l12));
  const double aI = is0 ? a.x : (is2 ? a.z : a.y), bI = is0 ? b.x : (is2 ? b.z : b.y), tI = is0 ? t.x : (is2 ? t.z : t.y);
  const double pt = dot(b, t) / dot(t, t) * (aI - bI) + bI;
  const double det_ap = aI - pt, det_pb = pt - bI;
  const bool F0 = same_sign(tI, -1 * det_ap), F1 = same_sign(tI, -1 * det_pb);
  const double q = det_ap * -1.0 / tI;
  ec = (F0 & F1) ? 0 : (!F0 ? 1 : 2);
  l0 = (F0 & F1) ? q : 1.0;
  l1 = 1 - q;
}

// triangle {v0 = C, v1 = B, v2 = A} (openGJK.c:168-393) on wave-uniform operands; labels are not kept (nothing above the
// sub-algorithm reads them)
__device__ __forceinline__ void gjk_tri_uni(Simplex& s, int lane) {
  const V3 c = s.v0, b = s.v1, a = s.v2;
  // the three edges, one per lane (lane & 3 = 0: {B,A}, 1: {C,A}, 2 and 3: {C,B}); consumed only if the origin is outside the face
  const int e = lane & 3;
  int ec; double eq0, eq1;
  seg_core(sel3(e == 0, b, c), sel3(e >= 2, b, a), ec, eq0, eq1);
  const V3 s21{b.x - a.x, b.y - a.y, b.z - a.z}, s31{c.x - a.x, c.y - a.y, c.z - a.z};
  const double nu0 = 1.0 * (b.y * c.z + a.y * b.z + c.y * a.z - b.y * a.z - c.y * b.z - a.y * c.z);
  const double nu1 = -1.0 * (b.z * c.x + a.z * b.x + c.z * a.x - b.z * a.x - c.z * b.x - a.z * c.x);
  const double nu2 = 1.0 * (b.x * c.y + a.x * b.y + c.x * a.y - b.x * a.y - c.x * b.y - a.x * c.y);
  const double f0 = fabs(nu0), f1 = fabs(nu1), f2 = fabs(nu2);
  // projection plane: 0 = drop x (J = y,z), 1 = drop y (J = x,z), 2 = drop z (J = x,y), 3 = the reference's fall-through
  // (I = 1 with J0 = -1, an out-of-bounds read that yields 0 -- oracle/orc_gjk.cpp -- and J1 = 0)
  int proj;
  if (uni(f0 > f1)) proj = uni(f0 > f2) ? 0 : 2;
  else if (uni(f0 < f1)) proj = uni(f1 > f2) ? 1 : 2;
  else proj = uni(f0 < f2) ? 2 : 3;
  const double nu_max = proj == 0 ? nu0 : (proj == 2 ? nu2 : nu1);
  V3 n;
  double nn = 0;
  n.x = s21.y * s31.z - s21.z * s31.y; nn += n.x * n.x;
  n.y = s21.z * s31.x - s21.x * s31.z; nn += n.y * n.y;
  n.z = s21.x * s31.y - s21.y * s31.x; nn += n.z * n.z;
  const double inv_len = 1 / sqrt(nn);
  n.x = n.x * inv_len; n.y = n.y * inv_len; n.z = n.z * inv_len;
  const double dna = dot(n, a);
  double B0, B1, B2;
  auto Bs = [&](double nJ0, double nJ1, double sa0, double sa1, double sb0, double sb1, double sc0, double sc1) {
    const double pp0 = dna * nJ0, pp1 = dna * nJ1;
    B0 = pp0 * sb1 + pp1 * sc0 + sb0 * sc1 - pp0 * sc1 - pp1 * sb0 - sc0 * sb1;
    B1 = pp0 * sc1 + pp1 * sa0 + sc0 * sa1 - pp0 * sa1 - pp1 * sc0 - sa0 * sc1;
    B2 = pp0 * sa1 + pp1 * sb0 + sa0 * sb1 - pp0 * sb1 - pp1 * sa0 - sb0 * sa1;
  };
  if (proj == 0) Bs(n.y, n.z, a.y, a.z, b.y, b.z, c.y, c.z);
  else if (proj == 1) Bs(n.x, n.z, a.x, a.z, b.x, b.z, c.x, c.z);
  else if (proj == 2) Bs(n.x, n.y, a.x, a.y, b.x, b.y, c.x, c.y);
  else Bs(0.0, n.x, 0.0, a.x, 0.0, b.x, 0.0, c.x);
  const bool pos = uni(nu_max > 0);
  const int F0 = uni(B0 > 0) == pos, F1 = uni(B1 > 0) == pos, F2 = uni(B2 > 0) == pos;
  const bool both = F1 + F2 == 0 || uni(isnan(n.x));
  if (!both && F0 + F1 + F2 == 3) {
    const double inv = 1 / nu_max;
    s.l0 = B2 * inv; s.l1 = B1 * inv; s.l2 = 1 - s.l0 - s.l1;
    s.n = 3;
    return;
  }
  // edge that becomes the simplex: {B,A} if only F2 failed; {C,A} if F1 failed (or, `both`, after {B,A} went to the auxiliary
  // simplex); {C,B} otherwise
  const int lane_cur = both ? 1 : (F2 == 0 ? 0 : (F1 == 0 ? 1 : 2));
  const int cc = __builtin_amdgcn_readlane(ec, lane_cur);
  const double q0 = gjk_rl(eq0, lane_cur), q1 = gjk_rl(eq1, lane_cur);
  if (lane_cur == 0) { s.v0 = b; s.v1 = a; } else if (lane_cur == 1) s.v1 = a;   // {C,B} is in place
  if (cc == 1) s.v0 = s.v1;
  s.n = cc == 0 ? 2 : 1; s.l0 = q0; s.l1 = q1;
  if (both) {   // keep the closer of the two edges' points; if the auxiliary one wins, its count and weights are taken but the vertices stay (reference quirk)
    const int ca = __builtin_amdgcn_readlane(ec, 0);
    const double a0 = gjk_rl(eq0, 0), a1 = gjk_rl(eq1, 0);
    const V3 av0 = ca == 1 ? a : b;
    V3 vt{0, 0, 0};
    vt.x += a0 * av0.x; vt.y += a0 * av0.y; vt.z += a0 * av0.z;
    if (ca == 0) { vt.x += a1 * a.x; vt.y += a1 * a.y; vt.z += a1 * a.z; }
    const V3 v = sx_point(s);
    if (!uni(dot(v, v) < dot(vt, vt))) { s.n = ca == 0 ? 2 : 1; s.l0 = a0; if (ca == 0) s.l1 = a1; }
  }
}

// witness vector of conv(b1) - conv(b2), computed by the whole wave; all lanes must call it with the same bodies
// and all lanes return the same vector.
// The loop is RESUMABLE: its whole state between two iterations is GjkState (the bodies' vertices are reloaded), all of it
// wave-uniform.  gjk_wave_run leaves the loop after iteration k_stop if the query has not ended by then (finished = false)
// and continues from a saved state (fresh = false) with the same instruction sequence -- the iterations of a query may be
// spread over two kernels (kernels_pairs.h: spec_pair_body) and arrive at the same bits.
template <class B> __device__ __forceinline__ V3 body_get_dyn(const B& b, int i) { return b.get(i); }
__device__ __forceinline__ V3 body_get_dyn(const BodyTri& b, int i) { return b.get_dyn(i); }
struct GjkState { Simplex s; V3 v; double wmax2; int c1, c2, k; };
template <class B1, class B2>
__device__ __forceinline__ V3 gjk_wave_run(const B1& b1, const B2& b2, int lane, GjkState& st, bool fresh, int k_stop, bool& finished, long long* prof = nullptr) {   // prof (timing builds): [0] support, [1..3] segment / triangle / tetrahedron steps (100 MHz ticks), [4..6] their counts
  static_assert(B1::N <= 16 && B2::N <= 16, "one body per row of 16 lanes");
  const double eps_rel2 = 1e-5 * 1e-5, eps_tot = 1e-15;
  // lanes 0..15 hold the vertices of body 1, lanes 16..31 those of body 2, once for the whole query
  const bool row1 = lane < 16, row2 = lane >= 16 && lane < 32;
  const int idx = lane & 15;
  const bool valid = (row1 && idx < B1::N) || (row2 && idx < B2::N);
  V3 p{0, 0, 0};
  if (row1 && idx < B1::N) p = body_get_dyn(b1, idx);
  if (row2 && idx < B2::N) p = body_get_dyn(b2, idx);
  const unsigned sgn = row1 ? 0x80000000u : 0u;   // body 1 is searched along -v, body 2 along v
  // the sticky supports as lane indices: a support is always a vertex of its body, and dot(vertex, dir) of the kept one is
  // bit for bit the `best` the reference starts its scan with -- so "some vertex is strictly better" == "the kept lane is not
  // among the lanes that attain the row maximum", and then the FIRST such lane wins (openGJK.c:714-737)
  int c1 = 0, c2 = 16;
  Simplex s;
  V3 v{gjk_rl(p.x, 0) - gjk_rl(p.x, 16), gjk_rl(p.y, 0) - gjk_rl(p.y, 16), gjk_rl(p.z, 0) - gjk_rl(p.z, 16)};
  s.n = 1; s.v0 = v;
  s.v1 = s.v2 = s.v3 = V3{0, 0, 0};
  s.w0 = s.w1 = s.w2 = s.w3 = 0; s.l0 = s.l1 = s.l2 = s.l3 = 0;
  double wmax2 = 0;
  int k = 0;
  if (!fresh) {   // member by member: a struct copy under a run-time condition leaves a stack object behind (k_mid then reserves scratch)
    s.n = st.s.n; s.w0 = st.s.w0; s.w1 = st.s.w1; s.w2 = st.s.w2; s.w3 = st.s.w3;
    s.v0.x = st.s.v0.x; s.v0.y = st.s.v0.y; s.v0.z = st.s.v0.z; s.v1.x = st.s.v1.x; s.v1.y = st.s.v1.y; s.v1.z = st.s.v1.z;
    s.v2.x = st.s.v2.x; s.v2.y = st.s.v2.y; s.v2.z = st.s.v2.z; s.v3.x = st.s.v3.x; s.v3.y = st.s.v3.y; s.v3.z = st.s.v3.z;
    s.l0 = st.s.l0; s.l1 = st.s.l1; s.l2 = st.s.l2; s.l3 = st.s.l3;
    v.x = st.v.x; v.y = st.v.y; v.z = st.v.z; wmax2 = st.wmax2; c1 = st.c1; c2 = st.c2; k = st.k;
  }
  finished = true;
  for (;;) {
    k++;
#ifdef TJ_PHASE_TIMING
    const long long tp0 = prof ? wall_clock64() : 0;
#endif
    const double sd0 = p.x * flip_sign(v.x, sgn) + p.y * flip_sign(v.y, sgn) + p.z * flip_sign(v.z, sgn);
    const double sd = valid ? sd0 : -INFINITY;
    const double m = gjk_row_max(sd);
    const unsigned long long hit = __ballot(valid && sd == m);
    const unsigned h1 = (unsigned)(hit & 0xFFFFull), h2 = (unsigned)((hit >> 16) & 0xFFFFull);
    if (h1 && !((h1 >> c1) & 1u)) c1 = __ffs(h1) - 1;
    if (h2 && !((h2 >> (c2 - 16)) & 1u)) c2 = 16 + __ffs(h2) - 1;
    const V3 w{gjk_rl(p.x, c1) - gjk_rl(p.x, c2), gjk_rl(p.y, c1) - gjk_rl(p.y, c2), gjk_rl(p.z, c1) - gjk_rl(p.z, c2)};
    if (uni((sq(v) - dot(v, w)) <= eps_rel2 * sq(v))) break;
    if (uni(sq(v) < eps_rel2)) break;
#ifdef TJ_PHASE_TIMING
    const int kind = s.n;
    asm volatile("" :: "v"(w.x), "v"(w.y));
    const long long tp1 = prof ? wall_clock64() : 0;
#endif
    if (s.n == 1) {
      s.v1 = w;
      int ec; double q0, q1;
      seg_core(s.v0, s.v1, ec, q0, q1);
      const int cc = __builtin_amdgcn_readfirstlane(ec);
      if (cc == 1) s.v0 = s.v1;
      s.n = cc == 0 ? 2 : 1; s.l0 = q0; s.l1 = q1;
    } else if (s.n == 2) {
      s.v2 = w;
      gjk_tri_uni(s, lane);
    } else {
      s.v3 = w; s.n = 4;
      gjk_tet_wave(s, lane);
      s.n = __builtin_amdgcn_readfirstlane(s.n);
    }
    v = sx_point(s);
#ifdef TJ_PHASE_TIMING
    if (prof) { asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z)); const long long tp2 = wall_clock64(); prof[0] += tp1 - tp0; prof[kind] += tp2 - tp1; prof[3 + kind] += 1; }
#endif
    { double t = sq(s.v0); if (uni(t > wmax2)) wmax2 = t; }
    if (s.n > 1) { double t = sq(s.v1); if (uni(t > wmax2)) wmax2 = t; }
    if (s.n > 2) { double t = sq(s.v2); if (uni(t > wmax2)) wmax2 = t; }
    if (s.n > 3) { double t = sq(s.v3); if (uni(t > wmax2)) wmax2 = t; }
    if (uni(sq(v) <= (eps_tot * eps_tot * wmax2))) break;
    if (!((s.n != 4) && (k != 50))) break;
    if (k == k_stop) { finished = false; break; }   // to be continued: the state below is that of "about to start iteration k + 1"
  }
  st.s = s; st.v = v; st.wmax2 = wmax2; st.c1 = c1; st.c2 = c2; st.k = k;
  return v;
}
template <class B1, class B2>
__device__ __forceinline__ V3 gjk_wave(const B1& b1, const B2& b2, int lane, int* iters_out = nullptr, long long* prof = nullptr) {
  GjkState st; bool fin;
  const V3 v = gjk_wave_run(b1, b2, lane, st, true, 50, fin, prof);
  if (iters_out) *iters_out = st.k;
  return v;
}

}  // namespace tj
