// oracle/orc_gjk.cpp -- TEST INFRASTRUCTURE ONLY (CPU checker, never shipped, never linked
// into the product library).
//
// CPU restatement of the GJK distance query with the signed-volumes sub-algorithm
// (Montanari, Petrinic, Barbieri, ACM TOG 36(3), 2017) *as patched and used by the
// reference*: /root/reference/lib/opengjk/src/openGJK.c.  The reference build is the
// "Fast" variant (no exact predicates, lib/opengjk/CMakeLists.txt:37-46) and its gjk()
// returns the witness vector v instead of the distance (openGJK.c:844-851).
//
// Because the result is only eps_rel=1e-5 accurate, *path* fidelity matters: every branch
// decision and every floating-point expression below keeps the association order of the
// reference so that, compiled without FMA contraction, the witness vector is bit-identical.
// Deliberately preserved quirks (file:line in openGJK.c):
//   * support(): keeps the previous support point unless a strictly larger dot is found,
//     first maximum wins (:721-736); first seed is vertex 0 of each body (:778-780)
//   * S2D degenerate branch compares two 1-simplices but, when the auxiliary one wins,
//     only lambdas/labels are taken over, NOT the vertices (:314-321)
//   * S2D axis choice leaves J = {-1, 0} when |nu0| == |nu1| >= |nu2| (:203,:229-248);
//     the reference then reads a[-1] (undefined); we read 0.0 there.  In the only case where
//     this is reachable in practice (collinear triangle, all nu == 0) the normal is NaN and
//     the value is never used.
//   * loop ends when the simplex has 4 vertices or after 50 iterations (:841)
// Pinned against the real reference by tests/test_oracle_vs_ref.py (ref_gjk KATs) and by
// the committed golden vectors tests/golden/gjk_kat.npz.
#include "orc.h"
#include <cmath>

namespace orc {

namespace {
struct Simplex {
  int n;
  double v[4][3];
  int wid[4];
  double lam[4];
};

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double sq3(const double* a) {
  double n2 = 0;
  for (int i = 0; i < 3; ++i) n2 += a[i] * a[i];
  return n2;
}
inline bool same_sign(double a, double b) { return (a > 0) == (b > 0); }

inline void combine(const Simplex& s, double* out) {
  for (int j = 0; j < 3; ++j) {
    out[j] = 0;
    for (int i = 0; i < s.n; ++i) out[j] += s.lam[i] * s.v[i][j];
  }
}

// closest point of a segment to the origin (openGJK.c:82-163)
void sub1(Simplex& s, double* out) {
  double a[3], b[3], t[3], f[3];
  for (int i = 0; i < 3; ++i) {
    b[i] = s.v[0][i];
    a[i] = s.v[1][i];
    t[i] = b[i] - a[i];
    f[i] = std::fabs(t[i]);
  }
  int I = 1;
  if (f[0] > f[1]) I = (f[0] > f[2]) ? 0 : 2;
  else if (f[0] < f[1]) I = (f[1] > f[2]) ? 1 : 2;
  else if (f[0] < f[2]) I = 2;
  else if (f[1] < f[2]) I = 2;

  double pt = dot3(b, t) / dot3(t, t) * (a[I] - b[I]) + b[I];
  double det_ap = a[I] - pt;
  double det_pb = pt - b[I];
  int f0 = same_sign(t[I], -1 * det_ap);
  int f1 = same_sign(t[I], -1 * det_pb);
  if (f0 + f1 == 2) {
    s.lam[0] = det_ap * -1.0 / t[I];
    s.lam[1] = 1 - s.lam[0];
    s.wid[0] = 0; s.wid[1] = 1; s.n = 2;
  } else if (f0 == 0) {
    s.lam[0] = 1; s.wid[0] = 0; s.n = 1;
    for (int i = 0; i < 3; ++i) s.v[0][i] = s.v[1][i];
  } else {
    s.lam[0] = 1; s.wid[0] = 1; s.n = 1;
  }
  combine(s, out);
}

// closest point of a triangle to the origin (openGJK.c:168-393)
void sub2(Simplex& s, double* out) {
  double a[3], b[3], c[3], s21[3], s31[3], nu[3], f[3], n[3], B[3], tmpv[3], v[3];
  for (int i = 0; i < 3; ++i) {
    c[i] = s.v[0][i]; b[i] = s.v[1][i]; a[i] = s.v[2][i];
    s21[i] = b[i] - a[i]; s31[i] = c[i] - a[i];
  }
  int k = 1, l = 2;
  for (int i = 0; i < 3; ++i) {
    double sg = (i == 1) ? -1.0 : 1.0;  // pow(-1.0, i)
    nu[i] = sg * (b[k] * c[l] + a[k] * b[l] + c[k] * a[l] - b[k] * a[l] - c[k] * b[l] - a[k] * c[l]);
    k = l; l = i;
  }
  for (int i = 0; i < 3; ++i) f[i] = std::fabs(nu[i]);
  int I = 1, J0 = -1, J1 = 0;
  if (f[0] > f[1]) {
    if (f[0] > f[2]) { I = 0; J0 = 1; J1 = 2; } else { J0 = 0; J1 = 1; I = 2; }
  } else if (f[0] < f[1]) {
    if (f[1] > f[2]) { J0 = 0; I = 1; J1 = 2; } else { J0 = 0; J1 = 1; I = 2; }
  } else if (f[0] < f[2]) { J0 = 0; J1 = 1; I = 2; }
  double nu_max = nu[I];

  double nn = 0;
  k = 1; l = 2;
  for (int i = 0; i < 3; ++i) {
    n[i] = s21[k] * s31[l] - s21[l] * s31[k];
    nn += n[i] * n[i];
    k = l; l = i;
  }
  double inv_len = 1 / std::sqrt(nn);
  for (int i = 0; i < 3; ++i) n[i] = n[i] * inv_len;
  double dna = dot3(n, a);
  auto at = [](const double* p, int j) { return j < 0 ? 0.0 : p[j]; };  // see header: a[-1]
  double pp0 = dna * at(n, J0), pp1 = dna * at(n, J1);
  double ss[3][2] = {{at(a, J0), at(a, J1)}, {at(b, J0), at(b, J1)}, {at(c, J0), at(c, J1)}};
  k = 1; l = 2;
  for (int i = 0; i < 3; ++i) {
    B[i] = pp0 * ss[k][1] + pp1 * ss[l][0] + ss[k][0] * ss[l][1] - pp0 * ss[l][1] - pp1 * ss[k][0] - ss[l][0] * ss[k][1];
    k = l; l = i;
  }
  int F[3];
  for (int i = 0; i < 3; ++i) F[i] = same_sign(nu_max, B[i]);

  if (F[1] + F[2] == 0 || std::isnan(n[0])) {
    Simplex aux;
    aux.n = 2; s.n = 2;
    for (int i = 0; i < 3; ++i) {
      aux.v[0][i] = s.v[1][i];
      aux.v[1][i] = s.v[2][i];
      s.v[1][i] = s.v[2][i];
    }
    sub1(aux, v);
    sub1(s, v);
    combine(aux, tmpv);
    combine(s, v);
    if (dot3(v, v) < dot3(tmpv, tmpv)) {
      for (int i = 1; i < s.n; ++i) s.wid[i] = s.wid[i] + 1;
    } else {
      s.n = aux.n;  // vertices intentionally not taken over (reference quirk)
      for (int i = 0; i < s.n; ++i) { s.lam[i] = aux.lam[i]; s.wid[i] = aux.wid[i]; }
    }
  } else if (F[0] + F[1] + F[2] == 3) {
    double inv = 1 / nu_max;
    s.lam[0] = B[2] * inv;
    s.lam[1] = B[1] * inv;
    s.lam[2] = 1 - s.lam[0] - s.lam[1];
    s.wid[0] = 0; s.wid[1] = 1; s.wid[2] = 2; s.n = 3;
  } else if (F[2] == 0) {  // faces segment AB
    s.n = 2;
    for (int i = 0; i < 3; ++i) { s.v[0][i] = s.v[1][i]; s.v[1][i] = s.v[2][i]; }
    sub1(s, v);
  } else if (F[1] == 0) {  // faces segment AC
    s.n = 2;
    for (int i = 0; i < 3; ++i) s.v[1][i] = s.v[2][i];
    sub1(s, v);
    for (int i = 1; i < s.n; ++i) s.wid[i] = s.wid[i] + 1;
  } else {  // faces segment BC
    s.n = 2;
    sub1(s, v);
  }
  combine(s, out);
}

// closest point of a tetrahedron to the origin (openGJK.c:398-711)
void sub3(Simplex& s, double* out) {
  static const int TRI[9] = {3, 3, 3, 1, 2, 2, 0, 0, 1};
  int F[4] = {1, 1, 1, 1};
  double a[3], b[3], c[3], d[3], B[4], v[3], tmpv[3];
  for (int i = 0; i < 3; ++i) { d[i] = s.v[0][i]; c[i] = s.v[1][i]; b[i] = s.v[2][i]; a[i] = s.v[3][i]; }
  B[0] = -1 * (b[0] * c[1] * d[2] + b[1] * c[2] * d[0] + b[2] * c[0] * d[1] - b[2] * c[1] * d[0] - b[1] * c[0] * d[2] - b[0] * c[2] * d[1]);
  B[1] = +1 * (a[0] * c[1] * d[2] + a[1] * c[2] * d[0] + a[2] * c[0] * d[1] - a[2] * c[1] * d[0] - a[1] * c[0] * d[2] - a[0] * c[2] * d[1]);
  B[2] = -1 * (a[0] * b[1] * d[2] + a[1] * b[2] * d[0] + a[2] * b[0] * d[1] - a[2] * b[1] * d[0] - a[1] * b[0] * d[2] - a[0] * b[2] * d[1]);
  B[3] = +1 * (a[0] * b[1] * c[2] + a[1] * b[2] * c[0] + a[2] * b[0] * c[1] - a[2] * b[1] * c[0] - a[1] * b[0] * c[2] - a[0] * b[2] * c[1]);
  double detM = B[0] + B[1] + B[2] + B[3];
  const double eps = 1e-13;
  if (std::fabs(detM) < eps) {
    if (std::fabs(B[2]) < eps && std::fabs(B[3]) < eps) F[1] = 0;
    else if (std::fabs(B[1]) < eps && std::fabs(B[3]) < eps) F[2] = 0;
    else if (std::fabs(B[1]) < eps && std::fabs(B[2]) < eps) F[3] = 0;
    else if (std::fabs(B[0]) < eps && std::fabs(B[3]) < eps) F[1] = 0;
    else if (std::fabs(B[0]) < eps && std::fabs(B[2]) < eps) F[1] = 0;
    else if (std::fabs(B[0]) < eps && std::fabs(B[1]) < eps) F[2] = 0;
    else for (int i = 0; i < 4; i++) F[i] = 0;
  } else {
    for (int i = 0; i < 4; ++i) F[i] = same_sign(detM, B[i]);
  }

  const int facing = F[1] + F[2] + F[3];
  if (F[0] + facing == 4) {  // origin inside
    double inv = 1 / detM;
    s.lam[3] = B[0] * inv;
    s.lam[2] = B[1] * inv;
    s.lam[1] = B[2] * inv;
    s.lam[0] = 1 - s.lam[1] - s.lam[2] - s.lam[3];
    for (int i = 0; i < 4; ++i) s.wid[i] = i;
    s.n = 4;
  } else if (facing == 0) {  // three candidate faces: pick the closest
    Simplex aux;
    int ids[4] = {0, 0, 0, 0}, nbest = 0;
    double lam_best[4] = {0, 0, 0, 0}, best = 0;
    for (int i = 0; i < 3; ++i) {
      aux.n = 3;
      for (int kk = 0; kk < 3; ++kk) {
        int vid = TRI[i + kk * 3];
        for (int j = 0; j < 3; ++j) aux.v[2 - kk][j] = s.v[vid][j];
      }
      sub2(aux, v);
      combine(aux, tmpv);
      double dd = dot3(tmpv, tmpv);
      if (i == 0 || dd < best) {
        best = dd;
        nbest = aux.n;
        for (int q = 0; q < nbest; ++q) { ids[q] = TRI[i + aux.wid[q] * 3]; lam_best[q] = aux.lam[q]; }
      }
    }
    double keep[4][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) keep[i][j] = s.v[i][j];
    s.n = nbest;
    for (int i = 0; i < s.n; ++i) {
      for (int j = 0; j < 3; ++j) s.v[nbest - 1 - i][j] = keep[ids[i]][j];
      s.lam[i] = lam_best[i];
      s.wid[nbest - 1 - i] = ids[i];
    }
  } else if (facing == 1) {  // two candidate faces
    Simplex aux;
    aux.n = 3;
    double best = 0;
    int used = 0, first = 0, second = 0;
    if (F[1] == 0) {  // ACD
      for (int i = 0; i < 3; ++i) { aux.v[0][i] = s.v[0][i]; aux.v[1][i] = s.v[1][i]; aux.v[2][i] = s.v[3][i]; }
      sub2(aux, v);
      combine(aux, tmpv);
      best = dot3(tmpv, tmpv);
      used = 1; first = 0;
    }
    if (F[2] == 0) {  // ABD
      if (!used) {
        for (int i = 0; i < 3; ++i) { aux.v[0][i] = s.v[0][i]; aux.v[1][i] = s.v[2][i]; aux.v[2][i] = s.v[3][i]; }
        sub2(aux, v);
        combine(aux, tmpv);
        best = dot3(tmpv, tmpv);
        first = 1;
      } else {
        s.n = 3;
        for (int i = 0; i < 3; ++i) { s.v[1][i] = s.v[2][i]; s.v[2][i] = s.v[3][i]; }
        sub2(s, v);
        second = 1;
      }
    }
    if (F[3] == 0) {  // ABC
      s.n = 3;
      for (int i = 0; i < 3; ++i) { s.v[0][i] = s.v[1][i]; s.v[1][i] = s.v[2][i]; s.v[2][i] = s.v[3][i]; }
      sub2(s, v);
      second = 2;
    }
    combine(s, v);
    if (dot3(v, v) < best) {
      // labels are rewritten in place while being read, exactly as the reference does
      for (int i = 0; i < s.n; ++i) s.wid[s.n - 1 - i] = TRI[second + s.wid[i] * 3];
    } else {
      s.n = aux.n;
      for (int i = 0; i < s.n; ++i) {
        for (int j = 0; j < 3; ++j) s.v[i][j] = aux.v[i][j];
        s.lam[i] = aux.lam[i];
        s.wid[aux.n - 1 - i] = TRI[first + aux.wid[i] * 3];
      }
    }
  } else if (facing == 2) {  // one candidate face
    if (F[1] == 0) {  // ACD
      s.n = 3;
      for (int i = 0; i < 3; ++i) s.v[2][i] = s.v[3][i];
      sub2(s, v);
    } else if (F[2] == 0) {  // ABD
      s.n = 3;
      for (int i = 0; i < 3; ++i) { s.v[1][i] = s.v[2][i]; s.v[2][i] = s.v[3][i]; }
      sub2(s, v);
      for (int i = 2; i < s.n; ++i) s.wid[i] = s.wid[i] + 1;
    } else if (F[3] == 0) {  // ABC
      s.n = 3;
      for (int i = 0; i < 3; ++i) { s.v[0][i] = s.v[1][i]; s.v[1][i] = s.v[2][i]; s.v[2][i] = s.v[3][i]; }
      sub2(s, v);
    }
  } else {  // BCD
    s.n = 3;
    sub2(s, v);
    for (int i = 0; i < s.n; ++i) s.wid[i] = s.wid[i] + 1;
  }
  combine(s, out);
}

// support mapping with the reference's "sticky" tie-break (openGJK.c:714-737)
inline void support(const double* pts, int n, const double* dir, double* cur) {
  int better = -1;
  double best = dot3(cur, dir);
  for (int i = 0; i < n; ++i) {
    double sdot = dot3(pts + 3 * i, dir);
    if (sdot > best) { best = sdot; better = i; }
  }
  if (better != -1) { cur[0] = pts[3 * better]; cur[1] = pts[3 * better + 1]; cur[2] = pts[3 * better + 2]; }
}
}  // namespace

// Witness vector of conv(p1) - conv(p2), points row-major n x 3 (openGJK.c:754-852).
void gjk(const double* p1, int n1, const double* p2, int n2, double* v_out, int* iters) {
  const int max_it = 50;
  const double eps_rel = 1e-5, eps_tot = 1e-15;
  const double eps_rel2 = eps_rel * eps_rel;
  Simplex s;
  double v[3], vm[3], w[3], s1[3], s2[3];
  double wmax2 = 0;
  int k = 0;
  s.n = 1;
  for (int i = 0; i < 3; ++i) {
    v[i] = p1[i] - p2[i];
    s1[i] = p1[i]; s2[i] = p2[i];
    s.v[0][i] = v[i];
  }
  do {
    k++;
    vm[0] = -v[0]; vm[1] = -v[1]; vm[2] = -v[2];
    support(p1, n1, vm, s1);
    support(p2, n2, v, s2);
    w[0] = s1[0] - s2[0]; w[1] = s1[1] - s2[1]; w[2] = s1[2] - s2[2];
    if ((sq3(v) - dot3(v, w)) <= eps_rel2 * sq3(v)) break;
    if (sq3(v) < eps_rel2) break;
    int i = s.n;
    s.v[i][0] = w[0]; s.v[i][1] = w[1]; s.v[i][2] = w[2];
    s.n++;
    switch (s.n) {
      case 4: sub3(s, v); break;
      case 3: sub2(s, v); break;
      case 2: sub1(s, v); break;
    }
    for (i = 0; i < s.n; i++) {
      double t = sq3(s.v[i]);
      if (t > wmax2) wmax2 = t;
    }
    if (sq3(v) <= (eps_tot * eps_tot * wmax2)) break;
  } while ((s.n != 4) && (k != max_it));
  v_out[0] = v[0]; v_out[1] = v[1]; v_out[2] = v[2];
  if (iters) *iters = k;
}

}  // namespace orc
