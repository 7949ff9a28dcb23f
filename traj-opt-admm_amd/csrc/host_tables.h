// host_tables.h -- host-side precompute of the product library: constant tables of the quintic
// Bezier formulation and the static BVH over the obstacle cloud.  Runs once per problem, never on
// the per-iteration path.
//
// Tables replace the reference's table globals (CCDUtils.cpp:5-44) as filled by
//   Combination<40>::value                      CCDUtils.h:110-135
//   Conversion<5>::convert_matrix               CCDUtils.h:137-170   (time_weight == 1 everywhere)
//   Dynamic3D<5,3>::dynamic_matrix              CCDUtils.h:172-227
//   Blossom<5>::coefficient + init_variable     CCDUtils.h:229-315, Main/admmPathPlanning3D.cpp:300-314
// The Kronecker selector lists A_list / A_vel_list / A_acc_list (admmPathPlanning3D.cpp:316-345)
// are not built at all: the kernels apply basis rows directly.
// The BVH replaces BVH::InitPointcloud (BVH.cpp:53-93, an incrementally balanced dynamic tree,
// 95 ms for 20k points on the CPU) by a Morton sort + implicit 8-ary box pyramid.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace tj {

struct HostTables {
  std::vector<double> basis, convert;  // [S][36], [P][36] row-major
  double mdyn[36];
  double kdop[49 * 3];
  std::vector<double> pow08;
};

inline void build_tables(int P, int res, int loop_cap, HostTables& t) {
  const int n = 5, der = 3;
  // Pascal triangle up to 10 is all this order needs
  long binom[11][11] = {{0}};
  for (int i = 0; i <= 10; i++) { binom[i][0] = 1; for (int j = 1; j <= i; j++) binom[i][j] = binom[i - 1][j - 1] + (j <= i - 1 ? binom[i - 1][j] : 0); }

  // C^2 junction maps between neighbouring pieces with equal durations: the last two rows of piece i
  // and the first two rows of piece i+1 are tied to the three shared control points.
  t.convert.assign((size_t)P * 36, 0.0);
  for (int i = 0; i < P; i++) for (int d = 0; d < 6; d++) t.convert[i * 36 + d * 7] = 1.0;
  for (int i = 0; i + 1 < P; i++) {
    double* a = &t.convert[i * 36];
    double* b = &t.convert[(i + 1) * 36];
    a[4 * 6 + 3] = 0.5; a[4 * 6 + 4] = 0.5; a[4 * 6 + 5] = 0.0;
    a[5 * 6 + 3] = 0.25; a[5 * 6 + 4] = 0.5; a[5 * 6 + 5] = 0.25;
    b[0 * 6 + 0] = 0.25; b[0 * 6 + 1] = 0.5; b[0 * 6 + 2] = 0.25;
    b[1 * 6 + 0] = 0.0; b[1 * 6 + 1] = 0.5; b[1 * 6 + 2] = 0.5;
  }

  // Gram matrix of the third derivative of the degree-5 Bernstein basis on [0,1]; the order of
  // the floating-point operations follows CCDUtils.h:180-218 so the entries are bit-identical.
  for (int i = 0; i <= n; i++)
    for (int j = 0; j <= n; j++) {
      double sum = 0;
      for (int k0 = 0; k0 <= der; k0++)
        for (int k1 = 0; k1 <= der; k1++) {
          const int r = i - k0, c = j - k1;
          if (r < 0 || c < 0 || r > n - der || c > n - der) continue;
          double term = ((k0 + k1) & 1) ? -1.0 : 1.0;
          term *= (binom[der][k0] * binom[der][k1] * binom[n - der][r] * binom[n - der][c]) / (double)binom[2 * (n - der)][r + c];
          for (int s = 0; s < der; s++) term *= (n - s) * (n - s);
          term /= (double)(2 * (n - der) + 1);
          sum += term;
        }
      t.mdyn[i * 6 + j] = sum;
    }
  for (int d = 0; d < 6; d++) t.mdyn[d * 7] += 1e-8;

  // Subdivision of a quintic Bezier curve to [t0,t1] by blossoming: row r is the blossom with
  // n-r arguments t0 and r arguments t1.  All entries are dyadic rationals for res = 2^k, hence
  // exact in any evaluation order.
  const int S = P * res;
  t.basis.assign((size_t)S * 36, 0.0);
  for (int k = 0; k < res; k++) {
    const double t0 = k / double(res), t1 = (k + 1) / double(res);
    double p0[6], p1[6], q0[6], q1[6];
    p0[0] = p1[0] = q0[0] = q1[0] = 1.0;
    for (int e = 1; e <= n; e++) { p0[e] = p0[e - 1] * t0; p1[e] = p1[e - 1] * t1; q0[e] = q0[e - 1] * (1 - t0); q1[e] = q1[e - 1] * (1 - t1); }
    double sub[36];
    for (int r = 0; r <= n; r++)
      for (int c = 0; c <= n; c++) {
        // choose a of the n-r "t0" slots and b of the r "t1" slots to be the variable, a+b = c
        double acc = 0;
        if (r + c < n) {
          for (int b = 0; b <= std::min(r, c); b++)
            acc += binom[n - r][c - b] * binom[r][b] * q0[n - r - c + b] * q1[r - b] * p0[c - b] * p1[b];
        } else {
          for (int a = 0; a <= std::min(n - r, n - c); a++)
            acc += binom[n - r][a] * binom[r][n - c - a] * q0[a] * q1[n - c - a] * p0[n - r - a] * p1[r + c - n + a];
        }
        sub[r * 6 + c] = acc;
      }
    for (int i = 0; i < P; i++) {
      double* out = &t.basis[(size_t)(i * res + k) * 36];
      const double* cv = &t.convert[i * 36];
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
          double acc = 0;
          for (int m = 0; m < 6; m++) acc += sub[r * 6 + m] * cv[m * 6 + c];
          out[r * 6 + c] = acc;
        }
    }
  }

  // 49 k-DOP directions: axes, cube diagonals, face diagonals and the {1,2}-mixed families of
  // CCDUtils.cpp:56-119 (same order), normalised like admmPathPlanning3D.cpp:403-410
  static const signed char raw[49][3] = {
      {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {1, -1, -1}, {0, 1, 1}, {0, 1, -1}, {1, 0, 1},
      {1, 0, -1}, {1, 1, 0}, {1, -1, 0}, {0, 2, 1}, {0, 2, -1}, {0, 1, 2}, {0, 1, -2}, {2, 0, 1}, {2, 0, -1}, {1, 0, 2},
      {1, 0, -2}, {2, 1, 0}, {2, -1, 0}, {1, 2, 0}, {1, -2, 0}, {1, 2, 1}, {1, 2, -1}, {1, -2, 1}, {-1, 2, 1}, {1, 1, 2},
      {1, 1, -2}, {1, -1, 2}, {-1, 1, 2}, {2, 1, 1}, {2, 1, -1}, {2, -1, 1}, {-2, 1, 1}, {2, 2, 1}, {2, 2, -1}, {2, -2, 1},
      {-2, 2, 1}, {2, 1, 2}, {2, 1, -2}, {2, -1, 2}, {-2, 1, 2}, {1, 2, 2}, {1, 2, -2}, {1, -2, 2}, {-1, 2, 2}};
  for (int k = 0; k < 49; k++) {
    const double x = raw[k][0], y = raw[k][1], z = raw[k][2];
    const double len = std::sqrt(x * x + y * y + z * z);
    t.kdop[3 * k] = x / len; t.kdop[3 * k + 1] = y / len; t.kdop[3 * k + 2] = z / len;
  }
  t.pow08.assign(loop_cap + 1, 1.0);
  for (int k = 1; k <= loop_cap; k++) t.pow08[k] = t.pow08[k - 1] * 0.8;  // step *= 0.8 (Step.h:93)
}

// ---- static BVH ----------------------------------------------------------------------------
// Obstacle primitives (prim = 1: points, prim = 3: triangles given as 3 vertices each) sorted along a 63-bit Morton curve
// of their centroids, under an implicit 8-ary pyramid of boxes.  Boxes are stored in fp32 rounded OUTWARD.
struct HostBvh {
  int prim = 1;
  std::vector<double> px, py, pz;   // Morton order (prim == 1)
  std::vector<double> tri;          // [n][9] Morton order (prim == 3)
  std::vector<float> leafbox;       // [n][6] outward-rounded box of each triangle (prim == 3)
  std::vector<int> order;           // sorted position -> original index
  std::vector<float> boxes;         // all levels, [node][6]
  std::vector<int> lvl_off, lvl_n;
};

inline uint64_t spread21(uint64_t v) {
  v &= 0x1fffff;
  v = (v | v << 32) & 0x1f00000000ffffULL;
  v = (v | v << 16) & 0x1f0000ff0000ffULL;
  v = (v | v << 8) & 0x100f00f00f00f00fULL;
  v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}
// largest float <= x / smallest float >= x
inline float f32_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -INFINITY); return f; }
inline float f32_up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, INFINITY); return f; }

// 63-bit Morton key of a centroid inside [lo, hi]  (shared with the device build, kernels_bvh.h: same expression, same bits)
inline uint64_t morton_key(const double* c, const double* lo, const double* hi) {
  uint64_t code = 0;
  for (int k = 0; k < 3; k++) {
    const double ext = hi[k] - lo[k];
    const double f = ext > 0 ? (c[k] - lo[k]) / ext : 0.0;
    const uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, f * 2097152.0));
    code |= spread21(q) << k;
  }
  return code;
}
inline void prim_centroid(const double* v, int prim, double* c) {
  for (int k = 0; k < 3; k++) c[k] = prim == 1 ? v[k] : (v[k] + v[3 + k] + v[6 + k]) / 3.0;
}

// verts: [n][prim][3]
inline void build_bvh(const double* verts, int n, int prim, HostBvh& b) {
  b = HostBvh();
  b.prim = prim;
  if (n <= 0) return;
  const size_t st = 3 * (size_t)prim;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  std::vector<double> cen((size_t)n * 3);
  for (int i = 0; i < n; i++) {
    prim_centroid(verts + st * i, prim, &cen[3 * (size_t)i]);
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], cen[3 * (size_t)i + k]); hi[k] = std::max(hi[k], cen[3 * (size_t)i + k]); }
  }
  std::vector<std::pair<uint64_t, int>> key(n);
  for (int i = 0; i < n; i++) key[i] = {morton_key(&cen[3 * (size_t)i], lo, hi), i};
  std::sort(key.begin(), key.end());   // ties broken by the original index: the order is a function of the input alone
  b.order.resize(n);
  if (prim == 1) { b.px.resize(n); b.py.resize(n); b.pz.resize(n); } else { b.tri.resize((size_t)n * 9); b.leafbox.resize((size_t)n * 6); }
  std::vector<double> plo((size_t)n * 3), phi((size_t)n * 3);   // exact fp64 box of every primitive, sorted order
  for (int i = 0; i < n; i++) {
    const int o = key[i].second; b.order[i] = o;
    const double* v = verts + st * o;
    for (int k = 0; k < 3; k++) {
      double l = INFINITY, h = -INFINITY;
      for (int j = 0; j < prim; j++) { l = std::min(l, v[3 * j + k]); h = std::max(h, v[3 * j + k]); }
      plo[3 * (size_t)i + k] = l; phi[3 * (size_t)i + k] = h;
    }
    if (prim == 1) { b.px[i] = v[0]; b.py[i] = v[1]; b.pz[i] = v[2]; }
    else {
      for (int k = 0; k < 9; k++) b.tri[(size_t)i * 9 + k] = v[k];
      for (int k = 0; k < 3; k++) { b.leafbox[(size_t)i * 6 + k] = f32_down(plo[3 * (size_t)i + k]); b.leafbox[(size_t)i * 6 + 3 + k] = f32_up(phi[3 * (size_t)i + k]); }
    }
  }
  // level 0: boxes over 8 consecutive primitives; higher levels: boxes over 8 consecutive boxes.  Unions are taken in fp64
  // and rounded outward once per box.
  std::vector<double> cl, ch;   // current level, fp64
  int cnt = (n + 7) / 8;
  cl.assign((size_t)cnt * 3, INFINITY); ch.assign((size_t)cnt * 3, -INFINITY);
  for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { cl[3 * (size_t)(i / 8) + k] = std::min(cl[3 * (size_t)(i / 8) + k], plo[3 * (size_t)i + k]); ch[3 * (size_t)(i / 8) + k] = std::max(ch[3 * (size_t)(i / 8) + k], phi[3 * (size_t)i + k]); }
  for (;;) {
    const int off = (int)(b.boxes.size() / 6);
    b.lvl_off.push_back(off); b.lvl_n.push_back(cnt);
    b.boxes.resize((size_t)(off + cnt) * 6);
    for (int g = 0; g < cnt; g++) for (int k = 0; k < 3; k++) { b.boxes[(size_t)(off + g) * 6 + k] = f32_down(cl[3 * (size_t)g + k]); b.boxes[(size_t)(off + g) * 6 + 3 + k] = f32_up(ch[3 * (size_t)g + k]); }
    if (cnt <= 64) break;
    const int nn = (cnt + 7) / 8;
    std::vector<double> nl((size_t)nn * 3, INFINITY), nh((size_t)nn * 3, -INFINITY);
    for (int i = 0; i < cnt; i++) for (int k = 0; k < 3; k++) { nl[3 * (size_t)(i / 8) + k] = std::min(nl[3 * (size_t)(i / 8) + k], cl[3 * (size_t)i + k]); nh[3 * (size_t)(i / 8) + k] = std::max(nh[3 * (size_t)(i / 8) + k], ch[3 * (size_t)i + k]); }
    cl.swap(nl); ch.swap(nh); cnt = nn;
  }
}

}  // namespace tj
