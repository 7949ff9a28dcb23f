// dev_optplane.h -- "optimal_plane":1 : the per-plane Newton refinements of the reference, one plane per lane.
//
//   opt_plane_obstacle   Optimal_plane::optimal_cd       (Optimal_plane.h:160-293; barrier_energy :93-116,
//                        barrier_grad :118-158, current_c / current_d :74-91)  hull vs one cloud point, 2 unknowns
//   opt_plane_pair       Optimal_plane::self_optimal_cd  (Optimal_plane.h:620-773; self_barrier_energy :518-552,
//                        self_barrier_grad :554-618)                           hull vs hull, 3 unknowns
//
// Both build a tangent frame (c0, c1) of the normal c, take a Newton step in the frame angles (theta, phi) [and the
// offset d], clamp it to a quarter turn and back off by 0.8 until the barrier energy decreases.  The Hessians the
// reference writes down have a structurally zero (phi, phi) entry, so Eigen's LLT always fails on the 3x3 and the
// matrix is repaired to lambda_min = 1e-8 from SelfAdjointEigenSolver's eigenvalue: that eigenvalue is reproduced with
// Eigen's own arithmetic (scaling, the closed-form 3x3 tridiagonalisation, implicit symmetric QR with Wilkinson shift,
// Golub & Van Loan alg. 8.3.2 in the operation order of Eigen 3.3.7), because a 1-ulp change of it moves the repaired
// eigenvalue by 1e-8 relative.  Summation orders follow Eigen: reductions over a row of a dynamic matrix are sequential,
// fixed-size Vector3d reductions pair (e0 + e1) + e2, `c.dot(row)` pairs e0 + (e1 + e2).
// The loops are unbounded in the reference (self_optimal_cd zig-zags for thousands of rounds on some pairs); here they
// stop after PLANE_NEWTON_CAP rounds / PLANE_BACKOFF_CAP back-offs and report it.
#pragma once
#include "dev_common.h"
#include "dev_crmath.h"   // log / sin / cos that round like glibc's: one ulp here is 1e-8 after the reference's eigenvalue repair

namespace tj {

constexpr int PLANE_NEWTON_CAP = 200000;
// back-offs of one Newton round: the offset component of the step is not clamped, a direction of 1e8 (the repaired eigenvalue
// is 1e-8) needs ~250 factors of 0.8 before the trial point equals the start in fp64 and the reference's loop ends
constexpr int PLANE_BACKOFF_CAP = 2000;
constexpr double TJ_PI = 3.14159265358979323846;

__device__ __forceinline__ double op_dot_row(const double* r, double c0, double c1, double c2) { return (r[0] * c0 + r[1] * c1) + r[2] * c2; }
__device__ __forceinline__ double op_sum3v(double e0, double e1, double e2) { return (e0 + e1) + e2; }
__device__ __forceinline__ void op_normalize(double& x, double& y, double& z) {  // MatrixBase::normalize
  const double s = op_sum3v(x * x, y * y, z * z);
  if (s > 0) { const double n = sqrt(s); x /= n; y /= n; z /= n; }
}

// ---- Eigen::SelfAdjointEigenSolver<Matrix{2,3}d>::eigenvalues()(0) -------------------------------------------------
__device__ __forceinline__ double op_hypot(double x, double y) {  // numext::hypot (MathFunctionsImpl.h:74-85)
  x = fabs(x); y = fabs(y);
  const double p = x < y ? y : x;
  if (p == 0) return 0;
  const double qp = (x < y ? x : y) / p;
  return p * sqrt(1.0 + qp * qp);
}
__device__ __forceinline__ void op_givens(double p, double q, double& c, double& s) {  // JacobiRotation::makeGivens (Jacobi.h)
  if (q == 0) { c = p < 0 ? -1.0 : 1.0; s = 0; }
  else if (p == 0) { c = 0; s = q < 0 ? 1.0 : -1.0; }
  else if (fabs(p) > fabs(q)) { const double t = q / p; double u = sqrt(1.0 + t * t); if (p < 0) u = -u; c = 1.0 / u; s = -t * c; }
  else { const double t = p / q; double u = sqrt(1.0 + t * t); if (q < 0) u = -u; s = -1.0 / u; c = -t * s; }
}
// one implicit QR step on the unreduced block [start, end] of a tridiagonal matrix with n <= 3 (tridiagonal_qr_step);
// d0..d2 / s0,s1 are kept in scalars so that nothing is indexed dynamically
__device__ __forceinline__ void op_qr_step(double& d0, double& d1, double& d2, double& s0, double& s1, int start, int end) {
  // block is {0,1}, {1,2} or {0,1,2}
  const double dl = end == 1 ? d0 : d1, de = end == 1 ? d1 : d2, e = end == 1 ? s0 : s1;
  const double td = (dl - de) * 0.5;
  double mu = de;
  if (td == 0) mu -= fabs(e);
  else {
    const double e2 = e * e, h = op_hypot(td, e);
    if (e2 == 0) mu -= (e / (td + (td > 0 ? 1.0 : -1.0))) * (e / h);
    else mu -= e2 / (td + (td > 0 ? h : -h));
  }
  double x = (start == 0 ? d0 : d1) - mu, z = start == 0 ? s0 : s1;
  for (int k = start; k < end; k++) {
    double c, s; op_givens(x, z, c, s);
    double& dk = k == 0 ? d0 : d1; double& dk1 = k == 0 ? d1 : d2; double& sk = k == 0 ? s0 : s1;
    const double sdk = s * dk + c * sk, dkp1 = s * sk + c * dk1;
    dk = c * (c * dk - s * sk) - s * (c * sk - s * dk1);
    dk1 = s * sdk + c * dkp1;
    sk = c * sdk - s * dkp1;
    if (k > start) s0 = c * s0 - s * z;   // only k == 1, start == 0
    x = sk;
    if (k < end - 1) { z = -s * s1; s1 = c * s1; }  // only k == 0, end == 2
  }
}
// lower triangle a00, a10, a11 [, a20, a21, a22]
template <int N>
__device__ inline double op_min_eig(double a00, double a10, double a11, double a20, double a21, double a22) {
  double scale = fmax(fabs(a00), fmax(fabs(a10), fabs(a11)));
  if (N == 3) scale = fmax(scale, fmax(fabs(a20), fmax(fabs(a21), fabs(a22))));
  if (scale == 0) scale = 1;
  a00 /= scale; a10 /= scale; a11 /= scale;
  double d0 = a00, d1 = a11, d2 = 0, s0 = a10, s1 = 0;
  if (N == 3) {
    a20 /= scale; a21 /= scale; a22 /= scale;
    const double tol = 2.2250738585072014e-308;
    const double v1norm2 = a20 * a20;
    if (v1norm2 <= tol) { d1 = a11; d2 = a22; s0 = a10; s1 = a21; }
    else {
      const double beta = sqrt(a10 * a10 + v1norm2), invBeta = 1.0 / beta;
      const double m01 = a10 * invBeta, m02 = a20 * invBeta;
      const double q = 2.0 * m01 * a21 + m02 * (a22 - a11);
      d1 = a11 + m02 * q; d2 = a22 - m02 * q; s0 = beta; s1 = a21 - m01 * q;
    }
  }
  int end = N - 1, start = 0, iter = 0;
  const double tiny = 2.2250738585072014e-308, prec = 2 * 2.220446049250313e-16;
  while (end > 0) {
    if (start <= 0 && 0 < end) if (fabs(s0) <= (fabs(d0) + fabs(d1)) * prec || fabs(s0) <= tiny) s0 = 0;
    if (N == 3 && start <= 1 && 1 < end) if (fabs(s1) <= (fabs(d1) + fabs(d2)) * prec || fabs(s1) <= tiny) s1 = 0;
    while (end > 0 && (end == 1 ? s0 : s1) == 0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > 30 * N) break;
    start = end - 1;
    while (start > 0 && s0 != 0) start--;   // start can only be 1 here
    op_qr_step(d0, d1, d2, s0, s1, start, end);
  }
  double ev = d0;
  if (iter <= 30 * N) { if (d1 < ev) ev = d1; if (N == 3 && d2 < ev) ev = d2; }
  return ev * scale;
}

// ---- Optimal_plane::optimal_cd --------------------------------------------------------------------------------------
// P: the segment's hull [6][3]; q: the obstacle point; (c, d) in/out.  Returns false when an iteration cap was hit.
__device__ inline bool opt_plane_obstacle(const double* P, double qx, double qy, double qz, double m, double off, double& cx, double& cy, double& cz, double& d) {
  for (int guard = 0; guard < PLANE_NEWTON_CAP; guard++) {
    double c0x = cy, c0y = -cx, c0z = 0.0; op_normalize(c0x, c0y, c0z);
    double c1x = c0y * cz - c0z * cy, c1y = c0z * cx - c0x * cz, c1z = c0x * cy - c0y * cx; op_normalize(c1x, c1y, c1z);
    double g0 = 0, g1 = 0, h00 = 0, h01 = 0, h11 = 0;
    for (int j = 0; j < 6; j++) {
      const double r[3] = {P[3 * j] + -qx, P[3 * j + 1] + -qy, P[3 * j + 2] + -qz};
      const double pc = op_dot_row(r, cx, cy, cz);
      const double dist = pc - off;
      if (dist < m) {
        const double pc0 = op_dot_row(r, c0x, c0y, c0z), pc1 = op_dot_row(r, c1x, c1y, c1z);
        const double lg = cr_log(dist / m);
        const double e1 = -(2 * (dist - m) * lg + (dist - m) * (dist - m) / dist);
        const double e2 = -(2 * lg + 4 * (dist - m) / dist - (dist - m) * (dist - m) / (dist * dist));
        g0 += e1 * pc0; g1 += 0;
        h00 += e2 * pc0 * pc0 - e1 * pc; h01 += e1 * pc1; h11 += 0;
      }
    }
    const double qd = qx, qe = qy, qf = qz;
    auto cur_d = [&](double ax, double ay, double az) { return -(ax * qd + (ay * qe + az * qf)) - off; };  // c.dot(row): fixed-size lhs
    if (sqrt(g0 * g0 + g1 * g1) < 1e-2) { d = cur_d(cx, cy, cz); return true; }
    h00 = h00 + 1e-2 * 1.0; h01 = h01 + 1e-2 * 0.0; h11 = h11 + 1e-2 * 1.0;
    // LLT (in place, lower); a failed factorisation is used as it stands, like Eigen's solve() after NumericalIssue
    double l00 = h00, l10 = h01, l11 = h11;
    auto llt2 = [&]() {
      if (l00 <= 0) return false;
      l00 = sqrt(l00); l10 /= l00;
      const double x = l11 - l10 * l10;
      if (x <= 0) return false;
      l11 = sqrt(x);
      return true;
    };
    if (!llt2()) {
      const double ev = op_min_eig<2>(h00, h01, h11, 0, 0, 0);
      if (ev < 0) { h00 = h00 - ev * 1.0 + 1e-8 * 1.0; h01 = h01 - ev * 0.0 + 1e-8 * 0.0; h11 = h11 - ev * 1.0 + 1e-8 * 1.0; }
      l00 = h00; l10 = h01; l11 = h11;
      llt2();
    }
    double y0 = g0 / l00, y1 = (g1 - l10 * y0) / l11;
    y1 = y1 / l11; y0 = (y0 - l10 * y1) / l00;
    const double dir0 = -y0, dir1 = -y1;
    const double w = -(g0 * dir0 + g1 * dir1);
    double step = 1.0;
    if (fabs(dir0) > 0.5 * TJ_PI || fabs(dir1) > 0.5 * TJ_PI) { const double a = 0.5 * fabs(TJ_PI / dir0), b = 0.5 * fabs(TJ_PI / dir1); step = 0.95 * (b < a ? b : a); }
    double tx, ty, tz;
    auto cur_c = [&](double th, double ph) {
      double ct, st, cp, sp; cr_sincos(th, &st, &ct); cr_sincos(ph, &sp, &cp);
      tx = ct * cx + st * (cp * c0x + sp * c1x); ty = ct * cy + st * (cp * c0y + sp * c1y); tz = ct * cz + st * (cp * c0z + sp * c1z);
    };
    auto energy = [&]() {
      const double dd = cur_d(tx, ty, tz);
      double e = 0;
      for (int j = 0; j < 6; j++) {
        const double dist = op_dot_row(P + 3 * j, tx, ty, tz) + dd;
        if (dist <= 0) return (double)INFINITY;
        if (dist < m) e += -(dist - m) * (dist - m) * cr_log(dist / m);
      }
      return e;
    };
    cur_c(0.0, 0.0);
    const double e0 = energy();
    cur_c(0.0 + step * dir0, 0.0 + step * dir1);
    double e1v = energy();
    int bo = 0;
    while (e0 - 1e-4 * w * step < e1v) {
      if (++bo > PLANE_BACKOFF_CAP) return false;
      step *= 0.8; cur_c(0.0 + step * dir0, 0.0 + step * dir1); e1v = energy();
    }
    cx = tx; cy = ty; cz = tz;
    d = cur_d(cx, cy, cz);
    if (fabs((e1v - e0) / e0) < 1e-1) return true;
  }
  return false;
}

// ---- Optimal_plane::self_optimal_cd ---------------------------------------------------------------------------------
// Shared by the per-lane and the wave-cooperative form (identical arithmetic => identical results):
// tangent frame of c (Optimal_plane.h:635-639)
struct OpFrame { double c0x, c0y, c0z, c1x, c1y, c1z; };
__device__ __forceinline__ OpFrame op_frame(double cx, double cy, double cz) {
  OpFrame f;
  f.c0x = cy; f.c0y = -cx; f.c0z = 0.0; op_normalize(f.c0x, f.c0y, f.c0z);
  f.c1x = f.c0y * cz - f.c0z * cy; f.c1y = f.c0z * cx - f.c0x * cz; f.c1z = f.c0x * cy - f.c0y * cx; op_normalize(f.c1x, f.c1y, f.c1z);
  return f;
}
// one barrier term of self_barrier_grad (:554-618) for point r of the first (second = false) or second body
struct OpTerm { double g0, g2, h00, h10, h20, h22; };
__device__ __forceinline__ OpTerm op_pair_term(const double* r, bool second, const OpFrame& f, double cx, double cy, double cz, double d, double m, double off) {
  OpTerm t{0, 0, 0, 0, 0, 0};
  const double sg = second ? -1.0 : 1.0;
  const double dc = op_dot_row(r, cx, cy, cz);
  const double dist = second ? -dc - d - 0.5 * off : dc + d - 0.5 * off;
  if (dist < m) {
    const double pc = sg * dc, pc0 = sg * op_dot_row(r, f.c0x, f.c0y, f.c0z), pc1 = sg * op_dot_row(r, f.c1x, f.c1y, f.c1z);
    const double lg = cr_log(dist / m);
    const double e1 = -(2 * (dist - m) * lg + (dist - m) * (dist - m) / dist);
    const double e2 = -(2 * lg + 4 * (dist - m) / dist - (dist - m) * (dist - m) / (dist * dist));
    t.g0 = e1 * pc0; t.g2 = sg * e1;
    t.h00 = e2 * pc0 * pc0 - e1 * pc; t.h10 = e1 * pc1; t.h20 = sg * e2 * pc0; t.h22 = e2;
  }
  return t;
}
// one term of self_barrier_energy (:518-552): returns false when the point is on the wrong side (energy = INFINITY)
__device__ __forceinline__ bool op_pair_energy_term(const double* r, bool second, double tx, double ty, double tz, double td, double m, double off, double& e) {
  const double dc = op_dot_row(r, tx, ty, tz);
  const double dist = second ? -dc - td - 0.5 * off : dc + td - 0.5 * off;
  e = 0;
  if (dist <= 0) return false;
  if (dist < m) e = -(dist - m) * (dist - m) * cr_log(dist / m);
  return true;
}
// LLT (always fails on this matrix: h11 == 0), eigenvalue repair, solve, clamp (:652-732): Newton direction and first step
__device__ inline void op_pair_direction(double g0, double g1, double g2, double h00, double h10, double h20, double h11, double h21, double h22,
                                         double& dir0, double& dir1, double& dir2, double& w, double& step) {
  double l00 = h00, l10 = h10, l20 = h20, l11 = h11, l21 = h21, l22 = h22;
  auto llt3 = [&]() {  // Eigen llt_inplace<Lower>::unblocked; on a pivot <= 0 the matrix stays as it is at that point
    if (l00 <= 0) return false;
    l00 = sqrt(l00); l10 /= l00; l20 /= l00;
    double x = l11 - l10 * l10;
    if (x <= 0) return false;
    l11 = x = sqrt(x);
    l21 -= l20 * l10; l21 /= x;
    x = l22 - (l20 * l20 + l21 * l21);
    if (x <= 0) return false;
    l22 = sqrt(x);
    return true;
  };
  if (!llt3()) {
    const double ev = op_min_eig<3>(h00, h10, h11, h20, h21, h22);
    if (ev < 0) {
      h00 = h00 - ev * 1.0 + 1e-8 * 1.0; h11 = h11 - ev * 1.0 + 1e-8 * 1.0; h22 = h22 - ev * 1.0 + 1e-8 * 1.0;
      h10 = h10 - ev * 0.0 + 1e-8 * 0.0; h20 = h20 - ev * 0.0 + 1e-8 * 0.0; h21 = h21 - ev * 0.0 + 1e-8 * 0.0;
    }
    l00 = h00; l10 = h10; l20 = h20; l11 = h11; l21 = h21; l22 = h22;
    llt3();
  }
  double y0 = g0 / l00;
  double y1 = (g1 - l10 * y0) / l11;
  double y2 = (g2 - (l20 * y0 + l21 * y1)) / l22;
  y2 = y2 / l22;
  y1 = (y1 - l21 * y2) / l11;
  y0 = (y0 - (l10 * y1 + l20 * y2)) / l00;
  dir0 = -y0; dir1 = -y1; dir2 = -y2;
  w = -op_sum3v(g0 * dir0, g1 * dir1, g2 * dir2);
  step = 1.0;
  if (fabs(dir0) > 0.5 * TJ_PI || fabs(dir1) > 0.5 * TJ_PI) { const double a = 0.5 * fabs(TJ_PI / dir0), b = 0.5 * fabs(TJ_PI / dir1); step = 0.95 * (b < a ? b : a); }
}
__device__ __forceinline__ void op_rotate(double ct, double st, double cp, double sp, double cx, double cy, double cz, const OpFrame& f, double& tx, double& ty, double& tz) {  // current_c (:74-80)
  tx = ct * cx + st * (cp * f.c0x + sp * f.c1x); ty = ct * cy + st * (cp * f.c0y + sp * f.c1y); tz = ct * cz + st * (cp * f.c0z + sp * f.c1z);
}

// A: hull of the lower robot index, B: hull of the higher; (c, d) in/out (d before the -+offset/2 split).  One plane per lane.
__device__ inline bool opt_plane_pair(const double* A, const double* B, double m, double off, double& cx, double& cy, double& cz, double& d, int* rounds = nullptr) {
  int guard = 0;
  bool ok = false;
  for (; guard < PLANE_NEWTON_CAP; guard++) {
    const OpFrame f = op_frame(cx, cy, cz);
    double g0 = 0, g1 = 0, g2 = 0, h00 = 0, h10 = 0, h20 = 0, h11 = 0, h21 = 0, h22 = 0;
    for (int j = 0; j < 12; j++) {
      const bool second = j >= 6;
      const OpTerm t = op_pair_term(second ? B + 3 * (j - 6) : A + 3 * j, second, f, cx, cy, cz, d, m, off);
      g0 += t.g0; g1 += 0; g2 += t.g2; h00 += t.h00; h10 += t.h10; h20 += t.h20; h11 += 0; h21 += 0; h22 += t.h22;
    }
    if (sqrt(op_sum3v(g0 * g0, g1 * g1, g2 * g2)) < 1e-2) { ok = true; break; }
    double dir0, dir1, dir2, w, step;
    op_pair_direction(g0, g1, g2, h00, h10, h20, h11, h21, h22, dir0, dir1, dir2, w, step);
    double tx, ty, tz, td;
    auto energy = [&]() {
      double e = 0;
      for (int j = 0; j < 12; j++) {
        const bool second = j >= 6;
        double t;
        if (!op_pair_energy_term(second ? B + 3 * (j - 6) : A + 3 * j, second, tx, ty, tz, td, m, off, t)) return (double)INFINITY;
        e += t;
      }
      return e;
    };
    auto rotate = [&](double th, double ph) { double ct, st, cp, sp; cr_sincos(th, &st, &ct); cr_sincos(ph, &sp, &cp); op_rotate(ct, st, cp, sp, cx, cy, cz, f, tx, ty, tz); };
    rotate(0.0, 0.0); td = d;
    const double e0 = energy();
    rotate(0.0 + step * dir0, 0.0 + step * dir1); td = d + step * dir2;
    double e1v = energy();
    int bo = 0;
    bool stuck = false;
    while (e0 - 1e-4 * w * step < e1v) {
      if (++bo > PLANE_BACKOFF_CAP) { stuck = true; break; }
      step *= 0.8;
      rotate(0.0 + step * dir0, 0.0 + step * dir1); td = d + step * dir2;
      e1v = energy();
    }
    if (stuck) break;
    cx = tx; cy = ty; cz = tz; d = td;
  }
  if (rounds) *rounds = guard;
  return ok;
}

// The same refinement computed by a whole wave for ONE plane (all 64 lanes call it with the same arguments): the 12 barrier
// terms -- the logarithms and divisions, i.e. nearly all of the work -- sit on lanes 0..11 and are summed in the reference's
// order (an inactive term adds +0.0, which is exact); the 3x3 algebra is uniform.  Same expressions as opt_plane_pair =>
// identical results.
// The Armijo search of a round is the long part: a step clamped to a quarter turn, or an offset component of 1e8 behind the repaired 1e-8
// eigenvalue, is backed off tens to hundreds of times, each time a rotation (two sin / cos pairs) and an energy (12 logarithms) -- measured
// on SCN-C: 20 - 40 rounds of ~7 us each for the slowest plane of a launch, which is what k_keep lasts.  Here FIVE evaluations run side by
// side: lanes 12 s .. 12 s + 11 hold the 12 barrier terms of slot s (the first pass: e0 and candidates 0..3; later passes: five more
// candidates), lanes 2 s and 2 s + 1 take the slot's two sin / cos pairs, the slots' sums are formed by five lanes in the reference's
// order out of LDS, and the candidates are then looked at IN ORDER -- the first that passes the Armijo test is the one the sequential
// loop would have stopped at, its energy the same expression on the same operands.  te64: 64 doubles of wave-private LDS.
__device__ inline bool opt_plane_pair_wave(const double* A, const double* B, double m, double off, int lane, double& cx, double& cy, double& cz, double& d, double* te64, int* rounds = nullptr, long long* dbgout = nullptr) {
  const int j = lane < 12 ? lane : 0;
  const bool second = j >= 6;
  const double* rp = second ? B + 3 * (j - 6) : A + 3 * j;
  const double r[3] = {rp[0], rp[1], rp[2]};
  // the same point again for the slot layout of the Armijo passes: lane 12 s + q holds point q of slot s
  const int slot = min(lane / 12, 4), q5 = lane - 12 * slot;
  const bool live = lane < 60;
  const bool second5 = q5 >= 6 && live;
  const double* rp5 = live ? (second5 ? B + 3 * (q5 - 6) : A + 3 * q5) : A;
  const double r5[3] = {rp5[0], rp5[1], rp5[2]};
  int guard = 0;
  bool ok = false;
#ifdef TJ_PHASE_TIMING
  long long tt[4] = {0, 0, 0, 0}; int npass = 0; long long tq;
#define OPT_T0 tq = wall_clock64()
#define OPT_T(i) do { const long long n_ = wall_clock64(); tt[i] += n_ - tq; tq = n_; } while (0)
#else
#define OPT_T0 do {} while (0)
#define OPT_T(i) do {} while (0)
#endif
  for (; guard < PLANE_NEWTON_CAP; guard++) {
    OPT_T0;
    const OpFrame f = op_frame(cx, cy, cz);
    OpTerm t = op_pair_term(r, second, f, cx, cy, cz, d, m, off);
    if (lane >= 12) t = OpTerm{0, 0, 0, 0, 0, 0};
    TJ_ORDER(t.h22); OPT_T(0);
    double g0 = 0, g1 = 0, g2 = 0, h00 = 0, h10 = 0, h20 = 0, h11 = 0, h21 = 0, h22 = 0;
#pragma unroll
    for (int q = 0; q < 12; q++) {
      g0 += gjk_rl(t.g0, q); g1 += 0; g2 += gjk_rl(t.g2, q); h00 += gjk_rl(t.h00, q); h10 += gjk_rl(t.h10, q); h20 += gjk_rl(t.h20, q); h11 += 0; h21 += 0; h22 += gjk_rl(t.h22, q);
    }
    TJ_ORDER(h22); OPT_T(1);
    if (sqrt(op_sum3v(g0 * g0, g1 * g1, g2 * g2)) < 1e-2) { ok = true; break; }
    double dir0, dir1, dir2, w, step;
    op_pair_direction(g0, g1, g2, h00, h10, h20, h11, h21, h22, dir0, dir1, dir2, w, step);
    TJ_ORDER(step); OPT_T(2);
    // one pass: slot s evaluates the energy at step `sv[s]` (NaN marks "angles exactly 0, offset d": e0)
    double e0 = 0, tx = cx, ty = cy, tz = cz, td = d;
    bool have_e0 = false, stuck = false, accepted = false;
    int gi = 0;   // index of the next candidate in the sequential loop's order (0 = the first trial step, then the back-offs)
    while (!accepted && !stuck) {
#ifdef TJ_PHASE_TIMING
      npass++;
#endif
      double sv[5];
      int first = 0;
      if (!have_e0) { sv[0] = 0.0; first = 1; }
      for (int s_ = first; s_ < 5; s_++) { sv[s_] = step; step *= 0.8; }   // the sequential loop's step *= 0.8, five at a time (uniform)
      const bool is_e0 = !have_e0 && slot == 0;
      const double my_step = slot == 0 ? sv[0] : (slot == 1 ? sv[1] : (slot == 2 ? sv[2] : (slot == 3 ? sv[3] : sv[4])));
      // sin / cos: lane 2 s takes theta of slot s, lane 2 s + 1 its phi
      const int as = min(lane >> 1, 4);
      const double a_step = as == 0 ? sv[0] : (as == 1 ? sv[1] : (as == 2 ? sv[2] : (as == 3 ? sv[3] : sv[4])));
      const bool a_e0 = !have_e0 && as == 0;
      const double ang = a_e0 ? 0.0 : ((lane & 1) ? 0.0 + a_step * dir1 : 0.0 + a_step * dir0);
      double cvv, svv; cr_sincos(ang, &svv, &cvv);
      const double ct = __shfl(cvv, 2 * slot), st = __shfl(svv, 2 * slot), cp = __shfl(cvv, 2 * slot + 1), sp = __shfl(svv, 2 * slot + 1);
      double ux, uy, uz;
      op_rotate(ct, st, cp, sp, cx, cy, cz, f, ux, uy, uz);
      const double ud = is_e0 ? d : d + my_step * dir2;
      double te;
      const bool fine = op_pair_energy_term(r5, second5, ux, uy, uz, ud, m, off, te) || !live;
      if (!live) te = 0;
      const unsigned long long notfine = __ballot(!fine);
      te64[lane] = te;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
      double es = 0;
      if (lane < 5) {
#pragma unroll
        for (int q = 0; q < 12; q++) es += te64[12 * lane + q];
        if ((notfine >> (12 * lane)) & 0xfffull) es = INFINITY;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
      if (!have_e0) { e0 = gjk_rl(es, 0); have_e0 = true; }
      for (int s_ = first; s_ < 5; s_++) {   // uniform: in the sequential loop's order
        if (gi > PLANE_BACKOFF_CAP) { stuck = true; break; }
        const double e1v = s_ == 1 ? gjk_rl(es, 1) : (s_ == 2 ? gjk_rl(es, 2) : (s_ == 3 ? gjk_rl(es, 3) : (s_ == 4 ? gjk_rl(es, 4) : gjk_rl(es, 0))));
        if (!(e0 - 1e-4 * w * sv[s_] < e1v)) {
          const int src = 12 * s_;
          tx = gjk_rl(ux, src); ty = gjk_rl(uy, src); tz = gjk_rl(uz, src); td = gjk_rl(ud, src);
          accepted = true;
          break;
        }
        gi++;
      }
    }
    TJ_ORDER(td); OPT_T(3);
    if (stuck) break;
    cx = tx; cy = ty; cz = tz; d = td;
  }
#ifdef TJ_PHASE_TIMING
  if (dbgout && lane == 0 && guard >= 8) { dbgout[0] = guard; dbgout[1] = npass; dbgout[2] = tt[0]; dbgout[3] = tt[1]; dbgout[4] = tt[2]; dbgout[5] = tt[3]; }
#endif
  if (rounds) *rounds = guard;
  return ok;
}

}  // namespace tj
