// kernels_ls.h -- Armijo line search on the x-objective, one workgroup per robot.
//
// Replaces spline_line_search (Optimization3D_multi.h:754-811, Optimization3D_admm.h:505-557) and
// Energy_admm::spline_energy / plane_barrier_energy / bound_energy (Energy_admm.h:16-170).
//
// The reference evaluates E(x), then E(x + s d) for s = s0, 0.8 s0, 0.8^2 s0, ... one after the other
// and stops at the first s that satisfies the Armijo test.  Each evaluation is tiny (a few hundred
// barrier terms) but strictly sequential -- on a GPU that is a chain of ~10 us latencies.  Here the
// 512-thread workgroup is split into 8 groups of 64 lanes (one wave each) and each evaluates ONE candidate:
// round 0 covers E(x) and the first 7 trial steps, later rounds 8 steps each.  The accepted step is
// the first one in the reference's order that passes the test, so the result is the one the
// sequential loop would produce; E(.) is a pure function of its inputs here (fixed reduction tree
// inside a group), i.e. it does not depend on which group or round evaluates it.
// Everything an evaluation reads more than once (basis, this robot's planes, slack/dual blocks)
// is staged in LDS once per launch.
#pragma once
#include "dev_common.h"
#include "dev_linalg.h"

namespace tj {

constexpr int LS_THREADS = 512;
constexpr int LS_GROUPS = 8;    // 16 candidates per round (1024 threads, 128-VGPR cap) measured SLOWER: 44.7 vs 40.5 us over the first 20 iterations
constexpr int LS_GSIZE = 64;   // one wave per candidate: group-private LDS needs only wave-local ordering

struct LsLayout {  // offsets in doubles into dynamic LDS
  size_t basis, convert, slack, lambda, tsl, tla, net, dir, gnet, ghull, gcons, res, planes, pltr, hn, hd, wseg, total;
  int plane_cap;
  int affine;   // hulls of a trial step come from hull(net) + step * hull(dir), both formed once per launch (see x_energy_group)
  int groups;   // Armijo candidates evaluated side by side (one wave each): 8 where the per-candidate hull buffers fit LDS,
                // fewer for long trajectories (piece_num > 10); the accepted step is the same, only the rounds get shorter
};
__host__ __device__ inline LsLayout ls_layout_g(int S, int T, int P, size_t lds_budget_bytes, int G, int affine = 0) {
  LsLayout L;
  L.groups = G;
  L.affine = affine;
  size_t o = 0;
  L.basis = o; o += (size_t)S * 36;
  L.convert = o; o += (size_t)P * 36;
  L.slack = o; o += (size_t)18 * P;
  L.lambda = o; o += (size_t)18 * P;
  L.tsl = o; o += P;
  L.tla = o; o += P;
  L.net = o; o += 3 * (size_t)T;
  L.dir = o; o += 3 * (size_t)T;
  L.gnet = o; o += (size_t)G * 3 * T;
  L.ghull = o; o += (size_t)G * S * 18;
  L.gcons = o; o += (size_t)G * 24 * P;   // per group: delta[18P], then 6 consensus/dual terms per piece
  L.res = o; o += 2 * LS_GROUPS + 8;
  L.hn = o; o += affine ? (size_t)S * 18 : 0;
  L.hd = o; o += affine ? (size_t)S * 18 : 0;
  L.wseg = o; o += S;   // seg_weight per segment (two divisions and a modulo per use otherwise)
  L.planes = o;
  const size_t used = o * 8;
  size_t room = lds_budget_bytes > used ? (lds_budget_bytes - used) / 36 : 0;  // 32 B plane + 4 B segment id
  if (room > 4096) room = 4096;
  L.plane_cap = (int)room;
  o += (size_t)L.plane_cap * 4;
  L.pltr = o; o += ((size_t)L.plane_cap + 1) / 2;
  L.total = o;
  return L;
}
// lds_budget_bytes: soft budget that also sizes the LDS-resident plane list; hard_bytes: what one workgroup may allocate.  The
// widest group count whose fixed part fits is taken (8 up to piece_num = 10 with the shipped res = 8).
__host__ __device__ inline LsLayout ls_layout(int S, int T, int P, size_t lds_budget_bytes, size_t hard_bytes = 155 * 1024) {
  LsLayout L = ls_layout_g(S, T, P, lds_budget_bytes, LS_GROUPS, 1);
  if (L.total * 8 <= hard_bytes) return L;
  L = ls_layout_g(S, T, P, lds_budget_bytes, LS_GROUPS);
  for (int G = LS_GROUPS / 2; G >= 1 && L.total * 8 > hard_bytes; G /= 2) L = ls_layout_g(S, T, P, lds_budget_bytes, G);
  return L;
}

// E(net, pt) for robot u, evaluated by ONE group = one wave (gl = lane within the group).
// Returns the value in every lane of the group.  Must be called by the whole workgroup (contains a
// block barrier).
// one control point coordinate of a segment's hull: row j of the segment's basis times the piece's 6 control points
__device__ __forceinline__ double ls_hull_entry(const Dev& D, const double* basis, const double* net, int idx) {
  const int tr = idx / 18, e = idx % 18, j = e / 3, a = e % 3;
  const double* B = basis + (size_t)tr * 36 + j * 6;
  const double* col = net + div_small(tr, D.res) * 3 + D.T * a;
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) acc += B[k] * col[k];
  return acc;
}

// `trial`: 0 = the hulls are those of `net` itself (E(x)), 1 = a trial step `step` along the direction.  With L.affine the
// trial hulls are hull(x) + step * hull(d) (the hull map is linear; the two images are formed once per launch by ls_stage)
// instead of 6 multiply-adds per entry and candidate: the hull pass was 2.9 of the 10.4 us of an evaluation.  The values
// differ from basis * (x + step d) by rounding (~1e-16 relative) -- far below any Armijo margin seen -- and the hulls that
// are PUBLISHED for the next iteration are recomputed exactly from the accepted control net.
__device__ inline double x_energy_group(const Dev& D, int u, const double* sm, const LsLayout& L, const double* net, double pt, double* hulls,
                                        double* cons, int M, bool planes_in_lds, const int* pref, int gl, int trial, double step) {
  const int S = D.S, T = D.T;
  const double* basis = sm + L.basis;
  const double* wsg = sm + L.wseg;
  TJ_TIC(D, K_BEGIN, 0);
  if (L.affine) {
    const double* hn = sm + L.hn; const double* hd = sm + L.hd;
    for (int idx = gl; idx < S * 18; idx += LS_GSIZE) hulls[idx] = trial ? hn[idx] + step * hd[idx] : hn[idx];
  } else {
    for (int idx = gl; idx < S * 18; idx += LS_GSIZE) hulls[idx] = ls_hull_entry(D, basis, net, idx);
  }
  __syncthreads();  // all 8 groups run this function in lock step (uniform trip counts)
  TJ_TIC(D, K_BEGIN, 1);
  const double m = D.margin;
  double part = 0, partb = 0;
  int bad = 0;
  // velocity / acceleration barriers (Energy_admm.h:98-170); two loops so that a wave never runs
  // both formulas
  for (int it = gl; it < S * 5; it += LS_GSIZE) {
    const int tr = it / 5, b = it % 5;
    const double w = wsg[tr];
    const double* Pp = hulls + tr * 18;
    const double vx = 5 * (Pp[3 * (b + 1)] - Pp[3 * b]), vy = 5 * (Pp[3 * (b + 1) + 1] - Pp[3 * b + 1]), vz = 5 * (Pp[3 * (b + 1) + 2] - Pp[3 * b + 2]);
    const double d = D.vel_limit - norm3(vx, vy, vz) / (w * pt);
    if (d <= 0) bad = 1;
    else if (d < m) partb += barrier(w, d, m);
  }
  for (int it = gl; it < S * 4; it += LS_GSIZE) {
    const int tr = it / 4, j = it % 4;
    const double w = wsg[tr];
    const double* Pp = hulls + tr * 18;
    const double ax = 20 * (Pp[3 * (j + 2)] - 2 * Pp[3 * (j + 1)] + Pp[3 * j]), ay = 20 * (Pp[3 * (j + 2) + 1] - 2 * Pp[3 * (j + 1) + 1] + Pp[3 * j + 1]),
                 az = 20 * (Pp[3 * (j + 2) + 2] - 2 * Pp[3 * (j + 1) + 2] + Pp[3 * j + 2]);
    const double d = D.acc_limit - norm3(ax, ay, az) / (w * w * pt * pt);
    if (d <= 0) bad = 1;
    else if (d < m) partb += barrier(w, d, m);
  }
  // A violated velocity / acceleration limit already makes the energy +infinity (Energy_admm.h:137-138,154-155): skip the plane
  // terms and the consensus sums.  In the first iterations every robot backs off 10-20 times on exactly this (measured:
  // tests/devtools/armijo_hist.py), so most candidates of those rounds end here.  The group = one wave: the decision is uniform.
  if (__ballot(bad != 0) != 0ull) return INFINITY;
  TJ_TIC(D, K_BEGIN, 2);
  // plane barrier (Energy_admm.h:46-96)
  const double* pl_lds = sm + L.planes;
  const int* pltr = (const int*)(sm + L.pltr);
  if (planes_in_lds) {
    // one (plane, hull point) term per lane and pass: a robot of the headline scene carries ~30 planes, so a lane per PLANE left
    // half the wave idle while the others took six logarithms one after the other
    // FOUR passes at a time while every lane has a term in each of them: a pass is a chain of two dependent LDS reads, a dot product and a logarithm (~0.5 us on its
    // own), and a robot next to an obstacle slab walks 45 of them per candidate (config 5: 28 us per round of eight).  The four terms are formed side by side --
    // an inactive one is +0.0, which changes no bit of the sum (x + 0.0 == x; the sums start at +0 and every term is positive) -- and added in pass order.
    int pass = 0;
    const int full = (6 * M) / LS_GSIZE;
    for (; pass + 4 <= full; pass += 4) {
      double t4[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int it = gl + LS_GSIZE * (pass + q);
        const int ip = it / 6, j = it - 6 * ip, tr = pltr[ip];
        const double* Pp = hulls + tr * 18 + 3 * j; const double* pl = pl_lds + 4 * ip;
        const double d = Pp[0] * pl[0] + Pp[1] * pl[1] + Pp[2] * pl[2] + pl[3];
        if (d <= 0) bad = 1;
        const double b = barrier(wsg[tr], d, m);
        t4[q] = (d > 0 && d < m) ? b : 0.0;
      }
      part += t4[0]; part += t4[1]; part += t4[2]; part += t4[3];
    }
    for (int it = gl + LS_GSIZE * pass; it < 6 * M; it += LS_GSIZE) {
      const int ip = it / 6, j = it - 6 * ip, tr = pltr[ip];
      const double* Pp = hulls + tr * 18 + 3 * j; const double* pl = pl_lds + 4 * ip;
      const double d = Pp[0] * pl[0] + Pp[1] * pl[1] + Pp[2] * pl[2] + pl[3];
      if (d <= 0) bad = 1;
      else if (d < m) part += barrier(wsg[tr], d, m);
    }
  } else for (int it = gl; it < M; it += LS_GSIZE) {
    int lo = 0, hi = S;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pref[mid] <= it) lo = mid; else hi = mid; }
    const int tr = lo;
    const int k = it - pref[tr], no = pref[512 + tr];
    const double* pl = k < no ? D.oplanes + (((size_t)u * S + tr) * D.cap_obs + k) * 4 : D.splanes + (((size_t)u * S + tr) * D.cap_self + (k - no)) * 4;
    const double c0 = pl[0], c1 = pl[1], c2 = pl[2], dk = pl[3];
    const double w = wsg[tr];
    const double* Pp = hulls + tr * 18;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double d = Pp[3 * j] * c0 + Pp[3 * j + 1] * c1 + Pp[3 * j + 2] * c2 + dk;
      if (d <= 0) bad = 1;
      else if (d < m) part += barrier(w, d, m);
    }
  }
  TJ_TIC(D, K_BEGIN, 3);
  // fixed butterfly inside the group
#pragma unroll
  for (int off = LS_GSIZE / 2; off > 0; off >>= 1) {
    part += __shfl_xor(part, off, LS_GSIZE);
    partb += __shfl_xor(partb, off, LS_GSIZE);
    bad |= __shfl_xor(bad, off, LS_GSIZE);
  }
  TJ_TIC(D, K_BEGIN, 4);
  double e = D.lambda * part + D.lambda * partb;
  // augmented-Lagrangian terms (Energy_admm.h:24-41).  The 18P entries of C x - z are spread over
  // the lanes, lane sp then forms the six terms of piece sp (Eigen's reduction order inside each),
  // and every lane adds the 6P terms in the reference's statement order.  cons is private to this
  // group = this wave, so ordering points suffice.
  const int P6 = 6 * D.P;
  const double* cv = sm + L.convert; const double* sl = sm + L.slack; const double* la = sm + L.lambda;
  double* delta = cons; double* terms = cons + 18 * D.P;
  for (int it = gl; it < 18 * D.P; it += LS_GSIZE) {
    const int sp = it / 18, r = it % 18, a = r / 6, j = r % 6;
    const double* C = cv + (size_t)sp * 36 + j * 6;
    double acc = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) acc += C[k] * net[sp * 3 + k + T * a];
    delta[it] = acc - sl[sp * 6 + j + P6 * a];
  }
  blk_sync<true>();
  for (int sp = gl; sp < D.P; sp += LS_GSIZE) {
    const double* dl = delta + 18 * sp;
    double prod[18];
#pragma unroll
    for (int i = 0; i < 18; i++) prod[i] = dl[i] * dl[i];
    double* t = terms + 6 * sp;
    t[0] = D.mu / 2.0 * esum(prod, 18);
    const double dt = pt - sm[L.tsl + sp];
    t[1] = D.mu / 2.0 * (dt * dt);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double pr[6];
#pragma unroll
      for (int j = 0; j < 6; j++) pr[j] = la[sp * 6 + j + P6 * a] * dl[j + 6 * a];
      t[2 + a] = esum(pr, 6);
    }
    t[5] = sm[L.tla + sp] * (pt - sm[L.tsl + sp]);
  }
  blk_sync<true>();
  for (int i = 0; i < 6 * D.P; i++) e += terms[i];
  TJ_TIC(D, K_BEGIN, 5);
  if (bad) e = INFINITY;
  return e;
}

// ---- the same E(net, pt), evaluated by a TEAM of four waves (round 0 of a robot that accepted the full step last time) ----------
// In the steady phase of a run nearly every robot accepts candidate 0, so all a launch has to decide is E(x) against E(x + s0 d) --
// and x_energy_group walks each of them through 7 serial velocity / acceleration passes and ~3 plane passes on ONE wave while six
// other waves evaluate candidates nobody looks at.  Here a pass is a UNIT: unit u (velocity passes, then acceleration passes, then
// plane passes, the order x_energy_group meets them in) is taken by wave u & 3 of the team, which leaves its lane's TERM
// (0.0 where the record is inactive: x + 0.0 == x bit for bit, the sums start at +0) in terms[u][lane]; wave 3 also forms the
// consensus / dual terms.  After one block barrier wave 0 of the team adds every lane's terms in unit order -- the very sequence of
// additions x_energy_group performs in that lane -- and finishes with the same butterfly and the same statement order.  Same
// operands, same operations, same order: the value is BITWISE x_energy_group's (test: TJ_LS_FAST=0 changes no bit), so it does not
// matter which shape evaluated E(x) when a later round compares candidates against it -- which the loops that end by rounding
// (stages_stack030) rely on.  Both teams run this in lock step (uniform trip counts; a violated limit does not leave early here).
struct LsTeamUnits { int nv, na, np; __device__ int total() const { return nv + na + np; } };
__device__ __forceinline__ LsTeamUnits ls_team_units(int S, int M) { return LsTeamUnits{(5 * S + 63) / 64, (4 * S + 63) / 64, (6 * M + 63) / 64}; }
// abort_word (helper blocks running a super-round ahead of the primary's word, k_linesearch): thread 0 of the block reads the robot's word once,
// mid-evaluation, and if the search is over by then (LS_WORD_DONE of this epoch) every wave leaves at the barrier -- *s_abort says so, the value is not used.
__device__ __forceinline__ double x_energy_team(const Dev& D, const double* sm, const LsLayout& L, const double* net, double pt, const double* hulls, double* terms,
                                                double* cons, int* bad_flag, double* out, int M, int tw, int gl,
                                                const unsigned long long* abort_word = nullptr, unsigned epoch = 0, int* s_abort = nullptr) {
  const int S = D.S, T = D.T;
  const double* wsg = sm + L.wseg;
  const double m = D.margin;
  const LsTeamUnits U4 = ls_team_units(S, M);
  const double* pl_lds = sm + L.planes;
  const int* pltr = (const int*)(sm + L.pltr);
  int bad = 0;
  unsigned long long w_abort = 0;
  for (int unit = tw; unit < U4.total(); unit += 4) {
    if (abort_word && threadIdx.x == 0 && unit + 4 >= U4.total()) w_abort = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // before this wave's last unit; used at the barrier
    double term = 0.0;
    if (unit < U4.nv) {
      const int it = gl + 64 * unit;
      if (it < S * 5) {
        const int tr = it / 5, b = it % 5;
        const double w = wsg[tr];
        const double* Pp = hulls + tr * 18;
        const double vx = 5 * (Pp[3 * (b + 1)] - Pp[3 * b]), vy = 5 * (Pp[3 * (b + 1) + 1] - Pp[3 * b + 1]), vz = 5 * (Pp[3 * (b + 1) + 2] - Pp[3 * b + 2]);
        const double d = D.vel_limit - norm3(vx, vy, vz) / (w * pt);
        if (d <= 0) bad = 1;
        else if (d < m) term = barrier(w, d, m);
      }
    } else if (unit < U4.nv + U4.na) {
      const int it = gl + 64 * (unit - U4.nv);
      if (it < S * 4) {
        const int tr = it / 4, j = it % 4;
        const double w = wsg[tr];
        const double* Pp = hulls + tr * 18;
        const double ax = 20 * (Pp[3 * (j + 2)] - 2 * Pp[3 * (j + 1)] + Pp[3 * j]), ay = 20 * (Pp[3 * (j + 2) + 1] - 2 * Pp[3 * (j + 1) + 1] + Pp[3 * j + 1]),
                     az = 20 * (Pp[3 * (j + 2) + 2] - 2 * Pp[3 * (j + 1) + 2] + Pp[3 * j + 2]);
        const double d = D.acc_limit - norm3(ax, ay, az) / (w * w * pt * pt);
        if (d <= 0) bad = 1;
        else if (d < m) term = barrier(w, d, m);
      }
    } else {
      const int it = gl + 64 * (unit - U4.nv - U4.na);
      if (it < 6 * M) {
        const int ip = it / 6, j = it - 6 * ip, tr = pltr[ip];
        const double* Pp = hulls + tr * 18 + 3 * j; const double* pl = pl_lds + 4 * ip;
        const double d = Pp[0] * pl[0] + Pp[1] * pl[1] + Pp[2] * pl[2] + pl[3];
        if (d <= 0) bad = 1;
        else if (d < m) term = barrier(wsg[tr], d, m);
      }
    }
    terms[unit * 64 + gl] = term;
  }
  if (__ballot(bad != 0) != 0ull && gl == 0) *bad_flag = 1;
  const int P6 = 6 * D.P;
  double* delta = cons; double* cterms = cons + 18 * D.P;
  if (tw == 3) {   // augmented-Lagrangian terms, statement for statement as in x_energy_group
    const double* cv = sm + L.convert; const double* sl = sm + L.slack; const double* la = sm + L.lambda;
    for (int it = gl; it < 18 * D.P; it += LS_GSIZE) {
      const int sp = it / 18, r = it % 18, a = r / 6, j = r % 6;
      const double* C = cv + (size_t)sp * 36 + j * 6;
      double acc = 0;
#pragma unroll
      for (int k = 0; k < 6; k++) acc += C[k] * net[sp * 3 + k + T * a];
      delta[it] = acc - sl[sp * 6 + j + P6 * a];
    }
    blk_sync<true>();
    for (int sp = gl; sp < D.P; sp += LS_GSIZE) {
      const double* dl = delta + 18 * sp;
      double prod[18];
#pragma unroll
      for (int i = 0; i < 18; i++) prod[i] = dl[i] * dl[i];
      double* t = cterms + 6 * sp;
      t[0] = D.mu / 2.0 * esum(prod, 18);
      const double dt = pt - sm[L.tsl + sp];
      t[1] = D.mu / 2.0 * (dt * dt);
#pragma unroll
      for (int a = 0; a < 3; a++) {
        double pr[6];
#pragma unroll
        for (int j = 0; j < 6; j++) pr[j] = la[sp * 6 + j + P6 * a] * dl[j + 6 * a];
        t[2 + a] = esum(pr, 6);
      }
      t[5] = sm[L.tla + sp] * (pt - sm[L.tsl + sp]);
    }
  }
  if (abort_word && threadIdx.x == 0) *s_abort = (unsigned)(w_abort >> 32) == epoch && (unsigned)w_abort == LS_WORD_DONE;
  __syncthreads();
  double e = 0;
  if (abort_word && *s_abort) return e;   // (uniform)
  if (tw == 0) {
    double part = 0, partb = 0;
    for (int u = 0; u < U4.nv + U4.na; u++) partb += terms[u * 64 + gl];
    for (int u = U4.nv + U4.na; u < U4.total(); u++) part += terms[u * 64 + gl];
#pragma unroll
    for (int off = LS_GSIZE / 2; off > 0; off >>= 1) {
      part += __shfl_xor(part, off, LS_GSIZE);
      partb += __shfl_xor(partb, off, LS_GSIZE);
    }
    e = D.lambda * part + D.lambda * partb;
    for (int i = 0; i < 6 * D.P; i++) e += cterms[i];
    if (*bad_flag) e = INFINITY;
    if (gl == 0) *out = e;
  }
  return e;
}

// Stage everything an evaluation reuses into LDS: tables, this robot's slack/dual blocks, control net,
// search direction and (if they fit) its planes with their segment ids.  Returns the plane count M and
// whether the planes are LDS resident.  Called by the whole block (contains barriers).
__device__ __forceinline__ int ls_stage(const Dev& D, const LsLayout& L, double* sm, int* pref, int u, int tid, int nth, bool& in_lds_out) {
  const int S = D.S, T = D.T, P = D.P, P6 = 6 * D.P;
  double* net = sm + L.net; double* dir = sm + L.dir;
  const double* gspline = D.spline + (size_t)u * 3 * T;
  TJ_TIC(D, K_LINESEARCH, 0);
  // plane counts of the first 64 segments: issued FIRST by wave 0 (its scan below is the only dependent step of this phase)
  int c_no = 0, c_all = 0;
  if (tid < 64 && tid < S) { c_no = D.ocount[u * S + tid]; c_all = c_no + (D.multi() ? D.scount[u * S + tid] : 0); }
  if (!L.affine) for (int i = tid; i < S * 36; i += nth) sm[L.basis + i] = D.basis[i];   // (affine layout: the only users of the basis -- hull(x), hull(d) and the published hulls -- read it from global memory)
  for (int i = tid; i < P * 36; i += nth) sm[L.convert + i] = D.convert[i];
  for (int i = tid; i < 18 * P; i += nth) { sm[L.slack + i] = D.p_slack[(size_t)u * 3 * P6 + i]; sm[L.lambda + i] = D.p_lambda[(size_t)u * 3 * P6 + i]; }
  for (int i = tid; i < P; i += nth) { sm[L.tsl + i] = D.t_slack[u * P + i]; sm[L.tla + i] = D.t_lambda[u * P + i]; }
  for (int i = tid; i < 3 * T; i += nth) { net[i] = gspline[i]; dir[i] = D.dirp(u)[i]; }
  for (int i = tid; i < S; i += nth) sm[L.wseg + i] = seg_weight(D, i);
  if (L.affine) {   // hull(x) and hull(d) straight from global memory, in the same round trip as everything above (they used to wait for the LDS copies of
                    // basis / net / dir behind the barrier below); same expression, same bits.  Consumed after the second barrier.
    const double* gdir = D.dirp(u);
    for (int idx = tid; idx < S * 18; idx += nth) { sm[L.hn + idx] = ls_hull_entry(D, D.basis, gspline, idx); sm[L.hd + idx] = ls_hull_entry(D, D.basis, gdir, idx); }
  }
  if (tid < 64) {  // plane-count prefix over the segments: lanes load, wave scan (a one-thread loop was 2S dependent loads, ~4 us)
    int run = 0;
    for (int base = 0; base < S; base += 64) {
      const int tr = base + tid;
      const int no_ = base == 0 ? c_no : (tr < S ? D.ocount[u * S + tr] : 0);
      const int c = base == 0 ? c_all : (tr < S ? no_ + (D.multi() ? D.scount[u * S + tr] : 0) : 0);
      if (tr < S) pref[512 + tr] = no_;   // obstacle planes of the segment: the plane gather below needs no second trip for it
      int x = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (tid >= off) x += y; }
      if (tr < S) pref[tr] = run + x - c;
      run += __shfl(x, 63);
    }
    if (tid == 0) pref[S] = run;
  }
  __syncthreads();
  TJ_TIC(D, K_LINESEARCH, 1);
  const int M = pref[S];
  const bool in_lds = M <= L.plane_cap;
  if (in_lds) {
    int* pltr = (int*)(sm + L.pltr);
    for (int it = tid; it < M; it += nth) {
      int lo = 0, hi = S;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pref[mid] <= it) lo = mid; else hi = mid; }
      const int tr = lo, k = it - pref[tr], no = pref[512 + tr];
      const double* pl = k < no ? D.oplanes + (((size_t)u * S + tr) * D.cap_obs + k) * 4 : D.splanes + (((size_t)u * S + tr) * D.cap_self + (k - no)) * 4;
      pltr[it] = tr;
      sm[L.planes + 4 * it] = pl[0]; sm[L.planes + 4 * it + 1] = pl[1]; sm[L.planes + 4 * it + 2] = pl[2]; sm[L.planes + 4 * it + 3] = pl[3];
    }
  }
  __syncthreads();
  in_lds_out = in_lds;
  return M;
}

// Hull cache of the NEXT iteration's robot-pair stage (layout of k_hullinfo, kernels_pairs.h: hull 18, AABB 6,
// 49 k-DOP intervals lo/hi), written from the hulls the accepted Armijo candidate was evaluated on -- they ARE the
// hulls of the control net just committed, so the separate k_hullinfo launch (and its place on the critical path)
// disappears from the single-GPU iteration graph.  Same expressions as k_hullinfo => identical bits.
constexpr int LS_HULL_STRIDE = 128;   // = HULL_INFO_STRIDE (kernels_sep.h)
__device__ __forceinline__ void ls_publish_hullinfo(const Dev& D, int u, const double* hulls, int tid, int nth) {
  const int S = D.S;
  double* o = D.hullinfo + (size_t)u * S * LS_HULL_STRIDE;
  for (int idx = tid; idx < S * 18; idx += nth) o[(size_t)(idx / 18) * LS_HULL_STRIDE + idx % 18] = hulls[idx];
  for (int idx = tid; idx < S * 3; idx += nth) {
    const int tr = idx / 3, a = idx % 3;
    const double* P = hulls + tr * 18;
    double lo = INFINITY, hi = -INFINITY;
    for (int j = 0; j < 6; j++) { const double v = P[3 * j + a]; if (v < lo) lo = v; if (v > hi) hi = v; }
    o[(size_t)tr * LS_HULL_STRIDE + 18 + a] = lo; o[(size_t)tr * LS_HULL_STRIDE + 21 + a] = hi;
    D.hbox[((size_t)tr * 6 + a) * D.U + u] = lo; D.hbox[((size_t)tr * 6 + 3 + a) * D.U + u] = hi;
  }
  for (int idx = tid; idx < S * 49; idx += nth) {
    const int tr = idx / 49, ax = idx % 49;
    const double* P = hulls + tr * 18;
    const double x = D.kdop[3 * ax], y = D.kdop[3 * ax + 1], z = D.kdop[3 * ax + 2];
    double up = -INFINITY, lo = INFINITY;
    for (int i = 0; i < 6; i++) { const double lv = x * P[3 * i] + y * P[3 * i + 1] + z * P[3 * i + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
    o[(size_t)tr * LS_HULL_STRIDE + 24 + ax] = lo; o[(size_t)tr * LS_HULL_STRIDE + 73 + ax] = up;
  }
}

// The work of k_begin, callable by any workgroup: the standalone kernel runs it before the first iteration of a batch, the
// last k_linesearch block to finish runs it for the NEXT iteration of the same batch (one kernel boundary less per iteration).
// quiet: 1 = every robot accepted the full step (exponent 0) in the line search that has just ended (k_linesearch's tickets carry the information; the
// standalone k_begin, which follows no line search of its batch, says 0): Ctl::ls_quiet counts such iterations in a row, and k_linesearch sends its helper
// blocks home while the count stands at LS_QUIET_ITERS or more.
// fa_started (asynchronous front, Dev::fa_seq): the iteration's k_front is already running on the other queue -- what IT counts with (the work-list counters, the hull
// records' completion counters) was reset by fa_early_begin before it was launched and is left alone here.
__device__ __forceinline__ bool begin_body(const Dev& D, int quiet = 0, bool fa_started = false) {   // returns whether the stop test has fired
  // stop test of the mains: iter>1 && gnorm<stop (Main/multiPathPlanning3D.cpp:633)
  __shared__ int done;
  if (threadIdx.x == 0) {
    Ctl h = *D.ctl;  // ONE wide read of the control block instead of a chain of dependent field loads
    if (h.pending) { h.iter++; h.pending = 0; }  // count the previous iteration (saves a launch)
    h.slack_now = h.slack_next; h.slack_next = 0;
    if (!h.done && D.stop > 0 && h.iter > 1 && h.gnorm < D.stop) h.done = 1;
    done = h.done;
    if (!done) { h.pending = 1; h.epoch++; h.slack_next = 1; }
    D.ctl->iter = h.iter; D.ctl->pending = h.pending; D.ctl->slack_now = h.slack_now; D.ctl->slack_next = h.slack_next;
    D.ctl->done = h.done; D.ctl->epoch = h.epoch; D.ctl->any_pair = 0;   // error bits and counters are only ever touched by atomics elsewhere
    D.ctl->gjk_max_sum = h.gjk_max_sum + (unsigned long long)h.gjk_max; D.ctl->gjk_prev = h.gjk_max; D.ctl->gjk_max = 0;
    D.ctl->ls_quiet = quiet ? min(h.ls_quiet + 1, 1 << 20) : 0;
  }
  __syncthreads();
  if (done) return true;
  for (int i = threadIdx.x; i < D.U; i += blockDim.x) { D.k_obs[i] = 0; D.k_self[i] = 0; }
  if (D.multi() && !fa_started) for (int i = threadIdx.x; i <= D.S; i += blockDim.x) D.pair_work_n[i] = 0;   // per-segment counts, [S] = cursor of the pair-solve waves
  if (threadIdx.x == 0 && !fa_started) *D.obs_work_n = 0;
  if (threadIdx.x < 3 && D.multi()) D.pair_ovf[threadIdx.x] = 0;
  if (threadIdx.x < 16) D.ctl->ccd_sub[threadIdx.x] = 0;   // arrival counters of k_ccd's selection blocks (folded pair replay)
  if (threadIdx.x == 0) D.ctl->c2_cnt = 0;
  if (D.xs_sync) for (int i = threadIdx.x; i <= 2 * D.U; i += blockDim.x) D.xs_sync[(size_t)i * 32] = 0;   // tickets, flags and the count of the asynchronous Newton solve
  if (D.xf || D.xs_async) for (int i = threadIdx.x + (fa_started ? D.S : 0); i < 2 * D.S; i += blockDim.x) D.xf_seg[(size_t)i * XF_SEG_STRIDE] = 0;   // ... and of the foreign-robot units of k_front / k_ccd (sharded contexts)
  if (D.ls_help > 1) for (int i = threadIdx.x; i < (D.u1 - D.u0) * LS_TAB_STRIDE; i += blockDim.x) ((unsigned long long*)D.ls_tab)[(size_t)D.u0 * LS_TAB_STRIDE + i] = LS_TAB_EMPTY;   // k_linesearch's helper posts
  if (D.keep_sync && threadIdx.x < 16) D.keep_sync[threadIdx.x * 32] = 0;   // completion counters of the asynchronous plane refinement
  if (threadIdx.x == 0 && D.optimal_plane && D.multi()) D.kpair_n[1] = D.kpair_n[0];  // planes stored before this iteration (k_keep part 2)
  return false;
}

// Asynchronous front (Dev::fa_seq > 0): what the NEXT iteration's k_front needs of its begin, by one block of k_linesearch at its very start -- before that
// block counts itself started, hence before the gate in front of k_front opens.  The stop test's inputs (iter, pending, gnorm: left by k_ccd) do not change
// while k_linesearch runs, so the decision is the one begin_body takes at the kernel's end; the control block itself is not touched (the blocks of this
// launch still read it).  Record: [1] epoch of the iteration k_front works for, [2] its stop flag.  What k_front counts with is zeroed here (consumed by
// k_mid of the running iteration, long finished): work-list counters and the hull records' completion counters.  Everything written through and waited for.
__device__ __forceinline__ void fa_early_begin(const Dev& D) {
  __shared__ int s_fdone;
  if (threadIdx.x == 0) {
    const Ctl h = *D.ctl;
    const int iter = h.iter + (h.pending ? 1 : 0);
    const int done = (h.done || (D.stop > 0 && iter > 1 && h.gnorm < D.stop)) ? 1 : 0;
    xf_store_i(D.fa_rec() + 1, done ? h.epoch : h.epoch + 1); xf_store_i(D.fa_rec() + 2, done);
    s_fdone = done;
  }
  __syncthreads();
  if (!s_fdone) {
    if (D.multi()) for (int i = threadIdx.x; i <= D.S; i += blockDim.x) xf_store_i(D.pair_work_n + i, 0);
    if (threadIdx.x == 0) xf_store_i(D.obs_work_n, 0);
    for (int i = threadIdx.x; i < D.S; i += blockDim.x) xf_store_i(D.xf_seg + (size_t)i * XF_SEG_STRIDE, 0);
  }
  sig_acked();   // (the signal: this block's count on the residency words, first thing in k_linesearch)
  __syncthreads();
  asm volatile("" ::: "memory");
}

// begin_next = 1: the last block to finish also starts the NEXT iteration (begin_body): the stop test and the counter resets
// need every block of this kernel to be done, which the ticket establishes; the host then omits the k_begin launch.
__global__ __launch_bounds__(LS_THREADS) void k_linesearch(Dev D, LsLayout L, int begin_next) {
  if (D.fa_seq > 0) {   // asynchronous front: the early begin (last block of the grid: a helper where there are helpers), then every block counts itself started
    if (blockIdx.x == gridDim.x - 1) fa_early_begin(D);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(D.fa_res(blockIdx.x), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  }
  if (TJ_DONE(D)) {
    // converged: the only begin work left for the next iteration is to retire the slack/dual update that k_mid has just paid
    if (begin_next && blockIdx.x == 0 && threadIdx.x == 0) { D.ctl->slack_now = D.ctl->slack_next; D.ctl->slack_next = 0; }
    return;
  }
  TJ_TIC_ENTRY(D, K_LINESEARCH);
  extern __shared__ double sm[];
  __shared__ int pref[1024];   // plane prefix per segment (S <= 511 checked on the host); [512 + tr]: obstacle planes of segment tr
  __shared__ int s_accept;
  __shared__ int s_bad[2];
  __shared__ int s_flag;        // super-rounds: 0 = go on, 1 = leave the team loop
  __shared__ int s_abort;       // super-rounds, helper ahead of the word: the search ended during this evaluation
  __shared__ double s_accst;    // super-rounds: step of the accepted candidate
  // ls_help blocks per robot (super-rounds, below): block h of robot ui is blockIdx = ui + h * owned -- the primaries (h = 0) lead the grid
  // In the steady phase of a run (every robot takes the full step, iteration after iteration) the helpers have nothing to contribute and cost ~1 us per
  // launch (their tickets, the table's reset): after LS_QUIET_ITERS such iterations in a row they leave at once and the primaries search on their own
  // (H = 1 below) until a robot backs off again.
  const int nown = D.u1 - D.u0;
  const int H = (D.ls_help > 1 && D.ctl->ls_quiet < LS_QUIET_ITERS) ? D.ls_help : 1;
  const int h = D.ls_help > 1 ? (int)blockIdx.x / nown : 0;
  if (h > 0 && H == 1) return;   // (no ticket: the primaries count among themselves then)
  const int tid = threadIdx.x, u = D.u0 + (int)blockIdx.x - h * nown, S = D.S, T = D.T, P = D.P;
  // G = L.groups candidates per round.  With G < 8 (long trajectories) the waves beyond G shadow the last group: they compute
  // the same candidate into the same buffers (identical values), which keeps every barrier uniform.
  const int G = L.groups;
  const int g = min(tid / LS_GSIZE, G - 1), gl = tid % LS_GSIZE;
  const bool shadow = tid / LS_GSIZE >= G;
  double* net = sm + L.net; double* dir = sm + L.dir;
  double* gnet = sm + L.gnet + (size_t)g * 3 * T;
  double* ghull = sm + L.ghull + (size_t)g * S * 18;
  double* res = sm + L.res;
  double* gspline = D.spline + (size_t)u * 3 * T;
  // scalars of the search first: two dependent global round trips that now overlap the staging below
  const double wolfe = D.wolfe(D.U - 1);  // reference quirk: the global left by the LAST robot (Optimization3D_multi.h:730,792)
  const double t_dir = D.tdir(u), t0 = D.piece_time[u];
  double step0 = D.pow08[min(STEP_CAP, max(D.k_obs[u], D.k_self[u]))];
  const int hist = D.ls_hist[u];   // exponent this robot accepted in the previous iteration (-1: none yet): the round-0 shape follows it
  if (h > 0 && D.ls_help_late > 0) {   // test hook (TJ_LS_HELP_LATE): this helper starts its staging late -- typically after the primary has committed
    const long long t_go = wall_clock64() + 100ll * D.ls_help_late;
    while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(32);
  }
  bool in_lds;
  const int M = ls_stage(D, L, sm, pref, u, tid, LS_THREADS, in_lds);
  TJ_TIC(D, K_LINESEARCH, 2);
  if (t0 + step0 * t_dir <= 0) step0 = -0.95 * t0 / t_dir;

  double e_base = 0, step_acc = step0, pt_acc = t0;
  int k_acc = -1, evals = 0, wg = 0;
  int k_first = -1;   // candidate of group 0 in the next generic round: -1 = E(x) is still to be evaluated (it is group 0's job in that round)
  // Rounds in the TEAM shape (x_energy_team): waves 0-3 evaluate one candidate, waves 4-7 the next.
  //  * one block per robot (ls_help = 1: fleets of more robots than compute units): round 0 only -- E(x) and candidate 0 -- for a robot that
  //    accepted the full step in the previous iteration (hist == 0; in the steady phase of a run that is nearly every robot, every iteration).
  //  * SUPER-ROUNDS (ls_help = H > 1: a fleet that leaves compute units idle, 64 robots on 256 CUs): the launch carries H blocks per robot,
  //    each on a CU of its own.  In super-round sr block h evaluates candidates 2 (h + H sr) - 1 and 2 (h + H sr) (candidate -1 = E(x): the
  //    primary's, h = 0, in super-round 0), so one super-round decides 2H candidates in the time of ONE team evaluation (3.8 us where a
  //    one-wave evaluation takes 8 - 10): the early iterations of a run, which back off 7 - 14 times on the velocity limit, take two super-rounds
  //    instead of two or three rounds of eight one-wave evaluations, and no robot waits for a history.  Helpers only EVALUATE: each posts its two
  //    energies (agent-scope stores into Dev::ls_tab; all-ones = not there yet) and then waits for the primary's word -- the next super-round
  //    or the end.  The primary walks the candidates in the reference's order (its own two, then the helpers' as they stand in the table), takes
  //    the first that passes and commits alone; it never depends on a helper: a post that is not there 10 us after its own evaluation (a GPU shared
  //    with another process, helpers not resident) sends it to the one-wave rounds below.  The order of stores that makes this safe:
  //    the primary's DONE word is performed (write-through, waited for) BEFORE any of its commit stores is issued, and a helper reads the word
  //    AFTER its staging loads have returned -- a helper that started late either sees DONE and leaves or has staged the state of before the commit.
  //    E(.) is a pure function of its inputs in either shape (bitwise: TJ_LS_FAST=0 / TJ_LS_HELP=1 change no bit), so it does not matter who evaluated what.
  const LsTeamUnits tu = ls_team_units(S, M);
  const bool team_ok = D.ls_fast && G == LS_GROUPS && in_lds && tu.total() * 64 <= 2 * S * 18;
  const bool coop = team_ok && H > 1;
  // A robot with hundreds of planes (next to an obstacle slab; the team shape does not hold its terms) is not a latency chain here but the fp64 issue rate of its compute
  // unit as well: a round of eight candidates takes it 57 us in the early iterations of config 5 (and ~30 later) where a round of TWO takes 29 (measured: ~20 us is one
  // wave's chain of 45 passes, ~4.7 us every further candidate on the unit).  In the steady phase of a run -- every robot of the fleet accepted the full step in the previous
  // iteration -- its FIRST round therefore evaluates only E(x) and the full step, the other waves idle at the barriers; if that fails, full rounds follow.  (Sizing the
  // first round by the robot's own last exponent while the fleet still backs off was measured too: a wrong guess costs what a right one saves.)  Which candidates share a
  // round changes no energy and not the order they are looked at in: same step.
  const bool narrow = !team_ok && G == LS_GROUPS && M >= LS_NARROW_MIN_PLANES && hist == 0 && D.ctl->ls_quiet >= 1 && D.ls_fast;
  int W = G;   // candidates of the current round
  double step = step0; int k_done = 0;                 // step = step0 * 0.8^k_done, kept across the rounds (each thread for the candidates of its team / group)
  bool commit_late = false;
  if (h > 0 && (!coop || D.ls_help_mute)) goto ticket;  // (uniform) a helper of a launch that searches in the one-wave shape: nothing to do  (ls_help_mute: test hook -- helpers
                                                        // that never post; every primary then runs into its 10 us timeout and finishes on its own)
  if (team_ok && (coop || hist == 0)) {
    const int team = tid >> 8, tw = (tid >> 6) & 3, tl = tid & 255;
    double* tnet = sm + L.gnet + (size_t)(4 * team) * 3 * T;
    double* thull = sm + L.ghull + (size_t)(4 * team) * S * 18;
    double* tterms = sm + L.ghull + (size_t)(4 * team + 1) * S * 18;   // the buffers of the team's groups 1 and 2: 2 * S * 18 doubles
    double* tcons = sm + L.gcons + (size_t)(4 * team) * 24 * P;
    const unsigned epoch = (unsigned)D.ctl->epoch;
    unsigned long long* word = D.ls_word + u;
    double* tab = D.ls_tab + (size_t)u * LS_TAB_STRIDE;   // three sets of LS_HELP_MAX x 2 slots: super-round sr uses set sr % 3
    // late-start guard of a helper (see above): the word is read now -- the staging barriers are behind us, every load of ls_stage has returned -- and
    // looked at before the first post (the round trip hides behind the evaluation)
    unsigned long long w_guard = 0;
    asm volatile("" ::: "memory");   // (compiler: the guard load stays behind ls_stage's loads and barriers; hardware: those loads have returned -- their values went through LDS and two s_barriers)
    if (h > 0 && tid == 0) w_guard = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::: "memory");
    for (int sr = 0;; sr++) {
      const int kb = 2 * (h + H * sr) - 1, k = kb + team;   // this block's two candidates of the super-round
      if (kb + 2 * H >= STEP_CAP) {   // (uniform; never in practice) the tail of a search that ends by rounding belongs to the one-wave rounds and their cap
        if (h > 0) break;
        if (tid == 0) { __hip_atomic_store(word, ((unsigned long long)epoch << 32) | LS_WORD_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_waitcnt(0); }
        k_first = kb;
        break;
      }
      for (; k_done < k; k_done++) step *= 0.8;          // same rounding as the reference's repeated step *= 0.8
      const double pt = k < 0 ? t0 : t0 + step * t_dir;
      for (int i = tl; i < 3 * T; i += 256) tnet[i] = k < 0 ? net[i] : net[i] + step * dir[i];
      if (tid < 2) s_bad[tid] = 0;
      if (L.affine) { const double* hn = sm + L.hn; const double* hd = sm + L.hd; for (int idx = tl; idx < S * 18; idx += 256) thull[idx] = k < 0 ? hn[idx] : hn[idx] + step * hd[idx]; }   // (needs nothing of tnet: one barrier for both)
      else { __syncthreads(); for (int idx = tl; idx < S * 18; idx += 256) thull[idx] = ls_hull_entry(D, sm + L.basis, tnet, idx); }
      __syncthreads();
      if (sr == 0) TJ_TIC(D, K_LINESEARCH, 3);
      const bool ahead = h > 0 && sr > 0;   // a helper beyond super-round 0 runs ahead of the primary's decision over the round before (below) and leaves mid-evaluation once the search is over
      x_energy_team(D, sm, L, tnet, pt, thull, tterms, tcons, &s_bad[team], &res[team], M, tw, gl, ahead ? word : nullptr, epoch, &s_abort);
      if (tl == 0) res[LS_GROUPS + team] = step;
      __syncthreads();
      if (ahead && s_abort) break;
      if (sr == 0) TJ_TIC(D, K_LINESEARCH, 4);
      const int set = (sr % 3) * LS_HELP_MAX * 2;
      if (h > 0) {   // helper: post, then wait for the primary's word
        if (sr == 0) {
          if (tid == 0) s_accept = (unsigned)(w_guard >> 32) == epoch && (unsigned)w_guard == LS_WORD_DONE;   // (its own word: s_flag is rewritten below without a barrier in between)
          __syncthreads();
          if (s_accept) break;
        }
        if (tl == 0) __hip_atomic_store(tab + set + 2 * h + team, res[team], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The word permits round sr + 1 as soon as the primary's own two candidates of round sr have failed -- before it has read our posts: that
        // evaluation runs ahead of the decision and is left at its barrier if the search ends meanwhile (x_energy_team: abort_word).
        if (tid == 0) {
          const long long t_end = wall_clock64() + 500000;   // 5 ms: a logic error must not hang the device
          int leave = 0;
          for (;;) {
            const unsigned long long w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == epoch) { if ((unsigned)w == LS_WORD_DONE) { leave = 1; break; } if ((unsigned)w > (unsigned)sr) break; }
            if (wall_clock64() > t_end) { leave = 1; atomicAdd(&D.ctl->ls_helper_timeouts, 1); break; }
            __builtin_amdgcn_s_sleep(1);
          }
          s_flag = leave;
        }
        __syncthreads();
        if (s_flag) break;
        continue;
      }
      // primary: the first candidate that passes, in the reference's order
      if (sr == 0) e_base = res[0];
      if (tid < 64) {
        int acc_k = -1, giveup = H == 1 ? 1 : 0; double acc_st = 0;
        for (int c = sr == 0 ? 1 : 0; c < 2 && acc_k < 0; c++) { const double st = res[LS_GROUPS + c]; if (!(e_base - 1e-4 * wolfe * st < res[c])) { acc_k = kb + c; acc_st = st; } }
        if (sr == 0) TJ_TIC(D, K_LS_COUPLED, 0);
        if (acc_k < 0 && H > 1) {
          // Our own two failed: PERMIT round sr + 1 at once (word = sr + 1) -- the helpers start it while we look at their posts of this round; if one of those
          // passes, they leave mid-evaluation.  (The set round sr + 1 posts into was emptied at the end of round sr - 2: long performed, the wait is free.)
          __builtin_amdgcn_s_waitcnt(0);
          if (tid == 0) __hip_atomic_store(word, ((unsigned long long)epoch << 32) | (unsigned)(sr + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          double v = 0.0;
          const long long t_end = wall_clock64() + 1000;   // 10 us
          for (;;) {
            const bool mine = tid >= 2 && tid < 2 * H;
            if (mine) v = __hip_atomic_load(tab + set + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(mine && (unsigned long long)__double_as_longlong(v) == LS_TAB_EMPTY) == 0ull) break;
            if (wall_clock64() > t_end) { giveup = 1; if (tid == 0) atomicAdd(&D.ctl->ls_giveups, 1); break; }
          }
          if (!giveup) {
            double st = res[LS_GROUPS + 1];
            for (int l = 2; l < 2 * H; l++) {
              st *= 0.8;
              const double e = __shfl(v, l);
              if (acc_k < 0 && !(e_base - 1e-4 * wolfe * st < e)) { acc_k = kb + l; acc_st = st; }
            }
          }
        }
        if (sr == 0) TJ_TIC(D, K_LS_COUPLED, 1);
        if (H > 1) {
          if (acc_k < 0 && !giveup) {   // on to the next super-round: empty the set just read -- it is round sr + 3's (not waited for: the permit of round sr + 2 will)
            if (tid >= 2 && tid < 2 * H) __hip_atomic_store((unsigned long long*)tab + set + tid, LS_TAB_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            if (tid == 0) __hip_atomic_store(word, ((unsigned long long)epoch << 32) | LS_WORD_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (giveup) __builtin_amdgcn_s_waitcnt(0);   // the one-wave rounds commit on their own: the word is performed before they start
          }
        }
        if (tid == 0) { s_accept = acc_k; s_accst = acc_st; s_flag = giveup; }
        if (sr == 0) TJ_TIC(D, K_LS_COUPLED, 2);
      }
      __syncthreads();
      if (sr == 0) TJ_TIC(D, K_LS_COUPLED, 3);
      const int acc = s_accept;
      if (acc >= 0) {
        k_acc = acc; step_acc = s_accst; pt_acc = t0 + step_acc * t_dir; evals = 2 + k_acc;
        if (acc - kb < 2) wg = 4 * (acc - kb);   // the accepting team's buffers hold the trial net (and its hulls)
        else {                                     // a helper's candidate: form its net here (same expression, same bits)
          wg = 4;
          double* wnet = sm + L.gnet + (size_t)4 * 3 * T;
          for (int i = tid; i < 3 * T; i += LS_THREADS) wnet[i] = net[i] + step_acc * dir[i];
          __syncthreads();
          if (!L.affine) { double* whl = sm + L.ghull + (size_t)4 * S * 18; for (int idx = tid; idx < S * 18; idx += LS_THREADS) whl[idx] = ls_hull_entry(D, sm + L.basis, wnet, idx); }
        }
        TJ_TIC(D, K_LS_COUPLED, 4);
        commit_late = true;   // the control net is stored after the hull cache (below): the DONE word has been performed by then
        __syncthreads();
        break;
      }
      if (s_flag) { k_first = kb + 2; break; }   // E(x) is known, the candidates up to kb + 1 have been looked at: one-wave rounds from here
    }
    if (h > 0) goto ticket;
  }
  for (int round = 0; k_acc < 0; round++, k_first += W) {
    W = (narrow && round == 0 && k_first < 0) ? hist + 2 : G;
    const bool act = g < W;
    // candidate of this group: -1 = E(x), otherwise trial index k >= 0
    const int k = k_first + g;
    if (act) for (; k_done < k; k_done++) step *= 0.8;          // same rounding as the reference's repeated step *= 0.8
    const double pt = k < 0 ? t0 : t0 + step * t_dir;
    if (act) for (int i = gl; i < 3 * T; i += LS_GSIZE) gnet[i] = k < 0 ? net[i] : net[i] + step * dir[i];
    __syncthreads();
    if (round == 0) TJ_TIC(D, K_LINESEARCH, 3);
    double e = 0.0;
    if (act) e = x_energy_group(D, u, sm, L, gnet, pt, ghull, sm + L.gcons + (size_t)g * 24 * P, M, in_lds, pref, gl, k >= 0, step);
    else __syncthreads();   // (the function's one block barrier)
    if (round == 0) TJ_TIC(D, K_LINESEARCH, 4);
    if (gl == 0 && !shadow && act) { res[g] = e; res[LS_GROUPS + g] = step; }
    if (tid == 0) s_accept = -1;
    __syncthreads();
    if (k_first < 0) e_base = res[0];
    if (tid == 0) {
      for (int c = (k_first < 0 ? 1 : 0); c < W; c++) {
        const double st = res[LS_GROUPS + c];
        if (!(e_base - 1e-4 * wolfe * st < res[c])) { s_accept = c; break; }
      }
    }
    __syncthreads();
    const int acc = s_accept;
    if (acc >= 0) {
      k_acc = k_first + acc;
      step_acc = res[LS_GROUPS + acc];
      pt_acc = t0 + step_acc * t_dir;
      evals = 2 + k_acc;
      // commit: the accepting group's trial net is the new control net
      wg = acc;
      const double* win = sm + L.gnet + (size_t)acc * 3 * T;
      if (D.fa_seq > 0) commit_late = true;   // asynchronous front: one commit site (below), written through and followed by the robot's flag
      else for (int i = tid; i < 3 * T; i += LS_THREADS) gspline[i] = win[i];
    } else if (W == G && k_first + G - 1 >= STEP_CAP) {
      // no acceptable step although step *= 0.8 has reached its fixed point (every further candidate is this one again): the
      // reference's loop would never end (Optimization3D_multi.h:792).  Take the last candidate and report.
      if (tid == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP);
      k_acc = k_first + G - 1;
      step_acc = res[LS_GROUPS + G - 1];
      pt_acc = t0 + step_acc * t_dir;
      evals = 2 + k_acc;
      wg = G - 1;
      const double* win = sm + L.gnet + (size_t)(G - 1) * 3 * T;
      if (D.fa_seq > 0) commit_late = true;
      else for (int i = tid; i < 3 * T; i += LS_THREADS) gspline[i] = win[i];
    }
    __syncthreads();
  }
  if (D.fuse && D.multi() && !D.fa_units) {   // (Dev::fa_units: the next k_front's units form the records themselves, from the committed control net; written through from here
                                                //  the 40 KB per robot put ~4 us in front of every commit flag -- measured)
    double* wh = sm + L.ghull + (size_t)wg * S * 18;
    if (L.affine) {   // the published hulls are exactly basis * (accepted control net), like k_hullinfo's
      const double* win = sm + L.gnet + (size_t)wg * 3 * T;
      for (int idx = tid; idx < S * 18; idx += LS_THREADS) wh[idx] = ls_hull_entry(D, D.basis, win, idx);
      __syncthreads();
    }
    ls_publish_hullinfo(D, u, wh, tid, LS_THREADS);
  }
  TJ_TIC(D, K_LS_COUPLED, 5);
  if (commit_late) {   // (uniform) accepted in a team round: commit now -- with helpers about, wave 0's DONE word must have been performed before the first of these stores is issued
    if (H > 1) { TJ_MARK_(0x2c1); if (tid < 64) __builtin_amdgcn_s_waitcnt(0); __syncthreads(); TJ_MARK_(0x2c2); }   // (s_waitcnt vmcnt(0): wave 0's DONE store has been acknowledged; the barrier hands that to the other waves; the compiler keeps the commit stores below)
    const double* win = sm + L.gnet + (size_t)wg * 3 * T;
    for (int i = tid; i < 3 * T; i += LS_THREADS) xs_out(D.fa_seq > 0, gspline + i, win[i]);
  }
  if (D.fa_seq > 0) {   // (uniform) asynchronous front: the control net is out (written through) and acknowledged -> the robot's commit flag; k_front's units on the other queue wait for it
    sig_acked();
    __syncthreads();
    asm volatile("" ::: "memory");
    if (tid == 0) __hip_atomic_store(D.fa_commit(u), D.fa_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  }
  TJ_TIC(D, K_LINESEARCH, 5);
  if (tid == 0) { D.piece_time[u] = pt_acc; D.step_out[u] = step_acc; D.ls_hist[u] = k_acc; D.blk_stats[(size_t)D.U * D.P + u] += (unsigned long long)evals; }   // per robot: one writer, no atomic in front of the ticket
  // direct exchange (sharded contexts): the committed control net goes straight into every peer's receive buffer -- the next iteration's k_front
  // there waits for it (inside a batch only: the first iteration of a batch is fed by k_begin, which pushes whatever the state is then)
  if (D.xch && begin_next) xch_push_robot<false>(D, 0, u, 3 * T, sm + L.gnet + (size_t)wg * 3 * T, 3 * T, tid, LS_THREADS);
ticket:
  if (h > 0 && begin_next) sig_acked();   // a helper's posts are performed before its ticket: begin_body's reset of the table cannot be overtaken by them
  if (begin_next) {
    // No fence: nothing another block of THIS kernel writes is read here (gnorm and the counters come from earlier kernels;
    // what begin_body resets was consumed by every block before its ticket), and what is written here is read by later
    // kernels only.  The agent-scope atomic alone orders the tickets.  (An agent-scope release here would write back the
    // XCD's L2 -- ~40 KB of hull cache per block -- from every block: measured +14 us.)
    __shared__ int s_last;
    __syncthreads();
    // the ticket also says whether this robot backed off (bit 16 up): the last block knows it of every robot without reading anything another block wrote
    if (tid == 0) {
      const int add = 1 + ((h == 0 && k_acc != 0) ? 0x10000 : 0), old = atomicAdd(&D.ctl->ticket, add);
      s_last = (old & 0xffff) == nown * H - 1 ? 1 + (((old + add) >> 16) != 0) : 0;
    }
    sig_sent();
    __syncthreads();
    if (s_last) {
      if (tid == 0) D.ctl->ticket = 0;
      begin_body(D, s_last == 1, D.fa_seq > 0);
      // asynchronous front: this launch does not end before the k_front it feeds has -- every block of it has counted itself done behind its acknowledged
      // (write-through) stores; k_mid, next on this queue, then finds everything in memory
      // (Dev::fa_mid: k_mid waits for k_front's end itself; here only until every k_front block has STARTED, i.e. is resident -- k_mid's waves cannot shut one out then)
      if (D.fa_seq > 0 && tid < 64) fa_wait16(D, D.fa_mid ? D.fa_fstart(0) : D.fa_fdone(0), (int)((unsigned)D.fa_seq * (unsigned)D.fa_nfront));
    }
  }
  TJ_TIC(D, K_LINESEARCH, 6);
}

// Energy_admm::spline_energy (Energy_admm.h:16-44) of the CURRENT state against the plane lists of the last iteration: what the
// mains could print next to gnorm.  Same staging and the same evaluation the line search performs for E(x); all eight groups
// evaluate it (the function's barrier is block wide), group 0 reports.
__global__ __launch_bounds__(LS_THREADS) void k_energy(Dev D, LsLayout L, double* out) {
  extern __shared__ double sm[];
  __shared__ int pref[1024];
  const int tid = threadIdx.x, u = D.u0 + blockIdx.x, S = D.S, T = D.T, P = D.P;
  const int G = L.groups;
  const int g = min(tid / LS_GSIZE, G - 1), gl = tid % LS_GSIZE;
  bool in_lds;
  const int M = ls_stage(D, L, sm, pref, u, tid, LS_THREADS, in_lds);
  double* net = sm + L.net;
  double* gnet = sm + L.gnet + (size_t)g * 3 * T;
  for (int i = gl; i < 3 * T; i += LS_GSIZE) gnet[i] = net[i];
  __syncthreads();
  const double e = x_energy_group(D, u, sm, L, gnet, D.piece_time[u], sm + L.ghull + (size_t)g * S * 18, sm + L.gcons + (size_t)g * 24 * P, M, in_lds, pref, gl, 0, 0.0);
  if (tid == 0) out[u] = e;
}

// ---- coupled mode ("decouple":0): Armijo search on the SUM of all robots' energies ---------------
// Optimization3D_multi::update_spline (Optimization3D_multi.h:587-636).  One step and one piece_time for
// every robot, accepted when e0 - 1e-4*wolfe*step >= sum_u E_u(x_u + step d_u, t + step t_dir).  The sum
// couples all robots, so a launch only EVALUATES: round r has every robot's block compute its energy at 8
// candidates (round 0: E(x) and steps 0.8^0..0.8^6, later rounds 8 more steps each) into ls_e; the next
// launch (or k_ls_commit) first forms the totals in robot order -- every block computes the same bits --
// and returns at once when an earlier round already holds the accepted step.
__device__ __forceinline__ int lsc_cand_k(int round, int c) { return round == 0 ? c - 1 : (LS_GROUPS - 1) + (round - 1) * LS_GROUPS + c; }

// step after the CCD clamps and the t > 0 guard (Optimization3D_multi.h:586-601); wave 0 computes, all threads get it
__device__ __forceinline__ double lsc_step0(const Dev& D, int tid, double t0, double t_dir, double* s_val) {
  if (tid < 64) {
    int kmax = D.k_self[0];
    const bool sharded = D.u1 - D.u0 != D.U;   // foreign robots' exponents arrive through exchange buffer 3
    for (int r = tid; r < D.U; r += 64) kmax = max(kmax, sharded ? (int)D.k_obs_f[r] : D.k_obs[r]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off));
    double step0 = D.pow08[min(STEP_CAP, kmax)];
    if (t0 + step0 * t_dir <= 0) step0 = -0.95 * t0 / t_dir;
    if (tid == 0) *s_val = step0;
  }
  __syncthreads();
  return *s_val;
}

// wave 0 only: first acceptable candidate among rounds [0, nrounds), in the reference's order.  acc[0] = round
// (-1 none), acc[1] = slot, accstep = its step.  stage: LDS scratch of 64 * LSC_ROUNDS * LS_GROUPS doubles.
// The energies of 64 robots (all rounds: they are contiguous per robot) come in with ONE coalesced pass per chunk, then lane
// (round, candidate) adds its column in robot order out of LDS -- the sum the reference forms (e += spline_energy(i)); one
// lane per candidate walking the robots through global memory was 64 dependent round trips per round, in every block of four
// launches per iteration.
// base (sharded contexts that follow the search, Dev::lsc_follow): the table holds the rounds [base, base + LSC_ROUNDS) -- candidates 0.8^(8 base - 1) and smaller; E(x) is
// the one formed with the first table (Ctl::lsc_e0).  acc[0] = absolute round.
template <bool FRESH = false>   // FRESH: the table is being written by other blocks of THIS launch (agent-scope stores): read it past the caches
__device__ __forceinline__ void lsc_decide(const Dev& D, int nrounds, double step0, int lane, int* acc, double* accstep, double* stage, int base = 0, double* e0_out = nullptr) {
  const double wolfe = D.ctl->wolfe_c;
  constexpr int RC = LSC_ROUNDS * LS_GROUPS;   // 32 columns
  const int r_l = lane / LS_GROUPS, c = lane % LS_GROUPS;
  double tot = 0;
  for (int u0 = 0; u0 < D.U; u0 += 64) {
    const int nu = min(64, D.U - u0), nval = nu * RC;
    blk_sync<true>();
    for (int i = lane; i < nval; i += 64) stage[i] = FRESH ? __hip_atomic_load(&D.ls_e[(size_t)u0 * RC + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : D.ls_e[(size_t)u0 * RC + i];
    blk_sync<true>();
    if (lane < RC) for (int j = 0; j < nu; j++) tot += stage[j * RC + lane];
  }
  double step = step0;
  if (lane < RC) { const int k = lsc_cand_k(base + r_l, c); for (int i = 0; i < k; i++) step *= 0.8; }   // same rounding as the reference's repeated step *= 0.8
  const double e0 = base == 0 ? __shfl(tot, 0) : D.ctl->lsc_e0;
  if (e0_out && lane == 0) *e0_out = e0;
  int found_r = -1, found_c = 0; double found_step = step0;
  for (int r = 0; r < nrounds && found_r < 0; r++) {
    const bool ok = r_l == r && lane < RC && !(base == 0 && r == 0 && c == 0) && !(e0 - 1e-4 * wolfe * step < tot);
    const unsigned long long mask = __ballot(ok);
    if (mask) { const int l = __ffsll((long long)mask) - 1; found_r = base + r; found_c = l - r * LS_GROUPS; found_step = __shfl(step, l); }
  }
  if (lane == 0) { acc[0] = found_r; acc[1] = found_c; *accstep = found_step; }
}

// The search BEYOND the rounds the launches evaluated (steps 0.8^31 and smaller): the reference's loop (Optimization3D_multi.h:623) has no bound, it ends by rounding
// like the decoupled one.  Met in no ordinary iteration (nine scenes, 30 iterations each: at most 24 back-offs) -- so it is the job of ONE block, which walks the
// robots one after the other (stage, evaluate eight candidates, add to the eight totals in robot order: the reference's e += spline_energy(i)), round after round,
// until a candidate passes or the step reaches the fixed point of `step *= 0.8`.  Slow (a round costs ~13 us per robot) and exact.  One context only: a sharded
// context decides on the gathered four-round table and still reports ERR_LS_RANGE beyond it.
__device__ __forceinline__ void lsc_continue(const Dev& D, const LsLayout& L, double* sm, int* pref, double step0, int tid, int* acc, double* accstep) {
  __shared__ double s_tot[LS_GROUPS], s_e0;
  __shared__ int s_hit;
  const int S = D.S, T = D.T, P = D.P, G = L.groups;
  const int g = min(tid / LS_GSIZE, G - 1), gl = tid % LS_GSIZE;
  const bool shadow = tid / LS_GSIZE >= G;
  double* net = sm + L.net; double* dir = sm + L.dir;
  double* gnet = sm + L.gnet + (size_t)g * 3 * T;
  double* ghull = sm + L.ghull + (size_t)g * S * 18;
  const double wolfe = D.ctl->wolfe_c;
  if (tid == 0) {   // E(x) summed in robot order: column (round 0, slot 0) of the table (written by other blocks of this launch with agent-scope stores)
    double e0 = 0;
    for (int u = 0; u < D.U; u++) e0 += __hip_atomic_load(&D.ls_e[((size_t)u * LSC_ROUNDS + 0) * LS_GROUPS + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_e0 = e0;
  }
  int k_first = lsc_cand_k(LSC_ROUNDS - 1, LS_GROUPS - 1) + 1;   // 31
  double step_first = step0;
  for (int i = 0; i < k_first; i++) step_first *= 0.8;            // same rounding as the reference's repeated step *= 0.8
  for (int round = LSC_ROUNDS;; round++, k_first += LS_GROUPS) {
    if (tid < LS_GROUPS) s_tot[tid] = 0.0;
    __syncthreads();
    for (int u = D.u0; u < D.u1; u++) {
      bool in_lds;
      const int M = ls_stage(D, L, sm, pref, u, tid, LS_THREADS, in_lds);
      const double t_dir = D.tdir(u), t0 = D.piece_time[u];
      for (int pass = 0; pass < LS_GROUPS / G; pass++) {
        const int slot = pass * G + g;
        double step = step_first;
        for (int i = 0; i < slot; i++) step *= 0.8;
        const double pt = t0 + step * t_dir;
        __syncthreads();
        for (int i = gl; i < 3 * T; i += LS_GSIZE) gnet[i] = net[i] + step * dir[i];
        __syncthreads();
        const double e = x_energy_group(D, u, sm, L, gnet, pt, ghull, sm + L.gcons + (size_t)g * 24 * P, M, in_lds, pref, gl, 1, step);
        if (gl == 0 && !shadow) s_tot[slot] += e;   // one writer per slot, robots in order
      }
      __syncthreads();
    }
    if (tid == 0) {
      int hit = -1; double st = step_first, hst = 0;
      for (int c = 0; c < LS_GROUPS; c++) { if (hit < 0 && !(s_e0 - 1e-4 * wolfe * st < s_tot[c])) { hit = c; hst = st; } st *= 0.8; }
      if (hit < 0 && k_first + LS_GROUPS - 1 >= STEP_CAP) {   // the fixed point of step *= 0.8: the reference's loop would never end -- take the last candidate and report
        atomicOr(&D.ctl->error, ERR_LOOP_CAP);
        hit = LS_GROUPS - 1; hst = step_first; for (int i = 0; i < hit; i++) hst *= 0.8;
      }
      s_hit = hit;
      if (hit >= 0) { acc[0] = round; acc[1] = hit; *accstep = hst; }
    }
    __syncthreads();
    if (s_hit >= 0) return;
    for (int i = 0; i < LS_GROUPS; i++) step_first *= 0.8;
  }
}

// wide (round 4): rounds covered by THIS launch.  A fleet that leaves compute units idle (owned robots x LSC_ROUNDS blocks fit the device) evaluates all four rounds
// at once, block (robot, h) takes round h: one launch and one decision instead of four launches of which the later ones mostly return at once (three kernel
// boundaries, ~8 us each with their staging).  Same energies into the same table, same decision.
// begin_next (wide launch of one context, inside a batch): the committing block also starts the NEXT iteration (begin_body), like k_linesearch's last block does
// in the decoupled chain -- the host then omits k_begin.
// base (sharded context, the caller follows the search: Dev::lsc_follow): the launch evaluates the rounds [base + round0, ...) into the table's rows [round0, ...)
__global__ __launch_bounds__(LS_THREADS) void k_ls_coupled(Dev D, LsLayout L, int round0, int wide, int begin_next, int base = 0) {
  if (D.fa_seq > 0) {   // asynchronous front (one context, all rounds in this launch): as in k_linesearch -- the early begin by the last block of the grid, then every block counts itself started
    if (blockIdx.x == gridDim.x - 1) fa_early_begin(D);
    if (threadIdx.x == 0) __hip_atomic_fetch_add(D.fa_res(blockIdx.x), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  }
  if (TJ_DONE(D)) {
    if (begin_next && blockIdx.x == 0 && threadIdx.x == 0) { D.ctl->slack_now = D.ctl->slack_next; D.ctl->slack_next = 0; }   // (as k_linesearch: retire the slack update k_mid has just paid)
    return;
  }
  extern __shared__ double sm[];
  __shared__ int pref[1024];
  __shared__ int s_acc[2];
  __shared__ double s_step0, s_accstep;
  const int nown = D.u1 - D.u0, hb = wide > 1 ? (int)blockIdx.x / nown : 0, round = base + round0 + hb;
  const int tid = threadIdx.x, u = D.u0 + (int)blockIdx.x - hb * nown, S = D.S, T = D.T, P = D.P;
  // G = L.groups candidates side by side (8 up to piece_num = 10; 4 or 2 for long trajectories, whose hull buffers are larger): a
  // round's LS_GROUPS candidates then take LS_GROUPS / G passes.  Waves beyond G shadow the last group (same values into the same
  // buffers), which keeps every barrier uniform.
  const int G = L.groups;
  const int g = min(tid / LS_GSIZE, G - 1), gl = tid % LS_GSIZE;
  const bool shadow = tid / LS_GSIZE >= G;
  // the early exit needs every robot's energies of the earlier rounds: a sharded context (u1 - u0 < U) only has its own until
  // the all-gather after the last round, so it evaluates every round (same decision, taken by k_ls_commit on the gathered table)
  // One context: the LAST block of a round to finish takes the decision over the rounds evaluated so far and leaves it in the
  // control block; the launches of the later rounds (and the commit) read one word instead of every block re-deriving it.
  const bool one_ctx = D.u1 - D.u0 == D.U;
  const int epoch = D.ctl->epoch;
  if (round0 > 0 && one_ctx && D.ctl->lsf_epoch == epoch) return;  // an earlier launch already holds the accepted step
  const double t_dir = D.tdir(u), t0 = D.piece_time[u];
  const double step0 = lsc_step0(D, tid, t0, t_dir, &s_step0);
  bool in_lds;
  const int M = ls_stage(D, L, sm, pref, u, tid, LS_THREADS, in_lds);
  double* net = sm + L.net; double* dir = sm + L.dir;
  double* gnet = sm + L.gnet + (size_t)g * 3 * T;
  double* ghull = sm + L.ghull + (size_t)g * S * 18;
  for (int pass = 0; pass < LS_GROUPS / G; pass++) {
    const int slot = pass * G + g;                      // this group's candidate of the round
    const int k = lsc_cand_k(round, slot);
    double step = step0;
    for (int i = 0; i < k; i++) step *= 0.8;
    const double pt = k < 0 ? t0 : t0 + step * t_dir;
    __syncthreads();                                    // the previous pass is through with the group buffers
    for (int i = gl; i < 3 * T; i += LS_GSIZE) gnet[i] = k < 0 ? net[i] : net[i] + step * dir[i];
    __syncthreads();
    const double e = x_energy_group(D, u, sm, L, gnet, pt, ghull, sm + L.gcons + (size_t)g * 24 * P, M, in_lds, pref, gl, k >= 0, step);
    if (gl == 0 && !shadow) {
      double* dst = &D.ls_e[((size_t)u * LSC_ROUNDS + (round - base)) * LS_GROUPS + slot];
      if (one_ctx) __hip_atomic_store(dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *dst = e;   // one context: write-through, read by the deciding block of THIS launch
    }
  }
  if (!one_ctx) return;
  // the stores were performed before this wave arrives at the barrier; then the block's ticket
  __builtin_amdgcn_s_waitcnt(0);
  __shared__ int s_last;
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&D.ctl->ls_ticket, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) D.ctl->ls_ticket = 0;
  if (tid < 64) lsc_decide<true>(D, round0 + wide, step0, tid, s_acc, &s_accstep, sm);   // sm: the evaluation is over
  __syncthreads();
  // none of the 31 steps of the evaluated rounds passes (never met outside constructed states): this block goes on alone, to the reference's own end
  if (s_acc[0] < 0 && round0 + wide >= LSC_ROUNDS) { lsc_continue(D, L, sm, pref, step0, tid, s_acc, &s_accstep); __syncthreads(); }
  if (tid == 0 && s_acc[0] >= 0) { D.ctl->lsf_r = s_acc[0]; D.ctl->lsf_c = s_acc[1]; D.ctl->lsf_step = s_accstep; D.ctl->lsf_epoch = epoch; }   // read by LATER kernels only
  if (wide < LSC_ROUNDS) return;
  // Every round has been evaluated and every other block is through (the ticket): this block also COMMITS -- k_ls_commit's work for all robots, 3T values
  // each; the host omits that launch.  Same expressions as k_ls_commit.
  __syncthreads();
  const double step = s_accstep;
  const int kacc = lsc_cand_k(s_acc[0], s_acc[1]);   // (lsc_continue always leaves a decision)
  for (int idx = tid; idx < nown * 3 * T; idx += LS_THREADS) {
    const int uu = D.u0 + idx / (3 * T), i = idx % (3 * T);
    double* gs = D.spline + (size_t)uu * 3 * T;
    xs_out(D.fa_seq > 0, gs + i, gs[i] + step * D.dirp(uu)[i]);
  }
  if (D.fa_seq > 0) {   // (uniform) asynchronous front: every robot's control net is out (written through) and acknowledged -> every robot's commit flag
    sig_acked();
    __syncthreads();
    asm volatile("" ::: "memory");
    for (int uu = D.u0 + tid; uu < D.u1; uu += LS_THREADS) __hip_atomic_store(D.fa_commit(uu), D.fa_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sig_sent();
  }
  for (int uu = D.u0 + tid; uu < D.u1; uu += LS_THREADS) {
    D.piece_time[uu] = D.piece_time[uu] + step * D.tdir(uu); D.step_out[uu] = step; D.blk_stats[(size_t)D.U * D.P + uu] += (unsigned long long)(2 + kacc);
  }
  if (begin_next) {
    __syncthreads();
    begin_body(D, 0, D.fa_seq > 0);
    if (D.fa_seq > 0 && tid < 64) fa_wait16(D, D.fa_mid ? D.fa_fstart(0) : D.fa_fdone(0), (int)((unsigned)D.fa_seq * (unsigned)D.fa_nfront));   // (as k_linesearch's last block)
  }
}

// commit: x_u += step d_u for every robot, the shared piece_time advances by step * t_direction
// base: see k_ls_coupled.  A sharded context whose caller follows the search (Dev::lsc_follow) commits NOTHING when none of the gathered candidates passes: it leaves
// Ctl::lsc_pending = 1 (every rank takes the same decision on the same gathered table) and the caller evaluates, gathers and decides the next rounds
// (tj_coupled_search_pending) -- on to the reference's own end, the fixed point of step *= 0.8 (Optimization3D_multi.h:623).
__global__ __launch_bounds__(64) void k_ls_commit(Dev D, int base = 0) {
  if (TJ_DONE(D)) return;
  __shared__ int s_acc[2];
  __shared__ double s_step0, s_accstep;
  const int tid = threadIdx.x, u = D.u0 + blockIdx.x, T = D.T;
  const double t_dir = D.tdir(u), t0 = D.piece_time[u];
  const double step0 = lsc_step0(D, tid, t0, t_dir, &s_step0);
  __shared__ double s_stage[64 * LSC_ROUNDS * LS_GROUPS];
  if (D.u1 - D.u0 == D.U) {   // one context: the last block of the deciding round left the decision (or none was acceptable)
    if (tid == 0) { const bool f = D.ctl->lsf_epoch == D.ctl->epoch; s_acc[0] = f ? D.ctl->lsf_r : -1; s_acc[1] = f ? D.ctl->lsf_c : 0; s_accstep = f ? D.ctl->lsf_step : step0; }
  } else {
    __shared__ double s_e0;
    lsc_decide(D, LSC_ROUNDS, step0, tid, s_acc, &s_accstep, s_stage, base, &s_e0);
    if (tid == 0 && blockIdx.x == 0 && base == 0) D.ctl->lsc_e0 = s_e0;   // (read by the launches of a continued search only: later kernels)
  }
  __syncthreads();
  double step = s_accstep;
  int kacc = s_acc[0] >= 0 ? lsc_cand_k(s_acc[0], s_acc[1]) : lsc_cand_k(base + LSC_ROUNDS - 1, LS_GROUPS - 1);
  if (tid == 0 && blockIdx.x == 0 && D.lsc_follow) D.ctl->lsc_pending = 0;
  if (s_acc[0] < 0) {  // no acceptable step within the evaluated range
    const bool at_end = kacc >= STEP_CAP;   // the fixed point of step *= 0.8 lies inside this table: the reference's loop would never end -- take the last candidate and report (like lsc_continue)
    if (D.lsc_follow && !at_end) {          // the caller goes on with the next rounds: nothing is committed
      if (tid == 0 && blockIdx.x == 0) D.ctl->lsc_pending = 1;
      return;
    }
    step = step0;
    for (int i = 0; i < kacc; i++) step *= 0.8;
    if (tid == 0 && blockIdx.x == 0) atomicOr(&D.ctl->error, at_end ? ERR_LOOP_CAP : (ERR_LOOP_CAP | ERR_LS_RANGE));
  }
  double* gspline = D.spline + (size_t)u * 3 * T;
  const double* dir = D.dirp(u);
  for (int i = tid; i < 3 * T; i += 64) gspline[i] = gspline[i] + step * dir[i];
  if (tid == 0) { D.piece_time[u] = t0 + step * t_dir; D.step_out[u] = step; D.blk_stats[(size_t)D.U * D.P + u] += (unsigned long long)(2 + kacc); }
}

}  // namespace tj
