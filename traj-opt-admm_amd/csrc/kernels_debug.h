// kernels_debug.h -- known-answer entry points: run the SAME device functions the hot-path kernels
// use (gjk, plane_obstacle, plane_pair, k-DOP tests, swept-hull CCD predicates, chol / min-eig) on
// caller-supplied batches, one case per lane (or per workgroup for the LDS linear algebra), so the
// parity tests can compare them bit-for-bit against vectors produced by the reference
// (tests/golden/gjk_kat.npz, prims_kat.npz).
#pragma once
#include "dev_common.h"
#include "dev_linalg.h"
#include "kernels_sep.h"
#include "dev_optplane.h"

namespace tj {

template <int K>
struct BodyPts {  // K explicit points, row-major [K][3]
  const double* p;
  static constexpr int N = K;
  __device__ __forceinline__ V3 get(int i) const { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
};

template <int K1, int K2>
__global__ __launch_bounds__(64) void k_dbg_gjk(int n, const double* a, const double* b, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 v = gjk(BodyPts<K1>{a + (size_t)i * 3 * K1}, BodyPts<K2>{b + (size_t)i * 3 * K2});
  out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
}

// wave-cooperative variant (gjk_wave): one case per workgroup = one wavefront
template <int K1, int K2>
__global__ __launch_bounds__(64) void k_dbg_gjk_wave(int n, const double* a, const double* b, double* out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const V3 v = gjk_wave(BodyPts<K1>{a + (size_t)i * 3 * K1}, BodyPts<K2>{b + (size_t)i * 3 * K2}, lane_id());
  if (threadIdx.x == 0) { out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z; }
}
// gjk_wave interrupted after k_stop iterations, its state taken through memory (20 doubles + 8 ints per case in `scratch`, the
// layout of the head start: kernels_pairs.h spec_pair_body) and the loop continued from there: out = witness vector, iterations
template <int K1, int K2>
__global__ __launch_bounds__(64) void k_dbg_gjk_wave_split(int n, const double* a, const double* b, int k_stop, double* scratch, double* out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const int lane = lane_id();
  const BodyPts<K1> ba{a + (size_t)i * 3 * K1}; const BodyPts<K2> bb{b + (size_t)i * 3 * K2};
  GjkState st; bool fin;
  V3 v = gjk_wave_run(ba, bb, lane, st, true, k_stop, fin);
  double* o = scratch + (size_t)i * 32;
  if (lane == 0) {
    o[0] = st.v.x; o[1] = st.v.y; o[2] = st.v.z;
    o[3] = st.s.v0.x; o[4] = st.s.v0.y; o[5] = st.s.v0.z; o[6] = st.s.v1.x; o[7] = st.s.v1.y; o[8] = st.s.v1.z;
    o[9] = st.s.v2.x; o[10] = st.s.v2.y; o[11] = st.s.v2.z; o[12] = st.s.v3.x; o[13] = st.s.v3.y; o[14] = st.s.v3.z;
    o[15] = st.s.l0; o[16] = st.s.l1; o[17] = st.s.l2; o[18] = st.s.l3; o[19] = st.wmax2;
    int* oi = (int*)(o + 20);
    oi[0] = st.s.n; oi[1] = st.s.w0; oi[2] = st.s.w1; oi[3] = st.s.w2; oi[4] = st.s.w3; oi[5] = st.c1; oi[6] = st.c2; oi[7] = st.k;
  }
  __threadfence();
  __syncthreads();
  if (!fin) {
    GjkState r;
    const int* oi = (const int*)(o + 20);
    r.v = V3{o[0], o[1], o[2]};
    r.s.v0 = V3{o[3], o[4], o[5]}; r.s.v1 = V3{o[6], o[7], o[8]}; r.s.v2 = V3{o[9], o[10], o[11]}; r.s.v3 = V3{o[12], o[13], o[14]};
    r.s.l0 = o[15]; r.s.l1 = o[16]; r.s.l2 = o[17]; r.s.l3 = o[18]; r.wmax2 = o[19];
    r.s.n = oi[0]; r.s.w0 = oi[1]; r.s.w1 = oi[2]; r.s.w2 = oi[3]; r.s.w3 = oi[4]; r.c1 = oi[5]; r.c2 = oi[6]; r.k = oi[7];
    bool fin2;
    v = gjk_wave_run(ba, bb, lane, r, false, 50, fin2);
    st.k = r.k;
  }
  if (threadIdx.x == 0) { out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = (double)st.k; }
}
// plane_pair_wave: one robot pair per wavefront (the form k_sep_self_solve uses)
__global__ __launch_bounds__(64) void k_dbg_pair_wave(Dev D, int n, const double* P, const double* Q, double dist, double* out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  __shared__ double A[18], B[18];
  if (threadIdx.x < 18) { A[threadIdx.x] = P[(size_t)i * 18 + threadIdx.x]; B[threadIdx.x] = Q[(size_t)i * 18 + threadIdx.x]; }
  __syncthreads();
  double e0 = 0, e1 = 0, e2 = 0, dpl = 0; bool capped;
  const bool ok = plane_pair_wave(A, B, dist, D.margin, D.offset, lane_id(), e0, e1, e2, dpl, capped);
  if (threadIdx.x == 0) { double* o = out + (size_t)i * 5; o[0] = ok; o[1] = e0; o[2] = e1; o[3] = e2; o[4] = dpl; }
}

// what: 0 plane_obstacle(P,q) -> out[5] = ok,c,d ; 1 plane_pair(P,Q) with Newton refine -> ok,c,d ;
//       2 k-DOP hull/point ; 3 k-DOP hull/hull   (out[0] = pass)
__global__ __launch_bounds__(64) void k_dbg_planes(Dev D, int what, int n, const double* P, const double* Q, double dist, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = P + (size_t)i * 18;
  double* o = out + (size_t)i * 5;
  if (what == 0) {
    const V3 q{Q[3 * i], Q[3 * i + 1], Q[3 * i + 2]};
    double c0 = 0, c1 = 0, c2 = 0, dd = 0;
    const bool ok = plane_obstacle(p, q, dist, D.offset, c0, c1, c2, dd);
    o[0] = ok; o[1] = c0; o[2] = c1; o[3] = c2; o[4] = dd;
  } else if (what == 1) {
    double e0 = 0, e1 = 0, e2 = 0, dpl = 0; bool capped;
    const bool ok = plane_pair(p, Q + (size_t)i * 18, dist, D.margin, D.offset, true, e0, e1, e2, dpl, capped);
    o[0] = ok; o[1] = e0; o[2] = e1; o[3] = e2; o[4] = dpl;
  } else if (what == 2) {
    double klo[49], khi[49];
    for (int k = 0; k < 49; k++) {
      const double x = D.kdop[3 * k], y = D.kdop[3 * k + 1], z = D.kdop[3 * k + 2];
      double up = -INFINITY, lo = INFINITY;
      for (int j = 0; j < 6; j++) { const double lv = x * p[3 * j] + y * p[3 * j + 1] + z * p[3 * j + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
      klo[k] = lo; khi[k] = up;
    }
    o[0] = kdop_point_pass(D, klo, khi, V3{Q[3 * i], Q[3 * i + 1], Q[3 * i + 2]}, dist);
  } else if (what == 3) {
    o[0] = kdop_hulls_pass(D, p, Q + (size_t)i * 18, dist);
  } else if (what == 5) {  // Optimal_plane::optimal_cd: o[1..4] = (c, d) in/out, o[0] = finished within the iteration caps
    double cx = o[1], cy = o[2], cz = o[3], d = o[4];
    o[0] = opt_plane_obstacle(p, Q[3 * i], Q[3 * i + 1], Q[3 * i + 2], D.margin, D.offset, cx, cy, cz, d);
    o[1] = cx; o[2] = cy; o[3] = cz; o[4] = d;
  } else {                 // Optimal_plane::self_optimal_cd
    double cx = o[1], cy = o[2], cz = o[3], d = o[4];
    o[0] = opt_plane_pair(p, Q + (size_t)i * 18, D.margin, D.offset, cx, cy, cz, d);
    o[1] = cx; o[2] = cy; o[3] = cz; o[4] = d;
  }
}

// Optimal_plane::self_optimal_cd by one wavefront per plane (opt_plane_pair_wave, the form k_keep uses for short lists)
__global__ __launch_bounds__(64) void k_dbg_optpair_wave(Dev D, int n, const double* P, const double* Q, double* out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  double* o = out + (size_t)i * 5;
  double cx = o[1], cy = o[2], cz = o[3], d = o[4];
  __shared__ double te64[64];
  const bool ok = opt_plane_pair_wave(P + (size_t)i * 18, Q + (size_t)i * 18, D.margin, D.offset, lane_id(), cx, cy, cz, d, te64);
  if (lane_id() == 0) { o[0] = ok; o[1] = cx; o[2] = cy; o[3] = cz; o[4] = d; }
}

// swept-hull CCD predicates at steps (t1,u1): out[0] = GJKCCD(P,D,q), out[1] = SelfGJKCCD(P,D,Q,E)
__global__ __launch_bounds__(64) void k_dbg_ccd(int n, const double* P, const double* Dd, const double* Q, const double* E, const double* q, const double* tu, double d, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double t1 = tu[2 * i], u1 = tu[2 * i + 1];
  const V3 v = gjk(BodySwept{P + (size_t)i * 18, Dd + (size_t)i * 18, t1}, BodyPoint{V3{q[3 * i], q[3 * i + 1], q[3 * i + 2]}});
  out[2 * i] = (v.x * v.x + v.y * v.y + v.z * v.z <= d * d);
  const V3 w = gjk(BodySwept{P + (size_t)i * 18, Dd + (size_t)i * 18, t1}, BodySwept{Q + (size_t)i * 18, E + (size_t)i * 18, u1});
  out[2 * i + 1] = (w.x * w.x + w.y * w.y + w.z * w.z <= d * d);
}

// triangle obstacle bodies (BASELINE config 5), one case per lane: out[i][8] = plane ok, c, d (hull P vs triangle, distance
// dist) | k-DOP pass hull/triangle at dist | k-DOP pass swept hull {P, P + t D}/triangle at off | GJK CCD hit at off
__global__ __launch_bounds__(64) void k_dbg_tri(Dev D, int n, const double* P, const double* Dd, const double* tri, const double* t, double dist, double off, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = P + (size_t)i * 18; const double* dd = Dd + (size_t)i * 18; const double* q = tri + (size_t)i * 9;
  const BodyTri tb{V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]}, V3{q[6], q[7], q[8]}};
  double* o = out + (size_t)i * 8;
  double c0 = 0, c1 = 0, c2 = 0, pd = 0;
  const bool ok = plane_from_witness_body(gjk(BodyHull{p}, tb), tb, dist, D.offset, c0, c1, c2, pd);
  o[0] = ok; o[1] = c0; o[2] = c1; o[3] = c2; o[4] = pd;
  double klo[49], khi[49];
  for (int pass = 0; pass < 2; pass++) {
    for (int k = 0; k < 49; k++) {
      const double x = D.kdop[3 * k], y = D.kdop[3 * k + 1], z = D.kdop[3 * k + 2];
      double up = -INFINITY, lo = INFINITY;
      for (int j = 0; j < 6; j++) { const double lv = x * p[3 * j] + y * p[3 * j + 1] + z * p[3 * j + 2]; if (lv < lo) lo = lv; if (lv > up) up = lv; }
      if (pass) for (int j = 0; j < 6; j++) {
        const double lv = x * (p[3 * j] + t[i] * dd[3 * j]) + y * (p[3 * j + 1] + t[i] * dd[3 * j + 1]) + z * (p[3 * j + 2] + t[i] * dd[3 * j + 2]);
        if (lv < lo) lo = lv; if (lv > up) up = lv;
      }
      klo[k] = lo; khi[k] = up;
    }
    o[5 + pass] = kdop_body_pass(D.kdop, klo, khi, tb, pass ? off : dist);
  }
  const V3 v = gjk(BodySwept{p, dd, t[i]}, tb);
  o[7] = (v.x * v.x + v.y * v.y + v.z * v.z <= off * off);
}

// broad-phase known answers: one wavefront per caller-supplied query box, raw candidate list (sorted primitive indices,
// traversal order) of aabb::Tree::query(box, margin) (AABB.cc:608-667) on the static BVH
template <int PRIM>
__global__ __launch_bounds__(64) void k_dbg_query(Dev D, int nq, const double* boxes, double m, int cap, int* out_ids, int* out_n) {
  const int b = blockIdx.x;
  if (b >= nq) return;
  __shared__ int fa[FRONT_CAP], fb[FRONT_CAP], cand[128];
  QBox q;
  for (int k = 0; k < 3; k++) { q.lo[k] = boxes[6 * (size_t)b + k]; q.hi[k] = boxes[6 * (size_t)b + 3 + k]; }
  int base = 0;
  unsigned long long visits = 0;
  bvh_query<4, PRIM>(D, q, m, fa, fb, cand, &visits, [&](int pt) {
    const bool ok = pt >= 0;
    const unsigned long long mask = ballot(ok);
    const int idx = base + prefix_count(mask);
    if (ok && idx < cap) out_ids[(size_t)b * cap + idx] = pt;
    base += __popcll(mask);
  });
  if (lane_id() == 0) out_n[b] = base;
}

// one workgroup per matrix: out[2*i] = LLT fails, out[2*i+1] = smallest eigenvalue
__global__ __launch_bounds__(64) void k_dbg_linalg(int nmat, int n, const double* mats, double* out) {
  extern __shared__ double sm[];
  double* A = sm; double* W = A + n * n; double* scr = W + n * n;
  const int tid = threadIdx.x;
  const double* src = mats + (size_t)blockIdx.x * n * n;
  for (int i = tid; i < n * n; i += 64) { A[i] = src[i]; W[i] = src[i]; }
  __syncthreads();
  bool ok = chol_lds(W, n, tid, 64);
  __syncthreads();
  bool agree = true;
  double r[19];
  const int row = min(tid, 18);
  if (n == 19) {  // the register variants k_grad uses must give the same verdict / the same bits
#pragma unroll
    for (int c = 0; c < 19; c++) r[c] = A[row * 19 + c];
    agree = chol_check_wave<19>(r) == ok;
#pragma unroll
    for (int c = 0; c < 19; c++) r[c] = A[row * 19 + c];
  }
  const double ev = min_eig_lds(A, n, scr, scr + n, scr + 2 * n, scr + 3 * n, tid, 64);
  if (n == 19) {   // the register copy fuses its products: same algorithm, agreement to rounding level of the matrix norm
    double nrm = 0;
    for (int i = 0; i < 361; i++) nrm = fmax(nrm, fabs(src[i]));
    agree = agree && fabs(min_eig_wave<19>(r, tid) - ev) <= 1e-13 * fmax(nrm, 1e-300);
  }
  if (tid == 0) { out[2 * blockIdx.x] = !agree ? 2.0 : ok ? 0.0 : 1.0; out[2 * blockIdx.x + 1] = ev; }
}

}  // namespace tj
