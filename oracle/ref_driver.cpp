// oracle/ref_driver.cpp  --  TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI glue around the *unmodified* reference sources that live under
// /root/reference.  It is compiled by oracle/Makefile (target `ref`) straight
// from those sources into oracle/_ref/libref.so; no reference source is copied
// into this repository.  The reference `main()`s cannot be compiled here (they
// include OMPL and libigl/GLFW headers that the image lacks), so this file
// re-creates only their *setup* steps (globals + init_variable) and then calls
// the reference's own hot-path functions:
//
//   Optimization3D_admm::optimization           (Optimization3D_admm.h:29)
//   Optimization3D_multi::optimization_decouple (Optimization3D_multi.h:29)
//   Optimization3D_multi::optimization          (Optimization3D_multi.h:120, coupled: "decouple":0)
//
// plus the public static stages of those classes so that intermediates can be
// dumped for golden fixtures (tests/golden/, made by tests/golden/make_golden.py).
//
// Setup steps mirrored (file:line in /root/reference):
//   kdop/aabb axis matrices        Main/admmPathPlanning3D.cpp:403-416
//   BVH::InitPointcloud            Main/admmPathPlanning3D.cpp:431-434
//   combination/ks/kt/convert      Main/admmPathPlanning3D.cpp:471-484, multiPathPlanning3D.cpp:590-600
//   init_variable (single)         Main/admmPathPlanning3D.cpp:249-353
//   init_variable (multi)          Main/multiPathPlanning3D.cpp:342-467
#include "HighOrderCCD/Optimization/Optimization3D_admm.h"
#include "HighOrderCCD/Optimization/Optimization3D_multi.h"
#include "HighOrderCCD/BVH/BVH.h"
#include <cstring>
#include <streambuf>

USE_PRJ_NAMESPACE
typedef Eigen::MatrixXd Data;

namespace {
struct NullBuf : std::streambuf { int overflow(int c) override { return c; } };
NullBuf g_nullbuf;
std::streambuf* g_old_cout = nullptr;

int g_mode = 0;  // 0 = single (Optimization3D_admm), 1 = multi decouple, 2 = multi coupled (shared piece_time)
BVH* g_bvh = nullptr;
std::vector<Eigen::RowVector3d> g_vertex_list;
std::vector<Data> g_spline, g_p_slack, g_p_lambda;
std::vector<Eigen::VectorXd> g_t_slack, g_t_lambda;
std::vector<double> g_piece_time;
bool g_axes_done = false;

// stage intermediates
std::vector<std::vector<std::vector<Eigen::Vector3d>>> g_c_lists;
std::vector<std::vector<std::vector<double>>> g_d_lists;
std::vector<Data> g_direction;
std::vector<double> g_t_direction, g_wolfe_each, g_gn_each, g_step;

void build_tables() {
  // shared tail of both init_variable()s
  M_dynamic = Dynamic3D<order_num, der_num>::dynamic_matrix();
  subdivide_tree.resize(piece_num * res);
  A_list.resize(piece_num * res);
  A_vel_list.resize(piece_num * res);
  A_acc_list.resize(piece_num * res);
  Eigen::MatrixXd basis, tmp_basis;
  Eigen::Matrix3d I; I.setIdentity();
  for (int k = 0; k < res; k++) {
    double a = k / double(res), b = (k + 1) / double(res);
    Blossom<order_num>::coefficient(basis, a, b);
    for (int i = 0; i < piece_num; i++) {
      std::pair<double, double> range(a, b);
      subdivide_tree[i * res + k] = std::make_tuple(i, range, basis * convert_list[i]);
      tmp_basis = basis * convert_list[i];
      A_list[i * res + k].resize(order_num + 1);
      A_vel_list[i * res + k].resize(order_num);
      A_acc_list[i * res + k].resize(order_num - 1);
      for (int j = 0; j <= order_num; j++) {
        Eigen::MatrixXd A = Eigen::kroneckerProduct(tmp_basis.row(j), I);
        A.transposeInPlace();
        A_list[i * res + k][j] = A;
        if (j < order_num) {
          A = Eigen::kroneckerProduct(tmp_basis.row(j + 1), I) - Eigen::kroneckerProduct(tmp_basis.row(j), I);
          A_vel_list[i * res + k][j] = A;
        }
        if (j < order_num - 1) {
          A = Eigen::kroneckerProduct(tmp_basis.row(j + 2), I) - 2 * Eigen::kroneckerProduct(tmp_basis.row(j + 1), I) +
              Eigen::kroneckerProduct(tmp_basis.row(j), I);
          A_acc_list[i * res + k][j] = A;
        }
      }
    }
  }
}
}  // namespace

extern "C" {

void ref_quiet(int on) {
  if (on && !g_old_cout) g_old_cout = std::cout.rdbuf(&g_nullbuf);
  if (!on && g_old_cout) { std::cout.rdbuf(g_old_cout); g_old_cout = nullptr; }
}

// params: lambda, margin, offset, mu, vel_limit, acc_limit, ks, kt
int ref_setup(int mode, int U, int P, int res_, const double* params, const double* cloud_rowmajor, int N) {
  ref_quiet(1);
  g_mode = mode;
  uav_num = U; piece_num = P; res = res_;
  lambda = params[0]; margin = params[1]; offset = params[2]; mu = params[3];
  vel_limit = params[4]; acc_limit = params[5]; ks = params[6]; kt = params[7];
  epsilon = 0.1; is_optimal_plane = false; automove = true; iter = 0; gnorm = 1;
  if (!g_axes_done) {
    int dim = kdop_axis.size();
    kdop_matrix.resize(3, dim);
    for (int k = 0; k < dim; k++) { kdop_axis[k].normalize(); kdop_matrix.col(k) = kdop_axis[k]; }
    aabb_matrix.resize(3, 3);
    for (int k = 0; k < 3; k++) aabb_matrix.col(k) = aabb_axis[k];
    g_axes_done = true;
  }
  Eigen::MatrixXd V(N, 3);
  for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) V(i, k) = cloud_rowmajor[3 * i + k];
  delete g_bvh; g_bvh = new BVH();
  if (N > 0) g_bvh->InitPointcloud(V);
  g_vertex_list.resize(N);
  for (int i = 0; i < N; i++) g_vertex_list[i] = V.row(i);
  time_weight.assign(piece_num, 1.0);
  whole_weight = piece_num;
  trajectory_num = (order_num + 1) + (piece_num - 1) * (order_num + 1 - 3);
  combination = Combination<40>::value();
  Conversion<order_num>::convert_matrix();
  build_tables();
  return 0;
}

// waypoints: U x (P+1) x 3 row-major (already scaled).
int ref_init_state(const double* wp, double piece_time0) {
  int U = uav_num, P = piece_num, T = trajectory_num;
  g_spline.assign(U, Data()); g_p_slack.assign(U, Data()); g_p_lambda.assign(U, Data());
  g_t_slack.assign(U, Eigen::VectorXd()); g_t_lambda.assign(U, Eigen::VectorXd());
  g_piece_time.assign(U, piece_time0);
  for (int u = 0; u < U; u++) {
    std::vector<Eigen::Vector3d> way_points(P + 1);
    for (int k = 0; k <= P; k++) way_points[k] = Eigen::Vector3d(wp[(u * (P + 1) + k) * 3], wp[(u * (P + 1) + k) * 3 + 1], wp[(u * (P + 1) + k) * 3 + 2]);
    Data spline(T, 3);
    if (g_mode == 0) {  // Main/admmPathPlanning3D.cpp:258-275
      spline.row(0) = way_points[0].transpose();
      for (int i = 0; i < P; i++) {
        Eigen::Vector3d head = 0.9 * way_points[i] + 0.1 * way_points[i + 1];
        Eigen::Vector3d tail = 0.9 * way_points[i + 1] + 0.1 * way_points[i];
        spline.row(i * (order_num - 2) + 1) = way_points[i].transpose();
        for (int j = 1; j < order_num - 2; j++)
          spline.row(j + i * (order_num - 2) + 1) = double(order_num - 3 - j) / (order_num - 4) * head.transpose() + (double)(j - 1) / (order_num - 4) * tail.transpose();
        spline.row((i + 1) * (order_num - 2) + 1) = way_points[i + 1].transpose();
      }
      spline.row(T - 1) = way_points[P].transpose();
    } else {  // Main/multiPathPlanning3D.cpp:363-375
      spline.row(0) = way_points[0].transpose();
      for (int k = 0; k < P; k++)
        for (int j = 0; j <= order_num - 2; j++)
          spline.row(j + k * (order_num - 2) + 1) = double(order_num - 2 - j) / (order_num - 2) * way_points[k].transpose() + (double)j / (order_num - 2) * way_points[k + 1].transpose();
      spline.row(T - 1) = way_points[P].transpose();
    }
    spline.row(1) = spline.row(0);
    spline.row(T - 2) = spline.row(T - 1);
    Data p_slack((order_num + 1) * P, 3), p_lambda((order_num + 1) * P, 3);
    p_lambda.setZero();
    for (int sp = 0; sp < P; sp++)
      p_slack.block<order_num + 1, 3>(sp * (order_num + 1), 0) = convert_list[sp] * spline.block<order_num + 1, 3>(sp * (order_num - 2), 0);
    Eigen::VectorXd t_slack(P), t_lambda(P);
    t_lambda.setZero();
    for (int sp = 0; sp < P; sp++) t_slack(sp) = piece_time0;
    g_spline[u] = spline; g_p_slack[u] = p_slack; g_p_lambda[u] = p_lambda;
    g_t_slack[u] = t_slack; g_t_lambda[u] = t_lambda;
  }
  iter = 0; gnorm = 1;
  return 0;
}

// "optimal_plane":1 -- persistent per-(segment, obstacle) / per-(segment, robot pair) plane caches, allocated
// the way the mains' init_variable does (Main/admmPathPlanning3D.cpp:343-351, Main/multiPathPlanning3D.cpp:450-464).
// Call after ref_setup / ref_init_state.
void ref_set_optimal_plane(int on) {
  is_optimal_plane = on != 0;
  const int S = piece_num * res, N = (int)g_vertex_list.size();
  is_seperate.assign(S, std::vector<bool>()); seperate_c.assign(S, {}); seperate_d.assign(S, {});
  is_self_seperate.assign(S, {}); self_seperate_c.assign(S, {}); self_seperate_d.assign(S, {});
  if (!on) return;
  for (int i = 0; i < S; i++) {
    is_seperate[i].assign(N, false); seperate_c[i].resize(N); seperate_d[i].resize(N);
    is_self_seperate[i].resize(uav_num); self_seperate_c[i].resize(uav_num); self_seperate_d[i].resize(uav_num);
    for (int j = 0; j < uav_num; j++) {
      is_self_seperate[i][j].assign(uav_num, false); self_seperate_c[i][j].resize(uav_num); self_seperate_d[i][j].resize(uav_num);
    }
  }
}

// the persistent caches, in the same exchange format as the oracle's orc_{get,set}_{obs,pair}_cache
int ref_get_obs_cache(int tr, int cap, int* ids, double* cd) {
  int n = 0;
  for (size_t ob = 0; ob < is_seperate[tr].size(); ob++)
    if (is_seperate[tr][ob]) {
      if (n < cap) { ids[n] = (int)ob; for (int a = 0; a < 3; a++) cd[4 * n + a] = seperate_c[tr][ob](a); cd[4 * n + 3] = seperate_d[tr][ob]; }
      n++;
    }
  return n;
}
void ref_set_obs_cache(int tr, int n, const int* ids, const double* cd) {
  is_seperate[tr].assign(is_seperate[tr].size(), false);
  for (int i = 0; i < n; i++) {
    is_seperate[tr][ids[i]] = true;
    seperate_c[tr][ids[i]] = Eigen::Vector3d(cd[4 * i], cd[4 * i + 1], cd[4 * i + 2]);
    seperate_d[tr][ids[i]] = cd[4 * i + 3];
  }
}
void ref_get_pair_cache(int* flags, double* cd) {
  const int S = piece_num * res, U = uav_num;
  for (int tr = 0; tr < S; tr++) for (int a = 0; a < U; a++) for (int b = 0; b < U; b++) {
    const size_t i = ((size_t)tr * U + a) * U + b;
    flags[i] = is_self_seperate[tr][a][b] ? 1 : 0;
    for (int k = 0; k < 3; k++) cd[4 * i + k] = flags[i] ? self_seperate_c[tr][a][b](k) : 0.0;
    cd[4 * i + 3] = flags[i] ? self_seperate_d[tr][a][b] : 0.0;
  }
}
void ref_set_pair_cache(const int* flags, const double* cd) {
  const int S = piece_num * res, U = uav_num;
  for (int tr = 0; tr < S; tr++) for (int a = 0; a < U; a++) for (int b = 0; b < U; b++) {
    const size_t i = ((size_t)tr * U + a) * U + b;
    is_self_seperate[tr][a][b] = flags[i] != 0;
    self_seperate_c[tr][a][b] = Eigen::Vector3d(cd[4 * i], cd[4 * i + 1], cd[4 * i + 2]);
    self_seperate_d[tr][a][b] = cd[4 * i + 3];
  }
}

// The motion validator of the reference's OMPL set-up (HighOrderCCD/OMPL/OMPL.cpp:36-98; the same test is edge_collision in
// Main/multiPathPlanning3D.cpp:123-160, which cannot be linked here because that file includes the OMPL headers): the
// reference's own BVH::EdgeCollision and CCD::GJKDCD in the same sequence.
void ref_edge_collision(int n, const double* edges, int n_prior, const double* prior, double d, int* hit) {
  for (int e = 0; e < n; e++) {
    Eigen::MatrixXd edge(2, 3);
    for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++) edge(r, k) = edges[6 * (size_t)e + 3 * r + k];
    bool h = false;
    if (!g_vertex_list.empty()) {
      std::vector<unsigned int> pairs;
      g_bvh->EdgeCollision(edge, pairs, d);
      for (unsigned int ob : pairs) { Eigen::RowVector3d p = g_vertex_list[ob]; if (CCD::GJKDCD(edge, p, d)) { h = true; break; } }
    }
    for (int j = 0; j < n_prior && !h; j++) {
      Eigen::MatrixXd pe(2, 3);
      for (int r = 0; r < 2; r++) for (int k = 0; k < 3; k++) pe(r, k) = prior[6 * (size_t)j + 3 * r + k];
      if (CCD::GJKDCD(edge, pe, d)) h = true;
    }
    hit[e] = h;
  }
}

int ref_T() { return trajectory_num; }

void ref_get_state(int u, double* spline, double* p_slack, double* p_lambda, double* t_slack, double* t_lambda, double* piece_time) {
  int P = piece_num, T = trajectory_num;
  memcpy(spline, g_spline[u].data(), sizeof(double) * T * 3);
  memcpy(p_slack, g_p_slack[u].data(), sizeof(double) * 6 * P * 3);
  memcpy(p_lambda, g_p_lambda[u].data(), sizeof(double) * 6 * P * 3);
  memcpy(t_slack, g_t_slack[u].data(), sizeof(double) * P);
  memcpy(t_lambda, g_t_lambda[u].data(), sizeof(double) * P);
  *piece_time = g_piece_time[u];
}

void ref_set_state(int u, const double* spline, const double* p_slack, const double* p_lambda, const double* t_slack, const double* t_lambda, double piece_time) {
  int P = piece_num, T = trajectory_num;
  memcpy(g_spline[u].data(), spline, sizeof(double) * T * 3);
  memcpy(g_p_slack[u].data(), p_slack, sizeof(double) * 6 * P * 3);
  memcpy(g_p_lambda[u].data(), p_lambda, sizeof(double) * 6 * P * 3);
  memcpy(g_t_slack[u].data(), t_slack, sizeof(double) * P);
  memcpy(g_t_lambda[u].data(), t_lambda, sizeof(double) * P);
  g_piece_time[u] = piece_time;
}

// One full ADMM iteration through the reference's own entry point.  Returns gnorm.
double ref_iterate() {
  if (g_mode == 0) {
    Optimization3D_admm::optimization(g_spline[0], g_piece_time[0], g_p_slack[0], g_t_slack[0], g_p_lambda[0], g_t_lambda[0], g_vertex_list, *g_bvh);
  } else if (g_mode == 1) {
    Optimization3D_multi::optimization_decouple(g_spline, g_piece_time, g_p_slack, g_t_slack, g_p_lambda, g_t_lambda, g_vertex_list, *g_bvh);
  } else {  // Main/multiPathPlanning3D.cpp:674-677: one piece_time shared by all robots
    double pt = g_piece_time[0];
    Optimization3D_multi::optimization(g_spline, pt, g_p_slack, g_t_slack, g_p_lambda, g_t_lambda, g_vertex_list, *g_bvh);
    for (double& v : g_piece_time) v = pt;
  }
  iter++;
  return gnorm;
}

void ref_get_tables(double* convert, double* Mdyn, double* basis) {
  for (int i = 0; i < piece_num; i++) memcpy(convert + 36 * i, convert_list[i].data(), 36 * sizeof(double));
  memcpy(Mdyn, M_dynamic.data(), 36 * sizeof(double));
  for (size_t s = 0; s < subdivide_tree.size(); s++) memcpy(basis + 36 * s, std::get<2>(subdivide_tree[s]).data(), 36 * sizeof(double));
}
void ref_get_kdop(double* axes /*3*49 col-major*/) { memcpy(axes, kdop_matrix.data(), sizeof(double) * 3 * 49); }

// ---------------- stages (same sequence as optimization / optimization_decouple) -------------
// Stage 1: separating planes (obstacle, then inter-robot in multi mode).
int ref_stage_planes() {
  int U = uav_num;
  g_c_lists.assign(U, {}); g_d_lists.assign(U, {});
  for (int i = 0; i < U; i++) {
    if (g_mode == 0) Optimization3D_admm::separate_plane(g_spline[i], g_vertex_list, g_c_lists[i], g_d_lists[i], *g_bvh);
    else Optimization3D_multi::separate_plane(g_spline[i], g_vertex_list, g_c_lists[i], g_d_lists[i], *g_bvh);
  }
  if (g_mode >= 1) Optimization3D_multi::separate_self(g_spline, g_c_lists, g_d_lists, *g_bvh);
  int total = 0;
  for (int i = 0; i < U; i++) for (auto& l : g_d_lists[i]) total += l.size();
  return total;
}
// counts: U*S ; planes: total*4 (cx,cy,cz,d) in (u, tr, k) order
void ref_get_planes(int* counts, double* planes) {
  int S = subdivide_tree.size(); size_t w = 0;
  for (int u = 0; u < uav_num; u++)
    for (int tr = 0; tr < S; tr++) {
      counts[u * S + tr] = g_d_lists[u][tr].size();
      for (size_t k = 0; k < g_d_lists[u][tr].size(); k++) {
        planes[4 * w] = g_c_lists[u][tr][k](0); planes[4 * w + 1] = g_c_lists[u][tr][k](1);
        planes[4 * w + 2] = g_c_lists[u][tr][k](2); planes[4 * w + 3] = g_d_lists[u][tr][k]; w++;
      }
    }
}
// Inject planes (teacher forcing): same layout as ref_get_planes.
void ref_set_planes(const int* counts, const double* planes) {
  int S = subdivide_tree.size(); size_t w = 0;
  g_c_lists.assign(uav_num, {}); g_d_lists.assign(uav_num, {});
  for (int u = 0; u < uav_num; u++) {
    g_c_lists[u].resize(S); g_d_lists[u].resize(S);
    for (int tr = 0; tr < S; tr++)
      for (int k = 0; k < counts[u * S + tr]; k++) {
        g_c_lists[u][tr].push_back(Eigen::Vector3d(planes[4 * w], planes[4 * w + 1], planes[4 * w + 2]));
        g_d_lists[u][tr].push_back(planes[4 * w + 3]); w++;
      }
  }
}

// Stage 2: descent direction for every robot.  Returns gnorm as the driver would see it.
double ref_stage_direction() {
  int U = uav_num;
  g_direction.assign(U, Data()); g_t_direction.assign(U, 0); g_wolfe_each.assign(U, 0); g_gn_each.assign(U, 0);
  gnorm = 0;
  for (int i = 0; i < U; i++) {
    double before = gnorm;
    if (g_mode == 0) {
      Optimization3D_admm::spline_descent_direction(g_spline[i], g_direction[i], g_piece_time[i], g_t_direction[i], g_p_slack[i], g_t_slack[i], g_p_lambda[i], g_t_lambda[i], g_c_lists[i], g_d_lists[i]);
      g_gn_each[i] = gnorm;
    } else {
      Optimization3D_multi::spline_descent_direction(g_spline[i], g_direction[i], g_piece_time[i], g_t_direction[i], g_p_slack[i], g_t_slack[i], g_p_lambda[i], g_t_lambda[i], g_c_lists[i], g_d_lists[i]);
      g_gn_each[i] = gnorm - before;
    }
    g_wolfe_each[i] = wolfe;
  }
  if (g_mode == 1) gnorm /= double(U);
  return gnorm;
}
void ref_get_direction(int u, double* direction /*T*3*/, double* t_direction, double* wolfe_u, double* gn_u) {
  memcpy(direction, g_direction[u].data(), sizeof(double) * trajectory_num * 3);
  *t_direction = g_t_direction[u]; *wolfe_u = g_wolfe_each[u]; *gn_u = g_gn_each[u];
}
// teacher forcing of the CCD / line-search stages: overwrite robot u's search direction record
void ref_set_direction(int u, const double* direction, double t_direction, double wolfe_u, double gn_u) {
  int U = uav_num;
  if ((int)g_direction.size() != U) { g_direction.assign(U, Data::Zero(trajectory_num, 3)); g_t_direction.assign(U, 0); g_wolfe_each.assign(U, 0); g_gn_each.assign(U, 0); }
  g_direction[u] = Eigen::Map<const Data>(direction, trajectory_num, 3);
  g_t_direction[u] = t_direction; g_wolfe_each[u] = wolfe_u; g_gn_each[u] = gn_u;
}
// per-piece local gradient/Hessian before the PSD shift (Gradient_admm.h:67-164)
void ref_local_grad(int u, int sp_id, double* g19, double* h361) {
  Eigen::VectorXd g; Eigen::MatrixXd h;
  Gradient_admm::local_spline_gradient(g_spline[u], g_piece_time[u], g_p_slack[u], g_t_slack[u], g_p_lambda[u], g_t_lambda[u], g_c_lists[u], g_d_lists[u], g, h, sp_id);
  memcpy(g19, g.data(), 19 * sizeof(double)); memcpy(h361, h.data(), 361 * sizeof(double));
}
// one segment's velocity / acceleration barrier terms (Gradient_admm.h:409-572): g[18], h[18*18] column-major, part[18]
void ref_bound_grad(int u, int tr, double* g18, double* h324, double* gt_ht, double* part18) {
  Eigen::VectorXd g, part; Eigen::MatrixXd h; double g_t, h_t;
  Gradient_admm::local_bound_gradient(tr, g_spline[u], g_piece_time[u], g, h, g_t, h_t, part);
  memcpy(g18, g.data(), 18 * sizeof(double)); memcpy(h324, h.data(), 324 * sizeof(double)); memcpy(part18, part.data(), 18 * sizeof(double));
  gt_ht[0] = g_t; gt_ht[1] = h_t;
}
// assembled (3T+1) gradient and Hessian after per-piece PSD projection (Gradient_admm.h:13-65)
void ref_global_grad(int u, double* g, double* h) {
  Eigen::VectorXd gg; Eigen::MatrixXd hh;
  Gradient_admm::global_spline_gradient(g_spline[u], g_piece_time[u], g_p_slack[u], g_t_slack[u], g_p_lambda[u], g_t_lambda[u], g_c_lists[u], g_d_lists[u], gg, hh);
  memcpy(g, gg.data(), gg.size() * sizeof(double)); memcpy(h, hh.data(), hh.size() * sizeof(double));
}

// Stage 3: CCD steps.  self (multi only) then position step; out[u] = min of both,
// self_out[u] / pos_out[u] the individual values.
void ref_stage_steps(double* self_out, double* pos_out) {
  int U = uav_num;
  g_step.assign(U, 1.0);
  std::vector<double> sl;
  if (g_mode == 1) { Step::self_step(g_spline, g_direction, sl, *g_bvh); }
  else sl.assign(U, 1.0);
  for (int i = 0; i < U; i++) {
    double ps = Step::position_step(g_spline[i], g_direction[i], g_vertex_list, *g_bvh);
    self_out[i] = sl[i]; pos_out[i] = ps;
    g_step[i] = std::min(sl[i], ps);
  }
}
// Stage 4: line search + commit.  `wolfe` is the global left by stage 2 (last robot's) --
// exactly what optimization_decouple sees.  Returns final Armijo step for each robot.
void ref_stage_linesearch(double* step_out) {
  int U = uav_num;
  if (!g_wolfe_each.empty()) wolfe = g_wolfe_each[U - 1];
  for (int i = 0; i < U; i++) {
    if (g_mode == 0) {
      // the single-UAV line search recomputes position_step itself
      Optimization3D_admm::spline_line_search(g_spline[i], g_direction[i], g_piece_time[i], g_t_direction[i], g_p_slack[i], g_t_slack[i], g_p_lambda[i], g_t_lambda[i], g_vertex_list, *g_bvh, g_c_lists[i], g_d_lists[i]);
      step_out[i] = 0;
    } else {
      double st = g_step[i];
      Optimization3D_multi::spline_line_search(g_spline[i], g_direction[i], g_piece_time[i], g_t_direction[i], g_p_slack[i], g_t_slack[i], g_p_lambda[i], g_t_lambda[i], g_c_lists[i], g_d_lists[i], st);
      step_out[i] = st;
    }
  }
}
// Stage 5: slack (z) update + dual update.
void ref_stage_slack() {
  for (int i = 0; i < uav_num; i++) {
    if (g_mode == 0) Optimization3D_admm::update_slack_lambda(g_spline[i], g_piece_time[i], g_p_slack[i], g_t_slack[i], g_p_lambda[i], g_t_lambda[i]);
    else Optimization3D_multi::update_slack_lambda(g_spline[i], g_piece_time[i], g_p_slack[i], g_t_slack[i], g_p_lambda[i], g_t_lambda[i]);
  }
}
// Coupled mode ("decouple":0): Newton system, CCD clamps and the Armijo search on the summed energy are ONE
// reference function (Optimization3D_multi::update_spline, Optimization3D_multi.h:508-639).  Uses the planes of
// ref_stage_planes / ref_set_planes.  Returns gnorm (= |G| / uav_num, :580); *wolfe_out = global wolfe (:558).
double ref_stage_update_spline(double* wolfe_out) {
  double pt = g_piece_time[0];
  Optimization3D_multi::update_spline(g_spline, pt, g_p_slack, g_t_slack, g_p_lambda, g_t_lambda, g_c_lists, g_d_lists, g_vertex_list, *g_bvh);
  for (double& v : g_piece_time) v = pt;
  if (wolfe_out) *wolfe_out = wolfe;
  return gnorm;
}
double ref_spline_energy(int u) {
  return Energy_admm::spline_energy(g_spline[u], g_piece_time[u], g_p_slack[u], g_t_slack[u], g_p_lambda[u], g_t_lambda[u], g_c_lists[u], g_d_lists[u]);
}

// ---------------- known-answer primitives ----------------
// GJK witness vector (openGJK.c:754) for two point sets given row-major n x 3.
void ref_gjk(int n1, const double* p1, int n2, const double* p2, double* v_out) {
  struct bd b1, b2; struct simplex s;
  std::vector<double*> r1(n1), r2(n2);
  std::vector<double> c1(p1, p1 + 3 * n1), c2(p2, p2 + 3 * n2);
  for (int i = 0; i < n1; i++) r1[i] = &c1[3 * i];
  for (int i = 0; i < n2; i++) r2[i] = &c2[3 * i];
  b1.coord = r1.data(); b1.numpoints = n1; b2.coord = r2.data(); b2.numpoints = n2; s.nvrtx = 0;
  double* c0 = gjk(b1, b2, &s);
  v_out[0] = c0[0]; v_out[1] = c0[1]; v_out[2] = c0[2];
}
// Separate::opengjk / selfgjk on column-major 6x3 hull (+ 1x3 point or 6x3 hull).
int ref_plane_obs(const double* P6x3, const double* q, double dist, double* cd) {
  Data P = Eigen::Map<const Data>(P6x3, 6, 3); Data Q(1, 3); Q << q[0], q[1], q[2];
  Eigen::Vector3d c; double d = 0;
  bool ok = Separate::opengjk(P, Q, dist, c, d);
  cd[0] = c(0); cd[1] = c(1); cd[2] = c(2); cd[3] = d; return ok;
}
int ref_plane_self(const double* P6x3, const double* Q6x3, double dist, int refine, double* cd) {
  Data P = Eigen::Map<const Data>(P6x3, 6, 3); Data Q = Eigen::Map<const Data>(Q6x3, 6, 3);
  Eigen::Vector3d c; double d = 0;
  bool ok = Separate::selfgjk(P, Q, dist, c, d);
  if (ok && refine) Optimal_plane::optimal_d(P, Q, c, d);
  cd[0] = c(0); cd[1] = c(1); cd[2] = c(2); cd[3] = d; return ok;
}
double ref_kat_min_eig_small(int n, const double* a) {
  if (n == 2) { Eigen::Matrix2d m; m << a[0], a[1], a[2], a[3]; Eigen::SelfAdjointEigenSolver<Eigen::Matrix2d> es(m); Eigen::MatrixXd ev = es.eigenvalues(); return ev(0); }
  Eigen::Matrix3d m; m << a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8];
  Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> es(m); Eigen::MatrixXd ev = es.eigenvalues(); return ev(0);
}
void ref_kat_optimal_cd(const double* P6x3, const double* q, double* cd) {
  Data P = Eigen::Map<const Data>(P6x3, 6, 3); Eigen::RowVector3d Q(q[0], q[1], q[2]);
  Eigen::Vector3d c(cd[0], cd[1], cd[2]); double d = cd[3];
  Optimal_plane::optimal_cd(P, Q, c, d);
  cd[0] = c(0); cd[1] = c(1); cd[2] = c(2); cd[3] = d;
}
void ref_kat_self_optimal_cd(const double* P6x3, const double* Q6x3, double* cd) {
  Data P = Eigen::Map<const Data>(P6x3, 6, 3); Data Q = Eigen::Map<const Data>(Q6x3, 6, 3);
  Eigen::Vector3d c(cd[0], cd[1], cd[2]); double d = cd[3];
  Optimal_plane::self_optimal_cd(P, Q, c, d);
  cd[0] = c(0); cd[1] = c(1); cd[2] = c(2); cd[3] = d;
}
int ref_kdop_dcd(const double* P6x3, const double* q, double d) {
  Data P = Eigen::Map<const Data>(P6x3, 6, 3); Data Q(1, 3); Q << q[0], q[1], q[2];
  return CCD::KDOPDCD(P, Q, d);
}
int ref_kdop_self_dcd(const double* P6x3, const double* Q6x3, double d) {
  Data P = Eigen::Map<const Data>(P6x3, 6, 3); Data Q = Eigen::Map<const Data>(Q6x3, 6, 3);
  return CCD::SelfKDOPDCD(P, Q, d);
}
int ref_kdop_ccd(const double* P, const double* D, const double* q, double d, double t0, double t1) {
  Data Pm = Eigen::Map<const Data>(P, 6, 3), Dm = Eigen::Map<const Data>(D, 6, 3); Data Q(1, 3); Q << q[0], q[1], q[2];
  return CCD::KDOPCCD(Pm, Dm, Q, d, t0, t1);
}
int ref_gjk_ccd(const double* P, const double* D, const double* q, double d, double t0, double t1) {
  Data Pm = Eigen::Map<const Data>(P, 6, 3), Dm = Eigen::Map<const Data>(D, 6, 3); Data Q(1, 3); Q << q[0], q[1], q[2];
  return CCD::GJKCCD(Pm, Dm, Q, d, t0, t1);
}
int ref_self_kdop_ccd(const double* P, const double* D, const double* Q, const double* E, double d, double t1, double u1) {
  Data Pm = Eigen::Map<const Data>(P, 6, 3), Dm = Eigen::Map<const Data>(D, 6, 3), Qm = Eigen::Map<const Data>(Q, 6, 3), Em = Eigen::Map<const Data>(E, 6, 3);
  return CCD::SelfKDOPCCD(Pm, Dm, Qm, Em, d, 0, t1, 0, u1);
}
int ref_self_gjk_ccd(const double* P, const double* D, const double* Q, const double* E, double d, double t1, double u1) {
  Data Pm = Eigen::Map<const Data>(P, 6, 3), Dm = Eigen::Map<const Data>(D, 6, 3), Qm = Eigen::Map<const Data>(Q, 6, 3), Em = Eigen::Map<const Data>(E, 6, 3);
  return CCD::SelfGJKCCD(Pm, Dm, Qm, Em, d, 0, t1, 0, u1);
}
// The same two predicates with body sizes taken from the arguments (CCD::KDOPDCD and CCD::GJKDCD loop over position.rows() /
// _position.rows(), CCD.h:17-114, :354-413): known answers for 3-vertex obstacle bodies (triangles) and 12-row swept hulls.
// A, B row-major n x 3.
int ref_kdop_general(int n1, const double* A, int n2, const double* B, double d) {
  Data Am(n1, 3), Bm(n2, 3);
  for (int i = 0; i < n1; i++) for (int k = 0; k < 3; k++) Am(i, k) = A[3 * i + k];
  for (int i = 0; i < n2; i++) for (int k = 0; k < 3; k++) Bm(i, k) = B[3 * i + k];
  return CCD::KDOPDCD(Am, Bm, d);
}
int ref_gjk_dcd_general(int n1, const double* A, int n2, const double* B, double d) {
  Data Am(n1, 3), Bm(n2, 3);
  for (int i = 0; i < n1; i++) for (int k = 0; k < 3; k++) Am(i, k) = A[3 * i + k];
  for (int i = 0; i < n2; i++) for (int k = 0; k < 3; k++) Bm(i, k) = B[3 * i + k];
  return CCD::GJKDCD(Am, Bm, d);
}
// The single-UAV Newton solve exactly as Optimization3D_admm.h:470-475 performs it: SimplicialLLT (AMD ordering) on the
// sparse view of a dense symmetric matrix.  H column-major n x n.  order = permutationPinv().indices().
int ref_sparse_llt_solve(int n, const double* H, const double* b, double* x, int* order) {
  Eigen::MatrixXd h0 = Eigen::Map<const Eigen::MatrixXd>(H, n, n);
  Eigen::VectorXd g0 = Eigen::Map<const Eigen::VectorXd>(b, n);
  Eigen::SparseMatrix<double> Hs = h0.sparseView();
  Eigen::SimplicialLLT<Eigen::SparseMatrix<double>> solver;
  solver.compute(Hs);
  Eigen::VectorXd xs = solver.solve(g0);
  for (int i = 0; i < n; i++) { x[i] = xs(i); if (order) order[i] = solver.permutationPinv().indices()(i); }
  return solver.info() == Eigen::Success;
}
// Broad phase known answers on the reference's own trees: prim = 1 BVH::InitPointcloud(V) + pc_tree.query, prim = 3
// BVH::InitObstacle(V, F) + ob_tree.query (the dormant triangle path, BVH.cpp:15-51).  verts row-major [n][prim][3];
// boxes [nq][6] = lo, hi.  ids are appended query after query; returns the total (may exceed cap: then call again).
int ref_query_kat(int prim, const double* verts, int n, int nq, const double* boxes, double d, int* counts, int* ids, int cap) {
  BVH bvh;
  NullBuf nb; std::streambuf* old = std::cout.rdbuf(&nb);
  if (prim == 1) {
    Eigen::MatrixXd V(n, 3);
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) V(i, k) = verts[3 * (size_t)i + k];
    bvh.InitPointcloud(V);
  } else {
    Eigen::MatrixXd V(3 * n, 3); Eigen::MatrixXi F(n, 3);
    for (int i = 0; i < 3 * n; i++) for (int k = 0; k < 3; k++) V(i, k) = verts[3 * (size_t)i + k];
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) F(i, j) = 3 * i + j;
    bvh.InitObstacle(V, F);
  }
  std::cout.rdbuf(old);
  int w = 0;
  for (int q = 0; q < nq; q++) {
    std::vector<double> lo(boxes + 6 * (size_t)q, boxes + 6 * (size_t)q + 3), hi(boxes + 6 * (size_t)q + 3, boxes + 6 * (size_t)q + 6);
    aabb::AABB box(lo, hi);
    std::vector<unsigned int> r = prim == 1 ? bvh.pc_tree.query(box, d) : bvh.ob_tree.query(box, d);
    counts[q] = (int)r.size();
    for (unsigned int id : r) { if (w < cap) ids[w] = (int)id; w++; }
  }
  return w;
}
// Broad phase: candidates of every segment of robot u (BVH::DCDCollision / CCDCollision).
// use_dir=0: DCD with margin d; use_dir=1: CCD using the stage-2 direction.
int ref_candidates(int u, int use_dir, double d, int* counts /*S*/, int* ids, int cap) {
  std::vector<std::vector<unsigned int>> pairs;
  if (use_dir) g_bvh->CCDCollision(g_spline[u], g_direction[u], pairs, d);
  else g_bvh->DCDCollision(g_spline[u], pairs, d);
  int w = 0;
  for (size_t s = 0; s < pairs.size(); s++) {
    counts[s] = pairs[s].size();
    for (unsigned int id : pairs[s]) { if (w < cap) ids[w] = id; w++; }
  }
  return w;
}
// Pair order of the dynamic-AABB-tree self query (AABB.cc:669-734) for n boxes (lo,hi row-major n x 3).
int ref_self_pairs(int n, const double* lo, const double* hi, double d, int* pairs, int cap) {
  aabb::Tree tree(3, 0.0, n, true);
  for (int i = 0; i < n; i++) {
    std::vector<double> l(lo + 3 * i, lo + 3 * i + 3), h(hi + 3 * i, hi + 3 * i + 3);
    tree.insertParticle(i, l, h);
  }
  auto pr = tree.query(d);
  int w = 0;
  for (auto& p : pr) { if (w < cap) { pairs[2 * w] = p.first; pairs[2 * w + 1] = p.second; } w++; }
  return w;
}
// Dense LLT + min eigenvalue, as the reference uses them (for KATs of the PSD shift).
int ref_llt_fails(int n, const double* h) {
  Eigen::MatrixXd H = Eigen::Map<const Eigen::MatrixXd>(h, n, n);
  Eigen::LLT<Eigen::MatrixXd> s; s.compute(H);
  return s.info() == Eigen::NumericalIssue;
}
double ref_min_eig(int n, const double* h) {
  Eigen::MatrixXd H = Eigen::Map<const Eigen::MatrixXd>(h, n, n);
  Eigen::SelfAdjointEigenSolver<Eigen::MatrixXd> es(H);
  return es.eigenvalues()(0);
}
}  // extern "C"
