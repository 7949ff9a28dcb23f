// cli_main.cpp -- `admmPathPlanning3D <mesh>` (-DTJ_CLI_SINGLE) and `multiPathPlanning3D <mesh>`
// (-DTJ_CLI_MULTI): the headless branches of the reference mains
// (Main/admmPathPlanning3D.cpp:355-547, Main/multiPathPlanning3D.cpp:470-695) with the per-iteration
// call replaced by the C ABI of libtrajadmm.so.  Same working-directory layout, same config keys,
// same result file.  `init:2` plans the way points with the library's device planner instead of OMPL,
// `decouple:0` selects the coupled multi-robot mode (one shared piece_time, Main/multiPathPlanning3D.cpp:674-677),
// `optimal_plane:1` the persistent-plane branch.  The GUI (`gui:1`) is outside the accelerated path and is rejected
// with a message instead of being silently ignored.  On convergence the trajectory duration and sampled arc length are printed
// like the mains' log_data ("ccd time:", "ccd len:").
//
// Extras (ours): --max-iter N, --dump-state FILE (control points; the reference never writes the
// trajectory), --sample-traj FILE (positions sampled like log_data, one "uav t x y z" per line),
// --batch N iterations per device batch (the stop test runs on the device before every iteration),
// --triangles (or the optional key "triangles":1 in 3D.json): obstacles are the TRIANGLES of the OBJ (`f` lines, tj_set_mesh)
// instead of its vertices as a point cloud.
// --gpus N / --devices a,b,.. (multi-UAV main only): the robots are sharded over N devices by the library (tj_group, trajadmm.h);
// the trajectory is bitwise the one-device one.
#include <chrono>
#include <sstream>
#include "../../include/trajadmm.h"
#include "cli_common.h"

#if defined(TJ_CLI_MULTI)
static const bool kMulti = true;
#else
static const bool kMulti = false;
#endif

int main(int argc, char** argv) {
  if (argc < 2) { std::cerr << "Syntax: " << argv[0] << " <mesh file> [--max-iter N] [--batch N] [--dump-state FILE] [--sample-traj FILE] [--triangles] [--gpus N | --devices a,b,..]" << std::endl; return -1; }
  const std::string mesh = argv[1];
  long max_iter = 1000000; int batch = 8; std::string dump, sample_file; bool triangles = false;
  std::vector<int> devices;   // empty: one context on device 0
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--max-iter" && i + 1 < argc) max_iter = atol(argv[++i]);
    else if (a == "--batch" && i + 1 < argc) batch = atoi(argv[++i]);
    else if (a == "--dump-state" && i + 1 < argc) dump = argv[++i];
    else if (a == "--sample-traj" && i + 1 < argc) sample_file = argv[++i];
    else if (a == "--triangles") triangles = true;
    else if (a == "--gpus" && i + 1 < argc) { const int n = atoi(argv[++i]); devices.clear(); for (int k = 0; k < n; k++) devices.push_back(k); }
    else if (a == "--devices" && i + 1 < argc) { devices.clear(); std::stringstream ss(argv[++i]); std::string t; while (std::getline(ss, t, ',')) devices.push_back(atoi(t.c_str())); }
    else { std::cerr << "unknown argument " << a << std::endl; return -1; }
  }
  tj_ctx* ctx = nullptr;
  tj_group* grp = nullptr;
  if (!kMulti && devices.size() > 1) { std::cerr << "error: a single UAV does not shard over devices" << std::endl; return 1; }
  try {
    auto j = tjcli::read_flat_json("Config_File/3D.json");
    const double lambda = tjcli::need(j, "lambda"), margin = tjcli::need(j, "margin"), offset = tjcli::need(j, "offset");
    const double mu = tjcli::need(j, "mu"), stop = tjcli::need(j, "stop"), vel = tjcli::need(j, "vel_limit"), acc = tjcli::need(j, "acc_limit");
    const int res = (int)tjcli::need(j, "res"), init = (int)tjcli::need(j, "init"), gui = (int)tjcli::need(j, "gui");
    const int optimal_plane = (int)tjcli::need(j, "optimal_plane"), if_exit = (int)tjcli::need(j, "exit"), init_ob = (int)tjcli::need(j, "init_ob");
    tjcli::need(j, "auto"); tjcli::need(j, "epsilon");
    const int decouple = (int)tjcli::need(j, "decouple");
    (void)if_exit;
    if (gui) throw std::runtime_error("gui:1 is not part of the accelerated path (use gui:0)");
    if (init != 1 && init != 2) throw std::runtime_error("init must be 1 (init/<mesh>_init_file.txt) or 2 (plan way points from start/goal pairs)");

    if (j.count("triangles") && j["triangles"] != 0) triangles = true;   // optional key (ours); the 16 reference keys stay mandatory
    const std::string model = std::string(kMulti ? "model/multiple/" : "model/single/") + mesh;
    std::vector<double> V; std::vector<int> F;
    if (triangles) tjcli::read_obj_mesh(model, V, F); else V = tjcli::read_obj_vertices(model);
    int U = 1, P = 0; std::vector<double> wp;
    if (kMulti) for (double& x : V) x *= 5;  // Main/multiPathPlanning3D.cpp:107
    const int N = init_ob ? (int)(triangles ? F.size() / 3 : V.size() / 3) : 0;
    auto set_obstacles = [&](tj_ctx* c) { return triangles ? tj_set_mesh(c, V.data(), (int)(V.size() / 3), F.data(), N) : tj_set_cloud(c, V.data(), N); };
    if (init == 1) {
      tjcli::read_waypoints("init/" + mesh + "_init_file.txt", kMulti, U, P, wp);
      if (kMulti) for (double& x : wp) x *= 5;  // :536
    } else {
      // "init":2 -- ompl_init (Main/multiPathPlanning3D.cpp:205-340, Main/admmPathPlanning3D.cpp:190-247) without OMPL: the
      // device planner of the library, in solver units, writing init/<mesh>_init_file.txt like the reference does
      std::vector<double> starts, goals;
      tjcli::read_start_goal("init/" + mesh + "_start_goal.txt", kMulti, starts, goals);
      U = (int)starts.size() / 3;
      tj_params pp; tj_ctx* pc = nullptr;
      tj_default_params(&pp, kMulti ? TJ_MODE_MULTI_DECOUPLE : TJ_MODE_SINGLE, U, 2);
      pp.margin = margin; pp.offset = offset;
      auto pchk = [&](int rc, const char* what) { if (rc < 0) { std::string m = std::string(what) + ": " + tj_last_error(pc); if (pc) tj_destroy(pc); throw std::runtime_error(m); } };
      pchk(tj_create(&pp, &pc), "tj_create");
      pchk(set_obstacles(pc), "tj_set_cloud / tj_set_mesh");
      const int cap = 256; int nw = 0;
      std::vector<double> buf((size_t)U * cap * 3);
      pchk(tj_plan_init(pc, U, starts.data(), goals.data(), 0.0, 0, 0, cap, buf.data(), &nw), "tj_plan_init");
      tj_destroy(pc);
      P = nw - 1;
      wp.assign((size_t)U * nw * 3, 0.0);
      for (int u = 0; u < U; u++) for (int k = 0; k < nw * 3; k++) wp[(size_t)u * nw * 3 + k] = buf[(size_t)u * cap * 3 + k];
      std::ofstream f("init/" + mesh + "_init_file.txt");   // one line per way point, all robots side by side (:324-333)
      for (int k = 0; k < nw; k++) { for (int u = 0; u < U; u++) for (int a = 0; a < 3; a++) f << wp[((size_t)u * nw + k) * 3 + a] << " "; f << std::endl; }
      std::cout << "ompl end\n";
    }
    if (kMulti) std::cout << "uav_num: " << U << "\n";
    std::cout << "time_obstacle build: " << N << (triangles ? " triangles" : " points") << std::endl;

    tj_params p;
    tj_default_params(&p, kMulti ? (decouple ? TJ_MODE_MULTI_DECOUPLE : TJ_MODE_MULTI_COUPLED) : TJ_MODE_SINGLE, U, P);
    p.optimal_plane = optimal_plane ? 1 : 0;  // persistent planes refined by Optimal_plane::optimal_cd / self_optimal_cd
    p.res = res; p.lambda = lambda; p.margin = margin; p.offset = offset; p.mu = mu; p.vel_limit = vel; p.acc_limit = acc; p.stop = stop;
    auto chk = [&](int rc, const char* what) { if (rc < 0) throw std::runtime_error(std::string(what) + ": " + (grp ? tj_group_last_error(grp) : tj_last_error(ctx))); };
    const bool group = devices.size() > 1;
    if (group) {
      if (tj_group_create(&p, (int)devices.size(), devices.data(), &grp) < 0) throw std::runtime_error(std::string("tj_group_create: ") + tj_group_last_error(nullptr));
      chk(triangles ? tj_group_set_mesh(grp, V.data(), (int)(V.size() / 3), F.data(), N) : tj_group_set_cloud(grp, V.data(), N), "tj_group_set_cloud / tj_group_set_mesh");
      chk(tj_group_init_state(grp, wp.data(), 20.0), "tj_group_init_state");
      std::cout << "devices: " << devices.size() << std::endl;
    } else {
      if (devices.size() == 1) p.device = devices[0];
      chk(tj_create(&p, &ctx), "tj_create");
      chk(set_obstacles(ctx), "tj_set_cloud / tj_set_mesh");
      chk(tj_init_state(ctx, wp.data(), 20.0), "tj_init_state");  // piece_time = 20 (admmPathPlanning3D.cpp:482)
    }
    auto get_state = [&](int u, double* s, double* pt) { return group ? tj_group_get_state(grp, u, s, nullptr, nullptr, nullptr, nullptr, pt) : tj_get_state(ctx, u, s, nullptr, nullptr, nullptr, nullptr, pt); };

    std::ofstream result("result/" + mesh + (kMulti ? "_result_file_multi.txt" : "_result_file_admm.txt"));
    double whole_ms = 0, gnorm = 1;
    int iter = 0, converged = 0;
    while (iter < max_iter && !converged) {
      const int n = (int)std::min<long>(batch, max_iter - iter);
      auto t0 = std::chrono::steady_clock::now();
      chk(group ? tj_group_iterate(grp, n, &gnorm, &iter, &converged) : tj_iterate(ctx, n, &gnorm, &iter, &converged), "tj_iterate");
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      whole_ms += ms;
      std::cout << "iter: " << iter << "\n" << "gnorm: " << gnorm << "\n" << "time:" << ms << std::endl;
    }
    if (converged) {
      result << "iter: " << iter << std::endl;
      result << "running time: " << whole_ms << std::endl;
      result << "point cloud size: " << V.size() / 3 << std::endl;
    }
    if (converged || !sample_file.empty()) {
      // log_data (Main/admmPathPlanning3D.cpp:33-77 called at :513; Main/multiPathPlanning3D.cpp:33-77 at :644-648).
      // Deviation: the multi main passes the never-updated global piece_time (20) in decoupled mode; each robot's
      // own piece_time is used here.
      const int T = 3 * P + 3;
      std::vector<double> conv((size_t)P * 36), s(3 * T);
      tj_host_tables(P, res, conv.data(), nullptr, nullptr, nullptr);
      std::ofstream sf;
      if (!sample_file.empty()) { sf.open(sample_file); sf.precision(17); }
      const double dt = kMulti ? 0.1 : 0.05;
      double whole_len = 0;
      for (int u = 0; u < U; u++) {
        double pt = 0, tt = 0, len = 0;
        chk(get_state(u, s.data(), &pt), "tj_get_state");
        std::vector<double> smp;
        tjcli::log_data(s.data(), P, conv.data(), pt, dt, tt, len, sample_file.empty() ? nullptr : &smp);
        std::cout << "ccd time:" << tt << std::endl << "ccd len:" << len << std::endl;
        whole_len += len;
        for (size_t k = 0; k + 2 < smp.size(); k += 3) sf << u << " " << (k / 3) * dt << " " << smp[k] << " " << smp[k + 1] << " " << smp[k + 2] << "\n";
      }
      (void)whole_len;
    }
    if (!dump.empty()) {
      std::ofstream df(dump);
      df.precision(17);
      const int T = 3 * P + 3;
      std::vector<double> s(3 * T); double pt = 0;
      df << "uav_num " << U << " piece_num " << P << " iter " << iter << " gnorm " << gnorm << " converged " << converged << "\n";
      for (int u = 0; u < U; u++) {
        chk(get_state(u, s.data(), &pt), "tj_get_state");
        df << "uav " << u << " piece_time " << pt << "\n";
        for (int r = 0; r < T; r++) df << s[r] << " " << s[r + T] << " " << s[r + 2 * T] << "\n";
      }
    }
    if (grp) tj_group_destroy(grp);
    if (ctx) tj_destroy(ctx);
    return converged ? 0 : 2;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    if (grp) tj_group_destroy(grp);
    if (ctx) tj_destroy(ctx);
    return 1;
  }
}
