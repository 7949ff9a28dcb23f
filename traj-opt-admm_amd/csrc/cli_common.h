// cli_common.h -- file formats of the reference's two command-line tools, kept verbatim so the
// binaries are drop-ins on the same working directory layout:
//   Config_File/3D.json          16 flat numeric keys (Main/admmPathPlanning3D.cpp:368-397)
//   model/{single,multiple}/<m>  OBJ, only `v` lines are used, reading stops at the first other
//                                line once more than 10 vertices were seen (CCDUtils.h:317-391)
//   init/<m>_init_file.txt       way points (admmPathPlanning3D.cpp:79-112, multiPathPlanning3D.cpp:78-121)
//   result/<m>_result_file_*.txt iter / running time / point cloud size (admmPathPlanning3D.cpp:507-510)
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace tjcli {

// flat JSON object with numeric values
inline std::map<std::string, double> read_flat_json(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::stringstream ss; ss << f.rdbuf();
  const std::string s = ss.str();
  std::map<std::string, double> out;
  size_t i = 0;
  while ((i = s.find('"', i)) != std::string::npos) {
    size_t j = s.find('"', i + 1);
    if (j == std::string::npos) break;
    std::string key = s.substr(i + 1, j - i - 1);
    size_t c = s.find(':', j);
    if (c == std::string::npos) break;
    char* end = nullptr;
    double v = std::strtod(s.c_str() + c + 1, &end);
    if (end == s.c_str() + c + 1) throw std::runtime_error("bad value for key " + key + " in " + path);
    out[key] = v;
    i = (size_t)(end - s.c_str());
  }
  return out;
}

inline double need(const std::map<std::string, double>& j, const char* key) {
  auto it = j.find(key);
  if (it == j.end()) throw std::runtime_error(std::string("3D.json: missing key \"") + key + "\"");  // all 16 keys are mandatory in the reference
  return it->second;
}

inline std::vector<double> read_obj_vertices(const std::string& path) {
  FILE* fp = fopen(path.c_str(), "r");
  if (!fp) throw std::runtime_error("cannot open " + path);
  std::vector<double> v;
  char line[2048], type[2048];
  int count = 0;
  while (fgets(line, sizeof(line), fp)) {
    if (sscanf(line, "%s", type) != 1) continue;
    if (strcmp(type, "v") == 0) {
      std::istringstream ls(line + 1);
      double x, y, z;
      if (ls >> x >> y >> z) { v.push_back(x); v.push_back(y); v.push_back(z); count++; }
    } else if (count > 10) break;
  }
  fclose(fp);
  return v;
}

// Triangle-mesh front end (ours: `--triangles` / "triangles":1 in 3D.json; BASELINE config 5).  The reference's reader drops
// the faces (above) although the library it links has a triangle broad phase (BVH::InitObstacle, BVH.cpp:15-51).  Reads the
// WHOLE file: every `v x y z` and every `f` line (1-based or negative indices, `i`, `i/t`, `i//n`, `i/t/n` tokens; polygons
// are fan-triangulated).  F holds 0-based vertex indices, 3 per triangle.
inline void read_obj_mesh(const std::string& path, std::vector<double>& V, std::vector<int>& F) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  V.clear(); F.clear();
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ls(line);
    std::string type;
    if (!(ls >> type)) continue;
    if (type == "v") {
      double x, y, z;
      if (ls >> x >> y >> z) { V.push_back(x); V.push_back(y); V.push_back(z); }
    } else if (type == "f") {
      std::vector<int> idx; std::string tok;
      const int nv = (int)(V.size() / 3);
      while (ls >> tok) {
        const long i = strtol(tok.c_str(), nullptr, 10);   // stops at the first '/'
        if (i == 0) throw std::runtime_error(path + ": bad face token '" + tok + "'");
        const long v = i > 0 ? i - 1 : nv + i;
        if (v < 0 || v >= nv) throw std::runtime_error(path + ": face refers to vertex " + std::to_string(i) + " of " + std::to_string(nv));
        idx.push_back((int)v);
      }
      for (size_t k = 1; k + 1 < idx.size(); k++) { F.push_back(idx[0]); F.push_back(idx[k]); F.push_back(idx[k + 1]); }
    }
  }
}

// single: one "x y z" per line.  multi: 3*U numbers per line, U from the first line.
inline void read_waypoints(const std::string& path, bool multi, int& U, int& P, std::vector<double>& wp /*[U][P+1][3]*/) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<std::vector<double>> rows;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    std::vector<double> r; double x;
    while (is >> x) r.push_back(x);
    if (!r.empty()) rows.push_back(r);
  }
  if (rows.size() < 3) throw std::runtime_error(path + ": need at least 3 way points");
  U = multi ? (int)rows[0].size() / 3 : 1;
  if (U < 1) throw std::runtime_error(path + ": first line has fewer than 3 numbers");
  P = (int)rows.size() - 1;
  wp.assign((size_t)U * (P + 1) * 3, 0.0);
  for (int k = 0; k <= P; k++) {
    if ((int)rows[k].size() < 3 * U) throw std::runtime_error(path + ": short line");
    for (int u = 0; u < U; u++) for (int a = 0; a < 3; a++) wp[((size_t)u * (P + 1) + k) * 3 + a] = rows[k][3 * u + a];
  }
}

// "init":2 -- start/goal pairs of the planner.  The reference hard-codes them (single: Main/admmPathPlanning3D.cpp:222-228,
// multi: four robots, Main/multiPathPlanning3D.cpp:251-267); init/<mesh>_start_goal.txt (ours: one "sx sy sz gx gy gz" line
// per robot, solver units) overrides them.
inline void read_start_goal(const std::string& path, bool multi, std::vector<double>& starts, std::vector<double>& goals) {
  starts.clear(); goals.clear();
  std::ifstream f(path);
  if (f) {
    double v[6];
    while (f >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5]) { starts.insert(starts.end(), v, v + 3); goals.insert(goals.end(), v + 3, v + 6); if (!multi) break; }
    if (starts.empty()) throw std::runtime_error(path + ": need six numbers per robot");
    return;
  }
  if (!multi) { starts = {2.7, 0, 0}; goals = {-2.7, 0, 0}; return; }
  starts = {2.5, 1.7, 0.5, 2.5, 1.7, -0.5, -2.5, 1.7, 0.5, -2.5, 1.7, -0.5};
  goals = {-2.5, 1.7, 0.5, -2.5, 1.7, -0.5, 2.5, 1.7, -0.5, 2.5, 1.7, 0.5};
}

// log_data (Main/admmPathPlanning3D.cpp:33-77, Main/multiPathPlanning3D.cpp:33-77): duration of a trajectory and
// the length of its polyline sampled every `dt` seconds of flight time (0.05 single, 0.1 multi), evaluated on the
// per-piece Bezier control points C_i x_i exactly like getPosFromBezier (:17-31).  spline is T x 3 column-major,
// convert is [P][36] row-major.  samples (optional) receives the sampled positions, 3 per point.
inline void log_data(const double* spline, int P, const double* convert, double piece_time, double dt,
                     double& time_out, double& len_out, std::vector<double>* samples = nullptr) {
  const int T = 3 * P + 3;
  static const double binom5[6] = {1, 5, 10, 10, 5, 1};
  std::vector<double> coeff((size_t)P * 18);  // [piece][axis][6]
  for (int i = 0; i < P; i++)
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 6; j++) {
        double acc = 0;
        for (int k = 0; k < 6; k++) acc += convert[(size_t)i * 36 + j * 6 + k] * spline[3 * i + k + T * a];
        coeff[(size_t)i * 18 + a * 6 + j] = acc;
      }
  time_out = 0;
  for (int i = 0; i < P; i++) time_out += 1.0 * piece_time;  // time_weight == 1 everywhere
  len_out = 0;
  double prev[3] = {0, 0, 0};
  bool have = false;
  for (double t = 0.0; t < P; t += dt / piece_time) {
    const int i = (int)std::floor(t);
    const double ct = t - i;
    double cur[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 6; j++) cur[a] += binom5[j] * coeff[(size_t)i * 18 + a * 6 + j] * std::pow(ct, j) * std::pow(1 - ct, 5 - j);
    if (have) len_out += std::sqrt((cur[0] - prev[0]) * (cur[0] - prev[0]) + (cur[1] - prev[1]) * (cur[1] - prev[1]) + (cur[2] - prev[2]) * (cur[2] - prev[2]));
    if (samples) { samples->push_back(cur[0]); samples->push_back(cur[1]); samples->push_back(cur[2]); }
    for (int a = 0; a < 3; a++) prev[a] = cur[a];
    have = true;
  }
}

}  // namespace tjcli
