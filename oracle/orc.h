// oracle/orc.h -- TEST INFRASTRUCTURE ONLY.  Declarations shared by the CPU restatement
// (oracle/orc_*.cpp -> oracle/liboracle.so).  Nothing under traj-opt-admm_amd/ includes
// this header; the product has its own HIP implementation.
#pragma once
#include <cstdint>
#include <vector>

namespace orc {

// ---- GJK (orc_gjk.cpp; reference lib/opengjk/src/openGJK.c:754) ----
void gjk(const double* p1, int n1, const double* p2, int n2, double* v_out, int* iters = nullptr);

// ---- dynamic AABB tree (orc_tree.cpp; reference HighOrderCCD/BVH/src/AABB.cc) ----
struct Box3 { double lo[3], hi[3]; };
class DynTree {
 public:
  static constexpr uint32_t NIL = 0xffffffffu;
  explicit DynTree(uint32_t capacity_hint = 16);
  void insert(uint32_t particle, const Box3& box);                                   // AABB.cc:389-443
  void query(const Box3& q, double margin, std::vector<uint32_t>& out) const;        // AABB.cc:608-667
  void self_query(double margin, std::vector<std::pair<uint32_t, uint32_t>>& out) const;  // AABB.cc:669-734
  uint32_t size() const { return count_; }
 private:
  struct Node { Box3 box; double area; uint32_t parent, left, right, next; int height; uint32_t particle; };
  std::vector<Node> nodes_;
  uint32_t root_, free_, count_, cap_;
  uint32_t alloc();
  void insert_leaf(uint32_t leaf);
  uint32_t balance(uint32_t node);
  static double area(const Box3& b);
  static void merge(Box3& out, double& out_area, const Box3& a, const Box3& b);
  bool leaf(uint32_t n) const { return nodes_[n].left == NIL; }
};

// ---- Eigen's SimplicialLLT with AMD ordering on the sparse view of a dense symmetric matrix (orc_amd.cpp) ----
void amd_order(int n, const std::vector<int>& colptr, const std::vector<int>& rowidx, std::vector<int>& order);
bool sparse_llt_solve(int n, const double* H, const double* b, double* x, int* order_out = nullptr);

// ---- tables (orc_tables.cpp; reference HighOrderCCD/Utils/CCDUtils.h:110-315) ----
// All 6x6 matrices are row-major m[row*6+col] in the oracle.
struct Tables {
  int P = 0, res = 0, S = 0;
  std::vector<double> convert;   // P x 36
  double Mdyn[36];
  std::vector<double> basis;     // S x 36  (blossom(k/res,(k+1)/res) * convert[piece]),  seg = piece*res + k
  double kdop[49][3];            // normalised 49 k-DOP axes (CCDUtils.cpp:56-119)
  void build(int P, int res);
};

}  // namespace orc
