// crmath_check.cpp -- TEST INFRASTRUCTURE: runs the host build of csrc/dev_crmath.h on a file of doubles.
//   crmath_check <in.bin> <out.bin>: for every x: cr_log(x), cr_sin(x), cr_cos(x), glibc log / sin / cos
// Compared against 80-digit values by tests/test_crmath.py (CPU) -- the device executes the same IEEE operations.
#include <cstdio>
#include <vector>
#include "../../traj-opt-admm_amd/csrc/dev_crmath.h"
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  fseek(f, 0, SEEK_END); const long n = ftell(f) / 8; fseek(f, 0, SEEK_SET);
  std::vector<double> x(n), o(6 * n);
  if (fread(x.data(), 8, n, f) != (size_t)n) return 4;
  fclose(f);
  for (long i = 0; i < n; i++) {
    double s, c; tj::cr_sincos(x[i], &s, &c);
    o[6 * i] = x[i] > 0 ? tj::cr_log(x[i]) : 0.0; o[6 * i + 1] = s; o[6 * i + 2] = c;
    o[6 * i + 3] = x[i] > 0 ? log(x[i]) : 0.0; o[6 * i + 4] = sin(x[i]); o[6 * i + 5] = cos(x[i]);
  }
  f = fopen(argv[2], "wb"); fwrite(o.data(), 8, 6 * n, f); fclose(f);
  return 0;
}
