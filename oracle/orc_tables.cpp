// oracle/orc_tables.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Precomputed tables of the quintic (order 5) Bezier formulation, restated from
//   binomials           HighOrderCCD/Utils/CCDUtils.h:110-135  (Combination<40>)
//   C2 junction blocks  CCDUtils.h:137-170                      (Conversion<5>, all time weights 1)
//   jerk Gram matrix    CCDUtils.h:172-227                      (Dynamic3D<5,3>, +1e-8 I at :218)
//   blossom subdivision CCDUtils.h:229-315                      (Blossom<5>)
//   per-segment basis   Main/admmPathPlanning3D.cpp:305-314     (blossom(k/res,(k+1)/res) * convert)
//   49 k-DOP axes       HighOrderCCD/Utils/CCDUtils.cpp:56-119, normalised at admmPathPlanning3D.cpp:403-410
#include "orc.h"
#include <cmath>

namespace orc {

static const int KDOP_RAW[49][3] = {
    {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
    {1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {1, -1, -1},
    {0, 1, 1}, {0, 1, -1}, {1, 0, 1}, {1, 0, -1}, {1, 1, 0}, {1, -1, 0},
    {0, 2, 1}, {0, 2, -1}, {0, 1, 2}, {0, 1, -2},
    {2, 0, 1}, {2, 0, -1}, {1, 0, 2}, {1, 0, -2},
    {2, 1, 0}, {2, -1, 0}, {1, 2, 0}, {1, -2, 0},
    {1, 2, 1}, {1, 2, -1}, {1, -2, 1}, {-1, 2, 1},
    {1, 1, 2}, {1, 1, -2}, {1, -1, 2}, {-1, 1, 2},
    {2, 1, 1}, {2, 1, -1}, {2, -1, 1}, {-2, 1, 1},
    {2, 2, 1}, {2, 2, -1}, {2, -2, 1}, {-2, 2, 1},
    {2, 1, 2}, {2, 1, -2}, {2, -1, 2}, {-2, 1, 2},
    {1, 2, 2}, {1, 2, -2}, {1, -2, 2}, {-1, 2, 2}};

void Tables::build(int P_, int res_) {
  P = P_; res = res_; S = P * res;
  const int N = 5, K = 3;
  // binomials, integer recurrence
  long comb[41][41] = {{0}};
  comb[0][0] = 1;
  for (int i = 1; i <= 40; ++i) {
    long long t = 1;
    for (int j = 0; j <= i; ++j) { comb[i][j] = t; t = t * (i - j) / (j + 1); }
  }
  // junction conversion: identity except two 2x3 blocks (p = q = 1/2)
  convert.assign((size_t)P * 36, 0.0);
  for (int i = 0; i < P; i++) for (int d = 0; d < 6; d++) convert[i * 36 + d * 6 + d] = 1.0;
  const double p = 0.5, q = 0.5;
  const double I0[2][3] = {{q * q, 2 * p * q, p * p}, {0, q, p}};
  const double I1[2][3] = {{q, p, 0}, {q * q, 2 * p * q, p * p}};
  for (int i = 0; i < P - 1; i++)
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 3; c++) {
        convert[i * 36 + (N - 1 + r) * 6 + (N - 2 + c)] = I1[r][c];
        convert[(i + 1) * 36 + r * 6 + c] = I0[r][c];
      }
  // jerk Gram matrix
  for (int i = 0; i <= N; i++)
    for (int j = 0; j <= N; j++) {
      double acc = 0;
      for (int k0 = 0; k0 <= K; k0++)
        for (int k1 = 0; k1 <= K; k1++)
          if (i - k0 <= N - K && j - k1 <= N - K && i - k0 >= 0 && j - k1 >= 0) {
            double t = ((k0 + k1) % 2 == 0) ? 1 : -1;
            t *= comb[K][k0] * comb[K][k1] * comb[N - K][i - k0] * comb[N - K][j - k1] / (double)comb[2 * N - K - K][i + j - k0 - k1];
            for (int s = 0; s < K; s++) t *= (N - s) * (N - s);
            t /= (double)(2 * N - K - K + 1);
            acc += t;
          }
      Mdyn[i * 6 + j] = acc;
    }
  for (int d = 0; d < 6; d++) Mdyn[d * 6 + d] = Mdyn[d * 6 + d] + 1e-8 * 1.0;
  // blossom subdivision of [a,b] then fold in the junction conversion
  basis.assign((size_t)S * 36, 0.0);
  for (int k = 0; k < res; k++) {
    double t0 = k / double(res), t1 = (k + 1) / double(res);
    double pt0[6], pt1[6], qt0[6], qt1[6];
    double a0 = 1, a1 = 1, b0 = 1, b1 = 1;
    for (int i = 0; i <= N; ++i) {
      pt0[i] = a0; a0 *= t0; qt0[i] = b0; b0 *= 1 - t0;
      pt1[i] = a1; a1 *= t1; qt1[i] = b1; b1 *= 1 - t1;
    }
    double M[36] = {0};
    for (int i = 0; i <= N; ++i)
      for (int j = 0; j <= N; ++j) {
        if (i + j < N) {
          int mk = i < j ? i : j;
          for (int kk = 0; kk <= mk; ++kk)
            M[i * 6 + j] += comb[N - i][j - kk] * comb[i][kk] * qt0[N - i - j + kk] * qt1[i - kk] * pt0[j - kk] * pt1[kk];
        } else {
          int mk = (N - i) < (N - j) ? (N - i) : (N - j);
          for (int kk = 0; kk <= mk; ++kk)
            M[i * 6 + j] += comb[N - i][kk] * comb[i][N - j - kk] * qt0[kk] * qt1[N - j - kk] * pt0[N - i - kk] * pt1[i + j - N + kk];
        }
      }
    for (int i = 0; i < P; i++) {
      double* B = &basis[(size_t)(i * res + k) * 36];
      const double* C = &convert[i * 36];
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
          double acc = 0;
          for (int m = 0; m < 6; m++) acc += M[r * 6 + m] * C[m * 6 + c];
          B[r * 6 + c] = acc;
        }
    }
  }
  for (int k = 0; k < 49; k++) {
    double x = KDOP_RAW[k][0], y = KDOP_RAW[k][1], z = KDOP_RAW[k][2];
    double len = std::sqrt(x * x + y * y + z * z);
    kdop[k][0] = x / len; kdop[k][1] = y / len; kdop[k][2] = z / len;
  }
}

}  // namespace orc
