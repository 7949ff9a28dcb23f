// kernels_bvh.h -- the obstacle BVH, built on the device.
//
// Replaces BVH::InitPointcloud / InitObstacle (HighOrderCCD/BVH/BVH.cpp:53-93, :15-51): the reference inserts the N
// primitives one by one into a dynamic AABB tree (insertLeaf + rotations, AABB.cc:846-1138: 95 ms for 20k points, minutes
// for 1M).  The tree SHAPE is free -- what the path needs is the candidate set of a box query -- so the device builds a
// static structure instead:
//   k_bvh_bounds      centroid of every primitive + grid-stride min/max reduction (order independent)
//   k_bvh_keys        63-bit Morton key of the centroid (21 bits per axis), value = the primitive's index
//   k_rsort_hist / k_rsort_scan / k_rsort_scatter
//                     least-significant-digit radix sort, 8 bits per pass, 8 passes, STABLE: equal keys keep their index
//                     order, i.e. the result is the lexicographic (key, index) order -- a function of the input alone
//   k_bvh_gather      primitives into sorted order (+ the fp32 outward-rounded box of every triangle)
//   k_bvh_level       boxes over 8 consecutive children, level by level: union in fp64, rounded outward to fp32 once
// Same expressions as the host build in host_tables.h (kept as the checker: TJ_BVH_HOST=1), hence the same bits.
//
// GPU shape: everything is a streaming pass over 1-2 words per primitive (HBM bound, coalesced); the only non-trivial
// kernel is the scatter, where each wave ranks its 64 keys by ballot matching (8 ballots give every lane the mask of lanes
// with the same digit) and the block turns per-wave digit counts into offsets in LDS -- no atomics, so the sort is stable.
#pragma once
#include "dev_common.h"

namespace tj {

constexpr int RS_THREADS = 256;          // 4 waves
constexpr int RS_ROUNDS = 8;             // a block sorts RS_THREADS * RS_ROUNDS consecutive elements of a pass
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;

__device__ __forceinline__ unsigned long long dev_spread21(unsigned long long v) {
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}
__device__ __forceinline__ float dev_f32_down(double x) { float f = (float)x; if ((double)f > x) f = nextafterf(f, -INFINITY); return f; }
__device__ __forceinline__ float dev_f32_up(double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, INFINITY); return f; }
__device__ __forceinline__ double prim_centroid_k(const double* v, int prim, int k) { return prim == 1 ? v[k] : (v[k] + v[3 + k] + v[6 + k]) / 3.0; }

// part[block][6] = min xyz, max xyz of the centroids of the block's grid-stride share
__global__ __launch_bounds__(256) void k_bvh_bounds(const double* verts, int n, int prim, double* part) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int k = 0; k < 3; k++) { const double c = prim_centroid_k(verts + (size_t)3 * prim * i, prim, k); lo[k] = fmin(lo[k], c); hi[k] = fmax(hi[k], c); }
  __shared__ double s[256][6];
  for (int k = 0; k < 3; k++) { s[threadIdx.x][k] = lo[k]; s[threadIdx.x][3 + k] = hi[k]; }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) for (int k = 0; k < 3; k++) { s[threadIdx.x][k] = fmin(s[threadIdx.x][k], s[threadIdx.x + off][k]); s[threadIdx.x][3 + k] = fmax(s[threadIdx.x][3 + k], s[threadIdx.x + off][3 + k]); }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[0][threadIdx.x];
}
__global__ void k_bvh_bounds_final(const double* part, int nblocks, double* lohi) {
  const int k = threadIdx.x;
  if (k >= 6) return;
  double r = part[k];
  for (int b = 1; b < nblocks; b++) r = k < 3 ? fmin(r, part[b * 6 + k]) : fmax(r, part[b * 6 + k]);
  lohi[k] = r;
}
__global__ __launch_bounds__(256) void k_bvh_keys(const double* verts, int n, int prim, const double* lohi, unsigned long long* key, int* val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long code = 0;
  for (int k = 0; k < 3; k++) {
    const double c = prim_centroid_k(verts + (size_t)3 * prim * i, prim, k);
    const double ext = lohi[3 + k] - lohi[k];
    const double f = ext > 0 ? (c - lohi[k]) / ext : 0.0;
    const unsigned long long q = (unsigned long long)fmin(2097151.0, fmax(0.0, f * 2097152.0));
    code |= dev_spread21(q) << k;
  }
  key[i] = code; val[i] = i;
}

// ---- stable LSD radix sort, one 8-bit digit per pass ----
// hist[bin * nblocks + block] = how many keys of the block's tile carry that digit
__global__ __launch_bounds__(RS_THREADS) void k_rsort_hist(const unsigned long long* key, int n, int shift, int nblocks, int* hist) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int base = blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ROUNDS; r++) {
    const int i = base + r * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(int)((key[i] >> shift) & 255ull)], 1);   // LDS atomics: counts only, order irrelevant
  }
  __syncthreads();
  hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of hist (bin-major, block-minor) in place: one block
__global__ __launch_bounds__(1024) void k_rsort_scan(int* hist, int total) {
  __shared__ int s[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < total; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < total ? hist[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < total) hist[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += s[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(RS_THREADS) void k_rsort_scatter(const unsigned long long* key, const int* val, int n, int shift, int nblocks, const int* offs,
                                                              unsigned long long* key_out, int* val_out) {
  __shared__ int wcount[RS_THREADS / 64][256];   // per wave: keys of this round with that digit
  __shared__ int run[256];                       // keys of earlier rounds of this tile with that digit
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  run[threadIdx.x] = 0;
  const int gbase = offs[threadIdx.x * nblocks + blockIdx.x];   // thread t owns digit t
  const int base = blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ROUNDS; r++) {
    for (int w = 0; w < RS_THREADS / 64; w++) wcount[w][threadIdx.x] = 0;
    __syncthreads();
    const int i = base + r * RS_THREADS + threadIdx.x;
    const bool live = i < n;
    const unsigned long long k = live ? key[i] : 0ull;
    const int d = (int)((k >> shift) & 255ull);
    // lanes of this wave with the same digit: intersect the 8 per-bit ballots
    unsigned long long same = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; b++) { const unsigned long long m = __ballot((d >> b) & 1); same &= ((d >> b) & 1) ? m : ~m; }
    const int rank = __popcll(same & ((1ull << lane) - 1ull));
    if (live && rank == 0) wcount[wave][d] = __popcll(same);   // the first lane of each digit group records the group size
    __syncthreads();
    if (live) {
      int before = run[d];
      for (int w = 0; w < wave; w++) before += wcount[w][d];
      const int pos = offs[d * nblocks + blockIdx.x] + before + rank;
      key_out[pos] = k; val_out[pos] = val[i];
    }
    __syncthreads();
    { int t = 0; for (int w = 0; w < RS_THREADS / 64; w++) t += wcount[w][threadIdx.x]; run[threadIdx.x] += t; }
    __syncthreads();
  }
  (void)gbase;
}

// sorted primitives: px/py/pz (prim 1) or tri[9] + fp32 box (prim 3); order[i] = original index
__global__ __launch_bounds__(256) void k_bvh_gather(const double* verts, const int* val, int n, int prim, double* px, double* py, double* pz, double* tri, float* leafbox) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* v = verts + (size_t)3 * prim * val[i];
  if (prim == 1) { px[i] = v[0]; py[i] = v[1]; pz[i] = v[2]; return; }
  for (int k = 0; k < 9; k++) tri[(size_t)i * 9 + k] = v[k];
  for (int k = 0; k < 3; k++) {
    double l = INFINITY, h = -INFINITY;
    for (int j = 0; j < 3; j++) { l = fmin(l, v[3 * j + k]); h = fmax(h, v[3 * j + k]); }
    leafbox[(size_t)i * 6 + k] = dev_f32_down(l); leafbox[(size_t)i * 6 + 3 + k] = dev_f32_up(h);
  }
}
// one level: box g over children [8g, 8g+8) of the level below (level 0: over sorted primitives).  cur64 = fp64 boxes of this
// level (scratch for the next one), out32 = the level's fp32 boxes inside the pyramid.
__global__ __launch_bounds__(256) void k_bvh_level(int level, int cnt, int nchild, int prim, const double* px, const double* py, const double* pz, const double* tri,
                                                   const double* child64, double* cur64, float* out32) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= cnt) return;
  double l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = 8 * g; i < min(nchild, 8 * g + 8); i++) {
    if (level == 0) {
      if (prim == 1) { const double p[3] = {px[i], py[i], pz[i]}; for (int k = 0; k < 3; k++) { l[k] = fmin(l[k], p[k]); h[k] = fmax(h[k], p[k]); } }
      else for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { const double c = tri[(size_t)i * 9 + 3 * j + k]; l[k] = fmin(l[k], c); h[k] = fmax(h[k], c); }
    } else for (int k = 0; k < 3; k++) { l[k] = fmin(l[k], child64[(size_t)i * 6 + k]); h[k] = fmax(h[k], child64[(size_t)i * 6 + 3 + k]); }
  }
  for (int k = 0; k < 3; k++) { cur64[(size_t)g * 6 + k] = l[k]; cur64[(size_t)g * 6 + 3 + k] = h[k]; out32[(size_t)g * 6 + k] = dev_f32_down(l[k]); out32[(size_t)g * 6 + 3 + k] = dev_f32_up(h[k]); }
}

}  // namespace tj
