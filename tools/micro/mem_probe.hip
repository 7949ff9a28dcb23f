// mem_probe.hip -- development probe: what ONE wavefront pays for a DEPENDENT global-memory round trip on gfx950, by where the
// line is found (bench.py prices every `mem` step of roofline.critical_path with these rows), and for a scalar
// branch on a fresh VALU comparison.  One wave of one block chases a pointer chain (every hop = one 8-byte load whose address is
// the value of the previous one; each entry on its own 128-byte line; a random single-cycle permutation):
//   l2        4 096 lines (512 KB), chased twice; the second pass is timed: hits in the XCD's own L2
//   mall      160 MB of lines touched once by a streaming kernel (the L2s are 4 MB per XCD, the Infinity Cache 256 MB),
//             then chased: L2 misses served by the Infinity Cache (MALL)
//   hbm       1.5 GB of lines, chased after a 1 GB flush: misses everywhere
//   xcd       the chain's lines are WRITTEN by blocks of a kernel that sit on the other XCDs (block b -> XCD b % 8), then chased by
//             block 0 of the NEXT kernel on the same queue: the situation of every chain kernel
//             of the ADMM iteration that reads what the previous kernel produced (work lists, hull caches, plane slots)
//   xcd_same  the same through the entries whose writer block sat on XCD 0 as well
// and, for the branch: a loop whose body is selected by s_cbranch on v_cmp + readfirstlane of the value just computed.
// Build on the GPU box:  hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o gpurun_out/mem_probe tools/micro/mem_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <numeric>
#include <random>
#include <vector>

constexpr int LINE = 16;   // 8-byte words per 128-byte line

// plain (cached) loads: what the kernels' own loads are
__global__ __launch_bounds__(64) void k_chase_cached(const unsigned long long* buf, unsigned long long start, int hops, long long* stamps, unsigned long long* sink) {
  if (blockIdx.x != 0) return;
  unsigned long long idx = start;
  const long long t0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < hops; i++) { idx = buf[idx * LINE]; asm volatile("" : "+v"(idx)); }
  const long long t1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = c1 - c0; *sink = idx; }
}
// writer: entry i is written by block i % gridDim.x (64 blocks: block b sits on XCD b % 8)
__global__ __launch_bounds__(64) void k_write(unsigned long long* buf, const unsigned long long* next, int n) {
  for (int i = blockIdx.x + gridDim.x * threadIdx.x; i < n; i += gridDim.x * 64) buf[(size_t)i * LINE] = next[i];
}
__global__ void k_touch(const unsigned long long* buf, size_t nlines, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nlines; i += (size_t)gridDim.x * blockDim.x) acc += buf[i * LINE];
  if (acc == 0x1234567ull) *sink = acc;
}
// branch probe: N iterations, each a v_cmp on the fresh value -> s_cbranch -> one of two fma bodies; against the same loop with a select
__global__ __launch_bounds__(64) void k_branch(double* out, long long* stamps, double seed, int n) {
  double a = seed + (threadIdx.x & 63) * 1e-9;
  const double b = 1.0000001, c = 1e-7;
  long long t0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < n; i++) {
    const bool up = __builtin_amdgcn_readfirstlane((int)(a > 1.5)) != 0;   // wave-uniform decision on the value just computed
    if (up) { a = fma(a, 0.5, c); asm volatile("" : "+v"(a)); } else { a = fma(a, b, 0.25); asm volatile("" : "+v"(a)); }
  }
  long long t1 = wall_clock64(), c1 = clock64();
  stamps[0] = t1 - t0; stamps[1] = c1 - c0;
  t0 = wall_clock64(); c0 = clock64();
  for (int i = 0; i < n; i++) {
    const double x = fma(a, 0.5, c), y = fma(a, b, 0.25);
    a = a > 1.5 ? x : y; asm volatile("" : "+v"(a));
  }
  t1 = wall_clock64(); c1 = clock64();
  stamps[2] = t1 - t0; stamps[3] = c1 - c0;
  // IEEE division and sqrt, each depending on the one before (the compiler cannot fold: the value passes through an opaque asm)
  t0 = wall_clock64(); c0 = clock64();
  for (int i = 0; i < n; i++) { a = 1.0 / (a + 2.0); asm volatile("" : "+v"(a)); }
  t1 = wall_clock64(); c1 = clock64();
  stamps[4] = t1 - t0; stamps[5] = c1 - c0;
  t0 = wall_clock64(); c0 = clock64();
  for (int i = 0; i < n; i++) { a = sqrt(a + 2.0); asm volatile("" : "+v"(a)); }
  t1 = wall_clock64(); c1 = clock64();
  stamps[6] = t1 - t0; stamps[7] = c1 - c0;
  out[threadIdx.x] = a;
}

static std::vector<unsigned long long> cycle(size_t n, unsigned seed) {   // next[] of a random single cycle over n entries
  std::vector<unsigned long long> perm(n), next(n);
  std::iota(perm.begin(), perm.end(), 0ull);
  std::mt19937_64 rng(seed);
  std::shuffle(perm.begin(), perm.end(), rng);
  for (size_t i = 0; i < n; i++) next[perm[i]] = perm[(i + 1) % n];
  return next;
}

int main() {
  long long* st; unsigned long long* sink; double* out;
  hipMalloc(&st, 64 * 8); hipMalloc(&sink, 64); hipMalloc(&out, 64 * 8);
  long long h[8];
  auto report = [&](const char* name, int hops) {
    hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f ns/hop  %8.1f clk/hop\n", name, h[0] * 10.0 / hops, (double)h[1] / hops);
  };
  auto build = [&](size_t nlines, unsigned seed, unsigned long long** dbuf) {
    std::vector<unsigned long long> next = cycle(nlines, seed), hostbuf(nlines * LINE, 0);
    for (size_t i = 0; i < nlines; i++) hostbuf[i * LINE] = next[i];
    hipMalloc(dbuf, nlines * LINE * 8);
    hipMemcpy(*dbuf, hostbuf.data(), nlines * LINE * 8, hipMemcpyHostToDevice);
  };
  {   // L2 hit
    unsigned long long* b; build(4096, 1, &b);
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k_chase_cached, dim3(1), dim3(64), 0, 0, b, 0ull, 4096, st, sink); hipDeviceSynchronize(); }
    hipLaunchKernelGGL(k_chase_cached, dim3(1), dim3(64), 0, 0, b, 0ull, 4096, st, sink); hipDeviceSynchronize();
    report("l2: 512 KB chain, warm (own XCD's L2), cached loads", 4096);
    hipFree(b);
  }
  {   // MALL hit
    const size_t nl = 160ull * 1024 * 1024 / 128;
    unsigned long long* b; build(nl, 2, &b);
    hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, b, nl, sink); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chase_cached, dim3(1), dim3(64), 0, 0, b, 0ull, 20000, st, sink); hipDeviceSynchronize();
    report("mall: 160 MB chain streamed once before (L2 miss, Infinity Cache)", 20000);
    hipFree(b);
  }
  {   // HBM
    const size_t nl = 1536ull * 1024 * 1024 / 128;
    unsigned long long* b; build(nl, 3, &b);
    unsigned long long* fl; const size_t fln = 1024ull * 1024 * 1024 / 128; hipMalloc(&fl, fln * 128); hipMemset(fl, 1, fln * 128);
    hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, 0, fl, fln, sink); hipDeviceSynchronize();
    hipLaunchKernelGGL(k_chase_cached, dim3(1), dim3(64), 0, 0, b, 0ull, 20000, st, sink); hipDeviceSynchronize();
    report("hbm: 1.5 GB chain after a 1 GB flush (miss everywhere)", 20000);
    hipFree(b); hipFree(fl);
  }
  for (int same = 0; same < 2; same++) {   // lines written by the previous kernel on other XCDs / on the same XCD
    const int n = 4096;
    // the chain runs through the entries whose WRITER block sits on XCD 0 (same) or on XCDs 1..7; every entry is written
    std::vector<int> members;
    for (int i = 0; i < n; i++) if ((((i % 64) % 8) == 0) == (same != 0)) members.push_back(i);
    std::vector<unsigned long long> sub = cycle(members.size(), 4 + same), next(n, 0);
    for (size_t j = 0; j < members.size(); j++) next[members[j]] = (unsigned long long)members[sub[j]];
    const int hops = (int)members.size();
    unsigned long long *b, *dn; hipMalloc(&b, (size_t)n * LINE * 8); hipMalloc(&dn, n * 8);
    hipMemcpy(dn, next.data(), n * 8, hipMemcpyHostToDevice);
    double best = 1e30, sum = 0; const int reps = 20;
    for (int rep = 0; rep < reps; rep++) {
      hipMemset(b, 0, (size_t)n * LINE * 8);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(k_write, dim3(64), dim3(64), 0, 0, b, dn, n);
      hipLaunchKernelGGL(k_chase_cached, dim3(8), dim3(64), 0, 0, b, (unsigned long long)members[0], hops, st, sink);   // block 0 (XCD 0) chases
      hipDeviceSynchronize();
      hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
      best = std::min(best, h[0] * 10.0 / hops); sum += h[0] * 10.0 / hops;
    }
    printf("%-58s %8.1f ns/hop (best of %d; mean %.1f)\n", same ? "xcd_same: lines written by the previous kernel on XCD 0" : "xcd: lines written by the previous kernel on XCDs 1..7", best, reps, sum / reps);
    hipFree(b); hipFree(dn);
  }
  {
    const int n = 4096;
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k_branch, dim3(1), dim3(64), 0, 0, out, st, 1.0, n); hipDeviceSynchronize(); }
    hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
    printf("%-58s %8.2f ns/iter %8.2f clk/iter\n", "branch: v_cmp + readfirstlane + s_cbranch + one fma", h[0] * 10.0 / n, (double)h[1] / n);
    printf("%-58s %8.2f ns/iter %8.2f clk/iter\n", "select: two fma + v_cmp + v_cndmask (same recurrence)", h[2] * 10.0 / n, (double)h[3] / n);
    printf("%-58s %8.2f ns/op   %8.2f clk/op\n", "IEEE division, dependent (a = 1 / (a + 2))", h[4] * 10.0 / n, (double)h[5] / n);
    printf("%-58s %8.2f ns/op   %8.2f clk/op\n", "IEEE sqrt, dependent (a = sqrt(a + 2))", h[6] * 10.0 / n, (double)h[7] / n);
  }
  return 0;
}
