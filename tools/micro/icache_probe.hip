// icache_probe.hip -- development probe (not part of the product): how much slower is straight-line code on its FIRST pass
// (instruction cache cold) than on later passes?  One wave per block runs a fully unrolled sequence of independent fp64 FMAs
// `reps` times and stamps s_memtime after every pass.  Build: hipcc -O3 --offload-arch=gfx950 -o icache_probe icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define F4(a,b,c,d) a = fma(a, x, y); b = fma(b, x, y); c = fma(c, x, y); d = fma(d, x, y);
#define F16(a,b,c,d) F4(a,b,c,d) F4(a,b,c,d) F4(a,b,c,d) F4(a,b,c,d)
#define F64(a,b,c,d) F16(a,b,c,d) F16(a,b,c,d) F16(a,b,c,d) F16(a,b,c,d)
#define F256(a,b,c,d) F64(a,b,c,d) F64(a,b,c,d) F64(a,b,c,d) F64(a,b,c,d)
#define F1K(a,b,c,d) F256(a,b,c,d) F256(a,b,c,d) F256(a,b,c,d) F256(a,b,c,d)
template <int KILO>
__global__ void probe(double* out, long long* stamps, int reps, double x, double y) {
  double a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  long long t = __builtin_readcyclecounter();
  if (threadIdx.x == 0) stamps[blockIdx.x * 16] = t;
  for (int r = 0; r < reps; r++) {
    if constexpr (KILO >= 1) { F1K(a, b, c, d) }
    if constexpr (KILO >= 2) { F1K(a, b, c, d) }
    if constexpr (KILO >= 4) { F1K(a, b, c, d) F1K(a, b, c, d) }
    if constexpr (KILO >= 8) { F1K(a, b, c, d) F1K(a, b, c, d) F1K(a, b, c, d) F1K(a, b, c, d) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) stamps[blockIdx.x * 16 + 1 + r] = t;
  }
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
template <int KILO>
void run(int blocks) {
  double* out; long long* st;
  hipMalloc(&out, blocks * 64 * 8); hipMalloc(&st, blocks * 16 * 8);
  for (int trial = 0; trial < 2; trial++) {
    hipMemset(st, 0, blocks * 16 * 8);
    hipLaunchKernelGGL(probe<KILO>, dim3(blocks), dim3(64), 0, 0, out, st, 4, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), st, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double pass[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; b++) for (int r = 0; r < 4; r++) pass[r] += double(h[b * 16 + 1 + r] - h[b * 16 + r]) / blocks;
    printf("%d k FMA (%d KB code), %4d blocks, launch %d: cycles per pass (s_memtime ticks, 100 MHz): first %.0f, then %.0f %.0f %.0f  -> per instruction first %.2f later %.2f ticks\n",
           KILO, KILO * 8, blocks, trial, pass[0], pass[1], pass[2], pass[3], pass[0] / (KILO * 1024.0), pass[3] / (KILO * 1024.0));
  }
  hipFree(out); hipFree(st);
}
int main() {
  run<1>(1); run<2>(1); run<4>(1); run<8>(1);
  run<4>(64); run<4>(320); run<8>(320);
  return 0;
}
