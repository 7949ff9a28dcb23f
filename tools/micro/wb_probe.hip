// wb_probe.hip -- development probe: what the END of a kernel costs by how many bytes the kernel stored (gfx950: every XCD's L2 is
// write-back, and the release at the end of a dispatch writes its dirty lines back before the next kernel of the queue starts).
// The ADMM iteration is a chain of six dependent kernels of 8 - 25 us that store 0.3 - 3.5 MB each (PMC table), so
// this is the price of the two caches the chain hands forward (swept-hull cache, hull cache).
// A pair of kernels is timed over many repetitions: k_store (512 blocks x 256 threads; every block stores its slice of X MB, then
// -- or before, `late` -- spins until 10 us have passed on the wall clock, so its own duration does not depend on X) followed by an
// empty dependent kernel.  Store flavours: plain, nontemporal, relaxed agent-scope atomic (write-through on this memory model).
// Build on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o gpurun_out/wb_probe tools/micro/wb_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__device__ __forceinline__ void st(double* p, double v) {
  if constexpr (MODE == 0) *p = v;
  else if constexpr (MODE == 1) __builtin_nontemporal_store(v, p);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_store(double* buf, size_t per_block, int late, int spin_ticks, double seed) {
  const long long t0 = wall_clock64();
  double* o = buf + (size_t)blockIdx.x * per_block;
  if (!late) for (size_t i = threadIdx.x; i < per_block; i += 256) st<MODE>(o + i, seed + i);
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(1);
  if (late) for (size_t i = threadIdx.x; i < per_block; i += 256) st<MODE>(o + i, seed + i);
}
__global__ void k_next(const double* buf, double* sink) { if (buf[0] == 12345.678) *sink = 1; }
// the dependent kernel with N doubles of private (scratch) memory per lane: a dynamically indexed array the compiler cannot keep in registers
template <int N>
__global__ __launch_bounds__(64) void k_next_scratch(const double* buf, double* sink, int j) {
  double a[N];
  for (int i = 0; i < N; i++) a[i] = buf[i] + i;
  asm volatile("" ::: "memory");
  double acc = 0;
  for (int i = 0; i < 4; i++) acc += a[(j + i * 7 + threadIdx.x) % N];
  if (acc == 12345.678) *sink = acc;
}
template <int N>
static double run_scratch(double* buf, double* sink, int blocks, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(k_store<0>, dim3(512), dim3(256), 0, 0, buf, (size_t)0, 0, 1000, (double)r);
      if constexpr (N == 0) hipLaunchKernelGGL(k_next, dim3(blocks), dim3(64), 0, 0, buf, sink);
      else hipLaunchKernelGGL(k_next_scratch<N>, dim3(blocks), dim3(64), 0, 0, buf, sink, r);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

template <int MODE>
static double run(double* buf, double* sink, size_t bytes, int late, int reps) {
  const size_t per_block = bytes / 8 / 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) {
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(k_store<MODE>, dim3(512), dim3(256), 0, 0, buf, per_block, late, 1000, (double)r);
      hipLaunchKernelGGL(k_next, dim3(1), dim3(64), 0, 0, buf, sink);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main() {
  double *buf, *sink; hipMalloc(&buf, 64ull << 20); hipMalloc(&sink, 64);
  const char* names[] = {"plain", "nontemporal", "atomic agent"};
  for (int late = 0; late < 2; late++) {
    printf("---- stores %s the 10 us spin; us per (k_store, k_next) pair ----\n", late ? "AFTER" : "BEFORE");
    printf("%-14s", "MB stored");
    for (double mb : {0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0}) printf("%8.2f", mb);
    printf("\n");
    for (int mode = 0; mode < 3; mode++) {
      printf("%-14s", names[mode]);
      for (double mb : {0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0}) {
        const size_t bytes = (size_t)(mb * 1048576);
        const double us = mode == 0 ? run<0>(buf, sink, bytes, late, 200) : mode == 1 ? run<1>(buf, sink, bytes, late, 200) : run<2>(buf, sink, bytes, late, 200);
        printf("%8.2f", us);
      }
      printf("\n");
    }
  }
  printf("---- the dependent kernel uses private (scratch) memory; us per (k_store of 0 MB, k_next) pair ----\n");
  printf("%-26s%10s%10s%10s%10s\n", "scratch bytes per lane", "0", "96", "416", "2048");
  for (int blocks : {1, 2881}) {
    printf("%5d blocks of 64        %10.2f%10.2f%10.2f%10.2f\n", blocks, run_scratch<0>(buf, sink, blocks, 200), run_scratch<12>(buf, sink, blocks, 200), run_scratch<52>(buf, sink, blocks, 200), run_scratch<256>(buf, sink, blocks, 200));
  }
  return 0;
}
