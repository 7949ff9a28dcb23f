// launch_probe.hip -- development probe: cost of a kernel boundary in a chain of dependent launches on one stream
// (the ADMM iteration is such a chain).  Empty kernels, kernels that dirty N MB of memory, grids of 1 / 64 / 2560 blocks.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_dirty(double* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i; }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double* buf; hipMalloc(&buf, 64 << 20);
  for (int grid : {1, 64, 2560}) {
    for (int rep = 0; rep < 2; rep++) {
      const int n = 2000;
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, s, (int*)nullptr);
      hipStreamSynchronize(s);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
      if (rep) printf("empty kernel, grid %5d x 64: %.2f us per dependent launch\n", grid, us);
    }
  }
  for (size_t mb : {0, 1, 4, 16}) {
    const int n = 1000;
    const size_t cnt = (mb << 20) / 8;
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) { if (cnt) hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, s, buf, cnt); hipLaunchKernelGGL(k_empty, dim3(64), dim3(64), 0, s, (int*)nullptr); }
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    printf("pair {write %zu MB, empty}: %.2f us per pair\n", mb, us);
  }
  return 0;
}
