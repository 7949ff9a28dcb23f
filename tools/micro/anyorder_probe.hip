// anyorder_probe.hip -- development probe: does a kernel launched with hipExtAnyOrderLaunch start while its predecessor on the SAME stream
// is still running (AQL barrier bit cleared)?  hip_ext.h says the flag is not supported on GFX9xx boards; this measures it.
// Kernel A spins ~40 us and stamps its start / end; kernel B (one block) stamps its start.  B.start < A.end <=> the launches overlapped.
// Build on the GPU box:  hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o gpurun_out/anyorder_probe tools/micro/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void k_a(long long* st, int ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (blockIdx.x == 0 && threadIdx.x == 0) { st[0] = t0; st[1] = wall_clock64(); }
}
__global__ void k_b(long long* st) { if (threadIdx.x == 0) st[2] = wall_clock64(); }

int main() {
  long long* st; hipMalloc(&st, 64);
  long long h[3];
  for (int flags = 0; flags < 2; flags++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(st, 0, 64); hipDeviceSynchronize();
      hipLaunchKernelGGL(k_a, dim3(64), dim3(64), 0, 0, st, 4000);
      hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, 0, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, st);
      hipDeviceSynchronize();
      hipMemcpy(h, st, 24, hipMemcpyDeviceToHost);
      printf("flags=%d  A ran %.1f us; B started %.1f us after A started (%s)\n", flags, (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, h[2] < h[1] ? "OVERLAPPED" : "after A ended");
    }
  }
  return 0;
}
