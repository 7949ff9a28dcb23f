// libm_probe.hip -- development probe: how often do the device's log / sin / cos (ocml) return the bits glibc returns on the host?
// Reads doubles from a file (argv[1]: n raw doubles), writes log(x), sin(x), cos(x) for each (argv[2]).  The comparison is done by
// tests/devtools/libm_agreement.py.  Build on the GPU box: hipcc -O3 --offload-arch=gfx950 -o gpurun_out/libm_probe tools/micro/libm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out[3 * i] = log(x[i]); out[3 * i + 1] = sin(x[i]); out[3 * i + 2] = cos(x[i]); }
}
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  fseek(f, 0, SEEK_END); const long n = ftell(f) / 8; fseek(f, 0, SEEK_SET);
  std::vector<double> h(n), o(3 * n);
  if (fread(h.data(), 8, n, f) != (size_t)n) return 4; fclose(f);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
  hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dout, (int)n);
  hipMemcpy(o.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost);
  f = fopen(argv[2], "wb"); fwrite(o.data(), 8, 3 * n, f); fclose(f);
  return 0;
}
