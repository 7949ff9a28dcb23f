// issue_probe.hip -- development probe: what ONE wavefront pays per instruction on gfx950 (the chain kernels of the ADMM
// iteration are single-wave dependent chains, so this is the unit their critical paths are priced in).
// Measures, with the 100 MHz wall clock and the shader clock, inside one wave of one block:
//   dependent / independent v_fma_f64, v_mul_f64 + v_add_f64, v_readlane -> v_fma (SGPR operand), v_rsq_f64, v_rcp_f64,
//   IEEE sqrt and division, v_cndmask pairs, DPP moves, LDS round trip, a second busy wave on the same SIMD.
// Build on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o gpurun_out/issue_probe tools/micro/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 2048;   // operations per measurement
#define STAMP(i) do { tw[i] = wall_clock64(); tc[i] = clock64(); } while (0)

__device__ __forceinline__ double rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

__global__ __launch_bounds__(256) void k_probe(double* out, long long* stamps, int nwaves_busy, double seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long tw[16], tc[16];
  double a = seed + lane * 1e-9, b = 1.0000001, c = 1e-7;
  __shared__ double lds[256];
  if (wave > 0) {   // optional co-resident busy waves (wave w sits on SIMD w & 3: waves 4.. share SIMD 0.. with wave 0..)
    if (wave <= nwaves_busy) { for (int i = 0; i < 40 * N; i++) a = fma(a, b, c); out[threadIdx.x] = a; }
    return;
  }
  STAMP(0);
#pragma unroll 16
  for (int i = 0; i < N; i++) a = fma(a, b, c);                       // dependent fma
  STAMP(1);
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
#pragma unroll 4
  for (int i = 0; i < N / 4; i++) { x0 = fma(x0, b, c); x1 = fma(x1, b, c); x2 = fma(x2, b, c); x3 = fma(x3, b, c); }   // 4 independent chains
  a = x0 + x1 + x2 + x3;
  STAMP(2);
#pragma unroll 16
  for (int i = 0; i < N / 2; i++) a = a * b + c;                      // dependent mul, add (contracted unless -ffp-contract=off)
  STAMP(3);
#pragma unroll 16
  for (int i = 0; i < N; i++) a = fma(a, rl(a, i & 31), c);           // readlane (2 x v_readlane_b32) -> fma through an SGPR pair
  STAMP(4);
#pragma unroll 16
  for (int i = 0; i < N / 8; i++) a = __builtin_amdgcn_rsq(a * a + 1.0);   // v_rsq_f64 (+ fma)
  STAMP(5);
#pragma unroll 16
  for (int i = 0; i < N / 8; i++) a = sqrt(a + 2.0);                  // IEEE sqrt sequence
  STAMP(6);
#pragma unroll 16
  for (int i = 0; i < N / 8; i++) { a = 1.0 / (a + 2.0); asm volatile("" : "+v"(a)); }   // IEEE division sequence (the opaque asm keeps every division: without it the
                                                                                         // compiler proved the recurrence's fixed point and round 3's row read 0.94 ns)
  STAMP(7);
  int sel = lane;
#pragma unroll 16
  for (int i = 0; i < N; i++) { a = (sel & 1) ? a : b + a; sel = sel * 3 + 1; asm volatile("" : "+v"(a)); }   // v_add_f64 + v_cndmask pair + integer ops
  STAMP(8);
#pragma unroll 16
  for (int i = 0; i < N / 4; i++) { lds[lane] = a; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); a = lds[(lane + 1) & 63] + c; }   // LDS write -> read round trip
  STAMP(9);
#pragma unroll 16
  for (int i = 0; i < N / 4; i++) a = log(a * a + 2.0);               // ocml log
  STAMP(10);
  out[lane] = a + sel;
  if (lane == 0) for (int i = 0; i <= 10; i++) { stamps[2 * i] = tw[i]; stamps[2 * i + 1] = tc[i]; }
}

int main() {
  double* out; long long* st;
  hipMalloc(&out, 4096); hipMalloc(&st, 64 * 8);
  const char* names[] = {"dependent v_fma_f64", "4 independent v_fma_f64 chains", "dependent v_mul_f64 + v_add_f64 (or fma)", "2 v_readlane + v_fma via SGPR (dependent)", "v_rsq_f64 + fma (dependent)",
                         "IEEE sqrt (dependent)", "IEEE division (dependent)", "v_add_f64 + cndmask pair + 2 int ops", "LDS write->read round trip", "log()"};
  const int ops[] = {N, N, N, N, N / 8, N / 8, N / 8, N, N / 4, N / 4};
  for (int busy : {0, 4, 7}) {
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, out, st, busy, 1.0);
      hipDeviceSynchronize();
    }
    std::vector<long long> h(64);
    hipMemcpy(h.data(), st, 64 * 8, hipMemcpyDeviceToHost);
    printf("---- %d busy co-resident waves (wave w on SIMD w & 3) ----\n", busy);
    for (int i = 0; i < 10; i++) {
      const double ns = (h[2 * (i + 1)] - h[2 * i]) * 10.0, cyc = (double)(h[2 * (i + 1) + 1] - h[2 * i + 1]);
      printf("%-44s %8.2f ns/op  %8.2f clk/op   (shader clock %.0f MHz)\n", names[i], ns / ops[i], cyc / ops[i], cyc / ns * 1e3);
    }
  }
  return 0;
}
