// probe_clock.hip — shader clock under a VALU-heavy load (all CUs busy): s_memtime ticks per s_memrealtime tick (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_spin(float* out, unsigned long long* clk, int iters)
{
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) { a = a * b + c; d = d * b + a; c = c * 0.999f + d; }
  }
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main()
{
  const int nb = 1024 * 2, iters = 20000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, nb * 256 * 4); hipMalloc(&clk, nb * 16);
  for (int rep = 0; rep < 3; rep++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_spin, dim3(nb), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(2 * nb);
    hipMemcpy(h.data(), clk, nb * 16, hipMemcpyDeviceToHost);
    double s0 = 0, s1 = 0; for (int i = 0; i < nb; i++) { s0 += h[2 * i]; s1 += h[2 * i + 1]; }
    printf("rep %d: kernel %.3f ms, memtime ticks/block %.0f, realtime ticks/block %.0f, ratio %.3f -> %.1f MHz if realtime=100MHz; VALU instr/cycle/SIMD est: %.3f\n",
           rep, ms, s0 / nb, s1 / nb, s0 / s1, 100.0 * s0 / s1, 0.0);
  }
  int clkrate = 0, mclk = 0; hipDeviceGetAttribute(&clkrate, hipDeviceAttributeClockRate, 0); hipDeviceGetAttribute(&mclk, hipDeviceAttributeWallClockRate, 0);
  printf("hipDeviceAttributeClockRate %d kHz, WallClockRate %d kHz\n", clkrate, mclk);
  return 0;
}
