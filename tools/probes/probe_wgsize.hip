// tools/probes/probe_wgsize.hip — how many single-wave workgroups does a CU run at a time?  Every wave spins for a fixed number of clock
// cycles; the grid holds 12288 waves either as 12288 workgroups of 64 threads or as 3072 of 256, with and without an LDS footprint.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_wgsize.hip -o /tmp/probe_wgsize && /tmp/probe_wgsize
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_BYTES>
__global__ void spin(long long cycles, unsigned* sink)
{
  __shared__ unsigned char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
  if (LDS_BYTES > 0) lds[threadIdx.x] = (unsigned char)threadIdx.x;
  const long long t0 = wall_clock64();
  long long t;
  do { __builtin_amdgcn_s_sleep(8); t = wall_clock64(); } while (t - t0 < cycles);
  if (t == 0x7fffffffffffffffll) sink[0] = lds[0];
}

template <int LDS_BYTES>
static float run(int wg, int waves, long long cyc)
{
  unsigned* d; hipMalloc(&d, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(spin<LDS_BYTES>, dim3(waves * 64 / wg), dim3(wg), 0, 0, cyc, d);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(spin<LDS_BYTES>, dim3(waves * 64 / wg), dim3(wg), 0, 0, cyc, d);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipFree(d);
  return ms * 1e3f;
}

int main()
{
  const int waves = 12288;
  const long long cyc = 800;    // wall_clock64 runs at 100 MHz: 800 ticks = 8 us
  printf("12288 waves, each alive for 8 us (ideal with every wave resident at once: ~8-16 us)\n");
  printf("wg  64, no LDS      : %8.1f us\n", run<0>(64, waves, cyc));
  printf("wg 256, no LDS      : %8.1f us\n", run<0>(256, waves, cyc));
  printf("wg  64, 7.5 KB LDS  : %8.1f us\n", run<7548>(64, waves, cyc));
  printf("wg 256, 30 KB LDS   : %8.1f us\n", run<30192>(256, waves, cyc));
  printf("wg  64, 20 KB LDS   : %8.1f us\n", run<20480>(64, waves, cyc));
  printf("wg 128, 15 KB LDS   : %8.1f us\n", run<15096>(128, waves, cyc));
  return 0;
}
