// probe_l2rw.hip — does a wave that writes a region with vector stores and reads it back through the scalar cache get the data from L2
// (no HBM fetch)?  4096 waves (4 per workgroup), each owns a private 57 KB region; per "symbol" it writes 1248 bytes (scattered 4-byte
// stores like the de-interleaver, or coalesced), waits vmcnt(0), and s_loads the completed 128-byte lines back.  Run under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_MISS_sum: mode 0 = write only, 1 = write + scalar read-back, 2 = write + vector read-back,
// 3 = coalesced 16-byte stores + scalar read-back, 4 = coalesced stores only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const float __attribute__((address_space(4))) c4float;
template <int MODE> __global__ __launch_bounds__(256) void k_probe(float* buf, float* sink, int nsym)
{
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63;
  float* out = buf + (size_t)w * 14592;           // 57 KB + pad, multiple of 32 floats
  float acc = 0.f;
  int done = 0, produced = 0;
  for (int s = 0; s < nsym; s++) {
    if (MODE >= 3) {                                    // coalesced: 16-byte stores, consecutive lanes consecutive addresses
      if (lane < 78) ((float4*)(out + s * 312))[lane] = make_float4((float)s, (float)lane, 1.f, 2.f);
      if (lane < 14) ((float4*)(out + s * 312))[64 + lane] = make_float4((float)s, (float)lane, 1.f, 2.f);
    } else
    if (lane < 52) for (int b = 0; b < 6; b++) out[s * 312 + ((lane * 6 + b) * 37) % 312] = (float)(s + lane + b);   // scattered like a de-interleaver
    produced += 312;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int safe = s + 1 == nsym ? produced : (produced & ~31);
    if (MODE == 1 || MODE == 3) {
      c4float* L = (c4float*)out;
      for (; done + 16 <= safe; done += 16) {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = L[done + i];
#pragma unroll
        for (int i = 0; i < 16; i++) acc += x[i];
      }
    } else if (MODE == 2) {
      for (; done + 64 <= safe; done += 64) acc += out[done + lane];
    }
  }
  if (acc == 12345.678f) sink[w] = acc;
}
int main(int argc, char** argv)
{
  float *buf, *sink; const int nw = 4096;
  hipMalloc(&buf, (size_t)nw * 14592 * 4); hipMalloc(&sink, nw * 4);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_probe<0>, dim3(nw / 4), dim3(256), 0, 0, buf, sink, 46);
    hipLaunchKernelGGL(k_probe<1>, dim3(nw / 4), dim3(256), 0, 0, buf, sink, 46);
    hipLaunchKernelGGL(k_probe<2>, dim3(nw / 4), dim3(256), 0, 0, buf, sink, 46);
    hipLaunchKernelGGL(k_probe<3>, dim3(nw / 4), dim3(256), 0, 0, buf, sink, 46);
    hipLaunchKernelGGL(k_probe<4>, dim3(nw / 4), dim3(256), 0, 0, buf, sink, 46);
  }
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
