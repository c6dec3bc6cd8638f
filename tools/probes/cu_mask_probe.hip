// tools/probes/cu_mask_probe.hip — which CUs of which XCD does a hipExtStreamCreateWithCUMask mask select?  hipcc --offload-arch=gfx950 -O2 -o tools/probes/cu_mask_probe tools/probes/cu_mask_probe.hip;
// run on the GPU box: cu_mask_probe N low | cu_mask_probe N spread.  (NOTES/frontend.md, last section)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
__global__ void k(unsigned* out)
{
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
  // stay a while so that workgroups spread
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 20000) {}
}
int main(int argc, char** argv)
{
  const int ncu = argc > 1 ? atoi(argv[1]) : 32;
  const bool low = argc > 2 && !strcmp(argv[2], "low");
  uint32_t mask[8] = {0};
  for (int i = 0; i < ncu; i++) { const int b = low ? i : i * 256 / ncu; mask[b >> 5] |= 1u << (b & 31); }
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask failed\n"); return 1; }
  const int nb = 4096;
  unsigned* d; hipMalloc(&d, 8 * nb); hipMemset(d, 0xff, 8 * nb);
  hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  unsigned* h = (unsigned*)malloc(8 * nb); hipMemcpy(h, d, 8 * nb, hipMemcpyDeviceToHost);
  int per_xcc[16] = {0}; bool seen[16][8][2][16]; memset(seen, 0, sizeof(seen));
  for (int i = 0; i < nb; i++) { const unsigned x = h[2 * i] & 15, hw = h[2 * i + 1]; per_xcc[x]++; seen[x][(hw >> 13) & 7][(hw >> 12) & 1][(hw >> 8) & 15] = true; }
  printf("mask %d %s:", ncu, low ? "low" : "spread");
  for (int x = 0; x < 8; x++) { int cus = 0; for (int a = 0; a < 8; a++) for (int b = 0; b < 2; b++) for (int c = 0; c < 16; c++) cus += seen[x][a][b][c]; printf(" xcc%d: %d wgs on %d CUs;", x, per_xcc[x], cus); }
  printf("\n");
  return 0;
}
