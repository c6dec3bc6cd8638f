// tools/probes/probe_stride.hip — what a k_finish-shaped read costs on MI355X: one wave per packet reads its 102 KB survivor area, 7 "tiles" of
// 57 rows x 256 bytes, 16 bytes per lane.  Variant 0: the rows of a tile 1792 bytes apart (block (btop_j - i) of every segment: today's
// layout).  Variant 1: the rows of a tile contiguous (a tile-major layout).  Dynamic LDS sets the waves per CU as in k_finish.
//   hipcc --offload-arch=gfx950 -O3 -o probe_stride tools/probes/probe_stride.hip && ./probe_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(64) void k_read(const uint4* __restrict__ base, uint32_t* __restrict__ out, int variant, int nrows, int ntile)
{
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x, frow = lane >> 4, fcol = lane & 15;
  const uint4* a = base + (size_t)blockIdx.x * (size_t)(nrows * ntile * 16);     // 16 uint4 per 256-byte block
  uint32_t acc = 0;
  for (int i = 0; i < ntile; i++) {
    uint4 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int row = j * 4 + frow;
      const int r = row < nrows ? row : 0;
      const size_t blk = variant == 0 ? (size_t)r * ntile + (ntile - 1 - i) : (size_t)i * nrows + r;
      v[j] = a[blk * 16 + fcol];
    }
#pragma unroll
    for (int j = 0; j < 16; j++) lds[(j * 4 + frow) * 17 + fcol] = v[j];
    __syncthreads();
    acc += lds[lane * 17 + (acc & 15)].x;
    __syncthreads();
  }
  out[blockIdx.x * 64 + lane] = acc;
}
int main()
{
  const int n = 4096, nrows = 57, ntile = 7;
  const size_t bytes = (size_t)n * nrows * ntile * 256;
  uint4* d; uint32_t* o;
  hipMalloc(&d, bytes); hipMalloc(&o, n * 64 * 4);
  hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int lds_kb : {21, 12, 6}) for (int variant = 0; variant < 2; variant++) {
    for (int rep = 0; rep < 3; rep++) k_read<<<n, 64, lds_kb * 1024>>>(d, o, variant, nrows, ntile);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; rep++) k_read<<<n, 64, lds_kb * 1024>>>(d, o, variant, nrows, ntile);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LDS %2d KB per wave, %s rows: %.1f us per launch, %.2f TB/s\n", lds_kb, variant ? "contiguous" : "1792-B strided", 100.0 * ms, bytes / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
