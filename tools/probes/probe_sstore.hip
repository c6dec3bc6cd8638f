// Stress probe: are scalar-store data SGPRs read at issue (safe to overwrite immediately) or later?
// Each wave stores N words (x2 and x4 variants), overwriting the data SGPRs with poison right after every store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k2(unsigned long long* out, int n)
{
  unsigned long long* p = out + (size_t)blockIdx.x * n;
  for (int i = 0; i < n; i++) {
    unsigned lo = 0x10000u + i, hi = 0xabc00000u + blockIdx.x;
    asm volatile(
        "s_mov_b32 s20, %0\n\ts_mov_b32 s21, %1\n\t"
        "s_store_dwordx2 s[20:21], %2, 0x0\n\t"
        "s_mov_b32 s20, 0xdeadbeef\n\ts_mov_b32 s21, 0xdeadbeef\n\t"
        "v_cmp_eq_u32_e64 s[20:21], v0, v0"
        :: "s"(lo), "s"(hi), "s"(p + i) : "s20", "s21", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void k4(unsigned long long* out, int n)
{
  unsigned long long* p = out + (size_t)blockIdx.x * n;
  for (int i = 0; i < n; i += 2) {
    unsigned lo = 0x10000u + i, hi = 0xabc00000u + blockIdx.x;
    asm volatile(
        "s_mov_b32 s20, %0\n\ts_mov_b32 s21, %1\n\ts_add_u32 s22, %0, 1\n\ts_mov_b32 s23, %1\n\t"
        "s_store_dwordx4 s[20:23], %2, 0x0\n\t"
        "s_mov_b32 s20, 0xdeadbeef\n\ts_mov_b32 s21, 0xdeadbeef\n\ts_mov_b32 s22, 0xdeadbeef\n\ts_mov_b32 s23, 0xdeadbeef\n\t"
        "v_cmp_eq_u32_e64 s[22:23], v0, v0"
        :: "s"(lo), "s"(hi), "s"(p + i) : "s20", "s21", "s22", "s23", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// data SGPRs produced by VALU (v_readfirstlane) right before the store: is VALU-def -> SMEM-store-data interlocked?
__global__ void k2v(unsigned long long* out, int n)
{
  unsigned long long* p = out + (size_t)blockIdx.x * n;
  for (int i = 0; i < n; i++) {
    unsigned lo = 0x10000u + i, hi = 0xabc00000u + blockIdx.x;
    asm volatile(
        "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\ts_mov_b32 s20, 0xdeadbeef\n\ts_mov_b32 s21, 0xdeadbeef\n\ts_nop 4\n\t"
        "v_readfirstlane_b32 s20, v10\n\tv_readfirstlane_b32 s21, v11\n\t"
        "s_store_dwordx2 s[20:21], %2, 0x0\n\t"
        :: "s"(lo), "s"(hi), "s"(p + i) : "s20", "s21", "v10", "v11", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void k4v(unsigned long long* out, int n)
{
  unsigned long long* p = out + (size_t)blockIdx.x * n;
  for (int i = 0; i < n; i += 2) {
    unsigned lo = 0x10000u + i, hi = 0xabc00000u + blockIdx.x;
    asm volatile(
        "v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_add_u32 v12, 1, v10\n\t"
        "s_mov_b32 s20, 0xdeadbeef\n\ts_mov_b32 s21, 0xdeadbeef\n\ts_mov_b32 s22, 0xdeadbeef\n\ts_mov_b32 s23, 0xdeadbeef\n\ts_nop 4\n\t"
        "v_readfirstlane_b32 s20, v10\n\tv_readfirstlane_b32 s21, v11\n\tv_readfirstlane_b32 s22, v12\n\tv_readfirstlane_b32 s23, v11\n\t"
        "s_store_dwordx4 s[20:23], %2, 0x0\n\t"
        :: "s"(lo), "s"(hi), "s"(p + i) : "s20", "s21", "s22", "s23", "v10", "v11", "v12", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

int main()
{
  const int nb = 4096, n = 2048;
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * nb * n);
  std::vector<unsigned long long> h((size_t)nb * n);
  for (int var = 0; var < 4; var++) {
    hipMemset(d, 0, sizeof(unsigned long long) * nb * n);
    if (var == 0) hipLaunchKernelGGL(k2, dim3(nb), dim3(64), 0, 0, d, n);
    else if (var == 1) hipLaunchKernelGGL(k4, dim3(nb), dim3(64), 0, 0, d, n);
    else if (var == 2) hipLaunchKernelGGL(k2v, dim3(nb), dim3(64), 0, 0, d, n);
    else hipLaunchKernelGGL(k4v, dim3(nb), dim3(64), 0, 0, d, n);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0, poison = 0, zero = 0;
    for (int b = 0; b < nb; b++) for (int i = 0; i < n; i++) {
      unsigned long long want = ((unsigned long long)(0xabc00000u + b) << 32) | (0x10000u + i);
      unsigned long long got = h[(size_t)b * n + i];
      if (got != want) { bad++; if ((got & 0xffffffffu) == 0xdeadbeefu || (got >> 32) == 0xdeadbeefu) poison++; if (got == 0) zero++; }
    }
    printf("%s s_store_dwordx%d: %ld wrong of %ld (poisoned %ld, never written %ld)\n", var < 2 ? "SALU-def" : "VALU-def", (var & 1) ? 4 : 2, bad, (long)nb * n, poison, zero);
  }
  return 0;
}
