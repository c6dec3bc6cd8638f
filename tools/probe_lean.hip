// Hardware probe for the cross-lane primitives the lean Viterbi kernel relies on (run on an MI355X):
//   partner_add<Q>(pre, bm) must equal pre[lane ^ (1<<Q)] + bm for Q = 0..5, and s_store_dwordx2 must land.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../gr-ieee80211_amd/csrc/c8p_viterbi_lean.h"

template <int Q> __device__ void one(float pre, float* out) { out[Q * 64 + lane_id()] = partner_add<Q>(pre, 0.5f); }

__global__ void k(float* out, unsigned long long* sv)
{
  const float pre = (float)(lane_id() * 3 + 1);
  one<0>(pre, out); one<1>(pre, out); one<2>(pre, out); one<3>(pre, out); one<4>(pre, out); one<5>(pre, out);
  for (int t = 0; t < 8; t++) {
    unsigned long long w = __ballot((lane_id() + t) & 1) ^ ((unsigned long long)t << 32);
    unsigned long long* p = sv + t;
    asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(w), "s"(p) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

int main()
{
  float* d; unsigned long long* sv;
  hipMalloc(&d, 6 * 64 * 4); hipMalloc(&sv, 64);
  hipMemset(sv, 0, 64);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sv);
  float h[6 * 64]; unsigned long long hs[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(hs, sv, sizeof(hs), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int q = 0; q < 6; q++) {
    int nb = 0;
    for (int l = 0; l < 64; l++) { float want = (float)((l ^ (1 << q)) * 3 + 1) + 0.5f; if (h[q * 64 + l] != want) nb++; }
    printf("Q=%d mismatches %d  (lane0 got %.1f want %.1f, lane63 got %.1f)\n", q, nb, h[q * 64], (float)((1 << q) * 3 + 1) + 0.5f, h[q * 64 + 63]);
    bad += nb;
  }
  for (int t = 0; t < 8; t++) {
    unsigned long long want = ((t & 1) ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull) ^ ((unsigned long long)t << 32);
    printf("s_store t=%d got %016llx want %016llx %s\n", t, hs[t], want, hs[t] == want ? "ok" : "BAD");
    if (hs[t] != want) bad++;
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad != 0;
}
