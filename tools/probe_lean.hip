// Hardware probe for the primitives of the lean Viterbi kernel (run on an MI355X):
//  - lean_step<Q> (DPP sub/max, permlane swaps, v_cmp_class) against a host emulation of one trellis step,
//  - s_store_dwordx2 / s_store_dwordx4 landing in memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../gr-ieee80211_amd/csrc/c8p_viterbi_lean.h"

struct Out { float nxt[6][64]; unsigned long long W[6]; };

template <int Q> __device__ void one(float pre, const LeanK* K, float t0, float t1, Out* o)
{
  unsigned long long W;
  LeanK k = K[Q * 64 + lane_id()];
  float n = lean_step<Q, true, true>(pre, k, t0, t1, W);
  o->nxt[Q][lane_id()] = n;
  if (lane_id() == 0) o->W[Q] = W;
}

__global__ void k(const float* pre_in, const LeanK* K, float t0, float t1, Out* o, unsigned long long* sv)
{
  const float pre = pre_in[lane_id()];
  one<0>(pre, K, t0, t1, o); one<1>(pre, K, t0, t1, o); one<2>(pre, K, t0, t1, o);
  one<3>(pre, K, t0, t1, o); one<4>(pre, K, t0, t1, o); one<5>(pre, K, t0, t1, o);
  {
    unsigned long long w0 = __ballot(lane_id() & 1) ^ (16ull << 32), w1 = ~w0;
    store_w2<0>(w0, w1, sv);
    unsigned long long w2 = __ballot((lane_id() + 1) & 1) ^ (17ull << 32), w3 = ~w2;
    store_w2<16>(w2, w3, sv);
    unsigned long long w4 = __ballot(lane_id() & 1) ^ (18ull << 32), w5 = ~w4;
    store_w2<32>(w4, w5, sv);
    unsigned long long w6 = __ballot((lane_id() + 1) & 1) ^ (19ull << 32), w7 = ~w6;
    store_w2<48>(w6, w7, sv);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

static int rotr6h(int x, int r) { r %= 6; return ((x >> r) | (x << (6 - r))) & 63; }
static int sv_out_h(int i, int b)
{
  int x0 = b, x1 = (i >> 5) & 1, x2 = (i >> 4) & 1, x3 = (i >> 3) & 1, x5 = (i >> 1) & 1, x6 = i & 1;
  return 2 * (x0 ^ x2 ^ x3 ^ x5 ^ x6) + (x0 ^ x1 ^ x2 ^ x3 ^ x6);
}

int main()
{
  float pre[64]; LeanK K[6 * 64]; int cK[6][64];
  srand(3);
  for (int l = 0; l < 64; l++) pre[l] = (float)((rand() % 64) - 32) * 0.25f;   // ties likely
  const float t0 = 0.75f, t1 = -1.25f;
  for (int q = 0; q < 6; q++) for (int l = 0; l < 64; l++) {
    int nst = rotr6h(l, (q + 1) % 6), m = nst & 31, b = nst >> 5;
    int ck = b ? sv_out_h(2 * m + 1, 1) : sv_out_h(2 * m, 0), cs = 3 - ck;
    cK[q][l] = ck;
    int c1 = (q >= 4 && b) ? cs : ck, c2 = (q >= 4 && b) ? ck : cs;
    LeanK& k = K[q * 64 + l];
    k.ka = (c1 >> 1) ? 1.f : 0.f; k.kb = (c1 & 1) ? 1.f : 0.f; k.sa = (c2 >> 1) ? 1.f : 0.f; k.sb = (c2 & 1) ? 1.f : 0.f;
    k.cls = b ? C8P_CLS_NEG : C8P_CLS_POS;
  }
  float* d_pre; LeanK* d_K; Out* d_o; unsigned long long* d_sv;
  hipMalloc(&d_pre, sizeof(pre)); hipMalloc(&d_K, sizeof(K)); hipMalloc(&d_o, sizeof(Out)); hipMalloc(&d_sv, 256);
  hipMemcpy(d_pre, pre, sizeof(pre), hipMemcpyHostToDevice); hipMemcpy(d_K, K, sizeof(K), hipMemcpyHostToDevice);
  hipMemset(d_sv, 0, 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_pre, d_K, t0, t1, d_o, d_sv);
  Out o; unsigned long long hs[8];
  hipMemcpy(&o, d_o, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(hs, d_sv, sizeof(hs), hipMemcpyDeviceToHost);
  int bad = 0;
  const float tab[4] = {0.f, t1, t0, t1 + t0};
  for (int q = 0; q < 6; q++) {
    int nb = 0; unsigned long long W = 0;
    for (int l = 0; l < 64; l++) {
      int bit = (l >> q) & 1, part = l ^ (1 << q);
      float own = pre[l] + tab[cK[q][l]], oth = pre[part] + tab[3 - cK[q][l]];   // partner's send uses its cS = 3 - cK[part] = 3 - cK[l]... same pair class
      float a0 = bit ? oth : own, a1 = bit ? own : oth;
      bool take = a1 > a0;
      float want = take ? a1 : a0;
      if (take) W |= 1ull << l;
      if (o.nxt[q][l] != want) nb++;
    }
    printf("Q=%d metric mismatches %d, W got %016llx want %016llx %s\n", q, nb, o.W[q], W, (nb == 0 && o.W[q] == W) ? "ok" : "BAD");
    if (nb || o.W[q] != W) bad++;
  }
  for (int t = 0; t < 4; t++) {
    unsigned long long w0 = ((t & 1) ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull) ^ ((unsigned long long)(t + 16) << 32);
    bool ok = hs[2 * t] == w0 && hs[2 * t + 1] == ~w0;
    printf("s_store x4 t=%d got %016llx %016llx want %016llx %016llx %s\n", t, hs[2 * t], hs[2 * t + 1], w0, ~w0, ok ? "ok" : "BAD");
    if (!ok) bad++;
  }
  printf(bad ? "PROBE FAILED\n" : "PROBE OK\n");
  return bad != 0;
}
