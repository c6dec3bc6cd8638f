// c8p_viterbi_lean.h — the fast forward pass of the K=7 soft Viterbi decoder for gfx950.
//
// Same recurrence as k_viterbi_fwd (decode_impl.cc:205-281), restructured so that a trellis step costs ~10 VALU
// instructions and touches neither LDS nor VGPR-held survivors:
//
//  * rotating state->lane map.  At time t state bit k lives in lane bit (k+t) mod 6, so the transition
//    n = (i>>1)|(b<<5) moves NO data except across the one lane bit q = t mod 6: lane D (time t+1) needs the
//    metrics of lanes D&~(1<<q) (even predecessor) and D|(1<<q) (odd predecessor) — itself and its XOR-partner.
//    The partner value arrives through a DPP modifier (q=0,1,3), two DPP hops (q=2) or a gfx950 permlane swap
//    (q=4,5); no ds_bpermute, no LDS.
//  * branch metrics from lane constants: tab[c] = A(c)*t0 + B(c)*t1 with A,B in {0,1} (SV_STATE_OUTPUT,
//    cloud80211phy.cc:1877), evaluated as fma(A,t0,B*t1): products with 0/1 are exact, one rounding = the
//    reference's t1+t0.  The odd-predecessor class is the complement (3-c).
//  * LLRs arrive through scalar loads (wave-uniform), puncturing is resolved at compile time per code rate.
//  * the survivor word of a step is the ballot of "odd predecessor wins" (one 64-bit word, lane order of time
//    t+1), written straight from SGPRs with s_store_dwordx2: 8 B per step.
//
// Exactness.  The kernel drops the reference's -1e15 clamp and its "neither candidate beat the pre-fill" record.
// For LLRs that are all finite with |LLR| <= 2^20 this is the identical function (DESIGN.md "lean = exact"):
// unreachable metrics stay exactly -1e15 (|bm| is far below half an ulp of 1e15), real metrics stay far above it,
// and the traceback from state 0 at T >= 6 only ever visits reachable states.  Every LLR is range-checked on the
// scalar unit while it is consumed; a packet that violates the bound is flagged and re-run by the exact kernel.
#pragma once
#include "c8p_dev.h"

typedef const float __attribute__((address_space(4))) c4float;   // constant address space: uniform index => s_load

#define C8P_LLR_BOUND_BITS 0x49800000u    // 2^20 as float bits

__device__ __forceinline__ int rotr6(int x, int r) { r %= 6; return ((x >> r) | (x << (6 - r))) & 63; }

template <int Q> __device__ __forceinline__ unsigned long long lane_bit_mask()
{
  return Q == 0 ? 0xAAAAAAAAAAAAAAAAull : Q == 1 ? 0xCCCCCCCCCCCCCCCCull : Q == 2 ? 0xF0F0F0F0F0F0F0F0ull :
         Q == 3 ? 0xFF00FF00FF00FF00ull : Q == 4 ? 0xFFFF0000FFFF0000ull : 0xFFFFFFFF00000000ull;
}

// v = pre[lane ^ (1<<Q)] + bm.  The DPP forms live in inline asm, which the compiler's hazard recogniser cannot
// see into: a VGPR written by the previous VALU instruction needs 2 wait states before a DPP read, hence "s_nop 1".
template <int Q> __device__ __forceinline__ float partner_add(float pre, float bm)
{
  float v;
  if (Q == 0) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(pre), "v"(bm));
  } else if (Q == 1) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(pre), "v"(bm));
  } else if (Q == 2) {
    float tq;   // lane^3 then lane^7 (mirror inside 8) = lane^4
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf" : "=v"(tq) : "v"(pre));
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(tq), "v"(bm));
  } else if (Q == 3) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(pre), "v"(bm));
  } else if (Q == 4) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(pre), __float_as_uint(pre), false, false);
    const float part = (lane_id() & 16) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
    v = part + bm;
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(pre), __float_as_uint(pre), false, false);
    const float part = (lane_id() & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
    v = part + bm;
  }
  return v;
}

struct LeanK {          // lane constants for one exchange phase q
  float a, b;           // own-candidate coefficients   tab[c]   = a*t0 + b*t1
  float a2, b2;         // partner-candidate            tab[3-c] = a2*t0 + b2*t1
};

// HAVE0/HAVE1: which of (t0, t1) exist at this trellis step (puncturing); a missing one is 0.0f in the reference.
template <int Q, bool HAVE0, bool HAVE1>
__device__ __forceinline__ float lean_step(float pre, const LeanK& k, float t0, float t1, unsigned long long* __restrict__ sv_word)
{
  float bmU, bmV;
  if (HAVE0 && HAVE1) { bmU = __builtin_fmaf(k.a, t0, k.b * t1); bmV = __builtin_fmaf(k.a2, t0, k.b2 * t1); }
  else if (HAVE0) { bmU = k.a * t0; bmV = k.a2 * t0; }
  else { bmU = k.b * t1; bmV = k.b2 * t1; }
  const float u = pre + bmU;                    // candidate through this lane's own metric
  const float v = partner_add<Q>(pre, bmV);     // candidate through the partner's metric
  // lanes with bit q clear: own = even predecessor, a0 = u, a1 = v; bit q set: a0 = v, a1 = u.  take1 = a1 > a0.
  const unsigned long long c1 = __ballot(v > u), c2 = __ballot(u > v);
  const unsigned long long M = lane_bit_mask<Q>();
  const unsigned long long W = (c1 & ~M) | (c2 & M);
  asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(W), "s"(sv_word) : "memory");
  return (v > u) ? v : u;                       // = the surviving metric in both cases (ties keep a0 = same value)
}

__device__ __forceinline__ unsigned bad_llr(float x) { return ((__float_as_uint(x) & 0x7fffffffu) > C8P_LLR_BOUND_BITS) ? 1u : 0u; }

// One super-iteration = PER trellis steps (lcm of the 6 exchange phases and the puncture period), NL LLRs.
template <int CR> struct RateInfo;
template <> struct RateInfo<C8P_CR_12> { static constexpr int PER = 6, NL = 12; };
template <> struct RateInfo<C8P_CR_23> { static constexpr int PER = 6, NL = 9; };
template <> struct RateInfo<C8P_CR_34> { static constexpr int PER = 6, NL = 8; };
template <> struct RateInfo<C8P_CR_56> { static constexpr int PER = 30, NL = 36; };

// position of (t0, t1) of step j inside the super-iteration's LLR block; -1 = punctured
template <int CR> __device__ __forceinline__ constexpr int idx0(int j)
{
  return CR == C8P_CR_12 ? 2 * j
       : CR == C8P_CR_23 ? 3 * (j / 2) + ((j % 2) ? 2 : 0)
       : CR == C8P_CR_34 ? ((j % 3) == 2 ? -1 : 4 * (j / 3) + ((j % 3) == 1 ? 2 : 0))
       : ((j % 5) == 0 ? 6 * (j / 5) : (j % 5) == 1 ? 6 * (j / 5) + 2 : (j % 5) == 3 ? 6 * (j / 5) + 4 : -1);
}
template <int CR> __device__ __forceinline__ constexpr int idx1(int j)
{
  return CR == C8P_CR_12 ? 2 * j + 1
       : CR == C8P_CR_23 ? ((j % 2) ? -1 : 3 * (j / 2) + 1)
       : CR == C8P_CR_34 ? ((j % 3) == 0 ? 4 * (j / 3) + 1 : (j % 3) == 2 ? 4 * (j / 3) + 3 : -1)
       : ((j % 5) == 0 ? 6 * (j / 5) + 1 : (j % 5) == 2 ? 6 * (j / 5) + 3 : (j % 5) == 4 ? 6 * (j / 5) + 5 : -1);
}

template <int CR, int J, bool GUARD>
__device__ __forceinline__ void lean_steps(float& pre, const LeanK (&K)[6], const float (&x)[RateInfo<CR>::NL],
                                           unsigned long long* __restrict__ sv, int tb, int Tn)
{
  if constexpr (J < RateInfo<CR>::PER) {
    if (!GUARD || tb + J < Tn) {
      constexpr int i0 = idx0<CR>(J), i1 = idx1<CR>(J);
      constexpr int Q = J % 6;
      const float t0 = i0 >= 0 ? x[i0 >= 0 ? i0 : 0] : 0.0f;
      const float t1 = i1 >= 0 ? x[i1 >= 0 ? i1 : 0] : 0.0f;
      pre = lean_step<Q, (i0 >= 0), (i1 >= 0)>(pre, K[Q], t0, t1, sv + tb + J);
      lean_steps<CR, J + 1, GUARD>(pre, K, x, sv, tb, Tn);
    }
  }
}

template <int CR>
__device__ __forceinline__ unsigned lean_forward(const float* __restrict__ Lg, int n_llr, int Tn, unsigned long long* __restrict__ sv)
{
  constexpr int PER = RateInfo<CR>::PER, NL = RateInfo<CR>::NL;
  const int lane = lane_id();
  LeanK K[6];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int nst = rotr6(lane, (q + 1) % 6);          // state held by this lane at time t+1, t mod 6 == q
    const int m = nst & 31, b = nst >> 5;
    const int cU = b ? sv_out(2 * m + 1, 1) : sv_out(2 * m, 0);
    K[q].a = (cU >> 1) ? 1.0f : 0.0f; K[q].b = (cU & 1) ? 1.0f : 0.0f;
    K[q].a2 = 1.0f - K[q].a; K[q].b2 = 1.0f - K[q].b;
  }
  c4float* L = (c4float*)Lg;
  float pre = lane == 0 ? 0.0f : C8P_NEG_INIT;      // state 0 sits in lane 0 under every rotation
  unsigned bad = 0;
  int tb = 0, lb = 0;
  float x[NL];
  // full super-iterations
  for (; tb + PER <= Tn; tb += PER, lb += NL) {
#pragma unroll
    for (int i = 0; i < NL; i++) { x[i] = L[lb + i]; }
#pragma unroll
    for (int i = 0; i < NL; i++) { bad |= bad_llr(x[i]); }
    lean_steps<CR, 0, false>(pre, K, x, sv, tb, Tn);
  }
  if (tb < Tn) {   // tail: fewer than PER steps left, loads clamped to the packet's own LLRs
#pragma unroll
    for (int i = 0; i < NL; i++) { const int li = lb + i; x[i] = li < n_llr ? L[li] : 0.0f; }
#pragma unroll
    for (int i = 0; i < NL; i++) { bad |= bad_llr(x[i]); }
    lean_steps<CR, 0, true>(pre, K, x, sv, tb, Tn);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
  return bad;
}
