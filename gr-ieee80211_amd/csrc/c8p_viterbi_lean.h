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
//  * the survivor bit of a step ("odd predecessor wins", lane order of time t+1) is shifted into a per-lane
//    accumulator (v_cmp -> vcc -> v_addc_co_u32 acc, acc, acc); every C8P_SV_BLK = 30 steps the wave stores its 64
//    accumulators with one coalesced 256-byte vector store.  Block b of a packet = 64 dwords, dword[lane] bit
//    (29 - k) = decision of step 30 b + k.  (The first version stored the ballot word of every step through
//    s_store_dwordx4: the scalar store path capped the kernel at 0.80 ms per 4096 packets against 0.47-0.57 ms
//    without stores; the accumulate costs one more VALU instruction per step and no scalar-memory traffic.)
//
// Exactness.  The kernel drops the reference's -1e15 clamp and its "neither candidate beat the pre-fill" record.
// For LLRs that are all finite with |LLR| <= 2^20 this is the identical function (DESIGN.md "lean = exact"):
// unreachable metrics stay exactly -1e15 (|bm| is far below half an ulp of 1e15), real metrics stay far above it,
// and the traceback from state 0 at T >= 6 only ever visits reachable states.  Every LLR is range-checked where it is
// produced (k_demod_sym) or handed in (k_llr_check); a flagged packet is skipped here and run by the exact kernel.
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

// ---- one trellis step ------------------------------------------------------------------------------------------
// Every lane owns one predecessor metric `pre` (even predecessor if its lane bit q is 0, odd if 1) and computes the
// two candidates that run through it:
//     keep = pre + tab[c_keep]   candidate for the next state that stays in this lane
//     send = pre + tab[c_send]   candidate for the partner lane's next state           (c_send = 3 - c_keep)
// then receives the partner's `send` as `recv`.  The reference takes the odd predecessor only on strict '>':
//     lane bit q = 0:  a0 = keep, a1 = recv   take1 = recv > keep
//     lane bit q = 1:  a0 = recv, a1 = keep   take1 = keep > recv
// New metric = max(keep, recv) in both cases.  With d = recv - keep the decision is a per-lane float-class test of d
// (positive classes for bit-0 lanes, negative classes for bit-1 lanes; d == 0 is a tie -> even predecessor), so one
// v_cmp_class produces the whole survivor word.  For q = 4,5 a permlane swap delivers a0 in one register and a1 in
// the other for every lane, and the decision is a plain a1 > a0.
//
// DPP operands sit inside inline asm, invisible to the compiler's hazard recogniser: a VGPR written by VALU needs
// 2 wait states before a DPP read, hence the leading "s_nop 1" of every DPP block.
typedef float v2f __attribute__((ext_vector_type(2)));
struct LeanK {          // lane constants for one exchange phase q, as the register pairs the packed instructions read
  v2f a;                // {ka, sa}: keep / send coefficient of t0   tab[c] = ka*t0 + kb*t1, tab[3-c] = sa*t0 + sb*t1
  v2f b;                // {kb, sb}: keep / send coefficient of t1
  int cls;              // v_cmp_class mask for d = recv - keep (q <= 3)
};

#define C8P_CLS_POS 0x380   // +denormal | +normal | +inf
#define C8P_CLS_NEG 0x01C   // -inf | -normal | -denormal

// keep and send are the same arithmetic on (ka,kb) and (sa,sb): one packed-f32 instruction does both
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are two independent IEEE operations, no cross-element effects).
template <bool HAVE0, bool HAVE1>
__device__ __forceinline__ v2f cand2(float pre, const LeanK& k, float t0, float t1)
{
  const v2f A = k.a, B = k.b, P = {pre, pre};
  if (HAVE0 && HAVE1) {
    const v2f T0 = {t0, t0}, T1 = {t1, t1};
    return P + __builtin_elementwise_fma(A, T0, B * T1);     // pre + tab[c], tab[3] = t1 + t0 rounded once
  }
  if (HAVE0) { const v2f T0 = {t0, t0}; return __builtin_elementwise_fma(A, T0, P); }   // pre + (t0 or 0): exact product
  const v2f T1 = {t1, t1};
  return __builtin_elementwise_fma(B, T1, P);
}

// The decision of a step (compare + shift into the accumulator) does not feed the metric recurrence, so it is issued
// one step late, in the two wait states a DPP / permlane read needs after the VALU write of its source (filled with
// s_nop before).  PQ = exchange phase of the step whose decision is still pending (-1: none); the pending operands are
//   PQ <= 3: (d, class mask) for v_cmp_class,   PQ >= 4: (a1, a0) for v_cmp_gt.
struct LeanPend { unsigned x, y; };
#define C8P_DEFER_NONE "s_nop 1\n\t"
#define C8P_DEFER_CLS  "v_cmp_class_f32_e32 vcc, %[px], %[py]\n\tv_addc_co_u32_e32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
#define C8P_DEFER_GT   "v_cmp_gt_f32_e32 vcc, %[px], %[py]\n\tv_addc_co_u32_e32 %[acc], vcc, %[acc], %[acc], vcc\n\t"
#define C8P_DPP2(NAME, DEFER, CTRL)                                                                                     \
  asm(DEFER "v_sub_f32_dpp %[d], %[send], %[keep] " CTRL " row_mask:0xf bank_mask:0xf\n\t"                             \
            "v_max_f32_dpp %[nxt], %[send], %[keep] " CTRL " row_mask:0xf bank_mask:0xf"                                \
      : [d] "=&v"(d), [nxt] "=&v"(nxt), [acc] "+v"(acc) : [send] "v"(send), [keep] "v"(keep), [px] "v"(pend.x), [py] "v"(pend.y) : "vcc")
#define C8P_DPP3(DEFER)                                                                                                 \
  asm(DEFER "v_mov_b32_dpp %[tq], %[send] quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"               \
            "v_sub_f32_dpp %[d], %[tq], %[keep] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                        \
            "v_max_f32_dpp %[nxt], %[tq], %[keep] row_half_mirror row_mask:0xf bank_mask:0xf"                          \
      : [tq] "=&v"(tq), [d] "=&v"(d), [nxt] "=&v"(nxt), [acc] "+v"(acc)                                                \
      : [send] "v"(send), [keep] "v"(keep), [px] "v"(pend.x), [py] "v"(pend.y) : "vcc")
#define C8P_SWAP(DEFER, OP)                                                                                             \
  asm(DEFER OP " %[a0], %[a1]" : [a0] "+v"(a0), [a1] "+v"(a1), [acc] "+v"(acc) : [px] "v"(pend.x), [py] "v"(pend.y) : "vcc")
#define C8P_BY_PQ(STMT_NONE, STMT_CLS, STMT_GT) \
  do { if constexpr (PQ < 0) { STMT_NONE; } else if constexpr (PQ <= 3) { STMT_CLS; } else { STMT_GT; } } while (0)

template <int Q, int PQ, bool HAVE0, bool HAVE1>
__device__ __forceinline__ float lean_step(float pre, const LeanK& k, float t0, float t1, unsigned& acc, LeanPend& pend)
{
  // for q = 4,5 the constants are laid out so that `keep` is the even-predecessor candidate slot (see lean_forward)
  const v2f ks = cand2<HAVE0, HAVE1>(pre, k, t0, t1);
  const float keep = ks.x, send = ks.y;
  float nxt;
  if constexpr (Q <= 3) {
    float d;
    if constexpr (Q == 0) {
      C8P_BY_PQ(C8P_DPP2(0, C8P_DEFER_NONE, "quad_perm:[1,0,3,2]"), C8P_DPP2(0, C8P_DEFER_CLS, "quad_perm:[1,0,3,2]"), C8P_DPP2(0, C8P_DEFER_GT, "quad_perm:[1,0,3,2]"));
    } else if constexpr (Q == 1) {
      C8P_BY_PQ(C8P_DPP2(1, C8P_DEFER_NONE, "quad_perm:[2,3,0,1]"), C8P_DPP2(1, C8P_DEFER_CLS, "quad_perm:[2,3,0,1]"), C8P_DPP2(1, C8P_DEFER_GT, "quad_perm:[2,3,0,1]"));
    } else if constexpr (Q == 2) {   // lane^3 inside the quad, then mirror inside 8 lanes (lane^7): together lane^4
      float tq;
      C8P_BY_PQ(C8P_DPP3(C8P_DEFER_NONE), C8P_DPP3(C8P_DEFER_CLS), C8P_DPP3(C8P_DEFER_GT));
    } else {
      C8P_BY_PQ(C8P_DPP2(3, C8P_DEFER_NONE, "row_ror:8"), C8P_DPP2(3, C8P_DEFER_CLS, "row_ror:8"), C8P_DPP2(3, C8P_DEFER_GT, "row_ror:8"));
    }
    pend.x = __float_as_uint(d); pend.y = (unsigned)k.cls;
  } else {
    // `keep`/`send` are here the candidates this lane computes for the a0-register / a1-register respectively;
    // the swap moves each to the lane that needs it: afterwards a0 / a1 hold those candidates in every lane.
    float a0 = keep, a1 = send;
    if constexpr (Q == 4) {
      C8P_BY_PQ(C8P_SWAP(C8P_DEFER_NONE, "v_permlane16_swap_b32"), C8P_SWAP(C8P_DEFER_CLS, "v_permlane16_swap_b32"), C8P_SWAP(C8P_DEFER_GT, "v_permlane16_swap_b32"));
    } else {
      C8P_BY_PQ(C8P_SWAP(C8P_DEFER_NONE, "v_permlane32_swap_b32"), C8P_SWAP(C8P_DEFER_CLS, "v_permlane32_swap_b32"), C8P_SWAP(C8P_DEFER_GT, "v_permlane32_swap_b32"));
    }
    asm("v_max_f32 %0, %1, %2" : "=v"(nxt) : "v"(a0), "v"(a1));
    pend.x = __float_as_uint(a1); pend.y = __float_as_uint(a0);
  }
  return nxt;
}
// the decision still pending after the last step of a block
template <int PQ> __device__ __forceinline__ void lean_flush(unsigned& acc, const LeanPend& pend)
{
  if constexpr (PQ < 0) return;
  else if constexpr (PQ <= 3) asm("v_cmp_class_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(pend.x), "v"(pend.y) : "vcc");
  else asm("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(pend.x), "v"(pend.y) : "vcc");
}

// range check applied where LLRs are produced (k_demod_sym) or handed in (k_llr_check), one flag per packet
__device__ __forceinline__ bool bad_llr(float x) { return (__float_as_uint(x) & 0x7fffffffu) > C8P_LLR_BOUND_BITS; }

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

#define C8P_SV_BLK 30      // trellis steps per survivor block (lcm of the exchange period 6 and the r=5/6 puncture period 5)

// J-th step of a super-iteration.  FIRST: no decision is pending when step 0 starts (first super-iteration of a block).
// GUARD (partial last block): every step settles its own decision at once.
template <int CR, int J, bool GUARD, bool FIRST>
__device__ __forceinline__ void lean_steps(float& pre, const LeanK (&K)[6], const float (&x)[RateInfo<CR>::NL], unsigned& acc, LeanPend& pend, int rem)
{
  if constexpr (J < RateInfo<CR>::PER) {
    if (!GUARD || J < rem) {
      constexpr int i0 = idx0<CR>(J), i1 = idx1<CR>(J);
      constexpr int PQ = GUARD ? -1 : (J == 0 ? (FIRST ? -1 : (RateInfo<CR>::PER - 1) % 6) : (J - 1) % 6);
      pre = lean_step<J % 6, PQ, (i0 >= 0), (i1 >= 0)>(pre, K[J % 6], i0 >= 0 ? x[i0 >= 0 ? i0 : 0] : 0.0f,
                                                        i1 >= 0 ? x[i1 >= 0 ? i1 : 0] : 0.0f, acc, pend);
      if constexpr (GUARD) lean_flush<J % 6>(acc, pend);
      lean_steps<CR, J + 1, GUARD, FIRST>(pre, K, x, acc, pend, rem);
    }
  }
}


#include "c8p_lean_asm.h"   // generated by tools/gen_lean_asm.py: lean_blocks_int<CR>, the integer-compare block loop

// ---- the integer-compare fast path: when is it the same function? --------------------------------------------------
// For POSITIVE floats the unsigned order of the bit patterns is the float order, so compare / max can be integer
// instructions — and an integer subtract has a borrow-IN, a per-lane tie rule that the float domain cannot express in one
// DPP instruction (tools/gen_lean_asm.py explains the step).  The loop may run a block only if every candidate metric
// computed inside it is > 0.  Every candidate of a block is a block-start metric plus a sub-sum of the block's LLRs, each
// addition rounded to nearest, so with   N >= sum over the block's LLRs of max(0, -LLR)   and all 64 start metrics >= b:
//     candidate >= (b - N) (1 - 2^-24)^30 > 0      whenever b >= 1.01 N + 1e-30.
// N comes from where the LLRs are produced: negb[p] >= the negative-part sum of ANY 60 consecutive LLRs of the packet (60 =
// the longest block, rate 1/2): k_demod_sym takes 2 (or 3, when a symbol holds fewer than 60 LLRs) times its symbol's
// negative-part sum, k_llr_check 60 times the largest -LLR.  Path metrics grow by about half the |LLR| mass per step
// (scrambled data), so a few hundred steps into a packet every block qualifies; the blocks that do not (the first ones, or
// streams that drive the metrics negative) run the float step below, which is the same function on any input.
#define C8P_NEGB_WINDOW 60

// lane constants of the integer loop: acc starts every block as `init` (bit 31-k = borrow-in of step k+1: 1 where this
// lane will hold the even predecessor), and even-holder lanes record the complement of the decision (`flip` undoes it)
__device__ __forceinline__ void lean_int_consts(int lane, unsigned& init, unsigned& flip)
{
  init = 0; flip = 0;
#pragma unroll
  for (int k = 0; k < C8P_SV_BLK; k++) {
    const int qn = (k + 1) % 6, q = k % 6;
    if (qn <= 3 && !((lane >> qn) & 1)) init |= 1u << (31 - k);
    if (q <= 3 && !((lane >> q) & 1)) flip |= 1u << (29 - k);
  }
}

// lane constants of all six exchange phases
__device__ __forceinline__ void lean_k_consts(int lane, LeanK (&K)[6])
{
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int nst = rotr6(lane, (q + 1) % 6);          // state held by this lane at time t+1, t mod 6 == q
    const int m = nst & 31, b = nst >> 5;              // b == lane bit q
    const int cK = b ? sv_out(2 * m + 1, 1) : sv_out(2 * m, 0);   // class of the candidate that stays in this lane
    const int cS = 3 - cK;                                         // class of the candidate sent to the partner
    // q <= 3: a.x, b.x = keep, a.y, b.y = send.
    // q >= 4: the swap exchanges the upper halves of the first register with the lower halves of the second, so the
    //         first register must hold {bit 0: keep (= a0), bit 1: send (partner's a0)} and the second
    //         {bit 0: send (partner's a1), bit 1: keep (= a1)}.
    const int c1 = (q >= 4 && b) ? cS : cK;
    const int c2 = (q >= 4 && b) ? cK : cS;
    K[q].a = v2f{(c1 >> 1) ? 1.0f : 0.0f, (c2 >> 1) ? 1.0f : 0.0f};
    K[q].b = v2f{(c1 & 1) ? 1.0f : 0.0f, (c2 & 1) ? 1.0f : 0.0f};
    K[q].cls = b ? C8P_CLS_NEG : C8P_CLS_POS;
  }
}

// The 24 coefficients (each 0.0 or 1.0) and the 4 class masks as ONE register: a kernel that does other work between two runs of
// the block loop (k_rx_1ss demodulates a symbol) keeps `bits` and the lane index instead of 28 VGPRs and unpacks them when it resumes.
__device__ __forceinline__ unsigned lean_k_pack(const LeanK (&K)[6])
{
  unsigned bits = 0;
#pragma unroll
  for (int q = 0; q < 6; q++)
    bits |= (K[q].a.x != 0.0f ? 1u : 0u) << (4 * q) | (K[q].a.y != 0.0f ? 2u : 0u) << (4 * q) | (K[q].b.x != 0.0f ? 4u : 0u) << (4 * q) |
            (K[q].b.y != 0.0f ? 8u : 0u) << (4 * q);
  return bits;
}
template <bool WITH_CLS = true>
__device__ __forceinline__ void lean_k_unpack(unsigned bits, int lane, LeanK (&K)[6])
{
#pragma unroll
  for (int q = 0; q < 6; q++) {
    // 1-bit field sign-extended (0 or all ones) AND the bit pattern of 1.0f: two instructions per coefficient
    K[q].a = v2f{__uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * q, 1) & 0x3f800000u),
                 __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * q + 1, 1) & 0x3f800000u)};
    K[q].b = v2f{__uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * q + 2, 1) & 0x3f800000u),
                 __uint_as_float((unsigned)__builtin_amdgcn_sbfe((int)bits, 4 * q + 3, 1) & 0x3f800000u)};
    K[q].cls = WITH_CLS ? (((lane >> q) & 1) ? C8P_CLS_NEG : C8P_CLS_POS) : 0;
  }
}

// the decoder's state between two calls (everything a wave carries from one survivor block to the next)
struct LeanState {
  float pre;                   // this lane's path metric
  unsigned acc;                // decision shift register
  LeanPend pend;               // float step: operands of the decision still to be recorded
  unsigned voff;               // byte offset of this lane's dword of the next survivor block
  int done;                    // full blocks finished
  int nflt;                    // of those, blocks run with the float step
  int segc;                    // blocks since the last traceback-segment boundary
};
__device__ __forceinline__ void lean_init(LeanState& S, int lane)
{
  S.pre = lane == 0 ? 0.0f : C8P_NEG_INIT;      // state 0 sits in lane 0 under every rotation
  S.acc = 0; S.pend = LeanPend{0u, 0u}; S.voff = (unsigned)lane * 4u; S.done = 0; S.nflt = 0; S.segc = 0;
}
__device__ __forceinline__ float lean_bound(float negb)    // wave-uniform (SGPR) threshold of the integer loop
{
  return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(fmaf(negb, 1.01f, 1e-30f))));
}

// full survivor blocks [S.done, upto) of a packet whose first `n_llr` LLRs (at least the ones those blocks consume) are at Lg
// Traceback hints: k_finish walks segments of `segn` survivor blocks in parallel (segn = ceil(T / (30 C8P_TB_SEGS)), c8p_seg_blocks) and has to
// assume the lane each segment ends in.  In the last block of every segment the forward pass sets bit 30 of the survivor dword
// (C8P_SV_HINT; a block holds 30 decisions in bits 29..0) on the lane(s) that hold the largest path metric at the block's end.
// Only a hint: k_finish checks every assumption against the neighbouring segment and walks again where it was wrong, so the
// decoded bits do not depend on it — but at cfg 2 / 30 dB nearly every hint is right and the survivors are read ~once instead
// of 1.3-1.7 times.
#define C8P_SV_HINT 0x40000000u
#ifndef C8P_TB_SEGS
#define C8P_TB_SEGS 40      // traceback segments per packet at most = rows of k_finish's LDS tile (a multiple of 4).  64 until late in round 3: every
                            // lane of k_finish a segment, 16 KB of tile.  With 40 the walk takes 10 tiles instead of 7 per cfg-2 packet (more instructions in
                            // k_finish), its exclusive time is the same (0.127 ms), and the PIPELINED step gains 1.5 % (7.10 against 6.99-7.00 M packets/s,
                            // five builds alternated three times on one box: 64 / 48 / 44 / 40 / 32 -> 7.00 / 7.08 / 7.08 / 7.10 / 7.10): 10 KB of LDS per
                            // k_finish workgroup leave room for one more decoder workgroup beside it on a CU
#endif
__device__ __forceinline__ int c8p_seg_blocks(int Tn) { return (Tn + C8P_TB_SEGS * C8P_SV_BLK - 1) / (C8P_TB_SEGS * C8P_SV_BLK); }

// TOUCH = false: no scalar-cache touches in the float step (k_viterbi_lean4 runs single blocks from inside two nested loops, where
// the compiler cannot keep the loop-carried touch registers in SGPRs)
template <int CR, bool TOUCH = true>
__device__ __forceinline__ void lean_blocks(LeanState& S, const LeanK (&K)[6], unsigned init, unsigned flip, float bound,
                                            const float* __restrict__ Lg, int n_llr, unsigned* __restrict__ sv, int upto, int segn)
{
  int segc = __builtin_amdgcn_readfirstlane(S.segc);           // (a local copy: asm "+s" operand)
  constexpr int PER = RateInfo<CR>::PER, NL = RateInfo<CR>::NL, REP = C8P_SV_BLK / PER, NLB = NL * REP;   // NLB: LLRs per block
  static_assert(NLB <= C8P_NEGB_WINDOW, "negb covers windows of C8P_NEGB_WINDOW LLRs");
  c4float* L = (c4float*)Lg;
  float x[NL];
  upto = __builtin_amdgcn_readfirstlane(upto);
  int left = __builtin_amdgcn_readfirstlane(upto - S.done);   // (wave-uniform; an asm "+s" operand)
  // touch registers: loop-carried ("+s") so that they can never share an SGPR with x[] — the touches are still in
  // flight when the next block's real loads are issued, and scalar loads may return in any order
  unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0;
  while (left > 0) {
    // integer compare: as many blocks as keep every metric above the bound
#if !(defined(C8P_EXPERIMENTS) && defined(C8P_EXP_NO_INT))
    if (__ballot(S.pre < bound) == 0ull) {
      int sc = __builtin_amdgcn_readfirstlane(segc), lf = __builtin_amdgcn_readfirstlane(left);   // (asm "+s" operands: loop-carried values
      lean_blocks_int<CR>(S.pre, S.acc, K, init, flip, bound, sc, segn, S.voff, Lg + (size_t)(upto - lf) * NLB, sv, lf);   // must not pick up a VGPR copy)
      segc = sc; left = lf;
      if (left == 0) break;
    }
#endif
    // one full block with the float step: REP super-iterations of PER steps each
    int lb = (upto - left) * NLB;
#pragma unroll
    for (int r = 0; r < REP; r++, lb += NL) {
#pragma unroll
      for (int i = 0; i < NL; i++) { x[i] = L[lb + i]; }
      if (TOUCH && r == 0) {
        // Scalar loads return out of order, so the only usable wait is lgkmcnt(0): a real prefetch of the next block
        // would be drained by the wait of this one.  Instead, right after that wait (the x[] inputs force it before
        // this statement), touch the scalar-cache lines of the NEXT block with throw-away loads: they complete during
        // the 30 steps below, and the next block's loads hit the scalar cache instead of stalling on L2 / HBM.
        const int nb = lb + NLB;                    // first LLR of the next block
        c4float* nx = L + (nb + NLB <= n_llr ? nb : lb);   // never touch past the LLRs that exist (no branch: a join
                                                           // would make the compiler drain the touches right away)
        if (NLB * 4 > 192)
          asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\t"
                       "s_load_dword %3, %5, 0xc0\n\ts_load_dword %4, %5, %8"
                       : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3), "+s"(d4) : "s"(nx), "s"(x[0]), "s"(x[NL - 1]), "n"(NLB * 4 - 4));
        else
          asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\t"
                       "s_load_dword %3, %4, %7"
                       : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3) : "s"(nx), "s"(x[0]), "s"(x[NL - 1]), "n"(NLB * 4 - 4));
      }
      if (r == 0) lean_steps<CR, 0, false, true>(S.pre, K, x, S.acc, S.pend, PER);
      else lean_steps<CR, 0, false, false>(S.pre, K, x, S.acc, S.pend, PER);
    }
    lean_flush<(PER - 1) % 6>(S.acc, S.pend);
    unsigned word = S.acc & ~(C8P_SV_HINT | 0x80000000u);      // (the float step keeps shifting: bits 30, 31 hold older decisions)
    if (++segc >= segn) {                                      // last block of a traceback segment: the hint (float compare)
      float m = S.pre;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (S.pre == m) word |= C8P_SV_HINT;
      segc = 0;
    }
    *(unsigned*)((char*)sv + S.voff) = word;
    if (TOUCH) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1), "+s"(d2), "+s"(d3), "+s"(d4));   // touches land before the integer loop reuses SGPRs
    S.voff += 256u; left--; S.nflt++;
  }
  S.done = upto; S.segc = __builtin_amdgcn_readfirstlane(segc);   // (wave-uniform: keep it out of the VGPRs)
}

// the last, partial block (steps nfull * 30 .. Tn): loads clamped to the packet's own LLRs, decisions left-aligned like a full block
template <int CR>
__device__ __forceinline__ void lean_tail(LeanState& S, const LeanK (&K)[6], const float* __restrict__ Lg, int n_llr, int Tn, unsigned* __restrict__ sv)
{
  constexpr int PER = RateInfo<CR>::PER, NL = RateInfo<CR>::NL, REP = C8P_SV_BLK / PER, NLB = NL * REP;
  c4float* L = (c4float*)Lg;
  const int nfull = Tn / C8P_SV_BLK, tb = nfull * C8P_SV_BLK;
  if (tb < Tn) {
    float x[NL];
    const int rem = Tn - tb;
    int lb = nfull * NLB;
#pragma unroll
    for (int r = 0; r < REP; r++, lb += NL) {
      if (r * PER < rem) {
#pragma unroll
        for (int i = 0; i < NL; i++) { const int li = lb + i; x[i] = li < n_llr ? L[li] : 0.0f; }
        lean_steps<CR, 0, true, true>(S.pre, K, x, S.acc, S.pend, rem - r * PER);
      }
    }
    *(unsigned*)((char*)sv + S.voff) = S.acc << (C8P_SV_BLK - rem);
  }
}

// ---- the same two pieces with the LLRs in LDS (k_rx_1ss_lds: the wave that demodulated them decodes them, nothing goes through HBM)
// lb: the wave's LDS buffer, lb[0] = first LLR of block S.done.  The integer loop reads it straight into VGPRs (lean_blocks_lds);
// the float step wants its LLRs in SGPRs: one LDS read (lane i fetches LLR i of the super-iteration) and a v_readlane per LLR —
// 5 % of the blocks at cfg 2 / 30 dB, so the extra 36 instructions per block do not show.
template <int CR>
__device__ __forceinline__ void lean_blocks_l(LeanState& S, const LeanK (&K)[6], unsigned init, unsigned flip, float bound,
                                              const float* lb, int lane, unsigned* __restrict__ sv, int upto, int segn)
{
  int segc = S.segc;                                          // (no readfirstlane on these wave-uniform counters: the intrinsic wants its
                                                              // operand in a VGPR, and the counters would then LIVE in — spilled — VGPRs)
  constexpr int PER = RateInfo<CR>::PER, NL = RateInfo<CR>::NL, REP = C8P_SV_BLK / PER, NLB = NL * REP;
  static_assert(NLB <= C8P_NEGB_WINDOW && NL <= 64, "negb covers windows of C8P_NEGB_WINDOW LLRs; one lane per LLR of a super-iteration");
  const unsigned la = (unsigned)(size_t)lb;                    // LDS byte address (low half of the flat address)
  const int n0 = upto - S.done;
  int left = n0;
  while (left > 0) {
    if (__ballot(S.pre < bound) == 0ull) {
      int sc = segc, lf = left;
      lean_blocks_lds<CR>(S.pre, S.acc, K, init, flip, bound, sc, segn, S.voff, la + (unsigned)((n0 - lf) * NLB * 4), sv, lf);
      segc = sc; left = lf;
      if (left == 0) break;
    }
    {
      // the class masks of the float step (phases 0..3), made here: the caller's constants come without them (24 coefficients are
      // unpacked every round, the masks are only needed by the ~5 % of blocks that take this path)
      const int lf = lane_fresh();                       // (not `lane`: the masks would be hoisted back out of this branch)
      LeanK Kf[6];
#pragma unroll
      for (int q = 0; q < 6; q++) { Kf[q].a = K[q].a; Kf[q].b = K[q].b; Kf[q].cls = ((lf >> q) & 1) ? C8P_CLS_NEG : C8P_CLS_POS; }
      lean_block_flt_lds<CR>(S.pre, S.acc, Kf, la + (unsigned)((n0 - left) * NLB * 4));
    }
    unsigned word = S.acc & ~(C8P_SV_HINT | 0x80000000u);
    if (++segc >= segn) {
      float m = S.pre;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if (S.pre == m) word |= C8P_SV_HINT;
      segc = 0;
    }
    *(unsigned*)((char*)sv + S.voff) = word;
    S.voff += 256u; left--; S.nflt++;
  }
  S.done = upto; S.segc = segc;
}
// lb[0] = first LLR of the partial block, n_avail of them exist (the rest read as 0.0f like the stream's end, lean_tail)
template <int CR>
__device__ __forceinline__ void lean_tail_l(LeanState& S, const LeanK (&K)[6], const float* lb, int n_avail, int lane, int Tn, unsigned* __restrict__ sv)
{
  constexpr int PER = RateInfo<CR>::PER, NL = RateInfo<CR>::NL, REP = C8P_SV_BLK / PER;
  const int nfull = Tn / C8P_SV_BLK, tb = nfull * C8P_SV_BLK;
  if (tb < Tn) {
    float x[NL];
    const int rem = Tn - tb;
    int li = 0;
#pragma unroll
    for (int r = 0; r < REP; r++, li += NL) {
      if (r * PER < rem) {
        const int k = li + (lane < NL ? lane : 0);
        const float v = k < n_avail ? lb[k] : 0.0f;
#pragma unroll
        for (int i = 0; i < NL; i++) x[i] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), i));
        lean_steps<CR, 0, true, true>(S.pre, K, x, S.acc, S.pend, rem - r * PER);
      }
    }
    *(unsigned*)((char*)sv + S.voff) = S.acc << (C8P_SV_BLK - rem);
  }
}

// ---- the SIG fields' rate-1/2 decoder on the lean step (round 3) -----------------------------------------------------------------
// svSigDecoder::decode (cloud80211phy.cc:2001-2088) is the data decoder's recurrence at rate 1/2 over L = 24 (L-SIG), 48 (HT-SIG,
// VHT-SIG-A) or 26 (VHT-SIG-B) steps, traced back from state 0.  The literal form (wave_sig_viterbi_exact: two ds_bpermute gathers, a
// three-way select per candidate, two ballots per step) costs 31 VALU instructions per step and was 60 % of k_classify's instructions;
// the lean step costs ~8 (lane = state under the rotating map, partner through DPP / permlane swap, decision into a per-lane shift
// register) and is the same function for finite |LLR| <= 2^20 and L >= 6 (top of this file).  The LLRs are checked first; anything
// else takes the literal form.  Decisions: step k in bit (29 - k) of accA for k < 30 when L > 30, else in bit (L - 1 - k) of acc.
// Traceback on the scalar unit: the lane index D at time u + 1 spells input bit u in its bit (u mod 6) (state bit k lives in lane bit
// (k + t) mod 6), and the predecessor's lane at time u is D with that bit replaced by the decision.
__device__ __forceinline__ unsigned long long wave_sig_viterbi(const float* llr, uint8_t* bits, int L, unsigned long long* w1, unsigned long long* w0)
{
  const int lane = lane_id();
  bool bad = false;
  for (int i = lane; i < 2 * L; i += 64) bad |= bad_llr(llr[i]);
  if (__ballot(bad) != 0ull || L < 6 || L > 60) return wave_sig_viterbi_exact(llr, bits, L, w1, w0);
  LeanK K[6];
  lean_k_consts(lane, K);
  float pre = lane == 0 ? 0.0f : C8P_NEG_INIT;
  unsigned acc = 0, accA = 0;
  LeanPend pend{0u, 0u};
  const int ngrp = L / 6, rem = L - 6 * ngrp;
  for (int g = 0; g <= ngrp; g++) {
    const int left = g < ngrp ? 6 : rem;               // steps of this group (the last one may be partial, or empty)
    if (left == 0) break;
    const int k = 12 * g + (lane < 12 ? lane : 0);
    const float v = k < 2 * L ? llr[k] : 0.0f;         // lane i holds LLR i of the group: broadcast to the scalar operands of the steps
    float x[12];
#pragma unroll
    for (int i = 0; i < 12; i++) x[i] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), i));
    lean_steps<C8P_CR_12, 0, true, true>(pre, K, x, acc, pend, left);     // (every step settles its own decision at once)
    if (g == 4) accA = acc;                            // 30 decisions: the shift register is full
  }
  int D = 0;
  unsigned long long word = 0;
  for (int u = L - 1; u >= 0; u--) {
    const int q = u % 6;
    word |= (unsigned long long)((D >> q) & 1) << u;
    const bool first = L > 30 && u < 30;
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)(first ? accA : acc), D);
    const unsigned d = (a >> (first ? 29 - u : L - 1 - u)) & 1u;
    D = (D & ~(1 << q)) | (int)(d << q);
  }
  __syncthreads();
  if (lane < L) bits[lane] = (uint8_t)((word >> lane) & 1ull);
  __syncthreads();
  return word;
}

template <int CR> struct LeanRate { static constexpr int NLB = RateInfo<CR>::NL * (C8P_SV_BLK / RateInfo<CR>::PER); };

// the whole forward pass of one packet whose LLRs are all there (k_viterbi_lean)
template <int CR>
__device__ __forceinline__ void lean_forward(const float* __restrict__ Lg, int n_llr, int Tn, unsigned* __restrict__ sv, float negb,
                                             uint32_t* __restrict__ n_float_blocks)
{
  const int lane = lane_id();
  LeanK K[6];
  lean_k_consts(lane, K);
  unsigned init, flip;
  lean_int_consts(lane, init, flip);
  LeanState S;
  lean_init(S, lane);
  lean_blocks<CR>(S, K, init, flip, lean_bound(negb), Lg, n_llr, sv, Tn / C8P_SV_BLK, c8p_seg_blocks(Tn));
  if (lane == 0) *n_float_blocks = (uint32_t)S.nflt;   // diagnostic (C8P_BUF_DECSTAT): blocks that did not qualify for the integer loop
  lean_tail<CR>(S, K, Lg, n_llr, Tn, sv);
}


// ================================================================================================ four packets per wave
// tools/gen_lean4_asm.py: a packet = one row of 16 lanes x four registers, index D = (lane & 15) + 16 r in the role of the lane index
// above.  Only the integer-compare block loop exists in this layout (lean4_int); float-step blocks, the partial last block and
// everything else run the code above, one packet at a time, after the metrics went through LDS into the lane = index layout.
#include "c8p_lean4_asm.h"

// lane constants of the row layout: the coefficient pairs of index D = l (register 0) without the permlane arrangement of q = 4, 5;
// register r's classes differ by a compile-time XOR (the generator's delta(q, r))
__device__ __forceinline__ void lean4_k_consts(int l, LeanK (&K)[6])
{
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const int nst = rotr6(l, (q + 1) % 6);
    const int m = nst & 31, b = nst >> 5;
    const int cK = b ? sv_out(2 * m + 1, 1) : sv_out(2 * m, 0), cS = 3 - cK;
    K[q].a = v2f{(cK >> 1) ? 1.0f : 0.0f, (cS >> 1) ? 1.0f : 0.0f};
    K[q].b = v2f{(cK & 1) ? 1.0f : 0.0f, (cS & 1) ? 1.0f : 0.0f};
    K[q].cls = 0;
  }
}
// decision registers of the row layout start a block as: registers 0..2 `initx` (bit 31-k = borrow-in of the NEXT register of the
// same step k: 1 where this lane holds the even predecessor in phase k mod 6), register 3 `inity` (borrow-in of step k+1) = `init`
// of lean_int_consts; `flip` is the same as there
__device__ __forceinline__ void lean4_int_consts(int l, unsigned& initx, unsigned& inity, unsigned& flip)
{
  lean_int_consts(l, inity, flip);
  initx = 0;
#pragma unroll
  for (int k = 0; k < C8P_SV_BLK; k++) {
    const int q = k % 6;
    if (q <= 3 && !((l >> q) & 1)) initx |= 1u << (31 - k);
  }
}
