// c8p_k_decode.h — the decode entry's kernels: LLR range check, k_viterbi_lean / lean4, the clamp-exact recurrence and the one-wave re-demodulation k_finish uses
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_k_demod.h"
#include "c8p_viterbi_lean.h"

// ================================================================================================ Viterbi forward
// LLR positions consumed by trellis step t for each puncture pattern (SV_PUNC_* cloud80211phy.cc:1857-1860);
// -1 = punctured (metric input 0.0f).
__device__ __forceinline__ void llr_index(int cr, int t, int& i0, int& i1)
{
  if (cr == C8P_CR_12) { i0 = 2 * t; i1 = 2 * t + 1; }
  else if (cr == C8P_CR_23) { int q = t >> 1, r = t & 1; i0 = 3 * q + (r ? 2 : 0); i1 = r ? -1 : 3 * q + 1; }
  else if (cr == C8P_CR_34) {
    int q = t / 3, r = t - 3 * q;
    i0 = r == 0 ? 4 * q : (r == 1 ? 4 * q + 2 : -1);
    i1 = r == 0 ? 4 * q + 1 : (r == 2 ? 4 * q + 3 : -1);
  } else {
    int q = t / 5, r = t - 5 * q;
    i0 = r == 0 ? 6 * q : r == 1 ? 6 * q + 2 : r == 3 ? 6 * q + 4 : -1;
    i1 = r == 0 ? 6 * q + 1 : r == 2 ? 6 * q + 3 : r == 4 ? 6 * q + 5 : -1;
  }
}
__device__ __forceinline__ int llr_needed(int cr, int T)   // LLRs consumed by T steps
{
  int i0, i1, m = 0;
  for (int t = T - 1; t >= 0 && t >= T - 5; t--) { llr_index(cr, t, i0, i1); if (i0 + 1 > m) m = i0 + 1; if (i1 + 1 > m) m = i1 + 1; }
  return m;
}
__device__ __forceinline__ bool decodable(const c8p_pkt_tags& t)   // decode_impl.cc:93-97 + stream long enough
{
  return t.status == C8P_ST_OK && t.len <= C8P_DECODE_B_MAX && t.trellis <= C8P_DECODE_T_MAX && t.trellis > 0 &&
         t.cr >= 0 && t.cr <= 3 && llr_needed(t.cr, t.trellis) <= t.total;
}

// The clamp-exact recurrence, one wave per packet, lane = next state (decode_impl.cc:205-281 literally: clamp, NaN behaviour, his = 0
// for untaken states).  Survivors: per step two ballot words
//   .x bit n: the odd predecessor 2(n&31)+1 won            (v_state_his = 2m+1)
//   .y bit n: neither candidate beat the -1e15 pre-fill   (v_state_his stays 0)
// Runs inside k_finish, for the packets the lean kernels left (range-check flag, C8P_EXACT_VITERBI=1).
__device__ __forceinline__ void viterbi_fwd_exact(const c8p_pkt_tags& t, const float* __restrict__ L, ulonglong2* __restrict__ SV)
{
  const int lane = lane_id();
  const int m = lane & 31, b = lane >> 5;
  const int cA = sv_out(2 * m, b), cB = sv_out(2 * m + 1, b);
  const int Tn = t.trellis;
  float pre = lane == 0 ? 0.0f : C8P_NEG_INIT;
  for (int t0 = 0; t0 < Tn; t0 += 64) {
    const int ts = t0 + lane;
    int i0, i1;
    llr_index(t.cr, ts, i0, i1);
    const float T0 = (ts < Tn && i0 >= 0) ? L[i0] : 0.0f;
    const float T1 = (ts < Tn && i1 >= 0) ? L[i1] : 0.0f;
    unsigned long long W1 = 0, W0 = 0;
    const int nst = min(64, Tn - t0);
    for (int l = 0; l < nst; l++) {
      bool take1, none;
      pre = acs_step(pre, 2 * m, cA, cB, readlane_f(T0, l), readlane_f(T1, l), take1, none);
      const unsigned long long b1 = __ballot(take1), b0 = __ballot(none);
      W1 = writelane_u64(b1, l, W1);
      W0 = writelane_u64(b0, l, W0);
    }
    if (ts < Tn) SV[ts] = make_ulonglong2(W1, W0);
  }
}

// Every data symbol of one packet demodulated into its LLR area by ONE wave (what k_demod_sym spreads over a wave per symbol).  Runs
// inside k_finish for the few packets whose LLRs are wanted in memory after all: the ones the on-chip kernels of c8p_rx flagged (an
// LLR outside the lean decoder's range) and, with the tap (c8p_cfg.keep_llr = 2), the ones those kernels do not touch.
__device__ __forceinline__ void redemod_packet(const c8p_pkt_in& P, const c8p_pkt_tags& t, int p, const cf* __restrict__ iq, const cf* __restrict__ chan,
                                               const cf* __restrict__ Hin, const Mimo* __restrict__ mimo, const DevTables* __restrict__ T,
                                               float* __restrict__ out, int s0 = 0)
{
  const int lane = lane_id();
  if (t.is_ndp) {                                  // 1024 dummy items (demod_impl.cc:265-271), zero-filled
    for (int i = lane; i < C8P_NDP_TOTAL; i += 64) out[i] = 0.0f;
    return;
  }
  const bool two = P.n_ant == 2 && t.nss == 2;
  const cf* x1 = iq + P.iq_off[0] + t.data_off + 8 + lane;
  const cf* x2 = iq + P.iq_off[two ? 1 : 0] + t.data_off + 8 + lane;
  const cf Hl = t.format == C8P_F_L ? chan[P.chan_off + lane] : Hin[(size_t)p * 64 + lane];
  const Mimo* M = &mimo[p];
#pragma unroll 1
  for (int s = s0; s < t.nsym; s++) {
    float neg = 0.0f;
    const cf f1 = wave_fft64(x1[(size_t)s * t.nsymsamp], T);
    if (!two) demod_sym_1ss(t, s, lane, f1, Hl, T, out + (size_t)s * t.ncbps, neg);
    else {
      const cf f2 = wave_fft64(x2[(size_t)s * t.nsymsamp], T);
      cf H4[4], INV4[4];
#pragma unroll
      for (int c = 0; c < 4; c++) { H4[c] = M->H[lane][c]; INV4[c] = M->INV[lane][c]; }
      demod_sym_2ss(t, s, lane, f1, f2, H4, INV4, M, T, out + (size_t)s * t.ncbps, neg);
    }
  }
}

// Range check of an LLR stream handed in by the caller (c8p_decode): flags[p] = 1 sends the packet to the exact kernel;
// flags[2n + p] = bound of the negative-part sum of any C8P_NEGB_WINDOW consecutive LLRs (c8p_viterbi_lean.h): window x the largest -LLR.
__global__ __launch_bounds__(64) void k_llr_check(const c8p_pkt_tags* __restrict__ tags, const float* __restrict__ llr,
                                                  uint32_t* __restrict__ flags, int n)
{
  const int p = blockIdx.x;
  if (p >= n) return;
  const c8p_pkt_tags t = tags[p];
  bool bad = false;
  float mneg = 0.0f;
  if (decodable(t)) {
    const float* L = llr + t.llr_off;
    // (eight loads in flight per lane: one wave reads its packet's 57 KB alone, and with one load per trip round the loop the kernel took
    // 170-190 us whatever the batch — 224 memory latencies in a row; OR and max do not care about the order)
    int i = lane_id();
    for (; i + 64 * 7 < t.total; i += 64 * 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = L[i + 64 * k];
#pragma unroll
      for (int k = 0; k < 8; k++) { bad |= bad_llr(v[k]); mneg = fmaxf(mneg, -v[k]); }
    }
    for (; i < t.total; i += 64) { const float v = L[i]; bad |= bad_llr(v); mneg = fmaxf(mneg, -v); }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mneg = fmaxf(mneg, __shfl_xor(mneg, o, 64));
  const unsigned long long any = __ballot(bad);
  if (lane_id() == 0) { flags[p] = any ? 1u : 0u; flags[2 * n + p] = any ? 0u : __float_as_uint(mneg * (float)C8P_NEGB_WINDOW); }
}

// The fast path (c8p_viterbi_lean.h).  Packets whose LLR stream failed the range check (flags[p] != 0) are left to
// viterbi_fwd_exact (inside k_finish), the clamp-exact recurrence.
// 4 packets per 256-thread workgroup: the four waves of a workgroup land on the four SIMDs of one CU, which keeps the
// 4096 equally long waves of a batch evenly spread (single-wave workgroups were observed to pile up on some SIMDs).
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(64))) void k_viterbi_lean(const c8p_pkt_tags* __restrict__ tags, const PktLayout* __restrict__ lay,
                                                      const float* __restrict__ llr, ulonglong2* __restrict__ surv,
                                                      uint32_t* __restrict__ flags, const float* __restrict__ negsym,
                                                      const c8p_pkt_in* __restrict__ pk_2x2_only, const uint8_t* __restrict__ done4, int n)
{
  const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (p >= n) return;
  if (flags[p] != 0) return;
  if (done4 && done4[p]) return;                   // decoded by k_viterbi_lean4
  const c8p_pkt_tags t = tags[p];
  if (!decodable(t)) return;
  if (pk_2x2_only && !(pk_2x2_only[p].n_ant == 2 && t.nss == 2)) return;   // c8p_rx: single-stream packets were decoded by k_rx_1ss
  // bound of the negative-part sum of any C8P_NEGB_WINDOW consecutive LLRs: the largest entry of the packet's symbols (c8p_rx: written
  // by k_demod_sym) or what k_llr_check left in flags[2n + p] (c8p_decode); kept there for the tests (C8P_BUF_DECSTAT)
  float negb;
  if (negsym) {
    const float* ns = negsym + lay[p].sym_base;
    negb = 0.0f;
    for (int s = lane_id(); s < t.nsym; s += 64) negb = fmaxf(negb, ns[s]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) negb = fmaxf(negb, __shfl_xor(negb, o, 64));
    if (lane_id() == 0) flags[2 * n + p] = __float_as_uint(negb);
  } else negb = __uint_as_float(flags[2 * n + p]);
  const float* L = llr + t.llr_off;
  unsigned* sv = (unsigned*)(surv + lay[p].surv_off);
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_CLOCK)   // experiment build (tools/exp_clock.py): shader cycles of this wave and where it ran, instead of the statistics
  const unsigned long long exp_t0 = clock64();
  unsigned exp_hw, exp_xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(exp_hw), "=s"(exp_xcc));
#endif
  switch (t.cr) {
    case C8P_CR_12: lean_forward<C8P_CR_12>(L, t.total, t.trellis, sv, negb, &flags[n + p]); break;
    case C8P_CR_23: lean_forward<C8P_CR_23>(L, t.total, t.trellis, sv, negb, &flags[n + p]); break;
    case C8P_CR_34: lean_forward<C8P_CR_34>(L, t.total, t.trellis, sv, negb, &flags[n + p]); break;
    default:        lean_forward<C8P_CR_56>(L, t.total, t.trellis, sv, negb, &flags[n + p]); break;
  }
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_CLOCK)
  if (lane_id() == 0) { flags[n + p] = (uint32_t)(clock64() - exp_t0); flags[2 * n + p] = (exp_hw & 0xffffu) | ((exp_xcc & 0xfu) << 16); }
#endif
}



// Four packets per wave (c8p_viterbi_lean.h, bottom; tools/gen_lean4_asm.py).  Wave w takes packets 4w .. 4w+3 if all four can be
// decoded by the lean path, share the code rate and the trellis length; it marks them in done4[] and k_viterbi_lean skips them.
// Otherwise it leaves them to k_viterbi_lean.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_viterbi_lean4(const c8p_pkt_tags* __restrict__ tags, const PktLayout* __restrict__ lay,
                                                       const float* __restrict__ llr, ulonglong2* __restrict__ surv,
                                                       uint32_t* __restrict__ flags, uint8_t* __restrict__ done4, int n)
{
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (4 * w + 3 >= n) return;                      // (a last, partial quad goes to k_viterbi_lean)
  const int lane = lane_id(), row = lane >> 4, l = lane & 15;
  const int p = 4 * w + row;
  const c8p_pkt_tags t = tags[p];                  // per row
  const c8p_pkt_tags t0 = tags[4 * w];             // wave-uniform copy of the first packet's tags
  const bool good = decodable(t) && flags[p] == 0 && t.cr == t0.cr && t.trellis == t0.trellis;
  if (__ballot(good) != ~0ull) return;
  if (l == 0) done4[p] = 1;
  const int Tn = __builtin_amdgcn_readfirstlane(t0.trellis), nfull = Tn / C8P_SV_BLK, segn = c8p_seg_blocks(Tn);
  const float bound = fmaf(__uint_as_float(flags[2 * n + p]), 1.01f, 1e-30f);     // per row (lean_bound)
  unsigned voff = (unsigned)(lay[p].surv_off * 16u) + (unsigned)l * 4u;             // byte offsets into the arenas (< 4 GB)
  unsigned laddr = (unsigned)(t.llr_off * 4u);
  unsigned* const svb = (unsigned*)surv;
  __shared__ float xch_all[4][4][64];              // [wave][packet][index D]: the bridge between the two layouts
  float (*xch)[64] = xch_all[threadIdx.x >> 6];
  float m[4];
#pragma unroll
  for (int r = 0; r < 4; r++) m[r] = (l == 0 && r == 0) ? 0.0f : C8P_NEG_INIT;
  int done = 0, segc = 0, nflt = 0;
  while (done < nfull) {
    // integer loop, all four packets at once, as long as every metric of every row stays above its row's bound
    const float mn = fminf(fminf(m[0], m[1]), fminf(m[2], m[3]));
    if (__ballot(mn < bound) == 0ull) {
      LeanK K4[6];
      lean4_k_consts(l, K4);
      unsigned initx, inity, flip;
      lean4_int_consts(l, initx, inity, flip);
      int left = __builtin_amdgcn_readfirstlane(nfull - done), sc = __builtin_amdgcn_readfirstlane(segc);   // (asm "+s" operands)
      switch (t0.cr) {
        case C8P_CR_12: lean4_int<C8P_CR_12>(m, K4, initx, inity, flip, bound, sc, segn, voff, laddr, llr, svb, left); break;
        case C8P_CR_23: lean4_int<C8P_CR_23>(m, K4, initx, inity, flip, bound, sc, segn, voff, laddr, llr, svb, left); break;
        case C8P_CR_34: lean4_int<C8P_CR_34>(m, K4, initx, inity, flip, bound, sc, segn, voff, laddr, llr, svb, left); break;
        default:        lean4_int<C8P_CR_56>(m, K4, initx, inity, flip, bound, sc, segn, voff, laddr, llr, svb, left); break;
      }
      segc = sc;
      done = nfull - left;
      if (done == nfull) break;
    }
    // one block of every packet with the lane = index code (it chooses the float or the integer step itself)
#pragma unroll
    for (int r = 0; r < 4; r++) xch[row][r * 16 + l] = m[r];
    for (int pp = 0; pp < 4; pp++) {
      const int q = __builtin_amdgcn_readfirstlane(4 * w + pp);
      LeanK K[6];
      lean_k_consts(lane, K);
      unsigned init, flip;
      lean_int_consts(lane, init, flip);
      LeanState S;
      lean_init(S, lane);
      S.pre = xch[pp][lane]; S.done = done; S.segc = segc;
      S.voff = (unsigned)lane * 4u + (unsigned)done * 256u;
      const c8p_pkt_tags tq = tags[q];
      const float* L = llr + tq.llr_off;
      unsigned* sv = (unsigned*)(surv + lay[q].surv_off);
      const float bq = lean_bound(__uint_as_float(flags[2 * n + q]));
      switch (t0.cr) {
        case C8P_CR_12: lean_blocks<C8P_CR_12, false>(S, K, init, flip, bq, L, tq.total, sv, done + 1, segn); break;
        case C8P_CR_23: lean_blocks<C8P_CR_23, false>(S, K, init, flip, bq, L, tq.total, sv, done + 1, segn); break;
        case C8P_CR_34: lean_blocks<C8P_CR_34, false>(S, K, init, flip, bq, L, tq.total, sv, done + 1, segn); break;
        default:        lean_blocks<C8P_CR_56, false>(S, K, init, flip, bq, L, tq.total, sv, done + 1, segn); break;
      }
      xch[pp][lane] = S.pre;
      if (pp == 3) segc = S.segc;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) m[r] = xch[row][r * 16 + l];
    done++; nflt++;
    voff += 256u;
    laddr += 4u * (unsigned)(t0.cr == C8P_CR_12 ? LeanRate<C8P_CR_12>::NLB : t0.cr == C8P_CR_23 ? LeanRate<C8P_CR_23>::NLB :
                             t0.cr == C8P_CR_34 ? LeanRate<C8P_CR_34>::NLB : LeanRate<C8P_CR_56>::NLB);
  }
  if (l == 0) flags[n + p] = (uint32_t)nflt;       // C8P_BUF_DECSTAT: blocks that went through the lane = index code
  // the partial last block, packet by packet
#pragma unroll
  for (int r = 0; r < 4; r++) xch[row][r * 16 + l] = m[r];
  for (int pp = 0; pp < 4; pp++) {
    const int q = __builtin_amdgcn_readfirstlane(4 * w + pp);
    LeanK K[6];
    lean_k_consts(lane, K);
    LeanState S;
    lean_init(S, lane);
    S.pre = xch[pp][lane];
    S.voff = (unsigned)lane * 4u + (unsigned)nfull * 256u;
    const c8p_pkt_tags tq = tags[q];
    const float* L = llr + tq.llr_off;
    unsigned* sv = (unsigned*)(surv + lay[q].surv_off);
    switch (t0.cr) {
      case C8P_CR_12: lean_tail<C8P_CR_12>(S, K, L, tq.total, Tn, sv); break;
      case C8P_CR_23: lean_tail<C8P_CR_23>(S, K, L, tq.total, Tn, sv); break;
      case C8P_CR_34: lean_tail<C8P_CR_34>(S, K, L, tq.total, Tn, sv); break;
      default:        lean_tail<C8P_CR_56>(S, K, L, tq.total, Tn, sv); break;
    }
  }
}
