// c8p_k_rx.h — c8p_rx: a packet demodulated AND decoded by one wave — k_rx_1ss (LLRs through the arena), k_rx_1ss_lds / k_rx_2ss_lds (LLRs through LDS), k_viterbi_lean_lds
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_k_decode.h"

// ================================================================================================ c8p_rx, single-stream packets
// Symbol demodulation AND the lean forward pass in one wave per packet: the wave demodulates OFDM symbol s into the packet's LLR
// area and right away runs the survivor blocks whose LLRs are complete, reading them back through the scalar cache.  k_demod_sym's
// 46 short waves per packet (and their launch) disappear, the next symbol's samples are in flight while the current one is decoded,
// and the decoder never waits for HBM: +3 % packets/s over the two kernels.  What it does NOT do on this part is save
// HBM reads: tools/probes/probe_l2rw.hip shows that a read of a line the same wave just wrote is a TCC hit and still an EA read request
// (FETCH_SIZE counts it) — keeping the LLRs out of the memory counters would take an LDS ring and VGPR-fed LLRs.
// (demod_impl.cc:279-314 -> decode_impl.cc:128-141, 205-281.)
// Scalar-cache coherence: a 64-byte line is only ever read when all of it has been written (`safe` below) and waited for
// (vmcnt(0): the vector stores are in L2); LLR areas start on 64-byte boundaries (plan_demod), so no line is shared by two packets.
// A packet whose LLRs fail the range check, or that cannot be decoded, is demodulated to the end; k_finish runs the clamp-exact
// forward pass of the flagged ones (viterbi_fwd_exact).
// Registers: the decoder's per-lane state is parked in LDS while a symbol is demodulated and its 28 lane constants travel as one
// packed word, so the kernel needs max(demodulator, decoder) + a few registers: 80 VGPRs, 6 waves per SIMD.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_rx_1ss(const c8p_pkt_in* __restrict__ pk, const cf* __restrict__ iq, const cf* __restrict__ chan,
                                                const PktLayout* __restrict__ lay, const DevTables* __restrict__ T,
                                                const c8p_pkt_tags* __restrict__ tags, const cf* __restrict__ Hin, float* __restrict__ llr,
                                                ulonglong2* __restrict__ surv, uint32_t* __restrict__ flags, int n)
{
  const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (p >= n) return;
  const c8p_pkt_tags t = tags[p];
  if (t.status != C8P_ST_OK) return;
  const c8p_pkt_in P = pk[p];
  if (P.n_ant == 2 && t.nss == 2) return;          // two streams: k_demod_sym + k_viterbi_lean
  const int lane = lane_id();
  const int segn = c8p_seg_blocks(t.trellis);
  float* out = llr + t.llr_off;
  if (t.is_ndp) {                                  // 1024 dummy items (demod_impl.cc:265-271), zero-filled
    for (int i = lane; i < C8P_NDP_TOTAL; i += 64) out[i] = 0.0f;
    return;
  }
  bool ok = decodable(t);                          // wave-uniform: this wave also decodes
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_CLOCK)   // experiment build (tools/exp_clock.py): shader cycles of this wave and where it ran, instead of the statistics
  const unsigned long long exp_t0 = clock64();
  unsigned exp_hw, exp_xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(exp_hw), "=s"(exp_xcc));
#endif
  const cf Hl = t.format == C8P_F_L ? chan[P.chan_off + lane] : Hin[(size_t)p * 64 + lane];
  const cf* x1 = iq + P.iq_off[0] + t.data_off + 8 + lane;
  unsigned* sv = (unsigned*)(surv + lay[p].surv_off);
  unsigned kbits;                                  // the decoder's 28 lane constants, packed while a symbol is demodulated
  {
    LeanK K0[6];
    lean_k_consts(lane, K0);
    kbits = lean_k_pack(K0);
  }
  unsigned init, flip;
  lean_int_consts(lane, init, flip);
  LeanState S;
  lean_init(S, lane);
  const int Tn = t.trellis, nfull = Tn / C8P_SV_BLK;
  const int nlb = t.cr == C8P_CR_12 ? LeanRate<C8P_CR_12>::NLB : t.cr == C8P_CR_23 ? LeanRate<C8P_CR_23>::NLB :
                  t.cr == C8P_CR_34 ? LeanRate<C8P_CR_34>::NLB : LeanRate<C8P_CR_56>::NLB;
  const float wfac = t.ncbps >= C8P_NEGB_WINDOW ? 2.0f : 3.0f;   // symbols a window of C8P_NEGB_WINDOW LLRs can touch
  float negb = 0.0f;                               // largest negative-part sum of the symbols so far, x wfac
  int produced = 0;
  cf xin = t.nsym > 0 ? x1[0] : mk(0.f, 0.f);
  // The decoder's per-lane state is parked in LDS while a symbol is demodulated (8 dwords per lane, 8 KB per workgroup): the two
  // phases then need max(demodulator, decoder) registers instead of their sum, which is what keeps 6 waves per SIMD resident.
  __shared__ uint4 park[4][2][64];
  uint4* const pk0 = &park[threadIdx.x >> 6][0][lane];
  uint4* const pk1 = &park[threadIdx.x >> 6][1][lane];
  for (int s = 0; s < t.nsym; s++) {
    *pk0 = make_uint4(__float_as_uint(S.pre), S.acc, S.pend.x, S.pend.y);
    *pk1 = make_uint4(S.voff, kbits, init, flip);
    asm volatile("; decoder state parked" : "=v"(S.pre), "=v"(S.acc), "=v"(S.pend.x), "=v"(S.pend.y), "=v"(S.voff), "=v"(kbits), "=v"(init), "=v"(flip));
    const cf xnext = x1[(size_t)(s + 1 < t.nsym ? s + 1 : s) * t.nsymsamp];   // next symbol's sample: in flight during this one
    // opaque per symbol: what depends only on the lane (FFT twiddle addresses, de-interleave offsets of all five constellations,
    // the channel as doubles) is otherwise hoisted out of the loop as 64-bit pairs, stays live across the decoder and is spilled
    // to scratch and reloaded every symbol
    int lane_s = lane;
    cf Hs = Hl;
    asm volatile("" : "+v"(lane_s), "+v"(Hs.x), "+v"(Hs.y));
    const cf f1 = wave_fft64(xin, T, lane_s);
    float neg = 0.0f;
    const bool bad = demod_sym_1ss(t, s, lane_s, f1, Hs, T, out + (size_t)s * t.ncbps, neg);
    produced += t.ncbps;
    xin = xnext;
    {
      const uint4 a = *pk0, b = *pk1;
      S.pre = __uint_as_float(a.x); S.acc = a.y; S.pend.x = a.z; S.pend.y = a.w;
      S.voff = b.x; kbits = b.y; init = b.z; flip = b.w;
    }
    if (__ballot(bad)) {
      if (ok && lane == 0) flags[p] = 1u;
      ok = false;
    }
    if (!ok) continue;
    neg = wave_sum_lane63(neg);
    negb = fmaxf(negb, __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(neg), 63)) * wfac);
    // survivor blocks whose LLRs lie in completely written 64-byte lines
    const int safe = s + 1 == t.nsym ? produced : (produced & ~15);   // whole 64-byte scalar-cache lines
    int upto = safe / nlb;
    if (upto > nfull) upto = nfull;
    if (upto == S.done) continue;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(kbits) :: "memory");   // this symbol's LLR stores have reached L2 (and: unpack HERE, per
    LeanK K[6];                                                    // symbol — the constants must not stay live across the demodulation)
    lean_k_unpack(kbits, lane, K);
    const float bound = lean_bound(negb);
    switch (t.cr) {
      case C8P_CR_12: lean_blocks<C8P_CR_12>(S, K, init, flip, bound, out, safe, sv, upto, segn); break;
      case C8P_CR_23: lean_blocks<C8P_CR_23>(S, K, init, flip, bound, out, safe, sv, upto, segn); break;
      case C8P_CR_34: lean_blocks<C8P_CR_34>(S, K, init, flip, bound, out, safe, sv, upto, segn); break;
      default:        lean_blocks<C8P_CR_56>(S, K, init, flip, bound, out, safe, sv, upto, segn); break;
    }
  }
  if (!ok) return;
  if (lane == 0) { flags[n + p] = (uint32_t)S.nflt; flags[2 * n + p] = __float_as_uint(negb); }   // C8P_BUF_DECSTAT
  LeanK K[6];
  lean_k_unpack(kbits, lane, K);
  switch (t.cr) {
    case C8P_CR_12: lean_tail<C8P_CR_12>(S, K, out, t.total, Tn, sv); break;
    case C8P_CR_23: lean_tail<C8P_CR_23>(S, K, out, t.total, Tn, sv); break;
    case C8P_CR_34: lean_tail<C8P_CR_34>(S, K, out, t.total, Tn, sv); break;
    default:        lean_tail<C8P_CR_56>(S, K, out, t.total, Tn, sv); break;
  }
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_CLOCK)
  if (lane == 0) { flags[n + p] = (uint32_t)(clock64() - exp_t0); flags[2 * n + p] = (exp_hw & 0xffffu) | ((exp_xcc & 0xfu) << 16); }
#endif
}

// The tap of the on-chip kernels (c8p_cfg.keep_llr = 2; parity tests): the symbol's de-interleaved LLRs, as the wave's lanes just left
// them in LDS, copied to the packet's LLR area before any block consumes them.  A kernel argument, not a second build of the kernel:
// what the parity tests check (1e-5) comes out of the very code that c8p_rx runs by default.
__device__ __forceinline__ void wave_lds_fence()   // LDS executes a wave's accesses in order; this only keeps the COMPILER from moving
{                                                  // one lane's read above another lane's write
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_tap(float* __restrict__ dst, const float* src, int n)
{
  wave_lds_fence();
  for (int i = lane_fresh(); i < n; i += 64) dst[i] = src[i];
}

// ---- the same with the LLR stream kept on chip (c8p_rx unless c8p_cfg.keep_llr == 1: C8P_BUF_LLR is then NOT filled) --------------
// The wave scatters a symbol's de-interleaved LLRs into its own LDS buffer (behind what the last round left over), runs every
// survivor block that is complete with the LLRs as VGPR operands (lean_blocks_lds: broadcast ds_read_b128 into a rotating set of 12
// registers that only exist inside the block loop) and moves the < 60 left-over LLRs to the front.  Neither the 235 MB of LLR
// stores nor their read-back (an EA read even when it hits L2, tools/probes/probe_l2rw.hip) exist any more: 1.58 -> 1.1 GB per
// 4096-packet step.  Buffer (C8P_LBUF floats): 59 left over + two 64-QAM symbols (one 256-QAM symbol = 416 LLRs).  A packet with an LLR outside the integer
// loop's range is rare: the wave flags it and stops; k_finish demodulates a flagged packet into the LLR arena (redemod_packet), runs the
// clamp-exact forward pass and goes on with its traceback.  Symbols that only carry padding behind the last trellis step are not demodulated at all.
#define C8P_LBUF 688
#ifndef C8P_RX1_WAVES
#define C8P_RX1_WAVES 6      // waves per SIMD k_rx_1ss_lds is compiled for (7 was tried in round 6: see NOTES/decoder.md)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C8P_RX1_WAVES, C8P_RX1_WAVES))) void k_rx_1ss_lds(const c8p_pkt_in* __restrict__ pk, const cf* __restrict__ iq, const cf* __restrict__ chan,
                                                const PktLayout* __restrict__ lay, const DevTables* __restrict__ T,
                                                const c8p_pkt_tags* __restrict__ tags, const cf* __restrict__ Hin, float* __restrict__ llr_tap,
                                                ulonglong2* __restrict__ surv, uint32_t* __restrict__ flags, int n,
                                                const uint32_t* __restrict__ order)
{
  // wave w of the launch takes packet order[w]: a batch of mixed lengths is laid out longest first, so that the four waves of a
  // workgroup (which hold its LDS until the last of them is done) carry packets of like length (nullptr: w itself)
  const int w_ = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (w_ >= n) return;
  const int p = order ? __builtin_amdgcn_readfirstlane((int)order[w_]) : w_;
  const c8p_pkt_tags t = tags[p];
  if (t.status != C8P_ST_OK) return;
  const c8p_pkt_in P = pk[p];
  if (P.n_ant == 2 && t.nss == 2) return;          // two streams: k_rx_2ss_lds
  if (t.is_ndp || !decodable(t)) return;           // nothing to decode (with the tap, k_finish fills the LLR arena of these packets)
  const int lane = lane_id();
  const int segn = c8p_seg_blocks(t.trellis);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __shared__ uint2 park[4][64];                    // the decoder's metric and survivor offset while a symbol is demodulated
  __shared__ uint4 parkK[4][64];                   // its lane constants (packed coefficients, init, flip): written once, read per symbol
  __shared__ cf parkH[4][64];                      // the channel: read back every symbol instead of living in two registers
  __shared__ cf parkX[4][64];                      // the next symbol's sample while the blocks run
  __shared__ __attribute__((aligned(16))) float lbuf[4][C8P_LBUF];
  parkH[wv][lane] = t.format == C8P_F_L ? chan[P.chan_off + lane] : Hin[(size_t)p * 64 + lane];
  const cf* x1 = iq + P.iq_off[0] + t.data_off + 8;               // wave-uniform; the lane is added per symbol
  unsigned* sv = (unsigned*)(surv + lay[p].surv_off);
  unsigned kbits;
  {
    LeanK K0[6];
    lean_k_consts(lane, K0);
    kbits = lean_k_pack(K0);
  }
  {
    unsigned init0, flip0;
    lean_int_consts(lane, init0, flip0);
    parkK[wv][lane] = make_uint4(kbits, init0, flip0, (unsigned)data_carrier_index(t, lane));
  }
  LeanState S;
  lean_init(S, lane);
  const int Tn = t.trellis, nfull = Tn / C8P_SV_BLK;
  const int nlb = t.cr == C8P_CR_12 ? LeanRate<C8P_CR_12>::NLB : t.cr == C8P_CR_23 ? LeanRate<C8P_CR_23>::NLB :
                  t.cr == C8P_CR_34 ? LeanRate<C8P_CR_34>::NLB : LeanRate<C8P_CR_56>::NLB;
  // ceil(65536 / nlb): (have * inv) >> 16 == have / nlb for have < 1499 (C8P_LBUF = 688).  Constants on purpose: a division is done by
  // the VALU, and a wave-uniform value that comes out of the VALU drags every counter it touches into (spilled) VGPRs
  const unsigned nlb_inv = t.cr == C8P_CR_12 ? 1093u : t.cr == C8P_CR_23 ? 1457u : t.cr == C8P_CR_34 ? 1639u : 1821u;
  const float wfac = __uint_as_float(t.ncbps >= C8P_NEGB_WINDOW ? 0x40000000u : 0x40400000u);   // 2 or 3 (selected as bits: stays in an SGPR)
  float negb = 0.0f;
  int have = 0;                                    // LLRs in the buffer: lb[0] = first LLR of block S.done
  cf xin = x1[lane];
  float* const lb = lbuf[wv];
  // Rounds: as many symbols as the buffer has room for are demodulated back to back (two at 64-QAM, one at 256-QAM, up to 12 at
  // BPSK), then their blocks run.  What a change of phase costs — parking, the constant unpack, the block loop's prologue — is paid
  // per round, not per symbol.
  const int needed = __builtin_amdgcn_readfirstlane(llr_needed(t.cr, Tn));   // LLRs the trellis consumes; symbols behind them are padding
  int s = 0;
  // Experiment build only (-DC8P_EXPERIMENTS -DC8P_EXP_PHASE=k, tools/probes/phase_account.py; NOTES/decoder.md "where the wave's time goes"):
  // k = 0: the wave's shader cycles (s_memtime) by phase — demodulating symbols / the block loop / the rest — go to C8P_BUF_DECSTAT instead of
  // the statistics; k = 1: the demodulator alone (no block loop, no tail); k = 2: the block loop alone (the buffer filled with a constant
  // instead of demodulated symbols).  1 and 2 are what the per-phase PMC passes run; their PDUs are meaningless (bench.py --no-check).
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE)
  const unsigned long long ph_t0 = __builtin_readcyclecounter();
  unsigned long long ph_dem = 0, ph_blk = 0;
#define C8P_PH_NOW() __builtin_readcyclecounter()
#endif
  while (s < t.nsym && s * t.ncbps < needed) {
    // Nothing that depends on the lane lives across a phase: the lane index is taken afresh (opaque) in each, the decoder's state
    // (metric, survivor offset) waits in LDS while the symbols are demodulated, its lane constants were written there once, and
    // the next symbol's sample waits there while the blocks run.  acc / pend are dead between blocks.
    {
      const int lane_p = lane_fresh();
      park[wv][lane_p] = make_uint2(__float_as_uint(S.pre), S.voff);
      asm volatile("; decoder state parked" : "=v"(S.pre), "=v"(S.voff));
    }
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE)
    const unsigned long long ph_a = C8P_PH_NOW();
#endif
#pragma unroll 1
    do {
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE) && C8P_EXP_PHASE == 2
      {                                              // no demodulator: ncbps LLRs of +3 (positive drift: the integer loop, as on real data)
        const int lane_s = lane_fresh();
        for (int i = lane_s; i < t.ncbps; i += 64) lb[have + i] = 3.0f;
        have += t.ncbps; s++;
        continue;
      }
#endif
      const int lane_s = lane_fresh();
      const cf xnext = x1[(uint32_t)((s + 1 < t.nsym ? s + 1 : s) * t.nsymsamp + lane_s)];
      cf Hs = parkH[wv][lane_s];
      asm volatile("" : "+v"(Hs.x), "+v"(Hs.y));
      const cf f1 = wave_fft64(xin, T, lane_s);
      float neg = 0.0f;
      const bool bad = demod_sym_1ss<true>(t, s, lane_s, f1, Hs, T, lb + have, neg, (int)*(volatile __attribute__((address_space(3))) unsigned*)&parkK[wv][lane_s].w);   // (address space spelled out: a
                                                                        // generic volatile pointer becomes a FLAT load)
      if (llr_tap) lds_tap(llr_tap + t.llr_off + (uint32_t)(s * t.ncbps), lb + have, t.ncbps);   // c8p_cfg.keep_llr = 2 (parity tests)
      have += t.ncbps;
      xin = xnext;
      s++;
      if (__ballot(bad)) {                         // the exact decoder takes this packet; it wants the LLRs in memory:
        if (lane_s == 0) flags[p] = 1u;            // k_finish demodulates it once more, into the LLR arena
        return;
      }
      neg = wave_sum_lane63(neg);
      negb = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(
          fmaxf(negb, __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(neg), 63)) * wfac))));
    } while (s < t.nsym && s * t.ncbps < needed && have + t.ncbps <= C8P_LBUF);
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE)
    const unsigned long long ph_b = C8P_PH_NOW();
    ph_dem += ph_b - ph_a;
#endif
    const int lane_d = lane_fresh();               // (again: the decoder's per-lane addresses and the pieces of the constant unpack must
                                                   // not be hoisted out of the loop either)
    {
      const uint2 a = park[wv][lane_d];
      S.pre = __uint_as_float(a.x); S.voff = a.y;
    }
    const int done = S.done;
    int nb = (int)(((unsigned)have * nlb_inv) >> 16);   // have / nlb: complete blocks in the buffer
    if (nb > nfull - done) nb = nfull - done;
    if (nb > 0) {
      wave_lds_fence();                              // the demodulating lanes' LLR writes before the decoder's (other lanes') reads
      parkX[wv][lane_d] = xin;
      asm volatile("; next sample parked" : "=v"(xin.x), "=v"(xin.y));
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 kc = *(volatile __attribute__((address_space(3))) u32x4*)&parkK[wv][lane_d];   // (volatile: read here, every time —
                                                                        // not hoisted and kept in registers)
      const unsigned init = kc.y, flip = kc.z;
      LeanK K[6];
      lean_k_unpack<false>(kc.x, lane_d, K);          // (the float step's class masks are made where a float block runs)
      const float bound = lean_bound(negb);
      const int upto = done + nb;
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE)
      const unsigned long long ph_c = C8P_PH_NOW();
#endif
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE) && C8P_EXP_PHASE == 1
      S.done = upto;                                 // (the demodulator alone: the blocks are counted, not run)
      asm volatile("" :: "v"(K[0].a), "v"(init), "v"(flip), "v"(bound));
#else
      switch (t.cr) {
        case C8P_CR_12: lean_blocks_l<C8P_CR_12>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
        case C8P_CR_23: lean_blocks_l<C8P_CR_23>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
        case C8P_CR_34: lean_blocks_l<C8P_CR_34>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
        default:        lean_blocks_l<C8P_CR_56>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
      }
#endif
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE)
      ph_blk += C8P_PH_NOW() - ph_c;
#endif
      // what is left over moves to the front (< nlb <= 60 LLRs unless the full blocks are all done; one lane each, 64 per pass)
      const int used = nb * nlb;
      have -= used;
      for (int i = 0; i < have; i += 64) {
        const float v = i + lane_d < have ? lb[used + i + lane_d] : 0.0f;
        wave_lds_fence();                            // every lane has read before any lane writes
        if (i + lane_d < have) lb[i + lane_d] = v;
      }
      wave_lds_fence();
      xin = parkX[wv][lane_fresh()];
    }
  }
  const int lane_t = lane_fresh();
  if (lane_t == 0) { flags[n + p] = (uint32_t)S.nflt; flags[2 * n + p] = __float_as_uint(negb); }   // C8P_BUF_DECSTAT
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE) && C8P_EXP_PHASE == 1
  return;
#endif
  LeanK K[6];
  lean_k_unpack(parkK[wv][lane_t].x, lane_t, K);
  switch (t.cr) {
    case C8P_CR_12: lean_tail_l<C8P_CR_12>(S, K, lb, have, lane_t, Tn, sv); break;
    case C8P_CR_23: lean_tail_l<C8P_CR_23>(S, K, lb, have, lane_t, Tn, sv); break;
    case C8P_CR_34: lean_tail_l<C8P_CR_34>(S, K, lb, have, lane_t, Tn, sv); break;
    default:        lean_tail_l<C8P_CR_56>(S, K, lb, have, lane_t, Tn, sv); break;
  }
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_PHASE)
  // [n + p]: the wave's cycles in all, [2n + p]: demodulation >> 6 in the high half, block loop >> 6 in the low half (a packet takes ~10^6)
  if (lane_t == 0) {
    const unsigned long long all = C8P_PH_NOW() - ph_t0;
    flags[n + p] = (uint32_t)all;
    flags[2 * n + p] = ((uint32_t)(ph_dem >> 6) << 16) | ((uint32_t)(ph_blk >> 6) & 0xffffu);
  }
#endif
}

// ---- two streams on two antennas (demod2's 2x2 path, BASELINE configs[2]) with the LLR stream kept on chip ------------------------
// One wave per packet, as k_rx_1ss_lds: per OFDM symbol two FFTs (antenna 0 / 1), the bin's zero-forcing equaliser from k_classify's
// Mimo record, the 8-term pilot phasor, two demaps scattered through the stream-1 / stream-2 de-interleave maps fused with the stream
// de-parser (deint2ss) into the wave's LDS buffer, then lean_blocks_lds on the blocks that are complete.  (htChanUpdate
// demod2_impl.cc:471-551, vhtChanUpdate :553-630, procSymDepasNL cloud80211phy.cc:2442, the loop :268-321.)  Before this kernel the
// two-stream packets of c8p_rx went through k_demod_sym -> 114 KB of LLRs per packet through HBM and back -> k_viterbi_lean.
// Registers and LDS: the Mimo record is 16 floats per lane — 4 KB per wave in LDS, read back with four ds_read_b128 per symbol — and
// one 256-QAM symbol of two streams is 832 LLRs (+ up to 59 left over): 7.6 KB of LDS per wave, 30 KB per workgroup, i.e. at most
// 5 workgroups per CU.  The kernel is therefore built for 4 waves per SIMD (128 VGPRs) and spends the registers that buys instead of
// parking the decoder's state in LDS as the single-stream kernel has to.
#define C8P_LBUF2 896
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rx_2ss_lds(const c8p_pkt_in* __restrict__ pk, const cf* __restrict__ iq,
                                                const PktLayout* __restrict__ lay, const DevTables* __restrict__ T,
                                                const c8p_pkt_tags* __restrict__ tags, const Mimo* __restrict__ mimo, float* __restrict__ llr_tap,
                                                ulonglong2* __restrict__ surv, uint32_t* __restrict__ flags, int n,
                                                const uint32_t* __restrict__ order)
{
  const int w_ = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (w_ >= n) return;
  const int p = order ? __builtin_amdgcn_readfirstlane((int)order[w_]) : w_;           // (see k_rx_1ss_lds)
  const c8p_pkt_tags t = tags[p];
  if (t.status != C8P_ST_OK) return;
  const c8p_pkt_in P = pk[p];
  if (!(P.n_ant == 2 && t.nss == 2)) return;       // one stream: k_rx_1ss_lds
  if (t.is_ndp || !decodable(t)) return;
  const int lane = lane_id();
  const int segn = c8p_seg_blocks(t.trellis);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __shared__ float4 mim[4][4][64];                 // [wave][H0 H1 | H2 H3 | INV0 INV1 | INV2 INV3][lane]
  __shared__ __attribute__((aligned(16))) float lbuf[4][C8P_LBUF2];
  const Mimo* M = &mimo[p];
  {
    const float4* h = (const float4*)&M->H[lane][0];
    const float4* v = (const float4*)&M->INV[lane][0];
    mim[wv][0][lane] = h[0]; mim[wv][1][lane] = h[1];
    mim[wv][2][lane] = v[0]; mim[wv][3][lane] = v[1];
  }
  const cf* x1 = iq + P.iq_off[0] + t.data_off + 8;               // wave-uniform; the lane is added per symbol
  const cf* x2 = iq + P.iq_off[1] + t.data_off + 8;
  unsigned* sv = (unsigned*)(surv + lay[p].surv_off);
  unsigned kbits, init, flip;
  {
    LeanK K0[6];
    lean_k_consts(lane, K0);
    kbits = lean_k_pack(K0);
    lean_int_consts(lane, init, flip);
  }
  const int jq = (!is_null_nl(lane) && !is_pilot(lane)) ? qidx_nl(lane) : -1;
  LeanState S;
  lean_init(S, lane);
  const int Tn = t.trellis, nfull = Tn / C8P_SV_BLK;
  const int nlb = t.cr == C8P_CR_12 ? LeanRate<C8P_CR_12>::NLB : t.cr == C8P_CR_23 ? LeanRate<C8P_CR_23>::NLB :
                  t.cr == C8P_CR_34 ? LeanRate<C8P_CR_34>::NLB : LeanRate<C8P_CR_56>::NLB;
  const unsigned nlb_inv = t.cr == C8P_CR_12 ? 1093u : t.cr == C8P_CR_23 ? 1457u : t.cr == C8P_CR_34 ? 1639u : 1821u;   // have / nlb for have < 1499
  static_assert(C8P_LBUF2 < 1499, "nlb_inv");
  const float wfac = __uint_as_float(t.ncbps >= C8P_NEGB_WINDOW ? 0x40000000u : 0x40400000u);   // 2 or 3
  float negb = 0.0f;
  int have = 0;
  cf xa = x1[lane], xb = x2[lane];
  float* const lb = lbuf[wv];
  const int needed = __builtin_amdgcn_readfirstlane(llr_needed(t.cr, Tn));   // LLRs the trellis consumes; symbols behind them are padding
  int s = 0;
  while (s < t.nsym && s * t.ncbps < needed) {
#pragma unroll 1
    do {
      const int lane = lane_fresh();               // (opaque per symbol: what depends only on the lane — FFT twiddle addresses, de-interleave
                                                   // offsets of all five constellations — must not be hoisted and held across the decoder)
      const uint32_t nx = (uint32_t)((s + 1 < t.nsym ? s + 1 : s) * t.nsymsamp + lane);
      const cf xna = x1[nx], xnb = x2[nx];          // the next symbol's samples: in flight during this one
      const cf f1 = wave_fft64(xa, T, lane);
      const cf f2 = wave_fft64(xb, T, lane_fresh());
      cf H4[4], INV4[4];
      {
        const float4 a = mim[wv][0][lane], b = mim[wv][1][lane], c = mim[wv][2][lane], d = mim[wv][3][lane];
        H4[0] = mk(a.x, a.y); H4[1] = mk(a.z, a.w); H4[2] = mk(b.x, b.y); H4[3] = mk(b.z, b.w);
        INV4[0] = mk(c.x, c.y); INV4[1] = mk(c.z, c.w); INV4[2] = mk(d.x, d.y); INV4[3] = mk(d.z, d.w);
      }
      float neg = 0.0f;
      const bool bad = demod_sym_2ss<true>(t, s, lane, f1, f2, H4, INV4, M, T, lb + have, neg, jq);
      if (llr_tap) lds_tap(llr_tap + t.llr_off + (uint32_t)(s * t.ncbps), lb + have, t.ncbps);   // c8p_cfg.keep_llr = 2 (parity tests)
      have += t.ncbps;
      xa = xna; xb = xnb;
      s++;
      if (__ballot(bad)) {                         // an LLR outside the integer loop's range: k_finish takes the packet from the start
        if (lane == 0) flags[p] = 1u;              // (demodulation into the LLR arena, clamp-exact forward pass)
        return;
      }
      neg = wave_sum_lane63(neg);
      negb = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(
          fmaxf(negb, __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(neg), 63)) * wfac))));
    } while (s < t.nsym && s * t.ncbps < needed && have + t.ncbps <= C8P_LBUF2);
    const int done = S.done;
    int nb = (int)(((unsigned)have * nlb_inv) >> 16);   // have / nlb: complete blocks in the buffer
    if (nb > nfull - done) nb = nfull - done;
    if (nb > 0) {
      wave_lds_fence();
      LeanK K[6];
      lean_k_unpack<false>(kbits, lane, K);
      const float bound = lean_bound(negb);
      const int upto = done + nb;
      switch (t.cr) {
        case C8P_CR_12: lean_blocks_l<C8P_CR_12>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
        case C8P_CR_23: lean_blocks_l<C8P_CR_23>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
        case C8P_CR_34: lean_blocks_l<C8P_CR_34>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
        default:        lean_blocks_l<C8P_CR_56>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
      }
      const int used = nb * nlb;
      have -= used;
      for (int i = 0; i < have; i += 64) {         // what is left over moves to the front (one lane each, 64 per pass; reads before writes)
        const float v = i + lane < have ? lb[used + i + lane] : 0.0f;
        wave_lds_fence();
        if (i + lane < have) lb[i + lane] = v;
      }
    }
  }
  if (lane == 0) { flags[n + p] = (uint32_t)S.nflt; flags[2 * n + p] = __float_as_uint(negb); }   // C8P_BUF_DECSTAT
  wave_lds_fence();
  LeanK K[6];
  lean_k_unpack(kbits, lane, K);
  switch (t.cr) {
    case C8P_CR_12: lean_tail_l<C8P_CR_12>(S, K, lb, have, lane, Tn, sv); break;
    case C8P_CR_23: lean_tail_l<C8P_CR_23>(S, K, lb, have, lane, Tn, sv); break;
    case C8P_CR_34: lean_tail_l<C8P_CR_34>(S, K, lb, have, lane, Tn, sv); break;
    default:        lean_tail_l<C8P_CR_56>(S, K, lb, have, lane, Tn, sv); break;
  }
}

// c8p_decode through the LDS-fed block loop (C8P_DECODE_LDS=1; tests): the caller's LLRs are copied into the wave's LDS buffer in
// rounds, as k_rx_1ss_lds's demodulator would have left them, and decoded by the same lean_blocks_l / lean_tail_l.  This is how the
// decode-entry parity tests (tie-only streams, trellis-length sweeps, bursts that leave the integer loop) reach the LDS flavours
// of all four code rates; nothing else uses it.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_viterbi_lean_lds(const c8p_pkt_tags* __restrict__ tags, const PktLayout* __restrict__ lay,
                                                      const float* __restrict__ llr, ulonglong2* __restrict__ surv, uint32_t* __restrict__ flags,
                                                      const uint8_t* __restrict__ done2, int n)
{
  const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (p >= n) return;
  if (flags[p] != 0) return;
  if (done2 && done2[p]) return;                   // decoded by k_viterbi_lean2_lds
  const c8p_pkt_tags t = tags[p];
  if (!decodable(t)) return;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __shared__ __attribute__((aligned(16))) float lbuf[4][C8P_LBUF];
  float* const lb = lbuf[wv];
  const int lane = lane_id();
  const float negb = __uint_as_float(flags[2 * n + p]);          // k_llr_check's bound
  const float bound = lean_bound(negb);
  const float* L = llr + t.llr_off;
  unsigned* sv = (unsigned*)(surv + lay[p].surv_off);
  const int Tn = t.trellis, nfull = Tn / C8P_SV_BLK, segn = c8p_seg_blocks(Tn);
  const int nlb = t.cr == C8P_CR_12 ? LeanRate<C8P_CR_12>::NLB : t.cr == C8P_CR_23 ? LeanRate<C8P_CR_23>::NLB :
                  t.cr == C8P_CR_34 ? LeanRate<C8P_CR_34>::NLB : LeanRate<C8P_CR_56>::NLB;
  const unsigned nlb_inv = t.cr == C8P_CR_12 ? 1093u : t.cr == C8P_CR_23 ? 1457u : t.cr == C8P_CR_34 ? 1639u : 1821u;
  const int needed = __builtin_amdgcn_readfirstlane(llr_needed(t.cr, Tn));
  LeanK K[6];
  lean_k_consts(lane, K);
  unsigned init, flip;
  lean_int_consts(lane, init, flip);
  LeanState S;
  lean_init(S, lane);
  int have = 0, got = 0;                           // LLRs in the buffer / copied so far
  while (got < needed) {
    int take = C8P_LBUF - have;                    // a round: as much as fits (the left-over differs from round to round)
    if (take > needed - got) take = needed - got;
    for (int i = lane; i < take; i += 64) lb[have + i] = L[got + i];
    have += take; got += take;
    int nb = (int)(((unsigned)have * nlb_inv) >> 16);
    if (nb > nfull - S.done) nb = nfull - S.done;
    if (nb > 0) {
      const int upto = S.done + nb;
      switch (t.cr) {
        case C8P_CR_12: lean_blocks_l<C8P_CR_12>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
        case C8P_CR_23: lean_blocks_l<C8P_CR_23>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
        case C8P_CR_34: lean_blocks_l<C8P_CR_34>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
        default:        lean_blocks_l<C8P_CR_56>(S, K, init, flip, bound, lb, lane, sv, upto, segn); break;
      }
      const int used = nb * nlb;
      have -= used;
      for (int i = 0; i < have; i += 64) {
        const float v = i + lane < have ? lb[used + i + lane] : 0.0f;
        wave_lds_fence();
        if (i + lane < have) lb[i + lane] = v;
      }
      wave_lds_fence();
    }
  }
  if (lane == 0) flags[n + p] = (uint32_t)S.nflt;
  switch (t.cr) {
    case C8P_CR_12: lean_tail_l<C8P_CR_12>(S, K, lb, have, lane, Tn, sv); break;
    case C8P_CR_23: lean_tail_l<C8P_CR_23>(S, K, lb, have, lane, Tn, sv); break;
    case C8P_CR_34: lean_tail_l<C8P_CR_34>(S, K, lb, have, lane, Tn, sv); break;
    default:        lean_tail_l<C8P_CR_56>(S, K, lb, have, lane, Tn, sv); break;
  }
}
