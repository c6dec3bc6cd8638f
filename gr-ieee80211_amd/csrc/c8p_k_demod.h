// c8p_k_demod.h — one OFDM data symbol: equalise, pilot phase, LLRs, de-interleave (demod_sym_1ss / demod_sym_2ss) and k_demod_sym (one wave per symbol)
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_k_classify.h"

// ================================================================================================ data symbols
// procSymQamToLlr (cloud80211phy.cc:2090-2148) for one subcarrier; o[0..nb)
template <int NB> __device__ __forceinline__ void qam_to_llr(cf q, float (&o)[NB])
{
  if constexpr (NB == 1) { o[0] = q.x; }
  else {
    constexpr float sc = NB == 2 ? 1.4142135623730951f : NB == 4 ? 3.1622776601683795f : NB == 6 ? 6.48074069840786f : 13.038404810405298f;
    constexpr int h = NB / 2;
    constexpr float A = NB == 4 ? 2.0f : NB == 6 ? 4.0f : 8.0f;
    q = cmulf(q, sc);
    const float r = q.x, i = q.y;
    o[0] = r; o[h] = i;
    if constexpr (NB >= 4) {
      const float r1 = A - fabsf(r), i1 = A - fabsf(i);
      o[1] = r1; o[h + 1] = i1;
      if constexpr (NB >= 6) {
        const float r2 = A * 0.5f - fabsf(r1), i2 = A * 0.5f - fabsf(i1);
        o[2] = r2; o[h + 2] = i2;
        if constexpr (NB == 8) { o[3] = 2.0f - fabsf(r2); o[h + 3] = 2.0f - fabsf(i2); }
      }
    }
  }
}

// LLRs of one subcarrier scattered through a de-interleave map (procSymDeint* cloud80211phy.cc:2150-2338): out[map[j*NB+b]] = llr[b]
// neg += sum of max(0, -LLR): the decoder's integer loop needs a bound on how far a block can pull a metric down
// Range check for the lean decoder (|LLR| <= 2^20, no NaN): only the two first-level values r, i are tested, against 2^20 - 8.  The
// others follow: o[k+1] = A / 2^k - |o[k]| with A <= 8, so |o[k+1]| <= max(A, |o[k]|) <= 2^20, and they are finite when r, i are.
// (A packet flagged a hair early takes the exact decoder, which gives the same bits.)
#define C8P_LLR_BOUND_M8_BITS 0x497fff80u     // 2^20 - 8 as float bits
template <int NB> __device__ __forceinline__ bool llr_scatter(cf q, int j, const int16_t* __restrict__ map, float* __restrict__ out, float& neg)
{
  float o[NB];
  qam_to_llr<NB>(q, o);
#pragma unroll
  for (int b = 0; b < NB; b++) {
    out[(uint32_t)(uint16_t)map[j * NB + b]] = o[b];   // unsigned 32-bit offsets from a uniform base: no 64-bit address per store
    neg += fmaxf(-o[b], 0.0f);
  }
  uint32_t m = __float_as_uint(o[0]) & 0x7fffffffu;
  if constexpr (NB >= 2) m = max(m, __float_as_uint(o[NB / 2]) & 0x7fffffffu);
  return m > C8P_LLR_BOUND_M8_BITS;
}
__device__ __forceinline__ bool llr_scatter_nb(int nb, cf q, int j, const int16_t* __restrict__ map, float* __restrict__ out, float& neg)
{
  switch (nb) {   // uniform per wave
    case 1: return llr_scatter<1>(q, j, map, out, neg);
    case 2: return llr_scatter<2>(q, j, map, out, neg);
    case 4: return llr_scatter<4>(q, j, map, out, neg);
    case 6: return llr_scatter<6>(q, j, map, out, neg);
    default: return llr_scatter<8>(q, j, map, out, neg);
  }
}
// wave sum on the DPP path (six VALU adds, total in lane 63): invalid source lanes read 0 (bound_ctrl), the row broadcasts
// only write the rows named by their row mask
__device__ __forceinline__ float wave_sum_lane63(float neg)
{
#define C8P_DPP_ADD(CTRL, ROWS) neg += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(neg), CTRL, ROWS, 0xf, true))
  C8P_DPP_ADD(0x111, 0xf);   // row_shr:1
  C8P_DPP_ADD(0x112, 0xf);   // row_shr:2
  C8P_DPP_ADD(0x114, 0xf);   // row_shr:4
  C8P_DPP_ADD(0x118, 0xf);   // row_shr:8   -> lane 15 of every row holds the row's sum
  C8P_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1, 3
  C8P_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's sum
#undef C8P_DPP_ADD
  return neg;
}
// One symbol's entry of negsym[]: an upper bound of the negative-part sum of any C8P_NEGB_WINDOW consecutive LLRs that touch this
// symbol (a window spans at most 2 symbols, 3 when a symbol holds fewer than C8P_NEGB_WINDOW LLRs).  The decoder takes the maximum
// over its packet's symbols (an atomicMax per symbol here cost 0.035 ms per 4096-packet batch: 46 waves per address).
__device__ __forceinline__ void publish_llr_stats(bool bad, float neg, int ncbps, uint32_t* __restrict__ flags, float* __restrict__ negsym,
                                                  uint32_t w, int p)
{
  neg = wave_sum_lane63(neg);
  const unsigned long long anybad = __ballot(bad);
  if (lane_id() == 63) {
    if (anybad) atomicOr(&flags[p], 1u);
    negsym[w] = neg * (ncbps >= C8P_NEGB_WINDOW ? 2.0f : 3.0f);
  }
}

// One OFDM data symbol of a single-stream packet received on antenna 0: FFT output f1 of this lane's bin, Hl = the channel of the
// bin (tag "chan" for legacy packets, k_classify's HT / VHT estimate otherwise) -> equalise, common pilot phase, LLRs scattered
// through the de-interleave map into out[0 .. ncbps).  legacyChanUpdate demod_impl.cc:507-539, nonLegacyChanUpdate :413-447
// (== the nSS = 1 branches of demod2), pilotShift :549-557 (base patterns cloud80211phy.cc:170-174).
// position of this lane's subcarrier in the q[] order of the packet's format, -1 for the null and pilot subcarriers (a kernel that
// demodulates many symbols of one packet computes it once: the nested range tests cost ~11 VALU instructions and a screen of scalar
// branches per symbol)
__device__ __forceinline__ int data_carrier_index(const c8p_pkt_tags& t, int lane)
{
  if (t.format == C8P_F_L) return (!is_null_l(lane) && !is_pilot(lane)) ? qidx_l(lane) : -1;
  return (!is_null_nl(lane) && !is_pilot(lane)) ? qidx_nl(lane) : -1;
}
// HAVE_JQ: the caller hands over data_carrier_index() of this lane in jq
template <bool HAVE_JQ = false>
__device__ __forceinline__ bool demod_sym_1ss(const c8p_pkt_tags& t, int s, int lane, cf f1, cf Hl, const DevTables* __restrict__ T,
                                              float* __restrict__ out, float& neg, int jq = 0)
{
  const int nb = t.nbpscs, mi = nbpsc_index(nb);
  const cf sig = cdiv(f1, Hl);
  bool bad = false;
  if (!HAVE_JQ) jq = data_carrier_index(t, lane);
  if (t.format == C8P_F_L) {
    const float Pp = T->pilotP[(1 + s) % 127];
    cf acc = cmulf(cmulf(shfl_c(sig, 7), 1.0f), Pp);
    acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 21), -1.0f), Pp));
    acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 43), 1.0f), Pp));
    acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 57), 1.0f), Pp));
    const cf ps = cconj(acc);
    const float pa = cabs_(ps);
    if (jq >= 0) {
      const cf q = cdivf(cmul(sig, ps), pa);
      bad = llr_scatter_nb(nb, q, jq, T->deintL[mi], out, neg);
    }
    return bad;
  }
  const int p0 = t.format == C8P_F_VHT ? 4 : 3;
  const float Pp = T->pilotP[(p0 + s) % 127];
  float pa_[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int kk = (k + s) & 3;
    pa_[k] = __uint_as_float(kk == 3 ? 0xbf800000u : 0x3f800000u);         // {1,1,1,-1} rotated left by s; as a bit pattern the
                                                                           // (wave-uniform) value stays in an SGPR
  }
  cf acc = cmulf(cmulf(shfl_c(sig, 7), pa_[2]), Pp);
  acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 21), pa_[3]), Pp));
  acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 43), pa_[0]), Pp));
  acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 57), pa_[1]), Pp));
  const cf ps = cconj(acc);
  const float pa = cabs_(ps);
  if (jq >= 0) {
    const cf q = cdivf(cmul(sig, ps), pa);
    bad = llr_scatter_nb(nb, q, jq, T->deintNL[mi], out, neg);
  }
  return bad;
}

// One OFDM data symbol of a two-stream packet received on two antennas: FFT outputs f1 / f2 of this lane's bin on antenna 0 / 1, the
// bin's channel H4 and (H^H H)^-1 INV4 (k_classify's Mimo record; M also holds the LTF pilot references) -> zero-forcing, the 8-term
// common pilot phase, both streams' LLRs scattered through the de-interleave maps fused with the stream de-parser into out[0 .. ncbps).
// htChanUpdate demod2_impl.cc:471-551 / vhtChanUpdate :553-630, pilotShift demod_impl.cc:549-557 (base patterns cloud80211phy.cc:170-174).
template <bool HAVE_JQ = false>
__device__ __forceinline__ bool demod_sym_2ss(const c8p_pkt_tags& t, int s, int lane, cf f1, cf f2, const cf (&H4)[4], const cf (&INV4)[4],
                                              const Mimo* __restrict__ M, const DevTables* __restrict__ T, float* __restrict__ out, float& neg, int jq = 0)
{
  const float Pp = T->pilotP[((t.format == C8P_F_VHT ? 4 : 3) + s) % 127];
  // pilot patterns rotated left by s
  float pa_[4], pb_[4];
  {
    const bool ht2 = t.format == C8P_F_HT;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int kk = (k + s) & 3;
      // (selected as bit patterns: the values are wave-uniform and stay in SGPRs; as float selects they took 8 VGPRs)
      const uint32_t l4 = kk == 3 ? 0xbf800000u : 0x3f800000u;               // {1,1,1,-1}
      const uint32_t h21 = kk >= 2 ? 0xbf800000u : 0x3f800000u;              // {1,1,-1,-1}
      const uint32_t h22 = (kk == 1 || kk == 2) ? 0xbf800000u : 0x3f800000u; // {1,-1,-1,1}
      pa_[k] = __uint_as_float(ht2 ? h21 : l4);
      pb_[k] = __uint_as_float((t.format == C8P_F_VHT) ? l4 : h22);
    }
  }
  const int nb = t.nbpscs, mi = nbpsc_index(nb);
  cf s1, s2;
  zf_2x2(f1, f2, H4, INV4, s1, s2);
  cf acc = cmul(cmulf(cmulf(shfl_c(s1, 7), pa_[2]), Pp), M->pil1[2]);
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s1, 21), pa_[3]), Pp), M->pil1[3]));
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s1, 43), pa_[0]), Pp), M->pil1[0]));
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s1, 57), pa_[1]), Pp), M->pil1[1]));
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s2, 7), pb_[2]), Pp), M->pil2[2]));
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s2, 21), pb_[3]), Pp), M->pil2[3]));
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s2, 43), pb_[0]), Pp), M->pil2[0]));
  acc = cadd(acc, cmul(cmulf(cmulf(shfl_c(s2, 57), pb_[1]), Pp), M->pil2[1]));
  const cf ps = cconj(acc);
  const float pa = cabs_(ps);
  if (!HAVE_JQ) jq = (!is_null_nl(lane) && !is_pilot(lane)) ? qidx_nl(lane) : -1;
  bool bad = false;
  if (jq >= 0) {
    const cf qa = cdivf(cmul(s1, ps), pa), qb = cdivf(cmul(s2, ps), pa);
    const bool b1 = llr_scatter_nb(nb, qa, jq, T->deint2ss[0][mi], out, neg);
    const bool b2 = llr_scatter_nb(nb, qb, jq, T->deint2ss[1][mi], out, neg);
    bad = b1 || b2;
  }
  return bad;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_demod_sym(const c8p_pkt_in* __restrict__ pk, const cf* __restrict__ iq,
                                                   const cf* __restrict__ chan, const uint32_t* __restrict__ sym2pkt,
                                                   const PktLayout* __restrict__ lay, const DevTables* __restrict__ T, const c8p_pkt_tags* __restrict__ tags,
                                                   const cf* __restrict__ Hin, const Mimo* __restrict__ mimo,
                                                   float* __restrict__ llr, uint32_t* __restrict__ flags, float* __restrict__ negsym,
                                                   int n, uint32_t total_syms, int only_2x2)
{
  // (the wave index is uniform, but the compiler cannot see that through threadIdx: without the readfirstlane the packet's
  // tags and descriptor — 170 bytes — are fetched per lane into VGPRs instead of once into SGPRs)
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
  if (w >= total_syms) return;
  const int lane = lane_id();
  const int p = (int)sym2pkt[w];                   // written by k_classify
  const int s = (int)(w - lay[p].sym_base);
  const c8p_pkt_tags t = tags[p];
  if (t.status != C8P_ST_OK) return;
  float* out = llr + t.llr_off;
  if (t.is_ndp) {                                  // 1024 dummy items (demod_impl.cc:265-271), zero-filled
    if (s == 0 && !only_2x2) for (int i = lane; i < C8P_NDP_TOTAL; i += 64) out[i] = 0.0f;
    return;
  }
  if (s >= t.nsym) return;
  const c8p_pkt_in P = pk[p];
  const bool two = P.n_ant == 2;
  if (only_2x2 && !(two && t.nss == 2)) return;    // c8p_rx: single-stream packets belong to k_rx_1ss
  const cf* x1 = iq + P.iq_off[0] + t.data_off + s * t.nsymsamp + 8;
  out += (size_t)s * t.ncbps;
  bool bad = false;                       // an LLR outside the lean decoder's exactness range (or NaN)
  float neg = 0.0f;                       // this lane's sum of max(0, -LLR)
  cf f1 = wave_fft64(x1[lane], T);
  if (!(two && t.nss == 2)) {
    const cf Hl = t.format == C8P_F_L ? chan[P.chan_off + lane] : Hin[(size_t)p * 64 + lane];
    bad = demod_sym_1ss(t, s, lane, f1, Hl, T, out, neg);
    publish_llr_stats(bad, neg, t.ncbps, flags, negsym, w, p);
    return;
  }

  // 2x2: htChanUpdate demod2_impl.cc:471-551 / vhtChanUpdate :553-630
  const cf* x2 = iq + P.iq_off[1] + t.data_off + s * t.nsymsamp + 8;
  cf f2 = wave_fft64(x2[lane], T);
  const Mimo* M = &mimo[p];
  cf H4[4], INV4[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { H4[c] = M->H[lane][c]; INV4[c] = M->INV[lane][c]; }
  bad = demod_sym_2ss(t, s, lane, f1, f2, H4, INV4, M, T, out, neg);
  publish_llr_stats(bad, neg, t.ncbps, flags, negsym, w, p);
}
