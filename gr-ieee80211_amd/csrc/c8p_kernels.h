// c8p_kernels.h — gfx950 kernels of the 802.11 OFDM RX hot path.
//
//   (arena layout is computed on the host from the descriptors and uploaded with them)
//   k_classify    wave / packet   format detection, SIG decode, channel estimate, SIG-B  (demod_impl.cc:106-219)
//   k_demod_sym   wave / symbol   FFT -> equalise -> pilot phase -> LLR -> de-interleave  (demod_impl.cc:279-314)
//   k_rx_1ss_lds / k_rx_2ss_lds   wave / packet   c8p_rx: symbol demodulation + lean forward pass, LLRs through LDS
//   k_viterbi_lean wave / packet  depuncture + 64-state ACS (c8p_viterbi_lean.h); viterbi_fwd_exact: the clamp-exact recurrence, inside k_finish
//   k_finish      wave / packet   traceback, descramble, A-MPDU walk, CRC-32, PDU records (decode_impl.cc:282-520)
//
// Every kernel is a single-wave (64-lane) program per unit of work: lane = FFT bin in the demod kernels,
// lane = trellis state in the decoder.  All reference citations are relative to /root/reference/lib/.
#pragma once
#include "c8p_dev.h"
#include "c8p_viterbi_lean.h"

// ================================================================================================ parsers (uniform code)
struct ModD {
  int format, sumu, ampdu, nSym, nSymSamp, nSS, nLTF, mcs, len, cr, nBPSCS, nDBPS, nCBPS, nCBPSS, valid;
};

// The decoded SIG bits are handed around as ONE wave-uniform 64-bit word (bit i = i-th bit): checks and field extraction
// then run on the scalar unit instead of 64 lanes each redoing them on bytes from LDS.
__device__ __forceinline__ unsigned long long sig_word(const uint8_t* b, int L)          // b in LDS, L <= 64
{
  const int lane = lane_id();
  return __ballot(lane < L && b[lane] != 0);
}
__device__ __forceinline__ bool crc8_check(unsigned long long w)   // checkBitCrc8 :1367-1403 over bits 0..33 against bits 34..41
{
  unsigned c = 0xff;
  for (int i = 0; i < 34; i++) {
    unsigned fb = (c >> 7) & 1u;
    c = (c << 1) & 0xffu;
    if (fb) c ^= 0x07u;
    if ((w >> i) & 1ull) c ^= 0x07u;
  }
  for (int i = 0; i < 8; i++) if ((w >> (34 + i)) & 1ull) c ^= (1u << (7 - i));
  return (c & 0xffu) == 0xffu;
}
__device__ __forceinline__ bool check_vht_a(unsigned long long w)   // signalCheckVhtA :753-771
{
  const unsigned long long ones = (1ull << 2) | (1ull << 23) | (1ull << 33);
  if ((w & ones) != ones) return false;
  if (!crc8_check(w)) return false;
  if (w & 3ull) return false;
  return true;
}
__device__ __forceinline__ bool check_ht(unsigned long long w)      // signalCheckHt :730-751
{
  if (!((w >> 26) & 1ull)) return false;
  if (!crc8_check(w)) return false;
  if (w & ((7ull << 5) | (7ull << 28) | (3ull << 32))) return false;       // bits 5-7, 28-30, 32-33
  return true;
}
__device__ __forceinline__ void set_rate(ModD& m)               // :1294-1322
{
  m.nCBPSS = m.nBPSCS * 52;
  m.nCBPS = m.nCBPSS * m.nSS;
  m.nDBPS = m.cr == C8P_CR_12 ? m.nCBPS / 2 : m.cr == C8P_CR_23 ? (m.nCBPS * 2) / 3 : m.cr == C8P_CR_34 ? (m.nCBPS * 3) / 4 : (m.nCBPS * 5) / 6;
  if (m.nSS == 1) m.nLTF = 1; else if (m.nSS == 2) m.nLTF = 2; else if (m.nSS <= 4) m.nLTF = 4;
}
__device__ __forceinline__ bool mod_vht(int mcs, ModD& m)       // modParserVht :1225-1323
{
  m.mcs = mcs;
  if (mcs < 0 || mcs > 9) { m.valid = 0; return false; }
  const int nb = mcs == 0 ? 1 : mcs <= 2 ? 2 : mcs <= 4 ? 4 : mcs <= 7 ? 6 : 8;
  const int cr = (mcs == 0 || mcs == 1 || mcs == 3) ? C8P_CR_12 : (mcs == 5) ? C8P_CR_23 : (mcs == 7 || mcs == 9) ? C8P_CR_56 : C8P_CR_34;
  m.nBPSCS = nb; m.cr = cr;
  set_rate(m);
  return true;
}
__device__ __forceinline__ void parse_legacy(int mcs, int len, ModD& m)   // signalParserL :773-850
{
  m = ModD{};
  m.valid = (mcs >= 0 && mcs <= 7);
  if (!m.valid) return;
  const int nb = mcs < 2 ? 1 : mcs < 4 ? 2 : mcs < 6 ? 4 : 6;
  const int ndbps = mcs == 0 ? 24 : mcs == 1 ? 36 : mcs == 2 ? 48 : mcs == 3 ? 72 : mcs == 4 ? 96 : mcs == 5 ? 144 : mcs == 6 ? 192 : 216;
  m.mcs = mcs; m.nBPSCS = nb; m.nDBPS = ndbps; m.nCBPS = 48 * nb; m.nCBPSS = m.nCBPS;
  m.cr = (mcs == 6) ? C8P_CR_23 : (mcs & 1) ? C8P_CR_34 : C8P_CR_12;
  m.len = len; m.nSS = 1; m.sumu = 0; m.nLTF = 0; m.format = C8P_F_L; m.nSymSamp = 80; m.ampdu = 0;
  m.nSym = (len * 8 + 22) / ndbps + (((len * 8 + 22) % ndbps) != 0);
}
__device__ __forceinline__ void parse_ht(unsigned long long w, ModD& m)    // signalParserHt :852-998
{
  m = ModD{};
  const int mcs = (int)(w & 0x7full), len = (int)((w >> 8) & 0xffffull);
  m.valid = 1; m.format = C8P_F_HT; m.sumu = 0;
  m.nSymSamp = ((w >> 31) & 1ull) ? 72 : 80;
  m.ampdu = (int)((w >> 27) & 1ull);
  m.mcs = mcs;
  const int k = mcs & 7;
  m.nBPSCS = k == 0 ? 1 : k <= 2 ? 2 : k <= 4 ? 4 : 6;
  m.cr = (k == 0 || k == 1 || k == 3) ? C8P_CR_12 : (k == 5) ? C8P_CR_23 : (k == 7) ? C8P_CR_56 : C8P_CR_34;
  m.len = len; m.nSS = mcs / 8 + 1;
  set_rate(m);
  m.nSym = (len * 8 + 22) / m.nDBPS + (((len * 8 + 22) % m.nDBPS) != 0);
}
__device__ __forceinline__ void parse_vht_a(unsigned long long w, ModD& m) // signalParserVhtA :1090-1178
{
  m = ModD{};
  m.valid = 1;
  const int gid = (int)((w >> 4) & 63ull);
  m.format = C8P_F_VHT;
  m.nSymSamp = ((w >> 24) & 1ull) ? 72 : 80;
  m.ampdu = 1;
  if (gid == 0 || gid == 63) {
    const int nsts = (int)((w >> 10) & 7ull), mcs = (int)((w >> 28) & 15ull);
    m.sumu = 0; m.nSS = nsts + 1;
    if (m.nSS > 4) m.valid = 0;
    mod_vht(mcs, m);
  } else {
    m.sumu = 1; m.nLTF = 2; m.nSS = 1;
  }
}
__device__ __forceinline__ void parse_vht_b(unsigned long long w, ModD& m) // signalParserVhtB :1180-1223
{
  if (m.sumu) {
    const int len = (int)(w & 0xffffull), mcs = (int)((w >> 16) & 15ull);
    if (!mod_vht(mcs, m)) return;
    m.len = len * 4;
    m.nSym = (m.len * 8 + 16 + 6) / m.nDBPS + (((m.len * 8 + 16 + 6) % m.nDBPS) != 0);
    m.nLTF = 2;
  } else {
    if (((w >> 17) & 7ull) == 7ull) {
      m.len = (int)(w & 0x1ffffull) * 4;
      m.nSym = (m.len * 8 + 16 + 6) / m.nDBPS + (((m.len * 8 + 16 + 6) % m.nDBPS) != 0);
    } else {
      if ((w & 0xfffffull) == 0x422E0ull) { m.len = 0; m.nSym = 0; } else { m.len = -1; m.nSym = -1; }
    }
  }
}

// gather the four pilot-bin values of a per-lane quantity and fold them in the reference's left-to-right order
__device__ __forceinline__ cf pilot_fold_sigb(cf s)   // conj(s7 - s21 + s43 + s57)  (demod_impl.cc:466, cloud80211phy.cc:632)
{
  cf a = shfl_c(s, 7), b = shfl_c(s, 21), c = shfl_c(s, 43), d = shfl_c(s, 57);
  return cconj(cadd(cadd(csub(a, b), c), d));
}

// ================================================================================================ signal front-end
// signal_impl::general_work (lib/signal_impl.cc:108-202) / signal2_impl: CFO de-rotation, L-LTF channel estimate,
// L-SIG demod + 24-step Viterbi + signalCheckLegacy, then the de-rotated packet items.  SURVEY.md §8f-1.
__device__ __forceinline__ cf rot(cf x, float th) { return cmul(x, mk(cosf(th), sinf(th))); }

__global__ __launch_bounds__(64) void k_signal(const c8p_sig_in* __restrict__ ev, const cf* __restrict__ raw,
                                               const uint64_t* __restrict__ out_off, const DevTables* __restrict__ T,
                                               c8p_pkt_in* __restrict__ pk_out, cf* __restrict__ iq_out,
                                               cf* __restrict__ chan_out, int n)
{
  __shared__ float s_llr[48];
  __shared__ uint8_t s_bits[24];
  __shared__ unsigned long long s_w1[49], s_w0[49];
  const int e = blockIdx.x;
  if (e >= n) return;
  const int lane = lane_id();
  const c8p_sig_in E = ev[e];
  const cf* in1 = raw + E.raw_off[0];
  const cf* in2 = raw + E.raw_off[1];
  const uint32_t cap = (E.n_avail >= 224u ? E.n_avail - 224u : 0u) + 320u;
  cf* o1 = iq_out + out_off[e];
  cf* o2 = o1 + cap;
  c8p_pkt_in P;
  memset(&P, 0, sizeof(P));
  P.n_ant = E.n_ant; P.snr = E.snr; P.rssi = E.rssi; P.cfo = E.cfo_rad * 3183098.8618379068f;   // signal_impl.cc:136
  P.iq_off[0] = out_off[e]; P.iq_off[1] = E.n_ant == 2 ? out_off[e] + cap : 0; P.chan_off = 64u * (uint32_t)e;
  cf h = mk(0.f, 0.f);
  int ok = 0, mcs = 0, len = 0, nsamp = 0;
  if (E.n_avail >= 224u) {
    cf f1 = wave_fft64(rot(in1[8 + lane], (float)(lane + 8) * E.cfo_rad), T);
    cf f2 = wave_fft64(rot(in1[72 + lane], (float)(lane + 72) * E.cfo_rad), T);
    cf fs = wave_fft64(rot(in1[152 + lane], (float)(lane + 152) * E.cfo_rad), T);
    const int idx = T->sigDemapNL[lane];
    const bool used_bin = idx >= 0 || is_pilot(lane);
    if (used_bin) h = cdivf(cadd(f1, f2), T->ltfL[lane] * 2.0f);      // procLHSigDemodDeint cloud80211phy.cc:609-627
    cf d = cdiv(fs, h);
    cf ps = pilot_fold_sigb(d);
    float pa = cabs_(ps);
    if (idx >= 0) s_llr[idx] = cdivf(cmul(d, ps), pa).x;
    __syncthreads();
    wave_sig_viterbi(s_llr, s_bits, 24, s_w1, s_w0);
    // signalCheckLegacy cloud80211phy.cc:650-728
    const uint8_t* b = s_bits;
    int par = 0;
    for (int i = 0; i < 17; i++) par += b[i];
    int rate = 0;
    for (int i = 0; i < 4; i++) rate |= ((int)b[i]) << i;
    for (int i = 0; i < 12; i++) len |= ((int)b[i + 5]) << i;
    int ndbps = 24;
    switch (rate) {
      case 11: mcs = 0; ndbps = 24; break; case 15: mcs = 1; ndbps = 36; break;
      case 10: mcs = 2; ndbps = 48; break; case 14: mcs = 3; ndbps = 72; break;
      case 9:  mcs = 4; ndbps = 96; break; case 13: mcs = 5; ndbps = 144; break;
      case 8:  mcs = 6; ndbps = 192; break; case 12: mcs = 7; ndbps = 216; break;
      default: mcs = 0; ndbps = 24; break;
    }
    ok = b[3] && !b[4] && !((par & 1) ^ b[17]) && !(len > 4095 || len < 14);
    nsamp = ((len * 8 + 22 + ndbps - 1) / ndbps) * 80;
    if (ok && (uint32_t)(224 + nsamp) > E.n_avail) {                  // the stream would have to wait for more items
      ok = 0;
      P.lsig_mcs = mcs; P.lsig_len = len; P.lsig_nsamp = nsamp; P.rsv[0] = C8P_SIG_NEED_MORE;
    }
  }
  chan_out[(size_t)e * 64 + lane] = h;
  if (ok) {
    for (int k = lane; k < nsamp; k += 64) {                           // S_COPY signal_impl.cc:164-192
      const float th = (float)(k + 224) * E.cfo_rad;
      o1[k] = rot(in1[224 + k], th);
      if (E.n_ant == 2) o2[k] = rot(in2[224 + k], th);
    }
    for (int k = nsamp + lane; k < nsamp + 320; k += 64) { o1[k] = mk(0.f, 0.f); if (E.n_ant == 2) o2[k] = mk(0.f, 0.f); }
    P.lsig_mcs = mcs; P.lsig_len = len; P.lsig_nsamp = nsamp; P.n_items = (uint32_t)nsamp + 320u;
  }
  if (lane == 0) pk_out[e] = P;
}

// ================================================================================================ classify
struct ClsShared {
  float llr_ht[96], llr_vht[96];
  float inted[52], coded[52];
  cf q0[52], q1[52];
  uint8_t bits[48], bbits[32], intd[52];
  unsigned long long w1[49], w0[49];
  float sssnr[2];
};


// SIG-B tail: de-interleave, 26-step Viterbi, re-encode, per-stream noise -> sssnr (demod_impl.cc:483-504, demod2_impl.cc:710-757)
__device__ __forceinline__ void sigb_finish(ClsShared& S, const DevTables* T, int nss)
{
  const int lane = lane_id();
  if (lane < 52) S.coded[T->deintSigB[lane]] = S.inted[lane];
  __syncthreads();
  wave_sig_viterbi(S.coded, S.bbits, 26, S.w1, S.w0);
  if (lane < 52) {
    const int k = lane >> 1;
    int st = 0;
    for (int j = 0; j < 7; j++) if (k - j >= 0 && S.bbits[k - j]) st |= 1 << j;
    const int enc = __popc(st & ((lane & 1) ? 0117 : 0155)) & 1;
    S.intd[T->intlSigB[lane]] = (uint8_t)enc;
  }
  __syncthreads();
  if (lane == 0) {
    double n0 = 0.0, n1 = 0.0;
    for (int i = 0; i < 52; i++) {
      const float ref = S.intd[i] ? 1.0f : -1.0f;
      float re = S.q0[i].x - ref, im = S.q0[i].y - 0.0f;
      n0 += (double)(re * re + im * im);
      if (nss == 2) {
        re = S.q1[i].x - ref; im = S.q1[i].y - 0.0f;
        n1 += (double)(re * re + im * im);
      }
    }
    S.sssnr[0] = (float)(log10(52.0 / n0) * 10.0);
    S.sssnr[1] = nss == 2 ? (float)(log10(52.0 / n1) * 10.0) : 0.0f;
  }
  __syncthreads();
}

// 2x2 channel estimate, lane = bin (demod2_impl.cc:372-463)
__device__ __forceinline__ void chan_est_2x2(cf f11, cf f21, cf f12, cf f22, bool vht, const DevTables* T,
                                             cf H[4], cf INV[4], cf pil1[4], cf pil2[4])
{
  const int lane = lane_id();
  const float l2 = T->ltfNL[lane] * 0.5f;
  H[0] = cmulf(csub(f11, f12), l2);
  H[1] = cmulf(csub(f21, f22), l2);
  H[2] = cmulf(cadd(f11, f12), l2);
  H[3] = cmulf(cadd(f21, f22), l2);
  if (vht) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      cf lo = shfl_c(H[c], (lane + 63) & 63), hi = shfl_c(H[c], (lane + 1) & 63);
      if (is_pilot(lane)) H[c] = cdivf(cadd(lo, hi), 2.0f);
    }
  }
  cf a = cadd(cmul(H[0], cconj(H[0])), cmul(H[1], cconj(H[1])));
  cf b = cadd(cmul(H[0], cconj(H[2])), cmul(H[1], cconj(H[3])));
  cf c = cadd(cmul(H[2], cconj(H[0])), cmul(H[3], cconj(H[1])));
  cf d = cadd(cmul(H[2], cconj(H[2])), cmul(H[3], cconj(H[3])));
  cf k = cdiv(mk(1.0f, 0.0f), csub(cmul(a, d), cmul(b, c)));
  INV[0] = cmul(k, d);
  INV[1] = cmul(cneg(k), b);
  INV[2] = cmul(cneg(k), c);
  INV[3] = cmul(k, a);
  cf t1 = cadd(cmul(f11, cconj(H[0])), cmul(f21, cconj(H[1])));
  cf t2 = cadd(cmul(f11, cconj(H[2])), cmul(f21, cconj(H[3])));
  cf s1 = cadd(cmul(t1, INV[0]), cmul(t2, INV[2]));
  cf s2 = cadd(cmul(t1, INV[1]), cmul(t2, INV[3]));
  if (lane == 57) { s1 = cneg(s1); s2 = cneg(s2); }
  s1 = cconj(s1); s2 = cconj(s2);
  pil1[2] = shfl_c(s1, 7);  pil2[2] = shfl_c(s2, 7);
  pil1[3] = shfl_c(s1, 21); pil2[3] = shfl_c(s2, 21);
  pil1[0] = shfl_c(s1, 43); pil2[0] = shfl_c(s2, 43);
  pil1[1] = shfl_c(s1, 57); pil2[1] = shfl_c(s2, 57);
}

__device__ __forceinline__ void zf_2x2(cf y1, cf y2, const cf H[4], const cf INV[4], cf& s1, cf& s2)   // demod2_impl.cc:515-518
{
  cf t1 = cadd(cmul(y1, cconj(H[0])), cmul(y2, cconj(H[1])));
  cf t2 = cadd(cmul(y1, cconj(H[2])), cmul(y2, cconj(H[3])));
  s1 = cadd(cmul(t1, INV[0]), cmul(t2, INV[2]));
  s2 = cadd(cmul(t1, INV[1]), cmul(t2, INV[3]));
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_classify(const c8p_pkt_in* __restrict__ pk, const cf* __restrict__ iq,
                                                 const cf* __restrict__ chan, const PktLayout* __restrict__ lay,
                                                 const DevTables* __restrict__ T, c8p_pkt_tags* __restrict__ tags,
                                                 cf* __restrict__ Hout, Mimo* __restrict__ mimo, cf* __restrict__ muchan,
                                                 uint32_t* __restrict__ flags, uint32_t* __restrict__ sym2pkt, int n)
{
  __shared__ ClsShared S;
  const int p = blockIdx.x;
  if (p >= n) return;
  const int lane = lane_id();
  const c8p_pkt_in P = pk[p];
  const bool two = P.n_ant == 2;
  const cf* in1 = iq + P.iq_off[0];
  const cf* in2 = two ? iq + P.iq_off[1] : in1;
  const int nitems = (int)P.n_items;
  const cf HL = chan[P.chan_off + lane];

  ModD m = ModD{};
  int status = C8P_ST_REJ_SHORT;
  int fmt = C8P_F_L;
  cf H = mk(0.f, 0.f);
  cf H4[4], INV4[4], pil1[4], pil2[4];
  for (int c = 0; c < 4; c++) { H4[c] = INV4[c] = pil1[c] = pil2[c] = mk(0.f, 0.f); }
  float sssnr0 = 0.f, sssnr1 = 0.f;
  int data_off = 0, trellis = 0, is_ndp = 0;
  bool done = false;

  if (lane == 0) flags[p] = 0;        // k_demod_sym raises it when an LLR fails the lean decoder's range check
  {                                   // symbol-wave -> packet map of k_demod_sym's grid (no search there)
    const uint32_t sb = lay[p].sym_base, nsb = lay[p].nsym_bound;
    for (uint32_t i = lane; i < nsb; i += 64) sym2pkt[sb + i] = (uint32_t)p;
  }
  muchan[(size_t)p * 128 + lane] = mk(0.f, 0.f);
  muchan[(size_t)p * 128 + 64 + lane] = mk(0.f, 0.f);

  // ---- DEMOD_S_FORMAT (demod_impl.cc:106-148)
  if (!(P.lsig_mcs > 0)) {
    if (nitems < 160) { done = true; }
    else {
      cf f1 = wave_fft64(in1[8 + lane], T);
      cf f2 = wave_fft64(in1[88 + lane], T);
      cf d1 = cdiv(f1, HL), d2 = cdiv(f2, HL);          // procNLSigDemodDeint cloud80211phy.cc:629-648
      cf p1 = pilot_fold_sigb(d1), p2 = pilot_fold_sigb(d2);
      float a1 = cabs_(p1), a2 = cabs_(p2);
      cf m1 = cdivf(cmul(d1, p1), a1), m2 = cdivf(cmul(d2, p2), a2);
      const int idx = T->sigDemapNL[lane];
      if (idx >= 0) {
        S.llr_ht[idx] = m1.y; S.llr_ht[idx + 48] = m2.y;
        S.llr_vht[idx] = m1.x; S.llr_vht[idx + 48] = m2.y;
      }
      __syncthreads();
      const unsigned long long wa = wave_sig_viterbi(S.llr_vht, S.bits, 48, S.w1, S.w0);
      if (check_vht_a(wa)) { parse_vht_a(wa, m); fmt = C8P_F_VHT; }
      else {
        __syncthreads();
        const unsigned long long wh = wave_sig_viterbi(S.llr_ht, S.bits, 48, S.w1, S.w0);
        if (check_ht(wh)) { parse_ht(wh, m); fmt = C8P_F_HT; }
      }
    }
  }

  if (!done && fmt == C8P_F_VHT) {
    // ---- DEMOD_S_VHT (demod_impl.cc:150-178, demod2_impl.cc:150-180)
    if (!m.valid && !m.sumu) { status = C8P_ST_REJ_UNSUP; done = true; }
    else if (m.nSS > 2) { status = C8P_ST_REJ_VHT; done = true; }
    else if (nitems < 160 + 80 + m.nLTF * 80 + 80) { done = true; }
    else if (two && m.sumu) { status = C8P_ST_REJ_UNSUP; done = true; }
    else {
      const cf* l1 = in1 + 240;
      const cf* l2 = in2 + 240;
      int nss_eff = 1;
      if (!two || m.nSS == 1) {
        cf f1 = wave_fft64(l1[8 + lane], T);
        if (!two && m.sumu) {                                   // demod_impl.cc:347-379
          cf f2 = wave_fft64(l1[88 + lane], T);
          if (P.mupos == 0) H = cdivf(csub(f1, f2), T->ltfNL[lane] * 2.0f);
          else H = cdivf(cadd(cdivf(f1, T->ltfNL[lane]), cdivf(f2, T->ltfNLvht22[lane])), 2.0f);
        } else {
          if (!two && !(m.nSS == 1 && m.nLTF == 1)) {           // :393-410 channel-sounding capture
            muchan[(size_t)p * 128 + lane] = l1[8 + lane];
            muchan[(size_t)p * 128 + 64 + lane] = l1[88 + lane];
          }
          H = cdivf(f1, T->ltfNL[lane]);
        }
        cf fb = wave_fft64(l1[m.nLTF * 80 + 8 + lane], T);      // vhtSigBDemod :449-481
        cf sig = cdiv(fb, H);
        cf ps = pilot_fold_sigb(sig);
        float pa = cabs_(ps);
        if (!is_null_nl(lane) && !is_pilot(lane)) {
          const int j = qidx_nl(lane);
          cf q = cdivf(cmul(sig, ps), pa);
          S.q0[j] = q; S.inted[j] = q.x;
        }
      } else {
        cf f11 = wave_fft64(l1[8 + lane], T), f21 = wave_fft64(l2[8 + lane], T);
        cf f12 = wave_fft64(l1[88 + lane], T), f22 = wave_fft64(l2[88 + lane], T);
        chan_est_2x2(f11, f21, f12, f22, true, T, H4, INV4, pil1, pil2);
        cf y1 = wave_fft64(l1[m.nLTF * 80 + 8 + lane], T), y2 = wave_fft64(l2[m.nLTF * 80 + 8 + lane], T);
        cf s1, s2;
        zf_2x2(y1, y2, H4, INV4, s1, s2);
        // demod2_impl.cc:679-688
        cf acc = cmul(shfl_c(s1, 7), pil1[2]);
        acc = csub(acc, cmul(shfl_c(s1, 21), pil1[3]));
        acc = cadd(acc, cmul(shfl_c(s1, 43), pil1[0]));
        acc = cadd(acc, cmul(shfl_c(s1, 57), pil1[1]));
        acc = cadd(acc, cmul(shfl_c(s2, 7), pil2[2]));
        acc = csub(acc, cmul(shfl_c(s2, 21), pil2[3]));
        acc = cadd(acc, cmul(shfl_c(s2, 43), pil2[0]));
        acc = cadd(acc, cmul(shfl_c(s2, 57), pil2[1]));
        cf ps = cconj(acc);
        float pa = cabs_(ps);
        if (!is_null_nl(lane) && !is_pilot(lane)) {
          const int j = qidx_nl(lane);
          cf qa = cdivf(cmul(s1, ps), pa), qb = cdivf(cmul(s2, ps), pa);
          S.q0[j] = qa; S.q1[j] = qb;
          S.inted[j] = (qa.x + qb.x) / 2.0f;
        }
        nss_eff = 2;
      }
      __syncthreads();
      sigb_finish(S, T, nss_eff);
      sssnr0 = S.sssnr[0]; sssnr1 = S.sssnr[1];
      parse_vht_b(sig_word(S.bbits, 26), m);
      const int nSymL = (P.lsig_len * 8 + 22 + 23) / 24;
      const bool len_ok = two ? (m.len > 0) : (m.len >= 0);
      if (!m.valid) { status = C8P_ST_REJ_UNSUP; done = true; }
      else if (!(len_ok && m.len <= 4095 && m.nSS <= 2 && (nSymL * 80) >= (m.nSym * m.nSymSamp + 160 + 80 + m.nLTF * 80 + 80))) {
        status = C8P_ST_REJ_VHT; done = true;
      } else {
        trellis = m.nSym * m.nDBPS;
        data_off = 160 + 80 + m.nLTF * 80 + 80;
        is_ndp = (m.nSym == 0);
        if (!two && m.nSS == 2 && !is_ndp) { status = C8P_ST_REJ_UNSUP; done = true; }
      }
    }
  } else if (!done && fmt == C8P_F_HT) {
    // ---- DEMOD_S_HT (demod_impl.cc:180-205, demod2_impl.cc:182-214)
    if (m.nSS > 2) { status = C8P_ST_REJ_HT; done = true; }
    else if (nitems < 160 + 80 + m.nLTF * 80) { done = true; }
    else {
      const cf* l1 = in1 + 240;
      const cf* l2 = in2 + 240;
      if (!two || m.nSS == 1) {
        cf f1 = wave_fft64(l1[8 + lane], T);
        H = cdivf(f1, T->ltfNL[lane]);
      } else {
        cf f11 = wave_fft64(l1[8 + lane], T), f21 = wave_fft64(l2[8 + lane], T);
        cf f12 = wave_fft64(l1[88 + lane], T), f22 = wave_fft64(l2[88 + lane], T);
        chan_est_2x2(f11, f21, f12, f22, false, T, H4, INV4, pil1, pil2);
      }
      const int nSymL = (P.lsig_len * 8 + 22 + 23) / 24;
      if (!(m.len > 0 && m.len <= 4095 && m.nSS <= 2 && (nSymL * 80) >= (m.nSym * m.nSymSamp + 160 + 80 + m.nLTF * 80))) {
        status = C8P_ST_REJ_HT; done = true;
      } else if (!two && m.nSS == 2) { status = C8P_ST_REJ_UNSUP; done = true; }
      else {
        trellis = m.len * 8 + 22;
        data_off = 160 + 80 + m.nLTF * 80;
      }
    }
  } else if (!done) {
    // ---- DEMOD_S_LEGACY (demod_impl.cc:207-219)
    parse_legacy(P.lsig_mcs, P.lsig_len, m);
    if (!m.valid) { status = C8P_ST_REJ_UNSUP; done = true; }
    else { trellis = m.len * 8 + 22; data_off = 0; }
  }

  if (!done) {
    if (!is_ndp && data_off + m.nSym * m.nSymSamp >= nitems) status = C8P_ST_REJ_SHORT;   // DEMOD loop bound :283
    else status = C8P_ST_OK;
  }

  // ---- DEMOD_S_WRTAG (demod_impl.cc:221-277)
  if (lane == 0) {
    c8p_pkt_tags t;
    memset(&t, 0, sizeof(t));
    t.status = status;
    t.cfo = P.cfo; t.snr = P.snr; t.rssi = P.rssi; t.seq = P.seq;
    t.llr_off = lay[p].llr_off;
    if (status == C8P_ST_OK) {
      t.format = m.format; t.mcs = m.mcs; t.len = m.len; t.cr = m.cr; t.ampdu = m.ampdu;
      t.trellis = trellis; t.total = is_ndp ? C8P_NDP_TOTAL : m.nSym * m.nCBPS;
      t.nsym = m.nSym; t.nss = m.nSS; t.nltf = m.nLTF; t.ncbps = m.nCBPS; t.ncbpss = m.nCBPSS;
      t.ndbps = m.nDBPS; t.nbpscs = m.nBPSCS; t.nsymsamp = m.nSymSamp; t.sumu = m.sumu;
      t.is_ndp = is_ndp; t.data_off = data_off; t.sssnr0 = sssnr0; t.sssnr1 = sssnr1;
    }
    tags[p] = t;
  }
  Hout[(size_t)p * 64 + lane] = H;
  if (two) {
    Mimo* M = &mimo[p];
#pragma unroll
    for (int c = 0; c < 4; c++) { M->H[lane][c] = H4[c]; M->INV[lane][c] = INV4[c]; }
    if (lane < 4) { M->pil1[lane] = pil1[lane]; M->pil2[lane] = pil2[lane]; }
  }
}

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
    for (int i = lane_id(); i < t.total; i += 64) { const float v = L[i]; bad |= bad_llr(v); mneg = fmaxf(mneg, -v); }
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
#ifdef C8P_EXP_CLOCK   // experiment build (tools/exp_clock.py): shader cycles of this wave and where it ran, instead of the statistics
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
#ifdef C8P_EXP_CLOCK
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

// ================================================================================================ c8p_rx, single-stream packets
// Symbol demodulation AND the lean forward pass in one wave per packet: the wave demodulates OFDM symbol s into the packet's LLR
// area and right away runs the survivor blocks whose LLRs are complete, reading them back through the scalar cache.  k_demod_sym's
// 46 short waves per packet (and their launch) disappear, the next symbol's samples are in flight while the current one is decoded,
// and the decoder never waits for HBM: +3 % packets/s over the two kernels.  What it does NOT do on this part is save
// HBM reads: tools/probe_l2rw.hip shows that a read of a line the same wave just wrote is a TCC hit and still an EA read request
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
#ifdef C8P_EXP_CLOCK   // experiment build (tools/exp_clock.py): shader cycles of this wave and where it ran, instead of the statistics
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
#ifdef C8P_EXP_CLOCK
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
// stores nor their read-back (an EA read even when it hits L2, tools/probe_l2rw.hip) exist any more: 1.58 -> 1.1 GB per
// 4096-packet step.  Buffer (C8P_LBUF floats): 59 left over + two 64-QAM symbols (one 256-QAM symbol = 416 LLRs).  A packet with an LLR outside the integer
// loop's range is rare: the wave flags it and stops; k_finish demodulates a flagged packet into the LLR arena (redemod_packet), runs the
// clamp-exact forward pass and goes on with its traceback.  Symbols that only carry padding behind the last trellis step are not demodulated at all.
#define C8P_LBUF 688
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_rx_1ss_lds(const c8p_pkt_in* __restrict__ pk, const cf* __restrict__ iq, const cf* __restrict__ chan,
                                                const PktLayout* __restrict__ lay, const DevTables* __restrict__ T,
                                                const c8p_pkt_tags* __restrict__ tags, const cf* __restrict__ Hin, float* __restrict__ llr_tap,
                                                ulonglong2* __restrict__ surv, uint32_t* __restrict__ flags, int n)
{
  const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (p >= n) return;
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
  while (s < t.nsym && s * t.ncbps < needed) {
    // Nothing that depends on the lane lives across a phase: the lane index is taken afresh (opaque) in each, the decoder's state
    // (metric, survivor offset) waits in LDS while the symbols are demodulated, its lane constants were written there once, and
    // the next symbol's sample waits there while the blocks run.  acc / pend are dead between blocks.
    {
      const int lane_p = lane_fresh();
      park[wv][lane_p] = make_uint2(__float_as_uint(S.pre), S.voff);
      asm volatile("; decoder state parked" : "=v"(S.pre), "=v"(S.voff));
    }
#pragma unroll 1
    do {
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
      switch (t.cr) {
        case C8P_CR_12: lean_blocks_l<C8P_CR_12>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
        case C8P_CR_23: lean_blocks_l<C8P_CR_23>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
        case C8P_CR_34: lean_blocks_l<C8P_CR_34>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
        default:        lean_blocks_l<C8P_CR_56>(S, K, init, flip, bound, lb, lane_d, sv, upto, segn); break;
      }
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
  LeanK K[6];
  lean_k_unpack(parkK[wv][lane_t].x, lane_t, K);
  switch (t.cr) {
    case C8P_CR_12: lean_tail_l<C8P_CR_12>(S, K, lb, have, lane_t, Tn, sv); break;
    case C8P_CR_23: lean_tail_l<C8P_CR_23>(S, K, lb, have, lane_t, Tn, sv); break;
    case C8P_CR_34: lean_tail_l<C8P_CR_34>(S, K, lb, have, lane_t, Tn, sv); break;
    default:        lean_tail_l<C8P_CR_56>(S, K, lb, have, lane_t, Tn, sv); break;
  }
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
                                                ulonglong2* __restrict__ surv, uint32_t* __restrict__ flags, int n)
{
  const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // wave-uniform
  if (p >= n) return;
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

// Two packets per wave (c8p_viterbi_lean.h, bottom; tools/gen_lean2_asm.py), c8p_decode entry (C8P_DECODE_LDS=2; tests and the
// decode-only measurement of the layout).  Wave w takes packets 2w, 2w+1 if both can be decoded by the lean path and share the code
// rate and the trellis length; it marks them in done2[] and k_viterbi_lean_lds skips them.  LLRs go through the packets' LDS
// buffers in rounds as in k_viterbi_lean_lds; blocks whose metrics are not provably positive and the partial last block go through
// LDS into the lane = index layout and run the one-packet code, one packet at a time.
__global__ __launch_bounds__(256) void k_viterbi_lean2_lds(const c8p_pkt_tags* __restrict__ tags, const PktLayout* __restrict__ lay,
                                                           const float* __restrict__ llr, ulonglong2* __restrict__ surv,
                                                           uint32_t* __restrict__ flags, uint8_t* __restrict__ done2, int n)
{
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (2 * w + 1 >= n) return;                      // (a last, single packet goes to k_viterbi_lean_lds)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = lane_id(), half = lane >> 5, l = lane & 31;
  const int p = 2 * w + half;
  const c8p_pkt_tags t0 = tags[2 * w], t1 = tags[2 * w + 1];     // wave-uniform (scalar loads)
  if (!(decodable(t0) && decodable(t1) && flags[2 * w] == 0 && flags[2 * w + 1] == 0 && t0.cr == t1.cr && t0.trellis == t1.trellis)) return;
  if (l == 0) done2[p] = 1;
  const int cr = __builtin_amdgcn_readfirstlane(t0.cr);
  const int Tn = __builtin_amdgcn_readfirstlane(t0.trellis), nfull = Tn / C8P_SV_BLK, segn = c8p_seg_blocks(Tn);
  const int nlb = cr == C8P_CR_12 ? LeanRate<C8P_CR_12>::NLB : cr == C8P_CR_23 ? LeanRate<C8P_CR_23>::NLB :
                  cr == C8P_CR_34 ? LeanRate<C8P_CR_34>::NLB : LeanRate<C8P_CR_56>::NLB;
  const unsigned nlb_inv = cr == C8P_CR_12 ? 1093u : cr == C8P_CR_23 ? 1457u : cr == C8P_CR_34 ? 1639u : 1821u;
  const int needed = __builtin_amdgcn_readfirstlane(llr_needed(cr, Tn));
  const float bound = fmaf(__uint_as_float(flags[2 * n + p]), 1.01f, 1e-30f);     // per half (lean_bound)
  unsigned voff = (unsigned)(lay[p].surv_off * 16u) + (unsigned)l * 4u;             // byte offset into the survivor arena (< 4 GB)
  unsigned* const svb = (unsigned*)surv;
  __shared__ __attribute__((aligned(16))) float lbuf[4][2][C8P_LBUF];
  __shared__ float xch_all[4][2][64];              // [wave][packet][index D]: the bridge between the two layouts
  float (*xch)[64] = xch_all[wv];
  float* const lbh = lbuf[wv][half];               // this half's buffer
  const unsigned la0 = (unsigned)(size_t)lbh;
  const float* const Lh = llr + (half ? t1.llr_off : t0.llr_off);
  float m0 = l == 0 ? 0.0f : C8P_NEG_INIT, m1 = C8P_NEG_INIT;
  int have = 0, got = 0, done = 0, segc = 0, nflt = 0;
  while (got < needed) {
    int take = C8P_LBUF - have;
    if (take > needed - got) take = needed - got;
    for (int i = l; i < take; i += 32) lbh[have + i] = Lh[got + i];
    have += take; got += take;
    int nb = (int)(((unsigned)have * nlb_inv) >> 16);
    if (nb > nfull - done) nb = nfull - done;
    const int upto = done + nb, done_r = done;
    unsigned laddr = la0;
    while (done < upto) {
      // integer loop, both packets at once, as long as every metric of both stays above its packet's bound
      if (__ballot(fminf(m0, m1) < bound) == 0ull) {
        const int l5 = lane_fresh() & 31;             // (afresh: the constants must not be hoisted and held across the other phases)
        LeanK K2[6];
        lean2_k_consts(l5, K2);
        unsigned initx, inity, flip;
        lean4_int_consts(l5, initx, inity, flip);
        int left = __builtin_amdgcn_readfirstlane(upto - done), sc = __builtin_amdgcn_readfirstlane(segc);   // (asm "+s" operands)
        switch (cr) {
          case C8P_CR_12: lean2_lds<C8P_CR_12>(m0, m1, K2, initx, inity, flip, bound, sc, segn, voff, laddr, svb, left); break;
          case C8P_CR_23: lean2_lds<C8P_CR_23>(m0, m1, K2, initx, inity, flip, bound, sc, segn, voff, laddr, svb, left); break;
          case C8P_CR_34: lean2_lds<C8P_CR_34>(m0, m1, K2, initx, inity, flip, bound, sc, segn, voff, laddr, svb, left); break;
          default:        lean2_lds<C8P_CR_56>(m0, m1, K2, initx, inity, flip, bound, sc, segn, voff, laddr, svb, left); break;
        }
        segc = sc;
        done = upto - left;
        if (done == upto) break;
      }
      // one block of each packet with the float step in the lane = index layout
      xch[half][l] = m0; xch[half][32 + l] = m1;
      const bool seg_end = segc + 1 >= segn;
#pragma unroll 1
      for (int pp = 0; pp < 2; pp++) {
        const int q = __builtin_amdgcn_readfirstlane(2 * w + pp);
        const int lane = lane_fresh();
        LeanK K[6];
        lean_k_consts(lane, K);
        float pre = xch[pp][lane];
        unsigned acc = 0;
        const unsigned la = (unsigned)(size_t)lbuf[wv][pp] + (unsigned)((done - done_r) * nlb * 4);
        switch (cr) {
          case C8P_CR_12: lean_block_flt_lds<C8P_CR_12>(pre, acc, K, la); break;
          case C8P_CR_23: lean_block_flt_lds<C8P_CR_23>(pre, acc, K, la); break;
          case C8P_CR_34: lean_block_flt_lds<C8P_CR_34>(pre, acc, K, la); break;
          default:        lean_block_flt_lds<C8P_CR_56>(pre, acc, K, la); break;
        }
        unsigned word = acc & ~(C8P_SV_HINT | 0x80000000u);
        if (seg_end) {
          float mx = pre;
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
          if (pre == mx) word |= C8P_SV_HINT;
        }
        unsigned* sv = (unsigned*)(surv + lay[q].surv_off);
        sv[done * 64 + lane] = word;
        xch[pp][lane] = pre;
      }
      m0 = xch[half][l]; m1 = xch[half][32 + l];
      segc = seg_end ? 0 : segc + 1;
      done++; nflt++;
      voff += 256u;
      laddr += (unsigned)(nlb * 4);
    }
    // what is left over moves to the front, each half its own buffer (one lane per LLR, 32 per pass; reads before writes)
    const int used = nb * nlb;
    have -= used;
    if (used > 0)
      for (int i = 0; i < have; i += 32) {
        const float v = i + l < have ? lbh[used + i + l] : 0.0f;
        if (i + l < have) lbh[i + l] = v;
      }
  }
  if (l == 0) flags[n + p] = (uint32_t)nflt;
  // the partial last block, packet by packet
  xch[half][l] = m0; xch[half][32 + l] = m1;
#pragma unroll 1
  for (int pp = 0; pp < 2; pp++) {
    const int q = __builtin_amdgcn_readfirstlane(2 * w + pp);
    const int lane = lane_fresh();
    LeanK K[6];
    lean_k_consts(lane, K);
    LeanState S;
    lean_init(S, lane);
    S.pre = xch[pp][lane];
    S.voff = (unsigned)lane * 4u + (unsigned)nfull * 256u;
    unsigned* sv = (unsigned*)(surv + lay[q].surv_off);
    switch (cr) {
      case C8P_CR_12: lean_tail_l<C8P_CR_12>(S, K, lbuf[wv][pp], have, lane, Tn, sv); break;
      case C8P_CR_23: lean_tail_l<C8P_CR_23>(S, K, lbuf[wv][pp], have, lane, Tn, sv); break;
      case C8P_CR_34: lean_tail_l<C8P_CR_34>(S, K, lbuf[wv][pp], have, lane, Tn, sv); break;
      default:        lean_tail_l<C8P_CR_56>(S, K, lbuf[wv][pp], have, lane, Tn, sv); break;
    }
  }
}

// ================================================================================================ finish
#define C8P_FIN_REDEMOD 1    // k_finish mode bits: flagged packets have no LLRs in memory yet (on-chip kernels of c8p_rx)
#define C8P_FIN_TAP     2    //                     c8p_cfg.keep_llr = 2: also fill the LLR area of NDPs and packets that are not decoded
struct FinCold { const c8p_pkt_in* pk; const cf* iq; const cf* chan; const cf* Hin; const Mimo* mimo; float* llr; };
#define CRC_POLY 0xedb88320u
__device__ __forceinline__ uint32_t multmodp(uint32_t a, uint32_t b)   // a(x)*b(x) mod P, reflected
{
  uint32_t mm = 1u << 31, pr = 0;
#pragma unroll 4
  for (int i = 0; i < 32; i++) {
    if (a & mm) pr ^= b;
    mm >>= 1;
    b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
  }
  return pr;
}
__device__ __forceinline__ uint32_t x8n_modp(uint32_t nbytes, const DevTables* T)   // x^(8 n) mod P
{
  if (nbytes < (uint32_t)C8P_X8N) return T->x8n[nbytes];    // one load instead of ~11 multiplications of 32 steps each
  uint32_t pr = 1u << 31;
  uint32_t k = 3;
  while (nbytes) {
    if (nbytes & 1u) pr = multmodp(T->x2n[k & 31], pr);
    nbytes >>= 1; k++;
  }
  return pr;
}
__device__ __forceinline__ uint32_t byte_at(const unsigned long long* bw, int bitoff)   // bitoff % 8 == 0
{
  return (uint32_t)((bw[bitoff >> 6] >> (bitoff & 63)) & 0xffull);
}
// zlib-compatible CRC-32 of `len` bytes starting at bit offset `bitoff` of the packed bit words, wave-parallel.
__device__ __forceinline__ uint32_t wave_crc32(const unsigned long long* bw, int bitoff, int len, const DevTables* T, const uint32_t* crcT)
{
  const int lane = lane_id();
  const int seg = (len + 63) >> 6;
  int beg = lane * seg, end = beg + seg;
  if (beg > len) beg = len;
  if (end > len) end = len;
  uint32_t c = lane == 0 ? 0xffffffffu : 0u;
  for (int i = beg; i < end; i++) c = crcT[(c ^ byte_at(bw, bitoff + 8 * i)) & 0xff] ^ (c >> 8);
  if (c != 0u && len - end > 0) c = multmodp(x8n_modp((uint32_t)(len - end), T), c);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o, 64);
  return c ^ 0xffffffffu;
}

__device__ __forceinline__ void emit_pdu(uint8_t* area, uint32_t cap, uint32_t& used, const unsigned long long* bw, int bitoff,
                                         int len, int format, int mcs)
{
  // record = {u16 n} + blob {format, len lo, len hi, MPDU[len], mcs}   (decode_impl.cc:359-361,414-419)
  const int lane = lane_id();
  const uint32_t n = (uint32_t)len + 4u;
  if (used + 2u + n > cap) return;
  uint8_t* r = area + used;
  if (lane == 0) {
    r[0] = (uint8_t)(n & 0xff); r[1] = (uint8_t)(n >> 8);
    r[2] = (uint8_t)format; r[3] = (uint8_t)(len % 256); r[4] = (uint8_t)(len / 256);
    r[5 + len] = (uint8_t)mcs;
  }
  for (int i = lane; i < len; i += 64) r[5 + i] = (uint8_t)byte_at(bw, bitoff + 8 * i);
  used += 2u + n;
}

__global__ __launch_bounds__(64) void k_finish(const c8p_pkt_tags* __restrict__ tags, const PktLayout* __restrict__ lay,
                                               ulonglong2* surv, const cf* __restrict__ muchan,
                                               const DevTables* __restrict__ T, c8p_pkt_out* __restrict__ outs,
                                               uint8_t* __restrict__ pdu, uint8_t* __restrict__ bits_o,
                                               uint8_t* __restrict__ sbits_o, const uint32_t* __restrict__ flags,
                                               int keep_bits, int n,
                                               // the packets the lean kernels left: their forward pass runs here (viterbi_fwd_exact on the LLR area),
                                               // after a demodulation into that area if the batch came through the on-chip kernels of c8p_rx
                                               // (what that needs comes through a record in memory, read only on that path: as kernel
                                               // arguments the seven values cost the traceback 25 spilled SGPRs)
                                               int mode, const FinCold* __restrict__ cold)
{
  constexpr int TS = 68;                                  // tile row stride in dwords (16-byte aligned rows, banks shifted by 4)
  // tile[row][lane of the forward pass]: the current survivor block of every segment.  After the traceback the same
  // memory holds bw[], the decoded bits packed 64 per word.
  __shared__ __attribute__((aligned(16))) uint32_t tile[64 * TS];
  // lane D at the entry of every 6-step group = 6 decoded bits; later the CRC-32 byte table.  Dynamic: sized by the launch for the
  // longest trellis of the batch (the launch passes it: 3.8 KB at cfg 2 instead of the 5.5 KB of the longest legal trellis), which lets
  // 7 instead of 6 of these one-wave workgroups share a CU's LDS
  extern __shared__ __attribute__((aligned(16))) uint8_t snaps[];
  __shared__ uint16_t s_btop[64];
  unsigned long long* bw = (unsigned long long*)tile;      // (C8P_DECODE_T_MAX + 63) / 64 + 2 words <= 64 * TS / 2
  static_assert(((C8P_DECODE_T_MAX + 63) / 64 + 2) * 8 <= 64 * TS * 4, "bw must fit in the tile");
  const int p = blockIdx.x;
  if (p >= n) return;
  const int lane = lane_id();
  const c8p_pkt_tags t = tags[p];
  const PktLayout Lp = lay[p];
  uint8_t* area = pdu + Lp.pdu_off;
  uint32_t used = 0;
  int n_pdu = 0, n_fail = 0, tb_passes = 0, n_refetch = 0;    // n_refetch: survivor blocks fetched by the passes after the first

  if (mode != 0 && t.status == C8P_ST_OK) {
    // LLRs that are wanted in memory and are not there yet: a packet the on-chip kernels flagged (C8P_FIN_REDEMOD: all of it, for the
    // clamp-exact forward pass below) and, with the tap (C8P_FIN_TAP), what those kernels skip — NDPs, packets that are not decoded,
    // the symbols that only carry padding behind the last trellis step
    const bool dec = decodable(t);
    const bool left = dec && !(flags && flags[p] == 0);
    int s0 = -1;
    if (left) s0 = (mode & C8P_FIN_REDEMOD) ? 0 : -1;
    else if (mode & C8P_FIN_TAP) {
      s0 = dec ? (llr_needed(t.cr, t.trellis) + t.ncbps - 1) / t.ncbps : 0;
      if (dec && s0 >= t.nsym) s0 = -1;
    }
    if (s0 >= 0) {
      const FinCold C = *cold;
      redemod_packet(C.pk[p], t, p, C.iq, C.chan, C.Hin, C.mimo, T, C.llr + t.llr_off, s0);
      __threadfence();                                     // this wave's LLR stores before its own loads through other lanes
    }
  }
  if (t.status == C8P_ST_OK && t.len <= C8P_DECODE_B_MAX && t.trellis <= C8P_DECODE_T_MAX && t.trellis == 0) {
    // NDP channel report (decode_impl.cc:100-121)
    const uint32_t nb = C8P_NDP_BLOB_BYTES;
    if (lane == 0) {
      area[0] = (uint8_t)(nb & 0xff); area[1] = (uint8_t)(nb >> 8);
      area[2] = C8P_F_VHT_CHAN; area[3] = 0; area[4] = 4;
    }
    const uint8_t* src = (const uint8_t*)(muchan + (size_t)p * 128);
    for (int i = lane; i < 1024; i += 64) area[5 + i] = src[i];
    used = 2u + nb; n_pdu = 1;
  } else if (decodable(t)) {
    const int Tn = t.trellis;
    const ulonglong2* SV = surv + Lp.surv_off;
    const int nch = (Tn + 63) >> 6;
    if (flags && flags[p] == 0) {
      // ---- lean survivors (blocks of 30 steps, dword[lane] per block, rotating lane map): segment-parallel traceback.
      // Lane j walks steps [j*S, (j+1)*S) backwards from an assumed end lane; the assumption of segment j is the
      // begin lane segment j+1 computed.  Iterating to a fixed point makes every boundary consistent with the
      // true end (lane 0 = state 0 at T), i.e. the result is the reference's full traceback (vstb_end :282-302).
      //  * state bit k lives in lane bit (k+t) mod 6, so lane D at time 6m+6 spells the six decoded bits of times
      //    6m..6m+5 (bit j = time 6m+j): the walk only records D at the entry of every 6-step group (snaps[]).
      //  * a re-walk stops as soon as it meets the previous iteration's path (same D at a tile entry): everything
      //    below is unchanged, so later iterations only touch the ~2 blocks it takes survivor paths to merge.
      //  * a tile = the current 256-byte block of every active segment, prefetched into registers with coalesced
      //    16-byte loads while the previous tile is walked, then staged in LDS; the walk reads tile[segment][D].
      const uint32_t* SW = (const uint32_t*)SV;
      // steps per segment: whole survivor blocks (30 steps = 5 groups), so that no block is fetched by two segments
      const int S = C8P_SV_BLK * c8p_seg_blocks(Tn);
      const int nseg = (Tn + S - 1) / S;
      const bool have = lane < nseg;
      const int tbeg = lane * S;
      const int tend = min(tbeg + S, Tn);
      const int btop = have ? (tend - 1) / C8P_SV_BLK : 0;                 // highest block of this segment
      const int nblk = have ? btop - tbeg / C8P_SV_BLK + 1 : 0;
      s_btop[lane] = (uint16_t)btop;
      int ntile = nblk;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) ntile = max(ntile, __shfl_xor(ntile, o, 64));
      int Dend = 0, Dbeg = 0;
      bool redo = have;
      __syncthreads();
      // rows this lane fetches: instruction j covers rows 4j .. 4j+3, 16 lanes x 16 bytes per row
      const int frow = lane >> 4, fcol = (lane & 15) * 4;
      const uint32_t fcol4 = (uint32_t)fcol * 4u;
      // 16 in-flight 16-byte loads per lane, kept in named registers (an indexed array ends up in scratch)
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define C8P_DECL(j) const int fb##j = s_btop[j * 4 + frow]; uint4 pf##j;
      C8P_ROWS16(C8P_DECL)
      // byte offset inside the packet's survivor area (32 bits, uniform base): block (fb_j - i) of an active row, block 0 of an
      // idle one.  mine_ = this lane's 16 row-activity bits, one per nibble
#define C8P_LOAD(j) pf##j = *(const uint4*)((const char*)SW + ((((uint32_t)(fb##j - i_) << 8) & (uint32_t)__builtin_amdgcn_sbfe((int)(j < 8 ? mlo_ : mhi_), ((j & 7) * 4), 1)) + fcol4));
#define C8P_PREFETCH(i, rows) do { const int i_ = (i); const unsigned long long m_ = (rows) >> frow; \
    const uint32_t mlo_ = (uint32_t)m_, mhi_ = (uint32_t)(m_ >> 32); C8P_ROWS16(C8P_LOAD) } while (0)
#define C8P_STAGE(j) *(uint4*)(&tile[(j * 4 + frow) * TS + fcol]) = pf##j;
      for (int it = 0; it < 66; it++) {
        int D = Dend;
        bool act = redo;
        unsigned long long am = __ballot(act && 0 < nblk);
        if (it > 0) n_refetch += __popcll(am);
        C8P_PREFETCH(0, am);                              // idle rows re-read block 0
        for (int i = 0; i < ntile && am; i++) {
          __syncthreads();
          C8P_ROWS16(C8P_STAGE)
          __syncthreads();
          const unsigned long long nxt_rows = __ballot(act && i + 1 < nblk);
          if (it > 0) n_refetch += __popcll(nxt_rows);
          C8P_PREFETCH(i + 1, nxt_rows);                   // in flight during the walk below
          if (it == 0 && i == 0 && have && lane < nseg - 1) {
            // the end lane this segment assumes: the forward pass marked the lane(s) with the largest path metric in bit 30 of
            // the segment's last block (c8p_viterbi_lean.h); lane 0 if nothing is marked.  The last segment ends in state 0.
            const uint32_t* row = &tile[lane * TS];
            int h = 0;
            for (int d = 63; d >= 0; d--) h = (row[d] & C8P_SV_HINT) ? d : h;
            D = Dend = h;
          }
          if (act && i < nblk) {
            const int blk = btop - i, b0 = blk * C8P_SV_BLK;
            const int hi = min(tend, b0 + C8P_SV_BLK) - 1 - b0, lo = max(tbeg, b0) - b0;   // active steps k in [lo, hi]
            const int gtop = blk * 5 + hi / 6;                                             // group of the first step walked
            if (it > 0 && snaps[gtop] == (uint8_t)D) {
              act = false;                                 // met the previous path
              D = Dbeg;
            } else {
              const uint32_t A = (hi >= 31 ? 0xffffffffu : ((2u << hi) - 1u)) & ~((1u << lo) - 1u);
              const uint32_t* row = &tile[lane * TS];
#pragma unroll
              for (int g = 4; g >= 0; g--) {
                if (A & (63u << (6 * g))) snaps[blk * 5 + g] = (uint8_t)D;
                // TWO steps per LDS round trip.  Step k (phase q = k mod 6) replaces lane bit q by the decision found in row[D]; the
                // next step, k - 1, reads row[D'] where D' differs from D in bit q at most — so both candidates, row[D with bit q = 0]
                // and row[D with bit q = 1], are fetched together (one ds_read2_b32) and the second step selects.  The walk is a chain
                // of dependent LDS reads (~100 cycles each, one wave per workgroup, 7 workgroups per CU): 15 instead of 30 per block.
#pragma unroll
                for (int q = 5; q >= 1; q -= 2) {
                  const int k = 6 * g + q;                 // steps b0 + k (phase q) and b0 + k - 1 (phase q - 1)
                  const uint32_t i0 = (uint32_t)D & ~(1u << q);
                  const uint32_t w0 = row[i0], w1 = row[i0 + (1u << q)];
                  const uint32_t own1 = ((uint32_t)D >> q) & 1u, own2 = ((uint32_t)D >> (q - 1)) & 1u;
                  uint32_t d1 = ((own1 ? w1 : w0) >> (C8P_SV_BLK - 1 - k)) & 1u;       // decision bit (29 - k)
                  d1 = ((A >> k) & 1u) ? d1 : own1;                                     // a step outside [lo, hi] leaves D alone
                  uint32_t d2 = ((d1 ? w1 : w0) >> (C8P_SV_BLK - k)) & 1u;             // decision bit (29 - (k - 1)) of row[D']
                  d2 = ((A >> (k - 1)) & 1u) ? d2 : own2;
                  D = (int)(((uint32_t)D & ~(3u << (q - 1))) | (d1 << q) | (d2 << (q - 1)));
                }
              }
            }
          }
          am = __ballot(act && i + 1 < nblk);
        }
        if (redo) Dbeg = D;
        int nd = __shfl_down(Dbeg, 1, 64);
        if (lane >= nseg - 1) nd = 0;
        redo = have && (nd != Dend);
        Dend = nd;
        tb_passes = it + 1;
        if (!__any(redo)) break;
      }
      // ---- decoded bits: word c = times 64c .. 64c+63, gathered from the 6-bit group snapshots
      __syncthreads();
      const int ngrp = (Tn + 5) / 6;                       // (the tile is dead from here on: its memory becomes bw[])
      for (int c = lane; c < nch; c += 64) {
        unsigned long long word = 0;
        const int base = 64 * c, g0 = base / 6;
        for (int g = g0; g < g0 + 12 && g < ngrp; g++) {
          const unsigned long long v = snaps[g] & 63u;
          const int sh = 6 * g - base;
          if (sh >= 64) break;
          word |= sh >= 0 ? (v << sh) : (v >> -sh);
        }
        if (base + 64 > Tn) word &= (1ull << (Tn - base)) - 1ull;
        bw[c] = word;
      }
    } else {
    // ---- a packet the lean kernels left: the clamp-exact forward pass, here
    viterbi_fwd_exact(t, cold->llr + t.llr_off, surv + Lp.surv_off);
    __threadfence();
    // ---- traceback from state 0 (vstb_end decode_impl.cc:282-302)
    int st = 0;
    for (int c = nch - 1; c >= 0; c--) {
      const int ts = c * 64 + lane;
      ulonglong2 wv = ts < Tn ? SV[ts] : make_ulonglong2(0ull, 0ull);
      const int last = min(63, Tn - 1 - c * 64);
      unsigned long long mask = 0;
      for (int l = last; l >= 0; l--) {
        const unsigned long long w1 = readlane_u64(wv.x, l), w0 = readlane_u64(wv.y, l);
        const int prev = ((w0 >> st) & 1ull) ? 0 : (2 * (st & 31) + (int)((w1 >> st) & 1ull));
        if (st == ((prev >> 1) | 32)) mask |= 1ull << l;
        st = prev;
      }
      if (lane == 0) bw[c] = mask;
    }
    }
    __syncthreads();
    if (keep_bits) {
      uint8_t* so = sbits_o + Lp.bits_off;
      for (int i = lane; i < Tn; i += 64) so[i] = (uint8_t)((bw[i >> 6] >> (i & 63)) & 1ull);
    }
    // ---- descramble (decode_impl.cc:304-323)
    int state = 0;
    {
      const unsigned long long w = bw[0];
      for (int i = 0; i < 7; i++) if ((w >> i) & 1ull) state |= 1 << (6 - i);
    }
    const int ph = T->scrPhase[state];
    __syncthreads();
    // word c = bits 64c..64c+63: scrambler output (ph + i - 7) mod 127 for bit i, 64 at a time out of the packed
    // doubled sequence scrW; bits 0..6 and bits >= T are 0
    for (int c = lane; c < nch; c += 64) {
      unsigned long long fbw = 0;
      if (state) {
        const int o = (ph + 120 + 64 * c) % 127, wi = o >> 6, sh = o & 63;
        fbw = T->scrW[wi] >> sh;
        if (sh) fbw |= T->scrW[wi + 1] << (64 - sh);
      }
      unsigned long long ub = bw[c] ^ fbw;
      if (c == 0) ub &= ~0x7full;
      if (64 * c + 64 > Tn) ub &= (1ull << (Tn - 64 * c)) - 1ull;
      bw[c] = ub;
    }
    if (lane == 0) bw[nch] = 0ull;
    // the CRC-32 byte table moves into LDS (snaps[] is dead by now): the per-lane CRC is a chain of dependent lookups
    uint32_t* crcT = (uint32_t*)snaps;
    for (int i = lane; i < 256; i += 64) crcT[i] = T->crc32[i];
    __syncthreads();
    if (keep_bits) {
      uint8_t* uo = bits_o + Lp.bits_off;
      for (int i = lane; i < Tn; i += 64) uo[i] = (uint8_t)((bw[i >> 6] >> (i & 63)) & 1ull);
    }
    // ---- packetAssemble (decode_impl.cc:325-520)
    if (t.format == C8P_F_VHT) {
      int procd = 16;
      if (procd < Tn) {
        int bp = 16;
        int tmpLen = 0;
        while (true) {
          procd += 32;
          if (procd > Tn) break;
          const uint32_t d = (uint32_t)((bw[bp >> 6] >> (bp & 63)) & 0xffffull);   // bp % 16 == 0: never straddles a word
          const int eof = d & 1;
          tmpLen |= (int)((d >> 2) & 1u) << 12;
          tmpLen |= (int)((d >> 3) & 1u) << 13;
          tmpLen |= (int)((d >> 4) & 0xfffu);
          const int padded = ((tmpLen / 4 + ((tmpLen % 4) != 0)) * 4 * 8);
          procd += padded;
          if (procd > Tn) break;
          bp += 32;
          const uint32_t crc = wave_crc32(bw, bp, tmpLen, T, crcT);
          if (crc != 558161692u) n_fail++;
          else {
            const uint32_t before = used;
            emit_pdu(area, Lp.pdu_cap, used, bw, bp, tmpLen, t.format, t.mcs);
            if (used != before) n_pdu++;
            tmpLen += 4;       // decode_impl.cc:414 reuses the variable for the PDU length; the next delimiter is OR-ed into it (:346-351)
          }
          bp += padded;
          if (eof) break;
        }
      }
    } else if (!t.ampdu) {
      const uint32_t crc = wave_crc32(bw, 16, t.len, T, crcT);
      if (crc != 558161692u) n_fail++;
      else {
        const uint32_t before = used;
        emit_pdu(area, Lp.pdu_cap, used, bw, 16, t.len, t.format, t.mcs);
        if (used != before) n_pdu++;
      }
    }
  }
  if (lane == 0) {
    c8p_pkt_out o;
    o.n_pdu = n_pdu; o.n_crc_fail = n_fail; o.pdu_bytes = used; o.rsv = (uint32_t)tb_passes | ((uint32_t)n_refetch << 8);   // diagnostics
    o.pdu_off = Lp.pdu_off; o.bits_off = Lp.bits_off;
    outs[p] = o;
  }
}
