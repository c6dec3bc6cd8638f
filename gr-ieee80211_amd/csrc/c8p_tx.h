// c8p_tx.h — transmit chain for one spatial stream at 20 MHz, long GI, on gfx950 (SURVEY.md §8f-4 "TX-lite"):
// what lib/encode_impl.cc + lib/modulation_impl.cc (cloud80211phy.cc:2594-3180) / tools/phy80211.py produce.
//
//   k_tx_bits   wave = packet.  SIG fields (L-SIG, HT-SIG, VHT-SIG-A/B: field bits in one 64-bit word, every lane its own
//               coded bits, CRC-8 on the word), then the data
//               field: service + PSDU (+ VHT EOF / zero padding) assembled 64 bits per lane-word, scrambled with the
//               packed doubled LFSR sequence, tail zeroed, and BCC + puncturing evaluated per OUTPUT bit (every coded
//               bit is a parity over a 7-bit window of the packed data bits) — nothing is sequential.
//   k_tx_sym    wave = 80 output samples, lane = subcarrier.  Field by field: training sequences, BPSK / QBPSK SIG
//               symbols, data symbols (interleaver map -> constellation point from the Gray recursion, pilots with the
//               127-periodic polarity and the HT/VHT rotation), 64-point inverse FFT across the wave in fp64, GI,
//               half-amplitude samples at every field junction (procConcat2Symbol), amplitude.
#pragma once
#include "c8p_dev.h"

struct TxPlan {            // c8p_tx_plan + where the packet's outputs go
  c8p_tx_plan p;
  unsigned long long iq_off, coded_off, psdu_off;
  unsigned int len;
  int format, mcs, seed, flags, nss;
  unsigned long long stride;   // chain 1 starts this many items after chain 0
  float mult;
  unsigned int unit_base;  // first 80-sample unit of this packet in k_tx_sym's grid
  int mu;                  // 0: single user; 1 / 2: user 0 / user 1 of a 2-user MU-MIMO packet (user 1's plan follows user 0's)
  int gid;                 // MU group id (VHT-SIG-A bits 4-9)
  unsigned long long q_off;   // MU: first item of the packet's steering matrix in the bfq arena ([64 bins, FFT order][q00 q01 q10 q11])
};
#define C8P_TX_SIGBITS 256  // per packet: L-SIG 48 | SIG-1 48 | SIG-2 48 | SIG-B 52 interleaved coded bits

__host__ __device__ __forceinline__ int tx_rate(int format, int mcs, c8p_tx_plan& r)
{
  const int nbL[8] = {1, 1, 2, 2, 4, 4, 6, 6}, crL[8] = {0, 2, 0, 2, 0, 2, 1, 2};
  const int nbN[10] = {1, 2, 2, 4, 4, 6, 6, 6, 8, 8}, crN[10] = {0, 0, 2, 0, 2, 1, 2, 3, 2, 3};
  if (format == C8P_F_L) { if (mcs < 0 || mcs > 7) return 0; r.nb = nbL[mcs]; r.cr = crL[mcs]; r.nsd = 48; }
  else if (format == C8P_F_HT || format == C8P_F_VHT) { if (mcs < 0 || mcs > (format == C8P_F_HT ? 7 : 8)) return 0; r.nb = nbN[mcs]; r.cr = crN[mcs]; r.nsd = 52; }
  else return 0;
  r.ncbps = r.nsd * r.nb;
  r.ndbps = r.cr == 0 ? r.ncbps / 2 : r.cr == 1 ? r.ncbps * 2 / 3 : r.cr == 2 ? r.ncbps * 3 / 4 : r.ncbps * 5 / 6;
  return 1;
}
// tools/phy80211header.py procPktLenNonAggre :427-459 / procPktLenAggre :461-506 (cloud80211phy.cc:2487-2524)
__host__ __device__ __forceinline__ int tx_plan_of(int format, int mcs, int nss, int len, c8p_tx_plan& p)
{
  p = c8p_tx_plan{};
  if (!tx_rate(format, mcs, p) || len < 0 || len > 4095 || nss < 1 || nss > 2 || (format == C8P_F_L && nss != 1)) return 0;
  if (len == 0 && format != C8P_F_VHT) return 0;
  p.ncbps *= nss; p.ndbps *= nss;
  const int nltf = nss;
  if (format == C8P_F_VHT) {
    if (len > 0) {
      p.nsym = (8 * len + 16 + 6 + p.ndbps - 1) / p.ndbps;
      p.psdu_len = (p.nsym * p.ndbps - 16 - 6) / 8;
      p.npad_eof = (p.psdu_len - len) / 4;
      p.npad_octet = p.psdu_len - len - p.npad_eof * 4;
    }
    p.legacy_len = ((20 + 8 + 4 + nltf * 4 + 4 + p.nsym * 4 - 20) / 4) * 3 - 3;
    p.n_pre = 2 + 1 + nltf + 1;
  } else {
    p.nsym = (8 * len + 16 + 6 + p.ndbps - 1) / p.ndbps;
    p.psdu_len = len;
    p.legacy_len = format == C8P_F_L ? len : ((20 + 8 + 4 + nltf * 4 + p.nsym * 4 - 20) / 4) * 3 - 3;
    p.n_pre = format == C8P_F_L ? 0 : 2 + 1 + nltf;
  }
  p.n_samp = 320 + 80 * (1 + p.n_pre + p.nsym);
  return 1;
}

// 2-user MU-MIMO, one stream per user: formatToModMu (cloud80211phy.cc:2558-2585) / procPktLenAggreMu (phy80211header.py:509-525)
__host__ __device__ __forceinline__ int tx_plan_mu(int mcs0, int len0, int mcs1, int len1, c8p_tx_plan& p0, c8p_tx_plan& p1)
{
  if (len0 <= 0 || len1 <= 0 || !tx_plan_of(C8P_F_VHT, mcs0, 1, len0, p0) || !tx_plan_of(C8P_F_VHT, mcs1, 1, len1, p1)) return 0;
  const int nsym = p0.nsym > p1.nsym ? p0.nsym : p1.nsym;
  // both users are padded to the longer one's symbols: the padded data field must still fit the decoder's trellis limit
  // (decode_impl.cc:93, 32782 steps) — encode2's d_bitsCoded[65728] (encode2_impl.h:74) holds two bits per data bit
  if (nsym * p0.ndbps > C8P_DECODE_T_MAX || nsym * p1.ndbps > C8P_DECODE_T_MAX) return 0;
  c8p_tx_plan* P[2] = {&p0, &p1};
  const int len[2] = {len0, len1};
  for (int u = 0; u < 2; u++) {
    c8p_tx_plan& p = *P[u];
    p.nsym = nsym;
    p.psdu_len = (nsym * p.ndbps - 16 - 6) / 8;
    p.npad_eof = (p.psdu_len - len[u]) / 4;
    p.npad_octet = p.psdu_len - len[u] - 4 * p.npad_eof;
    p.legacy_len = ((20 + 8 + 4 + 2 * 4 + 4 + nsym * 4 - 20) / 4) * 3 - 3;
    p.n_pre = 2 + 1 + 2 + 1;
    p.n_samp = 320 + 80 * (1 + p.n_pre + nsym);
  }
  return 1;
}

// coded bit q (A at even, B at odd positions) of the rate-1/2 encoder run over the field bits in `w` (bit i = i-th input)
__device__ __forceinline__ uint8_t tx_bcc_bit(unsigned long long w, int q)
{
  const int j = q >> 1, lo = j - 6;
  const unsigned win = (unsigned)(lo >= 0 ? (w >> lo) : (w << (-lo))) & 0x7fu;     // bit k = input bit j - 6 + k
  return (uint8_t)(__popc(win & ((q & 1) ? 0171u : 0133u)) & 1);
}
// genBitBitCrc8 over the first n bits of w; returns the 8 CRC bits, bit i = i-th bit in transmission order
__device__ __forceinline__ unsigned long long tx_crc8_word(unsigned long long w, int n)
{
  unsigned c = 0xff;
  for (int i = 0; i < n; i++) {
    const unsigned fb = ((c >> 7) & 1u) ^ (unsigned)((w >> i) & 1ull);
    c = (c << 1) & 0xffu;
    if (fb) c ^= 0x07u;
  }
  c ^= 0xffu;
  unsigned long long r = 0;
  for (int i = 0; i < 8; i++) r |= (unsigned long long)((c >> (7 - i)) & 1u) << i;
  return r;
}
// byte b of the padded PSDU: the caller's bytes, then (VHT) EOF delimiters {01 00 79 4e}, then zero octets
__device__ __forceinline__ unsigned tx_psdu_byte(const uint8_t* __restrict__ psdu, int b, int len, int npad_eof)
{
  if (b < 0) return 0u;
  if (b < len) return psdu[b];
  const int e = b - len;
  if (e < 4 * npad_eof) { const int k = e & 3; return k == 0 ? 0x01u : k == 1 ? 0x00u : k == 2 ? 0x79u : 0x4eu; }
  return 0u;
}

__global__ __launch_bounds__(64) void k_tx_bits(const TxPlan* __restrict__ plans, const uint8_t* __restrict__ psdu_arena,
                                                const DevTables* __restrict__ T, uint8_t* __restrict__ coded,
                                                uint8_t* __restrict__ sigbits, unsigned int* __restrict__ unit2pkt, int n)
{
  __shared__ unsigned long long dw[(C8P_DECODE_T_MAX + 63) / 64 + 2];   // scrambled data bits, 64 per word
  const int p = blockIdx.x;
  if (p >= n) return;
  const int lane = lane_id();
  const TxPlan P = plans[p];
  const c8p_tx_plan& m = P.p;
  uint8_t* sb = sigbits + (size_t)(P.mu == 2 ? p - 1 : p) * C8P_TX_SIGBITS;      // MU: both users share user 0's row
  if (P.mu != 2) for (unsigned i = lane; i < (unsigned)(m.n_samp / 80 * P.nss); i += 64) unit2pkt[P.unit_base + i] = (unsigned)p;   // k_tx_sym's grid
  // ---- SIG fields.  The field bits live in a 64-bit word (uniform); every lane produces coded bits on its own: the
  // rate-1/2 encoder output 2j / 2j+1 is the parity of the 7-bit window ending at input bit j under 133 / 171.
  unsigned long long crcb = 0;                        // VHT-SIG-B CRC (service bits 8..15), bit i = i-th CRC bit
  {
    // L-SIG  tools/phy80211.py:237-282
    const int rate = P.format == C8P_F_L ? P.mcs : 0;
    const unsigned rb = rate == 0 ? 0xBu : rate == 1 ? 0xFu : rate == 2 ? 0xAu : rate == 3 ? 0xEu : rate == 4 ? 0x9u : rate == 5 ? 0xDu : rate == 6 ? 0x8u : 0xCu;
    unsigned long long w = (unsigned long long)rb | ((unsigned long long)(m.legacy_len & 0xfff) << 5);     // C_LEGACY_RATE_BIT, LENGTH
    w |= (unsigned long long)(__popcll(w & 0x1ffffull) & 1) << 17;                                         // even parity
    if (lane < 48 && P.mu != 2) sb[3 * (lane % 16) + lane / 16] = tx_bcc_bit(w, lane);                     // procInterleaveSigL
    if (P.format != C8P_F_L && P.mu != 2) {
      if (P.format == C8P_F_HT) {                                                   // :284-350
        const int mf = P.mcs + 8 * (P.nss - 1);
        w = (unsigned long long)(mf & 0x7f) | ((unsigned long long)(m.psdu_len & 0xffff) << 8) | (7ull << 24);
      } else if (P.mu) {                                                            // :366-416 MU: group id, nSTS 1 + 1, reserved ones
        w = (1ull << 2) | ((unsigned long long)(P.gid & 63) << 4) | (1ull << 10) | (1ull << 13) | (1ull << 23) | (0x1full << 29);
      } else {                                                                      // :352-455, SU, group id 0, partial AID 0
        w = (1ull << 2) | ((unsigned long long)((P.nss - 1) & 7) << 10) | (1ull << 23) | ((unsigned long long)(P.mcs & 15) << 28) | (1ull << 33);
      }
      w |= tx_crc8_word(w, 34) << 34;
      for (int q = lane; q < 96; q += 64) { const int h = q / 48, k = q - 48 * h; sb[48 + 48 * h + 3 * (k % 16) + k / 16] = tx_bcc_bit(w, q); }
    }
    if (P.format == C8P_F_VHT) {                                                    // :512-569
      if (P.mu) {                                                                   // :571-598 MU: length / 4 (16 bits), MCS (4 bits)
        w = (unsigned long long)(((int)P.len / 4) & 0xffff) | ((unsigned long long)(P.mcs & 15) << 16);
        crcb = tx_crc8_word(w, 20);
      } else if (P.len > 0) {
        w = (unsigned long long)((((int)P.len + 3) / 4) & 0x1ffff) | (7ull << 17);
        crcb = tx_crc8_word(w, 20);
      } else w = 0x422E0ull;                                                        // NDP: C_NDP_SIG_B_20
      __shared__ uint8_t s_c52[64];
      if (lane < 52) s_c52[lane] = tx_bcc_bit(w, lane);
      __syncthreads();
      if (lane < 52) sb[144 + (P.mu == 2 ? 52 : 0) + lane] = s_c52[T->deintSigB[lane]];
    }
  }
  __syncthreads();
  // ---- data bits, 64 per word: bit i = service (16) | PSDU bits LSB first | zeros     :639-682
  const int nbits = m.nsym * m.ndbps;
  const int nw = (nbits + 63) >> 6;
  const uint8_t* psdu = psdu_arena + P.psdu_off;
  const int ph = T->scrPhase[P.seed & 127];
  for (int c = lane; c < nw; c += 64) {
    unsigned long long w = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) w |= (unsigned long long)tx_psdu_byte(psdu, 8 * c - 2 + k, (int)P.len, m.npad_eof) << (8 * k);
    if (c == 0 && P.format == C8P_F_VHT) w |= crcb << 8;                            // service bits 8..15 = SIG-B CRC
    // scramble: output bit of data bit i = seq[(ph + i) mod 127]                       procScramble :800-806
    const int o = (ph + 64 * c) % 127, wi = o >> 6, sh = o & 63;
    unsigned long long sq = T->scrW[wi] >> sh;
    if (sh) sq |= T->scrW[wi + 1] << (64 - sh);
    w ^= sq;
    if (64 * c + 64 > nbits) w &= (1ull << (nbits - 64 * c)) - 1ull;
    dw[c] = w;
  }
  if (lane == 0) dw[nw] = 0ull;
  __syncthreads();
  if (lane == 0) {        // tail bits are zero AFTER scrambling (:692-702)
    const int t0 = P.format == C8P_F_VHT ? nbits - 6 : 16 + 8 * m.psdu_len;
    for (int i = t0; i < t0 + 6; i++) dw[i >> 6] &= ~(1ull << (i & 63));
  }
  __syncthreads();
  // ---- BCC + puncturing, one coded bit per lane and pass                               procBcc :763-798
  const int ncoded = m.nsym * m.ncbps;
  uint8_t* out = coded + P.coded_off;
  for (int q = lane; q < ncoded; q += 64) {
    int j, second;                                     // input bit index, 0: output A (g0 = 133), 1: output B (g1 = 171)
    if (m.cr == C8P_CR_12) { j = q >> 1; second = q & 1; }
    else if (m.cr == C8P_CR_23) { const int g = q / 3, r = q - 3 * g; j = 2 * g + (r == 2); second = r == 1; }
    else if (m.cr == C8P_CR_34) { const int g = q >> 2, r = q & 3; j = 3 * g + (r >= 2 ? r - 1 : 0); second = (r == 1 || r == 3); }
    else { const int g = q / 6, r = q - 6 * g; j = 5 * g + (r <= 1 ? 0 : r - 1); second = (r == 1 || r == 3 || r == 5); }
    // window: bit k of `win` = data bit j - 6 + k (zero before the start)
    const int lo = j - 6;
    unsigned win;
    if (lo >= 0) {
      const int wi2 = lo >> 6, sh2 = lo & 63;
      unsigned long long v = dw[wi2] >> sh2;
      if (sh2 > 57) v |= dw[wi2 + 1] << (64 - sh2);
      win = (unsigned)v & 0x7fu;
    } else win = ((unsigned)dw[0] << (-lo)) & 0x7fu;
    out[q] = (uint8_t)(__popc(win & (second ? 0171u : 0133u)) & 1);
  }
}

// Gray recursion of C_QAM_MODU_TAB: h bits per axis, least significant chip bit first, level from the LLR recursion run backwards
__device__ __forceinline__ double tx_axis(const uint8_t* b, int h)
{
  const double A = h == 1 ? 0.0 : (double)(1 << (h - 1));
  double v = b[h - 1] ? 1.0 : -1.0;
  for (int k = h - 1; k >= 1; k--) { const double Tk = A / (double)(1 << (k - 1)); v = (b[k - 1] ? 1.0 : -1.0) * (Tk - v); }
  return v;
}

enum { TXF_STF, TXF_LLTF, TXF_NLLTF, TXF_SIG, TXF_SIGB, TXF_DATA };
struct TxField { int kind, sig_off, sig_q, dsym, ltf_i; };

// frequency-domain value of this lane's bin for one field of one spatial stream (before the cyclic shift); returns ntf
// U = the plan that owns the stream's bits (the packet's, or the user's for MU-MIMO); ss / nss = stream index / streams
// that share U's coded bits (MU: every user is a single stream, nss = 1, ss only selects the P-matrix row)
__device__ __forceinline__ double tx_field_x(const TxPlan& U, const TxField& f, int ss, int nss, const uint8_t* __restrict__ sb, int sigb_off,
                                             const uint8_t* __restrict__ coded, const DevTables* __restrict__ T, int lane, double& xr, double& xi)
{
  const c8p_tx_plan& m = U.p;
  const int k = lane < 32 ? lane : lane - 64;         // subcarrier of this lane's bin
  const int ak = k < 0 ? -k : k;
  const bool pilot_bin = ak == 7 || ak == 21;
  const int pil_i = k == -21 ? 0 : k == -7 ? 1 : k == 7 ? 2 : 3;
  const bool vht = U.format == C8P_F_VHT;
  double ntf = 52.0;
  xr = 0.0; xi = 0.0;
  if (f.kind == TXF_STF) {                            // C_STF_L / C_STF_HT / C_STF_VHT: +-(1+j)/sqrt(2) on every 4th subcarrier
    ntf = 12.0;
    if (ak >= 4 && ak <= 24 && (ak & 3) == 0) {
      const int mi = k < 0 ? (k + 24) / 4 : 6 + (k - 4) / 4;
      const int sgn = (0xF25u >> mi) & 1u ? 1 : -1;   // {1,-1,1,-1,-1,1, -1,-1,1,1,1,1}
      xr = xi = sgn * 0.70710678118654757;
    }
  } else if (f.kind == TXF_LLTF) { ntf = 52.0; xr = (double)T->ltfL[lane]; }
  else if (f.kind == TXF_NLLTF) {                     // C_P_LTF_VHT_4 rows {1,-1},{1,1}; VHT pilots follow C_R_LTF_VHT_4 = {1,-1}
    ntf = 56.0;
    const double pm = (ss == 0 && f.ltf_i == 1) ? -1.0 : 1.0, rm = f.ltf_i == 1 ? -1.0 : 1.0;
    xr = (double)T->ltfNL[lane] * ((vht && pilot_bin) ? rm : pm);
  }
  else if (f.kind == TXF_SIG) {
    ntf = 52.0;
    if (pilot_bin) xr = pil_i == 3 ? -1.0 : 1.0;
    else if (ak >= 1 && ak <= 26) { const double v = sb[f.sig_off + qidx_l(lane)] ? 1.0 : -1.0; if (f.sig_q) xi = v; else xr = v; }
  } else if (f.kind == TXF_SIGB) {
    ntf = 56.0;
    if (pilot_bin) xr = pil_i == 3 ? -1.0 : 1.0;
    else if (ak >= 1 && ak <= 28) xr = sb[sigb_off + qidx_nl(lane)] ? 1.0 : -1.0;
  } else {
    const bool leg = U.format == C8P_F_L;
    ntf = leg ? 52.0 : 56.0;
    const int lim = leg ? 26 : 28;
    const int p0 = leg ? 1 : (vht ? 4 : 3);
    const float pol = T->pilotP[(p0 + f.dsym) % 127];
    if (pilot_bin) {                                  // base pattern rotated left by one per symbol for HT / VHT (:811-818)
      const int idx = leg ? pil_i : ((pil_i + f.dsym) & 3);
      float base = idx == 3 ? -1.0f : 1.0f;                                       // {1,1,1,-1}
      if (U.format == C8P_F_HT && nss == 2) base = ss == 0 ? (idx >= 2 ? -1.0f : 1.0f)      // {1,1,-1,-1}
                                                           : ((idx == 1 || idx == 2) ? -1.0f : 1.0f);   // {1,-1,-1,1}
      xr = (double)(base * pol);
    } else if (ak >= 1 && ak <= lim) {
      const int j = leg ? qidx_l(lane) : qidx_nl(lane);
      const int mi = nbpsc_index(m.nb);
      const int16_t* map = leg ? T->deintL[mi] : (nss == 2 ? T->deint2ss[ss][mi] : T->deintNL[mi]);   // (2 chains: de-interleave fused with the stream parser)
      const uint8_t* cb = coded + U.coded_off + (size_t)f.dsym * m.ncbps;
      uint8_t bits[8];
      for (int b = 0; b < m.nb; b++) bits[b] = cb[map[j * m.nb + b]];
      if (m.nb == 1) xr = bits[0] ? 1.0 : -1.0;
      else {
        const int h = m.nb >> 1;
        const double sc = m.nb == 2 ? 1.4142135623730951 : m.nb == 4 ? 3.1622776601683795 : m.nb == 6 ? 6.4807406984078604 : 13.038404810405298;
        xr = tx_axis(bits, h) / sc; xi = tx_axis(bits + h, h) / sc;
      }
    }
  }
  return ntf;
}
// procLegacyCSD / procCSD: X[k] *= exp(-j 2 pi shift k / 64), shift in samples
__device__ __forceinline__ void tx_csd(int shift, int lane, const DevTables* __restrict__ T, double& xr, double& xi)
{
  if (!shift) return;
  const int k = lane < 32 ? lane : lane - 64;
  const int mm = (shift * k) & 63;                     // e^{-j 2 pi mm / 64} from the twiddle table
  const double2 w = T->tw[mm & 31];
  const double wr = mm < 32 ? w.x : -w.x, wi = mm < 32 ? w.y : -w.y;
  const double ar = xr * wr - xi * wi, ai = xr * wi + xi * wr;
  xr = ar; xi = ai;
}

__global__ __launch_bounds__(256) void k_tx_sym(const TxPlan* __restrict__ plans, const unsigned int* __restrict__ unit2pkt,
                                                const uint8_t* __restrict__ coded, const uint8_t* __restrict__ sigbits,
                                                const cf* __restrict__ bfq, const DevTables* __restrict__ T, cf* __restrict__ iq,
                                                unsigned int total_units)
{
  __shared__ double2 st[4][64];
  const unsigned int u = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));   // uniform: the plan comes through scalar loads
  if (u >= total_units) return;
  const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int p = (int)unit2pkt[u];
  const TxPlan P = plans[p];
  const c8p_tx_plan& m = P.p;
  const int upp = m.n_samp / 80;                      // units per chain
  const int ss = (int)(u - P.unit_base) / upp;        // transmit chain
  const int unit = (int)(u - P.unit_base) - ss * upp; // 80-sample unit inside the packet: 0,1 L-STF  2,3 L-LTF  4 L-SIG  5.. pre  then data
  const int nss = P.nss;
  int shift = 0;                                      // cyclic shift of this field on this chain, in samples (C_CYCLIC_SHIFT_L / _NL)
  const uint8_t* sb = sigbits + (size_t)p * C8P_TX_SIGBITS;
  TxField f = {TXF_STF, 0, 0, 0, 0};
  bool steered = false;                               // MU-MIMO: fields from the VHT-STF on go through the steering matrix
  const int shL = ss ? -4 : 0, shN = ss ? -8 : 0;     // -200 ns / -400 ns on the second chain
  if (unit < 2) { f.kind = TXF_STF; shift = shL; }
  else if (unit < 4) { f.kind = TXF_LLTF; shift = shL; }
  else if (unit == 4) { f.kind = TXF_SIG; f.sig_off = 0; shift = shL; }
  else {
    const int q = unit - 5;                           // index among the pre symbols
    if (q < m.n_pre) {
      if (q < 2) { f.kind = TXF_SIG; f.sig_off = 48 + 48 * q; f.sig_q = (P.format == C8P_F_HT || q == 1) ? 1 : 0; shift = shL; }   // HT-SIG: QBPSK x2, VHT-SIG-A: BPSK, QBPSK
      else if (q == 2) { f.kind = TXF_STF; shift = shN; steered = true; }
      else if (q < 3 + nss) { f.kind = TXF_NLLTF; f.ltf_i = q - 3; shift = shN; steered = true; }
      else { f.kind = TXF_SIGB; shift = shN; steered = true; }
    } else { f.kind = TXF_DATA; f.dsym = q - m.n_pre; shift = P.format == C8P_F_L ? shL : shN; steered = true; }
  }
  double xr, xi, ntf;
  if (P.mu && steered) {
    // chain ss = Q[ss][0] X_0 + Q[ss][1] X_1 per subcarrier (procSpatialMapping / procNss2SymBfQ); stream s carries user s,
    // the second stream with the VHT cyclic shift
    const TxPlan U1 = plans[p + 1];
    double x0r, x0i, x1r, x1i;
    ntf = tx_field_x(P, f, 0, 1, sb, 144, coded, T, lane, x0r, x0i);
    tx_field_x(U1, f, 1, 1, sb, 196, coded, T, lane, x1r, x1i);
    tx_csd(-8, lane, T, x1r, x1i);
    const cf qa = bfq[P.q_off + 4u * (unsigned)lane + 2u * (unsigned)ss], qb = bfq[P.q_off + 4u * (unsigned)lane + 2u * (unsigned)ss + 1u];
    xr = (double)qa.x * x0r - (double)qa.y * x0i + (double)qb.x * x1r - (double)qb.y * x1i;
    xi = (double)qa.x * x0i + (double)qa.y * x0r + (double)qb.x * x1i + (double)qb.y * x1r;
  } else {
    ntf = tx_field_x(P, f, ss, nss, sb, 144, coded, T, lane, xr, xi);
    tx_csd(shift, lane, T, xr, xi);
  }
  const double2 t = wave_ifft64_d(xr, xi, T);
  const double sc = 1.0 / sqrt(ntf * (double)nss);
  st[wv][lane] = make_double2(t.x * sc, t.y * sc);
  __syncthreads();
  // ---- 80 output samples: GI (or the 160-sample training field layout), junction halving, amplitude
  const int n_samp = m.n_samp;
  for (int mm = lane; mm < 80; mm += 64) {
    int src;
    if (unit < 4) { const int f2 = (unit & 1) * 80 + mm; src = f2 < 32 ? 32 + f2 : (f2 - 32) & 63; }   // t[32:] + t + t
    else src = mm < 16 ? 48 + mm : mm - 16;
    const int g = unit * 80 + mm;
    double2 v = st[wv][src];
    const bool junction = !(P.flags & C8P_TX_NO_WINDOW) && ((g == 159 || g == 160) || (g >= 319 && ((g + 1) % 80 == 0) && g + 1 < n_samp) || (g >= 320 && (g % 80 == 0)));
    const double a = (double)P.mult * (junction ? 0.5 : 1.0);
    iq[P.iq_off + (unsigned long long)ss * P.stride + (unsigned long long)g] = mk((float)(v.x * a), (float)(v.y * a));
  }
}
