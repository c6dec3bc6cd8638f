// c8p_k_parse.h — SIG-field checks and parsers (wave-uniform code) shared by k_signal and k_classify
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_dev.h"

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
