/*
 * c8p_oracle_tx.c — CPU restatement of the reference's transmit chain for one spatial stream at 20 MHz, long GI
 * ("TX-lite"): what tools/phy80211.py genFromMpdu / genFromAmpdu + genFinalSig produce, which is also
 * what lib/encode_impl.cc + lib/modulation_impl.cc + cloud80211phy.cc:2594-3180 produce in the flowgraph.
 *
 * TEST INFRASTRUCTURE ONLY (see c8p_oracle.h).  Pinned by the committed fixtures: tests/golden/packets_*.npz hold the
 * toolbox's own IQ ("tx") and punctured coded bits ("coded") for legacy MCS 0-7, HT MCS 0-7 and VHT MCS 0-8.
 *
 * Restated from tools/phy80211.py (line numbers of that file):
 *   __genLegacyTraining :223-235, __genLegacySignal :237-282, __genHtSignal :284-350, __genVhtSignalA :352-455,
 *   __genNonLegacyTraining :459-510, __genVhtSignalB :512-569, __genDataBits :639-682, __genCodedBits :684-710,
 *   __genInterleaveDataBits :736-746, __genConstellation :748-761, __genOfdmSignal :788-819, genFinalSig :1018-1035
 * and tools/phy80211header.py: procPktLenNonAggre :427-459, procPktLenAggre :461-506, procBcc :763-798,
 * procScramble :800-806, procInterleaveSigL :808-819, procConcat2Symbol :894-901, procPilotInsert :914-925,
 * procDcInsert :927-937, procNonDataSC :940-948, procFftMod :958-964, procToneScaling :966-967, procGi :969-977.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "c8p_oracle.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { double re, im; } cd;

static const signed char LTF_L53[53] = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                                        1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};
static const signed char STF_SIGN[12] = {1, -1, 1, -1, -1, 1, -1, -1, 1, 1, 1, 1};   /* subcarriers -24,-20,...,-4, 4,...,24 */

typedef struct { int nb, cr, ndbps, ncbps, nsd; } rate_t;

static int rate_of(int format, int mcs, rate_t* r)
{
  static const int nbL[8] = {1, 1, 2, 2, 4, 4, 6, 6}, crL[8] = {0, 2, 0, 2, 0, 2, 1, 2};
  static const int nbN[10] = {1, 2, 2, 4, 4, 6, 6, 6, 8, 8}, crN[10] = {0, 0, 2, 0, 2, 1, 2, 3, 2, 3};
  if (format == 0) { if (mcs < 0 || mcs > 7) return 0; r->nb = nbL[mcs]; r->cr = crL[mcs]; r->nsd = 48; }
  else { if (mcs < 0 || mcs > (format == 1 ? 7 : 8)) return 0; r->nb = nbN[mcs]; r->cr = crN[mcs]; r->nsd = 52; }
  r->ncbps = r->nsd * r->nb;
  r->ndbps = r->cr == 0 ? r->ncbps / 2 : r->cr == 1 ? r->ncbps * 2 / 3 : r->cr == 2 ? r->ncbps * 3 / 4 : r->ncbps * 5 / 6;
  return 1;
}

/* nss = 1 or 2 spatial streams (HT: mcs 0-7 with nss 1, mcs 8-15 with nss 2 — pass mcs % 8 and nss).  len = 0 with VHT: NDP. */
int c8po_tx_plan2(int format, int mcs, int nss, int len, c8p_tx_plan* p)
{
  rate_t r;
  memset(p, 0, sizeof(*p));
  if (!rate_of(format, mcs, &r) || len < 0 || len > 4095 || nss < 1 || nss > 2 || (format == 0 && nss != 1)) return 0;
  if (len == 0 && format != 2) return 0;
  r.ncbps *= nss; r.ndbps *= nss;
  p->nb = r.nb; p->cr = r.cr; p->ndbps = r.ndbps; p->ncbps = r.ncbps; p->nsd = r.nsd;
  const int nltf = nss;
  if (format == 2) {                                 /* procPktLenAggre :480-504 (len = A-MPDU bytes) */
    if (len > 0) {
      p->nsym = (8 * len + 16 + 6 + r.ndbps - 1) / r.ndbps;
      p->psdu_len = (p->nsym * r.ndbps - 16 - 6) / 8;
      p->npad_eof = (p->psdu_len - len) / 4;
      p->npad_octet = p->psdu_len - len - p->npad_eof * 4;
    }
    p->legacy_len = ((20 + 8 + 4 + nltf * 4 + 4 + p->nsym * 4 - 20) / 4) * 3 - 3;
    p->n_pre = 2 + 1 + nltf + 1;                     /* SIG-A x2, VHT-STF, VHT-LTFs, SIG-B */
  } else {                                           /* procPktLenNonAggre :431-456 */
    p->nsym = (8 * len + 16 + 6 + r.ndbps - 1) / r.ndbps;
    p->psdu_len = len;
    p->legacy_len = format == 0 ? len : ((20 + 8 + 4 + nltf * 4 + p->nsym * 4 - 20) / 4) * 3 - 3;
    p->n_pre = format == 0 ? 0 : 2 + 1 + nltf;       /* HT-SIG x2, HT-STF, HT-LTFs */
  }
  p->n_samp = 320 + 80 * (1 + p->n_pre + p->nsym);
  return 1;
}
int c8po_tx_plan(int format, int mcs, int len, c8p_tx_plan* p) { return len > 0 ? c8po_tx_plan2(format, mcs, 1, len, p) : 0; }

static void bcc(const uint8_t* in, int n, uint8_t* out2n)      /* procBcc :766-771, rate 1/2: A (0155) then B (0117) */
{
  unsigned st = 0;
  for (int j = 0; j < n; j++) {
    st = ((st << 1) & 0x7e) | in[j];
    out2n[2 * j] = (uint8_t)(__builtin_popcount(st & 0155) & 1);
    out2n[2 * j + 1] = (uint8_t)(__builtin_popcount(st & 0117) & 1);
  }
}
static int puncture(const uint8_t* c, int n2, int cr, uint8_t* out)   /* procBcc :772-795 */
{
  int k = 0;
  for (int i = 0; i < n2; i++) {
    int keep = 1;
    if (cr == 1) keep = (i % 4) != 3;
    else if (cr == 2) keep = !((i % 6) == 3 || (i % 6) == 4);
    else if (cr == 3) { const int m = i % 10; keep = !(m == 3 || m == 4 || m == 7 || m == 8); }
    if (keep) out[k++] = c[i];
  }
  return k;
}
static void crc8_bits(const uint8_t* b, int n, uint8_t* out8)  /* genBitBitCrc8: x^8+x^2+x+1, init ones, complemented, MSB first */
{
  unsigned c = 0xff;
  for (int i = 0; i < n; i++) {
    const unsigned fb = ((c >> 7) & 1u) ^ b[i];
    c = (c << 1) & 0xffu;
    if (fb) c ^= 0x07u;
  }
  c ^= 0xffu;
  for (int i = 0; i < 8; i++) out8[i] = (uint8_t)((c >> (7 - i)) & 1u);
}
static void interleave_sig48(const uint8_t* in, uint8_t* out)  /* procInterleaveSigL :808-816 */
{
  for (int k = 0; k < 48; k++) out[3 * (k % 16) + k / 16] = in[k];
}

/* one OFDM symbol: X[64] in "shifted" order (index i = subcarrier i - 32), scale 1/sqrt(ntf), 64-point IDFT with 1/64,
 * written as cp + 64 samples at dst */
/* procLegacyCSD / procCSD (phy80211header.py:950-956): X[k] *= exp(-j 2 pi shift k / 64), shift = cyclic shift in samples
 * (shift_ns * 20 MHz * 0.001: -200 ns = -4, -400 ns = -8) */
void c8po_ofdm_csd(const cd* X0, double ntf, int cp, double shift, cd* dst)
{
  cd t[64], X[64];
  for (int i = 0; i < 64; i++) {
    const double ph = -2.0 * M_PI * shift * (double)(i - 32) / 64.0;
    X[i].re = X0[i].re * cos(ph) - X0[i].im * sin(ph);
    X[i].im = X0[i].re * sin(ph) + X0[i].im * cos(ph);
  }
  for (int n = 0; n < 64; n++) {
    double sr = 0, si = 0;
    for (int i = 0; i < 64; i++) {
      const int k = i - 32;                            /* subcarrier */
      const double a = 2.0 * M_PI * (double)k * (double)n / 64.0;
      sr += X[i].re * cos(a) - X[i].im * sin(a);
      si += X[i].re * sin(a) + X[i].im * cos(a);
    }
    t[n].re = sr / 64.0 / sqrt(ntf); t[n].im = si / 64.0 / sqrt(ntf);
  }
  for (int n = 0; n < cp; n++) dst[n] = t[64 - cp + n];
  for (int n = 0; n < 64; n++) dst[cp + n] = t[n];
}
/* data / SIG symbol from nsd constellation points (ascending subcarrier, pilots and DC not included) */
static void place(const cd* q, int nsd, const double pil[4], cd* X)
{
  memset(X, 0, sizeof(cd) * 64);
  const int lim = nsd == 48 ? 26 : 28;
  int j = 0;
  for (int k = -lim; k <= lim; k++) {
    if (k == 0) continue;
    if (k == -21 || k == -7 || k == 7 || k == 21) { const int pi = k == -21 ? 0 : k == -7 ? 1 : k == 7 ? 2 : 3; X[k + 32].re = pil[pi]; continue; }
    X[k + 32] = q[j++];
  }
}
static cd qam_point(const uint8_t* bits, int nb)      /* C_QAM_MODU_TAB: chip = sum b_j 2^j, Gray levels, I from the low half */
{
  cd r = {0, 0};
  if (nb == 1) { r.re = bits[0] ? 1.0 : -1.0; return r; }
  const int h = nb / 2;
  const double A = h == 1 ? 0 : (double)(1 << (h - 1));
  double ax[2];
  for (int a = 0; a < 2; a++) {
    const uint8_t* b = bits + a * h;
    double v = b[h - 1] ? 1.0 : -1.0;
    for (int k = h - 1; k >= 1; k--) { const double T = A / (double)(1 << (k - 1)); v = (b[k - 1] ? 1.0 : -1.0) * (T - v); }
    ax[a] = v;
  }
  const double sc = nb == 2 ? sqrt(2.0) : nb == 4 ? sqrt(10.0) : nb == 6 ? sqrt(42.0) : sqrt(170.0);
  r.re = ax[0] / sc; r.im = ax[1] / sc;
  return r;
}

static int PS127[127];
static void init_ps(void)
{
  static int done = 0;
  if (done) return;
  int st = 0x7f;
  for (int n = 0; n < 127; n++) { const int fb = ((st >> 6) & 1) ^ ((st >> 3) & 1); PS127[n] = fb ? -1 : 1; st = ((st << 1) & 0x7e) | fb; }
  done = 1;
}

/* Test hook: bits XOR-ed into HT-SIG / VHT-SIG-A bits 0..33 before their CRC-8 is computed, so that tests can transmit
 * packets whose (valid, CRC-correct) signal field announces things the transmitter does not do — short GI, aggregation,
 * LDPC, STBC, 40 MHz, other stream counts, MU group ids — and compare how receivers take them. */
static unsigned long long g_sig_xor = 0;
void c8po_tx_test_sig_xor(unsigned long long mask) { g_sig_xor = mask & 0x3ffffffffull; }

/* one packet on nss transmit chains (no spatial mapping matrix: stream i = chain i, as genFromMpdu / genFromAmpdu).
 * out0 / out1: cap samples each (out1 may be NULL when nss == 1).  HT: mcs = 0..7 is the per-stream rate, the HT-SIG MCS
 * field carries mcs + 8 (nss - 1).  VHT with len == 0: NDP.  Returns the samples per chain, or -1. */
int c8po_tx_packet2(int format, int mcs, int nss, const uint8_t* psdu, int len, int scr_seed, float mult,
                    c8p_c32* out0, c8p_c32* out1, int cap, uint8_t* coded_out, int* n_coded)
{
  c8p_tx_plan P;
  c8po_init();
  init_ps();
  if (!c8po_tx_plan2(format, mcs, nss, len, &P) || P.n_samp > cap || (nss == 2 && !out1)) return -1;
  const double csdL[2] = {0.0, -4.0}, csdN[2] = {0.0, -8.0};       /* C_CYCLIC_SHIFT_L / _NL for 2 streams, in samples */
  const double pil0[4] = {1, 1, 1, -1};
  /* ---- bit level, shared by the chains */
  uint8_t lsig[48], sig12[96], sigb[52], sigb_crc[8] = {0};
  {
    static const uint8_t RATE[8][4] = {{1, 1, 0, 1}, {1, 1, 1, 1}, {0, 1, 0, 1}, {0, 1, 1, 1}, {1, 0, 0, 1}, {1, 0, 1, 1}, {0, 0, 0, 1}, {0, 0, 1, 1}};
    uint8_t b[48], c[96];
    memset(b, 0, sizeof(b));                           /* L-SIG :237-282 */
    memcpy(b, RATE[format == 0 ? mcs : 0], 4);
    for (int i = 0; i < 12; i++) b[5 + i] = (uint8_t)((P.legacy_len >> i) & 1);
    int par = 0; for (int i = 0; i < 17; i++) par += b[i];
    b[17] = (uint8_t)(par & 1);
    bcc(b, 24, c); interleave_sig48(c, lsig);
    if (format == 1) {                                 /* HT-SIG :284-350 */
      memset(b, 0, sizeof(b));
      const int m = mcs + 8 * (nss - 1);
      for (int i = 0; i < 7; i++) b[i] = (uint8_t)((m >> i) & 1);
      for (int i = 0; i < 16; i++) b[8 + i] = (uint8_t)((P.psdu_len >> i) & 1);
      b[24] = 1; b[25] = 1; b[26] = 1;
      for (int i = 0; i < 34; i++) b[i] ^= (uint8_t)((g_sig_xor >> i) & 1);
      crc8_bits(b, 34, b + 34);
      bcc(b, 48, c); interleave_sig48(c, sig12); interleave_sig48(c + 48, sig12 + 48);
    } else if (format == 2) {                          /* VHT-SIG-A :352-455, SU, group id 0, partial AID 0 */
      memset(b, 0, sizeof(b));
      b[2] = 1;
      for (int i = 0; i < 3; i++) b[10 + i] = (uint8_t)(((nss - 1) >> i) & 1);
      b[23] = 1;
      for (int i = 0; i < 4; i++) b[28 + i] = (uint8_t)((mcs >> i) & 1);
      b[33] = 1;
      for (int i = 0; i < 34; i++) b[i] ^= (uint8_t)((g_sig_xor >> i) & 1);
      crc8_bits(b, 34, b + 34);
      bcc(b, 48, c); interleave_sig48(c, sig12); interleave_sig48(c + 48, sig12 + 48);
      memset(b, 0, sizeof(b));                         /* VHT-SIG-B :512-569 */
      if (len > 0) {
        const int l4 = (len + 3) / 4;
        for (int i = 0; i < 17; i++) b[i] = (uint8_t)((l4 >> i) & 1);
        b[17] = b[18] = b[19] = 1;
        crc8_bits(b, 20, sigb_crc);
      } else {
        static const uint8_t NDP20[20] = {0, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        memcpy(b, NDP20, 20);
      }
      bcc(b, 26, c);
      int32_t map[64];
      c8po_table(4 /* C8P_TAB_DEINT_SIGB */, 0, map, 64);
      for (int r = 0; r < 52; r++) sigb[r] = c[map[r]];
    }
  }
  /* ---- data bits (:639-682), scrambling and coding (:684-710) */
  const int nbits = P.nsym * P.ndbps;
  uint8_t* d = (uint8_t*)calloc((size_t)nbits + 8, 1);
  uint8_t* c2 = (uint8_t*)calloc((size_t)nbits * 2 + 16, 1);
  uint8_t* cp = (uint8_t*)calloc((size_t)P.nsym * P.ncbps + 16, 1);
  int ncoded = 0;
  if (nbits > 0) {
    int w = 16;
    if (format == 2) memcpy(d + 8, sigb_crc, 8);
    for (int i = 0; i < len; i++) for (int k = 0; k < 8; k++) d[w++] = (uint8_t)((psdu[i] >> k) & 1);
    if (format == 2) {
      static const uint8_t EOF32[32] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0};
      for (int e = 0; e < P.npad_eof; e++) { memcpy(d + w, EOF32, 32); w += 32; }
    }
    unsigned st = (unsigned)scr_seed;                  /* procScramble :800-806 */
    for (int i = 0; i < nbits; i++) {
      const unsigned fb = ((st >> 6) & 1u) ^ ((st >> 3) & 1u);
      d[i] ^= (uint8_t)fb;
      st = ((st << 1) & 0x7e) | fb;
    }
    if (format == 2) { for (int i = nbits - 6; i < nbits; i++) d[i] = 0; }     /* tail after scrambling (:692-696) */
    else { for (int i = 0; i < 6; i++) d[16 + 8 * P.psdu_len + i] = 0; }       /* tail reset (:700-702) */
    bcc(d, nbits, c2);
    ncoded = puncture(c2, 2 * nbits, P.cr, cp);
  }
  if (n_coded) *n_coded = ncoded;
  if (coded_out) memcpy(coded_out, cp, (size_t)ncoded);
  /* ---- per transmit chain */
  const int mi = P.nb == 1 ? 0 : P.nb == 2 ? 1 : P.nb == 4 ? 2 : P.nb == 6 ? 3 : 4;
  int32_t map1[512], map2[512];
  c8po_table(format == 0 ? 0 : 1, mi, map1, 512);
  c8po_table(2 /* C8P_TAB_DEINT_NL2 */, mi, map2, 512);
  const int ncbpss = P.nsd * P.nb, sblk = P.nb / 2 > 1 ? P.nb / 2 : 1;
  cd* sig = (cd*)calloc((size_t)P.n_samp, sizeof(cd));
  for (int ss = 0; ss < nss; ss++) {
    cd X[64];
    int pos = 0;
    const double sL = csdL[nss == 2 ? ss : 0], sN = csdN[nss == 2 ? ss : 0], dn = (double)nss;
    /* L-STF, L-LTF (:223-235) */
    memset(X, 0, sizeof(X));
    for (int m = 0; m < 12; m++) { const int k = (m < 6 ? -24 + 4 * m : 4 + 4 * (m - 6)); const double v = STF_SIGN[m] * sqrt(0.5); X[k + 32].re = v; X[k + 32].im = v; }
    { cd t[80]; c8po_ofdm_csd(X, 12.0 * dn, 0, sL, t); for (int n = 0; n < 32; n++) sig[n] = t[32 + n]; for (int r = 0; r < 2; r++) for (int n = 0; n < 64; n++) sig[32 + 64 * r + n] = t[n]; }
    memset(X, 0, sizeof(X));
    for (int k = -26; k <= 26; k++) X[k + 32].re = LTF_L53[k + 26];
    { cd t[80]; c8po_ofdm_csd(X, 52.0 * dn, 0, sL, t); for (int n = 0; n < 32; n++) sig[160 + n] = t[32 + n]; for (int r = 0; r < 2; r++) for (int n = 0; n < 64; n++) sig[192 + 64 * r + n] = t[n]; }
    pos = 320;
    { cd q[48]; for (int i = 0; i < 48; i++) { q[i].re = lsig[i] ? 1.0 : -1.0; q[i].im = 0; }
      place(q, 48, pil0, X); c8po_ofdm_csd(X, 52.0 * dn, 16, sL, sig + pos); pos += 80; }
    if (format != 0) {
      for (int s2 = 0; s2 < 2; s2++) {                 /* HT-SIG: QBPSK x2; VHT-SIG-A: BPSK then QBPSK */
        cd q[48];
        const int quad = format == 1 || s2 == 1;
        for (int i = 0; i < 48; i++) { const double v = sig12[48 * s2 + i] ? 1.0 : -1.0; q[i].re = quad ? 0 : v; q[i].im = quad ? v : 0; }
        place(q, 48, pil0, X); c8po_ofdm_csd(X, 52.0 * dn, 16, sL, sig + pos); pos += 80;
      }
      /* HT / VHT STF + nss LTFs (:459-510) */
      memset(X, 0, sizeof(X));
      for (int m = 0; m < 12; m++) { const int k = (m < 6 ? -24 + 4 * m : 4 + 4 * (m - 6)); const double v = STF_SIGN[m] * sqrt(0.5); X[k + 32].re = v; X[k + 32].im = v; }
      c8po_ofdm_csd(X, 12.0 * dn, 16, sN, sig + pos); pos += 80;
      static const int PM[2][2] = {{1, -1}, {1, 1}}, RM[2] = {1, -1};           /* C_P_LTF_VHT_4, C_R_LTF_VHT_4 */
      for (int l = 0; l < nss; l++) {
        memset(X, 0, sizeof(X));
        for (int k = -28; k <= 28; k++) {
          if (k == 0) continue;
          double v = (k >= -26 && k <= 26) ? (double)LTF_L53[k + 26] : (k < 0 ? 1.0 : -1.0);
          const int pilot = k == -21 || k == -7 || k == 7 || k == 21;
          v *= (format == 2 && pilot) ? RM[l] : PM[ss][l];
          X[k + 32].re = v;
        }
        c8po_ofdm_csd(X, 56.0 * dn, 16, sN, sig + pos); pos += 80;
      }
    }
    if (format == 2) {
      cd q[52]; for (int i = 0; i < 52; i++) { q[i].re = sigb[i] ? 1.0 : -1.0; q[i].im = 0; }
      place(q, 52, pil0, X); c8po_ofdm_csd(X, 56.0 * dn, 16, sN, sig + pos); pos += 80;
    }
    /* data symbols: stream parse + interleave (:712-746), map (:748-761), pilots + OFDM (:788-819) */
    double pil[4] = {1, 1, 1, -1};
    if (format == 1 && nss == 2) { if (ss == 0) { pil[2] = -1; } else { pil[1] = -1; pil[2] = -1; pil[3] = 1; } }   /* C_PILOT_HT 2 streams */
    int pidx = format == 0 ? 1 : format == 1 ? 3 : 4;
    for (int s3 = 0; s3 < P.nsym; s3++) {
      cd q[52];
      uint8_t bits[8];
      for (int j = 0; j < P.nsd; j++) {
        for (int k = 0; k < P.nb; k++) {
          const int r = j * P.nb + k;
          int ci = ss == 0 ? map1[r] : map2[r];        /* position inside the stream after de-interleaving */
          if (nss == 2) ci = ((ci / sblk) * 2 + ss) * sblk + ci % sblk;      /* stream parser */
          bits[k] = cp[s3 * P.ncbps + ci];
        }
        q[j] = qam_point(bits, P.nb);
      }
      (void)ncbpss;
      double pp[4]; for (int k = 0; k < 4; k++) pp[k] = pil[k] * PS127[pidx];
      place(q, P.nsd, pp, X);
      c8po_ofdm_csd(X, (format == 0 ? 52.0 : 56.0) * dn, 16, format == 0 ? sL : sN, sig + pos); pos += 80;
      pidx = (pidx + 1) % 127;
      if (format != 0) { const double t = pil[0]; pil[0] = pil[1]; pil[1] = pil[2]; pil[2] = pil[3]; pil[3] = t; }
    }
    /* procConcat2Symbol at every junction (:894-901), amplitude (:839) */
    for (int b = 160; b < P.n_samp; b += (b < 320 ? 160 : 80)) { sig[b - 1].re *= 0.5; sig[b - 1].im *= 0.5; sig[b].re *= 0.5; sig[b].im *= 0.5; }
    c8p_c32* out = ss == 0 ? out0 : out1;
    for (int n = 0; n < P.n_samp; n++) { out[n].re = (float)(sig[n].re * (double)mult); out[n].im = (float)(sig[n].im * (double)mult); }
  }
  free(sig); free(d); free(c2); free(cp);
  return P.n_samp;
}

int c8po_tx_packet(int format, int mcs, const uint8_t* psdu, int len, int scr_seed, float mult, c8p_c32* out, int cap,
                   uint8_t* coded_out, int* n_coded)
{
  return len > 0 ? c8po_tx_packet2(format, mcs, 1, psdu, len, scr_seed, mult, out, NULL, cap, coded_out, n_coded) : -1;
}

/* ------------------------------------------------------------------------------------------------------------------
 * 2-user VHT MU-MIMO packet, one spatial stream per user, steering matrix Q per subcarrier:
 * tools/phy80211.py genAmpduMu :180-221 (__genVhtSignalA MU branch :366-416, __genNonLegacyTraining :459-510 with
 * procSpatialMapping, __genVhtSignalBMu :571-637, __genOfdmSignalMu :763-786; phy80211header.py procPktLenAggreMu
 * :509-525, procSpatialMapping :1364-1378) = lib/encode2_impl.cc:129-215 + lib/modulation2_impl.cc:176-205,356-364
 * (formatToModMu cloud80211phy.cc:2558, vhtSigB20BitsGenMU :3084, procNss2SymBfQ :2932).
 * Pinned by tests/golden/packets_mu_tx.npz (the toolbox's own IQ on both chains and both users' coded bits). */
int c8po_tx_plan_mu(int mcs0, int len0, int mcs1, int len1, c8p_tx_plan* p0, c8p_tx_plan* p1)
{
  c8p_tx_plan* P[2] = {p0, p1};
  const int mcs[2] = {mcs0, mcs1}, len[2] = {len0, len1};
  int nsym = 0;
  for (int u = 0; u < 2; u++) {
    if (len[u] <= 0 || !c8po_tx_plan2(2, mcs[u], 1, len[u], P[u])) return 0;       /* procPktLenAggre per user */
    if (P[u]->nsym > nsym) nsym = P[u]->nsym;
  }
  for (int u = 0; u < 2; u++) if (nsym * P[u]->ndbps > 32782) return 0;            /* decode_impl.cc:93 / encode2_impl.h:74 */
  for (int u = 0; u < 2; u++) {                                                    /* procPktLenAggreMu */
    c8p_tx_plan* p = P[u];
    p->nsym = nsym;
    p->psdu_len = (nsym * p->ndbps - 16 - 6) / 8;
    p->npad_eof = (p->psdu_len - len[u]) / 4;
    p->npad_octet = p->psdu_len - len[u] - 4 * p->npad_eof;
    p->legacy_len = ((20 + 8 + 4 + 2 * 4 + 4 + nsym * 4 - 20) / 4) * 3 - 3;
    p->n_pre = 2 + 1 + 2 + 1;
    p->n_samp = 320 + 80 * (1 + p->n_pre + nsym);
  }
  return 1;
}

static void sig_symbol_bits(const uint8_t* b, int nbits, uint8_t* out)             /* rate 1/2 + procInterleaveSigL per 48 */
{
  uint8_t c[96];
  bcc(b, nbits, c);
  for (int h = 0; h < 2 * nbits / 48; h++) interleave_sig48(c + 48 * h, out + 48 * h);
}

int c8po_tx_packet_mu(int mcs0, const uint8_t* psdu0, int len0, int mcs1, const uint8_t* psdu1, int len1, int group_id,
                      const c8p_c32* bfq, int scr_seed, float mult, c8p_c32* out0, c8p_c32* out1, int cap,
                      uint8_t* coded0, uint8_t* coded1, int* n_coded /* [2] */)
{
  c8p_tx_plan P[2];
  c8po_init();
  init_ps();
  if (!c8po_tx_plan_mu(mcs0, len0, mcs1, len1, &P[0], &P[1]) || P[0].n_samp > cap) return -1;
  const int mcs[2] = {mcs0, mcs1}, len[2] = {len0, len1};
  const uint8_t* psdu[2] = {psdu0, psdu1};
  uint8_t* coded_out[2] = {coded0, coded1};
  const int nsym = P[0].nsym, n_samp = P[0].n_samp;
  const double pil0[4] = {1, 1, 1, -1};
  uint8_t lsig[48], siga[96], sigb[2][52], sigb_crc[2][8];
  {
    uint8_t b[48];
    memset(b, 0, sizeof(b));                            /* L-SIG, rate 6 Mb/s */
    b[0] = 1; b[1] = 1; b[3] = 1;
    for (int i = 0; i < 12; i++) b[5 + i] = (uint8_t)((P[0].legacy_len >> i) & 1);
    int par = 0; for (int i = 0; i < 17; i++) par += b[i];
    b[17] = (uint8_t)(par & 1);
    sig_symbol_bits(b, 24, lsig);
    memset(b, 0, sizeof(b));                            /* VHT-SIG-A, MU */
    b[2] = 1;
    for (int i = 0; i < 6; i++) b[4 + i] = (uint8_t)((group_id >> i) & 1);
    b[10] = 1; b[13] = 1;                               /* nSTS of user 0 / 1 = 1, users 2, 3 absent */
    b[23] = 1;
    b[29] = b[30] = b[31] = b[32] = b[33] = 1;          /* b28 = user-1 BCC; reserved ones */
    crc8_bits(b, 34, b + 34);
    sig_symbol_bits(b, 48, siga);
    int32_t map[64];
    c8po_table(4 /* C8P_TAB_DEINT_SIGB */, 0, map, 64);
    for (int u = 0; u < 2; u++) {                       /* VHT-SIG-B per user: len/4 (16 bits), mcs (4), tail */
      uint8_t c[52];
      memset(b, 0, sizeof(b));
      for (int i = 0; i < 16; i++) b[i] = (uint8_t)(((len[u] / 4) >> i) & 1);
      for (int i = 0; i < 4; i++) b[16 + i] = (uint8_t)((mcs[u] >> i) & 1);
      crc8_bits(b, 20, sigb_crc[u]);
      bcc(b, 26, c);
      for (int r = 0; r < 52; r++) sigb[u][r] = c[map[r]];
    }
  }
  /* per-user data field: service (SIG-B CRC) + A-MPDU + EOF delimiters + zero octets + pad, scramble, tail, BCC, puncture */
  uint8_t* cp[2];
  for (int u = 0; u < 2; u++) {
    const int nbits = nsym * P[u].ndbps;
    uint8_t* d = (uint8_t*)calloc((size_t)nbits + 8, 1);
    uint8_t* c2 = (uint8_t*)calloc((size_t)nbits * 2 + 16, 1);
    cp[u] = (uint8_t*)calloc((size_t)nsym * P[u].ncbps + 16, 1);
    static const uint8_t EOF32[32] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0};
    int w = 16;
    memcpy(d + 8, sigb_crc[u], 8);
    for (int i = 0; i < len[u]; i++) for (int k = 0; k < 8; k++) d[w++] = (uint8_t)((psdu[u][i] >> k) & 1);
    for (int e = 0; e < P[u].npad_eof; e++) { memcpy(d + w, EOF32, 32); w += 32; }
    unsigned st = (unsigned)scr_seed;
    for (int i = 0; i < nbits; i++) { const unsigned fb = ((st >> 6) & 1u) ^ ((st >> 3) & 1u); d[i] ^= (uint8_t)fb; st = ((st << 1) & 0x7e) | fb; }
    for (int i = nbits - 6; i < nbits; i++) d[i] = 0;
    bcc(d, nbits, c2);
    const int nc = puncture(c2, 2 * nbits, P[u].cr, cp[u]);
    if (n_coded) n_coded[u] = nc;
    if (coded_out[u]) memcpy(coded_out[u], cp[u], (size_t)nc);
    free(d); free(c2);
  }
  /* ---- frequency-domain fields per stream, spatial mapping, OFDM per chain */
  cd* sig[2];
  for (int c = 0; c < 2; c++) sig[c] = (cd*)calloc((size_t)n_samp, sizeof(cd));
  const double csdL[2] = {0.0, -4.0}, csdN[2] = {0.0, -8.0};
  cd X[64], Z[64];
  memset(Z, 0, sizeof(Z));
  /* legacy part and SIG-A: the same symbol on both chains, legacy cyclic shift on the second (no Q) */
  for (int c = 0; c < 2; c++) {
    memset(X, 0, sizeof(X));
    for (int m = 0; m < 12; m++) { const int k = (m < 6 ? -24 + 4 * m : 4 + 4 * (m - 6)); const double v = STF_SIGN[m] * sqrt(0.5); X[k + 32].re = v; X[k + 32].im = v; }
    { cd t[80]; c8po_ofdm_csd(X, 24.0, 0, csdL[c], t); for (int n = 0; n < 32; n++) sig[c][n] = t[32 + n]; for (int r = 0; r < 2; r++) for (int n = 0; n < 64; n++) sig[c][32 + 64 * r + n] = t[n]; }
    memset(X, 0, sizeof(X));
    for (int k = -26; k <= 26; k++) X[k + 32].re = LTF_L53[k + 26];
    { cd t[80]; c8po_ofdm_csd(X, 104.0, 0, csdL[c], t); for (int n = 0; n < 32; n++) sig[c][160 + n] = t[32 + n]; for (int r = 0; r < 2; r++) for (int n = 0; n < 64; n++) sig[c][192 + 64 * r + n] = t[n]; }
    cd q[48];
    for (int i = 0; i < 48; i++) { q[i].re = lsig[i] ? 1.0 : -1.0; q[i].im = 0; }
    place(q, 48, pil0, X); c8po_ofdm_csd(X, 104.0, 16, csdL[c], sig[c] + 320);
    for (int s2 = 0; s2 < 2; s2++) {
      for (int i = 0; i < 48; i++) { const double v = siga[48 * s2 + i] ? 1.0 : -1.0; q[i].re = s2 ? 0 : v; q[i].im = s2 ? v : 0; }
      place(q, 48, pil0, X); c8po_ofdm_csd(X, 104.0, 16, csdL[c], sig[c] + 400 + 80 * s2);
    }
  }
  /* steered fields: per stream X_s (with the VHT cyclic shift applied in the frequency domain), chain c = sum_s Q[c][s] X_s */
  const int n_steer = 1 + 2 + 1 + nsym;                 /* VHT-STF, 2 LTFs, SIG-B, data */
  double pil[4] = {1, 1, 1, -1};
  int pidx = 4;
  int32_t map1[512];
  for (int f = 0; f < n_steer; f++) {
    cd Xs[2][64];
    double ntf = 56.0;
    for (int s = 0; s < 2; s++) {
      memset(X, 0, sizeof(X));
      if (f == 0) {
        ntf = 12.0;
        for (int m = 0; m < 12; m++) { const int k = (m < 6 ? -24 + 4 * m : 4 + 4 * (m - 6)); const double v = STF_SIGN[m] * sqrt(0.5); X[k + 32].re = v; X[k + 32].im = v; }
      } else if (f <= 2) {
        static const int PM[2][2] = {{1, -1}, {1, 1}}, RM[2] = {1, -1};
        const int l = f - 1;
        for (int k = -28; k <= 28; k++) {
          if (k == 0) continue;
          double v = (k >= -26 && k <= 26) ? (double)LTF_L53[k + 26] : (k < 0 ? 1.0 : -1.0);
          const int pilot = k == -21 || k == -7 || k == 7 || k == 21;
          X[k + 32].re = v * (pilot ? RM[l] : PM[s][l]);
        }
      } else if (f == 3) {
        cd q[52]; for (int i = 0; i < 52; i++) { q[i].re = sigb[s][i] ? 1.0 : -1.0; q[i].im = 0; }
        place(q, 52, pil0, X);
      } else {
        const int s3 = f - 4;
        const int mi = P[s].nb == 1 ? 0 : P[s].nb == 2 ? 1 : P[s].nb == 4 ? 2 : P[s].nb == 6 ? 3 : 4;
        c8po_table(1 /* C8P_TAB_DEINT_NL */, mi, map1, 512);                      /* both users: the first stream's interleaver */
        cd q[52];
        uint8_t bits[8];
        for (int j = 0; j < 52; j++) {
          for (int k = 0; k < P[s].nb; k++) bits[k] = cp[s][s3 * P[s].ncbps + map1[j * P[s].nb + k]];
          q[j] = qam_point(bits, P[s].nb);
        }
        double pp[4]; for (int k = 0; k < 4; k++) pp[k] = pil[k] * PS127[pidx];
        place(q, 52, pp, X);
      }
      for (int i = 0; i < 64; i++) {                    /* procCSD in the frequency domain */
        const double ph = -2.0 * M_PI * csdN[s] * (double)(i - 32) / 64.0;
        Xs[s][i].re = X[i].re * cos(ph) - X[i].im * sin(ph);
        Xs[s][i].im = X[i].re * sin(ph) + X[i].im * cos(ph);
      }
    }
    if (f >= 4) { pidx = (pidx + 1) % 127; const double t = pil[0]; pil[0] = pil[1]; pil[1] = pil[2]; pil[2] = pil[3]; pil[3] = t; }
    for (int c = 0; c < 2; c++) {
      for (int i = 0; i < 64; i++) {
        const int bin = (i - 32) & 63;                  /* bfq is in FFT order */
        const c8p_c32 qa = bfq[4 * bin + 2 * c], qb = bfq[4 * bin + 2 * c + 1];
        X[i].re = (double)qa.re * Xs[0][i].re - (double)qa.im * Xs[0][i].im + (double)qb.re * Xs[1][i].re - (double)qb.im * Xs[1][i].im;
        X[i].im = (double)qa.re * Xs[0][i].im + (double)qa.im * Xs[0][i].re + (double)qb.re * Xs[1][i].im + (double)qb.im * Xs[1][i].re;
      }
      c8po_ofdm_csd(X, ntf * 2.0, 16, 0.0, sig[c] + 560 + 80 * f);
    }
  }
  for (int c = 0; c < 2; c++) {
    for (int b = 160; b < n_samp; b += (b < 320 ? 160 : 80)) { sig[c][b - 1].re *= 0.5; sig[c][b - 1].im *= 0.5; sig[c][b].re *= 0.5; sig[c][b].im *= 0.5; }
    c8p_c32* out = c == 0 ? out0 : out1;
    for (int n = 0; n < n_samp; n++) { out[n].re = (float)(sig[c][n].re * (double)mult); out[n].im = (float)(sig[c][n].im * (double)mult); }
    free(sig[c]);
  }
  free(cp[0]); free(cp[1]);
  return n_samp;
}
