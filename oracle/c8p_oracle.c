/*
 * c8p_oracle.c — scalar CPU restatement of the reference hot path (TEST INFRASTRUCTURE, see c8p_oracle.h).
 *
 * Arithmetic model (what the reference's C++ does when built with g++ >= 11 on x86-64 without FMA):
 *   complex * complex   inline (ac - bd, ad + bc) in float
 *   complex / complex   libgcc __divsc3: ((ac+bd)/(cc+dd), (bc-ad)/(cc+dd)) evaluated in double, rounded once
 *   complex {*,/} float per component
 *   std::abs(complex)   hypotf
 * Build with -ffp-contract=off and without -ffast-math / -march=native.
 * All file:line citations are relative to /root/reference/lib/.
 */
#include "c8p_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

typedef c8p_c32 cf;

/* ------------------------------------------------------------------ complex helpers */
static inline cf C(float re, float im) { cf r; r.re = re; r.im = im; return r; }
static inline cf cadd(cf a, cf b) { return C(a.re + b.re, a.im + b.im); }
static inline cf csub(cf a, cf b) { return C(a.re - b.re, a.im - b.im); }
static inline cf cneg(cf a) { return C(-a.re, -a.im); }
static inline cf cconj(cf a) { return C(a.re, -a.im); }
static inline cf cmul(cf a, cf b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline cf cmulf(cf a, float f) { return C(a.re * f, a.im * f); }
static inline cf cdivf(cf a, float f) { return C(a.re / f, a.im / f); }
cf c8po_cdiv(cf a, cf b)
{
  double aa = a.re, bb = a.im, cc = b.re, dd = b.im;
  double denom = (cc * cc) + (dd * dd);
  double x = ((aa * cc) + (bb * dd)) / denom;
  double y = ((bb * cc) - (aa * dd)) / denom;
  return C((float)x, (float)y);
}
#define cdiv c8po_cdiv
float c8po_cabsf(cf a) { return hypotf(a.re, a.im); }
/* bulk forms for the arithmetic-model tests (the device primitives are compared with these on millions of operands) */
void c8po_cabsf_n(const cf* a, float* out, long n) { for (long i = 0; i < n; i++) out[i] = hypotf(a[i].re, a[i].im); }
void c8po_cdiv_n(const cf* ab, cf* out, long n) { for (long i = 0; i < n; i++) out[i] = c8po_cdiv(ab[2 * i], ab[2 * i + 1]); }
/* std::complex<float> division as the compiler emits it for the reference (libgcc __divsc3 through the _Complex operator) */
void c8po_cdiv_libgcc_n(const cf* ab, cf* out, long n)
{
  for (long i = 0; i < n; i++) {
    const float _Complex a = ab[2 * i].re + ab[2 * i].im * 1.0fi, b = ab[2 * i + 1].re + ab[2 * i + 1].im * 1.0fi, q = a / b;
    out[i].re = __real__ q; out[i].im = __imag__ q;
  }
}
#define cabsf_ c8po_cabsf

/* ------------------------------------------------------------------ tables (Appendix A of SURVEY.md) */
static int g_init = 0;
static int SV_NEXT[64][2], SV_OUT[64][2];
static const int NBPSC_L[4] = {1, 2, 4, 6};
static const int NBPSC_NL[5] = {1, 2, 4, 6, 8};
static int DEINT_L[4][288];
static int DEINT_NL[5][416], DEINT_NL2[5][416];
static int DEINT_SIGB[52], INTL_SIGB[52];
static int DSC_L[64];        /* C8P_LEGACY_D_SC  cloud80211phy.cc:37 */
static int SIG_DEMAP_NL[128];/* FFT_NL_SIG_DEMAP cloud80211phy.cc:31 */
static float PILOT_P[127];   /* cloud80211phy.cc:160 */
static float LTF_L[64], LTF_NL[64], LTF_NL_VHT22[64];
static const float PILOT_L4[4] = {1.f, 1.f, 1.f, -1.f};       /* PILOT_L / PILOT_HT_1 / PILOT_VHT :170-174 */
static const float PILOT_HT21[4] = {1.f, 1.f, -1.f, -1.f};
static const float PILOT_HT22[4] = {1.f, -1.f, -1.f, 1.f};
static double TW_C[64], TW_S[64];
static uint32_t CRC32_T[256];

static const signed char LLTF_M26_P26[53] = {
  1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1,
  0,
  1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};

static void build_deint(int* map, int ncbps, int nbpsc, int ncol, int rot2)
{
  int s = nbpsc / 2 > 1 ? nbpsc / 2 : 1;
  int nrow = ncbps / ncol;
  for (int k = 0; k < ncbps; k++) {
    int i = nrow * (k % ncol) + k / ncol;
    int j = s * (i / s) + (i + ncbps - (ncol * i) / ncbps) % s;
    int r = j;
    if (rot2) r = ((j - 2 * 11 * nbpsc) % ncbps + ncbps) % ncbps;
    map[r] = k;
  }
}

void c8po_init(void)
{
  if (g_init) return;
  for (int i = 0; i < 64; i++)
    for (int b = 0; b < 2; b++) {
      int x[7];
      x[0] = b;
      for (int k = 1; k < 7; k++) x[k] = (i >> (6 - k)) & 1;   /* x1 = i5 ... x6 = i0 */
      int A = x[0] ^ x[2] ^ x[3] ^ x[5] ^ x[6];
      int B = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[6];
      SV_NEXT[i][b] = (i >> 1) | (b << 5);
      SV_OUT[i][b] = 2 * A + B;
    }
  for (int m = 0; m < 4; m++) build_deint(DEINT_L[m], 48 * NBPSC_L[m], NBPSC_L[m], 16, 0);
  for (int m = 0; m < 5; m++) {
    build_deint(DEINT_NL[m], 52 * NBPSC_NL[m], NBPSC_NL[m], 13, 0);
    build_deint(DEINT_NL2[m], 52 * NBPSC_NL[m], NBPSC_NL[m], 13, 1);
  }
  for (int i = 0; i < 52; i++) { DEINT_SIGB[i] = DEINT_NL[0][i]; }
  for (int i = 0; i < 52; i++) { INTL_SIGB[DEINT_SIGB[i]] = i; }
  /* data-subcarrier rank in ascending subcarrier order without pilots / DC */
  {
    int rank = 0;
    for (int i = 0; i < 64; i++) DSC_L[i] = -1;
    for (int sc = -26; sc <= 26; sc++) {
      if (sc == 0 || sc == -21 || sc == -7 || sc == 7 || sc == 21) continue;
      DSC_L[(sc + 64) % 64] = rank++;
    }
    for (int i = 0; i < 64; i++) {
      int v = DSC_L[i] < 0 ? -1 : DEINT_L[0][DSC_L[i]];
      SIG_DEMAP_NL[i] = v;
      SIG_DEMAP_NL[i + 64] = v < 0 ? -1 : v + 48;
    }
  }
  {
    int st = 0x7f;
    for (int n = 0; n < 127; n++) {
      int fb = ((st >> 6) & 1) ^ ((st >> 3) & 1);
      PILOT_P[n] = fb ? -1.0f : 1.0f;
      st = ((st << 1) & 0x7e) | fb;
    }
  }
  for (int i = 0; i < 64; i++) { LTF_L[i] = 0; LTF_NL[i] = 0; }
  for (int sc = -26; sc <= 26; sc++) {
    LTF_L[(sc + 64) % 64] = (float)LLTF_M26_P26[sc + 26];
    LTF_NL[(sc + 64) % 64] = (float)LLTF_M26_P26[sc + 26];
  }
  LTF_NL[(-28 + 64)] = 1; LTF_NL[(-27 + 64)] = 1; LTF_NL[27] = -1; LTF_NL[28] = -1;
  for (int i = 0; i < 64; i++) LTF_NL_VHT22[i] = LTF_NL[i];
  LTF_NL_VHT22[7] = -LTF_NL[7]; LTF_NL_VHT22[21] = -LTF_NL[21];
  LTF_NL_VHT22[43] = -LTF_NL[43]; LTF_NL_VHT22[57] = -LTF_NL[57];
  for (int i = 0; i < 64; i++) {
    double a = 2.0 * 3.14159265358979323846 * (double)i / 64.0;
    TW_C[i] = cos(a); TW_S[i] = sin(a);
  }
  /* exact values on the axes and diagonals */
  TW_C[0] = 1; TW_S[0] = 0; TW_C[16] = 0; TW_S[16] = 1; TW_C[32] = -1; TW_S[32] = 0; TW_C[48] = 0; TW_S[48] = -1;
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    CRC32_T[n] = c;
  }
  g_init = 1;
}

int c8po_table(int which, int sub, int32_t* dst, int cap)
{
  c8po_init();
  int n = 0;
  int32_t tmp[416];
  switch (which) {
    case C8P_TAB_DEINT_L: if (sub < 0 || sub > 3) return -1; n = 48 * NBPSC_L[sub]; for (int i = 0; i < n; i++) tmp[i] = DEINT_L[sub][i]; break;
    case C8P_TAB_DEINT_NL: if (sub < 0 || sub > 4) return -1; n = 52 * NBPSC_NL[sub]; for (int i = 0; i < n; i++) tmp[i] = DEINT_NL[sub][i]; break;
    case C8P_TAB_DEINT_NL2: if (sub < 0 || sub > 4) return -1; n = 52 * NBPSC_NL[sub]; for (int i = 0; i < n; i++) tmp[i] = DEINT_NL2[sub][i]; break;
    case C8P_TAB_SIG_DEMAP_NL: n = 128; for (int i = 0; i < n; i++) tmp[i] = SIG_DEMAP_NL[i]; break;
    case C8P_TAB_DEINT_SIGB: n = 52; for (int i = 0; i < n; i++) tmp[i] = DEINT_SIGB[i]; break;
    case C8P_TAB_SV_OUT: n = 128; for (int i = 0; i < 64; i++) { tmp[2 * i] = SV_OUT[i][0]; tmp[2 * i + 1] = SV_OUT[i][1]; } break;
    case C8P_TAB_PILOT_P: n = 127; for (int i = 0; i < n; i++) tmp[i] = (int)PILOT_P[i]; break;
    case C8P_TAB_LTF_NL: n = 64; for (int i = 0; i < n; i++) tmp[i] = (int)LTF_NL[i]; break;
    case C8P_TAB_LTF_NL_VHT22: n = 64; for (int i = 0; i < n; i++) tmp[i] = (int)LTF_NL_VHT22[i]; break;
    default: return -1;
  }
  for (int i = 0; i < n && i < cap; i++) dst[i] = tmp[i];
  return n;
}

/* ------------------------------------------------------------------ FFT (demod_impl.cc:541-547 → gr::fft::fft_complex_fwd(64)) */
/* The pinned model (0) is the exact DFT rounded once to float.  The reference's transform is FFTW3f, whose float rounding depends on
 * the plan it picks and which cannot be built here; models 1-3 are float32 transforms of three different operation orders, there ONLY so
 * that tests/test_oracle_golden.py can measure how far the LLR stream moves when the FFT rounds in float (NOTES/parity.md): radix-2
 * decimation in time, radix-4 decimation in frequency, and the plain 64-term sum in float (the worst of the three by construction). */
static int FFT_MODEL = 0;
void c8po_set_fft_model(int m) { FFT_MODEL = m; }
int c8po_get_fft_model(void) { return FFT_MODEL; }
static void fft64_f32_radix2_dit(const cf* in, cf* out)
{
  float xr[64], xi[64];
  for (int n = 0; n < 64; n++) {
    int r = 0;
    for (int b = 0; b < 6; b++) r |= ((n >> b) & 1) << (5 - b);
    xr[r] = in[n].re; xi[r] = in[n].im;
  }
  for (int len = 2; len <= 64; len <<= 1) {
    const int half = len >> 1, step = 64 / len;
    for (int i = 0; i < 64; i += len)
      for (int k = 0; k < half; k++) {
        const float c = (float)TW_C[k * step], s = (float)-TW_S[k * step];
        const float ur = xr[i + k], ui = xi[i + k];
        const float vr = xr[i + k + half] * c - xi[i + k + half] * s, vi = xr[i + k + half] * s + xi[i + k + half] * c;
        xr[i + k] = ur + vr; xi[i + k] = ui + vi;
        xr[i + k + half] = ur - vr; xi[i + k + half] = ui - vi;
      }
  }
  for (int k = 0; k < 64; k++) { out[k].re = xr[k]; out[k].im = xi[k]; }
}
static void fft64_f32_radix4_dif(const cf* in, cf* out)
{
  float xr[64], xi[64];
  for (int n = 0; n < 64; n++) { xr[n] = in[n].re; xi[n] = in[n].im; }
  for (int len = 64; len >= 4; len >>= 2) {           /* three stages: 64, 16, 4 */
    const int q = len >> 2, step = 64 / len;
    for (int i = 0; i < 64; i += len)
      for (int k = 0; k < q; k++) {
        const float ar = xr[i + k], ai = xi[i + k], br = xr[i + k + q], bi = xi[i + k + q];
        const float cr = xr[i + k + 2 * q], ci = xi[i + k + 2 * q], dr = xr[i + k + 3 * q], di = xi[i + k + 3 * q];
        const float s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
        const float s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
        const float y0r = s0r + s2r, y0i = s0i + s2i;               /* bin 0 of the 4-point DFT */
        const float y2r = s0r - s2r, y2i = s0i - s2i;               /* bin 2 */
        const float y1r = s1r + s3i, y1i = s1i - s3r;               /* bin 1: s1 - j s3 */
        const float y3r = s1r - s3i, y3i = s1i + s3r;               /* bin 3: s1 + j s3 */
        const float c1 = (float)TW_C[k * step], n1 = (float)-TW_S[k * step];
        const float c2 = (float)TW_C[(2 * k * step) & 63], n2 = (float)-TW_S[(2 * k * step) & 63];
        const float c3 = (float)TW_C[(3 * k * step) & 63], n3 = (float)-TW_S[(3 * k * step) & 63];
        xr[i + k] = y0r; xi[i + k] = y0i;
        xr[i + k + q] = y1r * c1 - y1i * n1; xi[i + k + q] = y1r * n1 + y1i * c1;
        xr[i + k + 2 * q] = y2r * c2 - y2i * n2; xi[i + k + 2 * q] = y2r * n2 + y2i * c2;
        xr[i + k + 3 * q] = y3r * c3 - y3i * n3; xi[i + k + 3 * q] = y3r * n3 + y3i * c3;
      }
  }
  for (int n = 0; n < 64; n++) {                       /* base-4 digit reversal */
    const int r = ((n & 3) << 4) | (n & 12) | ((n >> 4) & 3);
    out[r].re = xr[n]; out[r].im = xi[n];
  }
}
static void fft64_f32_direct(const cf* in, cf* out)
{
  for (int k = 0; k < 64; k++) {
    float sr = 0.0f, si = 0.0f;
    for (int n = 0; n < 64; n++) {
      const int t = (k * n) & 63;
      const float c = (float)TW_C[t], s = (float)-TW_S[t];
      sr += in[n].re * c - in[n].im * s;
      si += in[n].re * s + in[n].im * c;
    }
    out[k].re = sr; out[k].im = si;
  }
}
void c8po_fft64(const cf* in, cf* out)
{
  c8po_init();
  if (FFT_MODEL == 1) { fft64_f32_radix2_dit(in, out); return; }
  if (FFT_MODEL == 2) { fft64_f32_radix4_dif(in, out); return; }
  if (FFT_MODEL == 3) { fft64_f32_direct(in, out); return; }
  for (int k = 0; k < 64; k++) {
    double sr = 0.0, si = 0.0;
    for (int n = 0; n < 64; n++) {
      int t = (k * n) & 63;
      double c = TW_C[t], s = -TW_S[t];      /* e^{-j 2 pi k n / 64} */
      sr += (double)in[n].re * c - (double)in[n].im * s;
      si += (double)in[n].re * s + (double)in[n].im * c;
    }
    out[k].re = (float)sr; out[k].im = (float)si;
  }
}

/* ------------------------------------------------------------------ CRC-8 (cloud80211phy.cc:1367-1403) */
int c8po_crc8_check(const uint8_t* bits, int len, const uint8_t* crcbits)
{
  unsigned c = 0xff;
  for (int i = 0; i < len; i++) {
    unsigned fb = ((c >> 7) & 1u);
    c = (c << 1) & 0xffu;
    if (fb) c ^= 0x07u;
    if (bits[i]) c ^= 0x07u;
  }
  for (int i = 0; i < 8; i++) if (crcbits[i]) c ^= (1u << (7 - i));
  return (c & 0xffu) == 0xffu;
}

/* bccEncoder cloud80211phy.cc:2622-2645 */
void c8po_bcc_encode(const uint8_t* in, uint8_t* out, int len)
{
  int state = 0;
  for (int i = 0; i < len; i++) {
    state = ((state << 1) & 0x7e) | (in[i] & 1);
    out[2 * i] = (uint8_t)(__builtin_popcount(state & 0155) & 1);
    out[2 * i + 1] = (uint8_t)(__builtin_popcount(state & 0117) & 1);
  }
}

#ifndef C8PO_BUILD_FLAGS
#define C8PO_BUILD_FLAGS "unknown"
#endif
const char* c8po_build_flags(void) { return C8PO_BUILD_FLAGS; }

uint32_t c8po_crc32(const uint8_t* p, int n)
{
  c8po_init();
  uint32_t c = 0xffffffffu;
  for (int i = 0; i < n; i++) c = CRC32_T[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

/* ------------------------------------------------------------------ Viterbi core shared by SIG and data decoders
 * recurrence of svSigDecoder::decode (cloud80211phy.cc:2019-2063) == vstb_update (decode_impl.cc:231-274) */
#define NEG_INIT (-1000000000000000.0f)

void c8po_sig_viterbi(const float* llr, uint8_t* bits, int L)
{
  c8po_init();
  if (L < 0 || L > 48) return;
  float m0[64], m1[64];
  float *pre = m0, *cur = m1;
  int his[64][49];
  int seq[49];
  for (int i = 0; i < 64; i++) {
    for (int j = 0; j <= L; j++) his[i][j] = 0;
    m0[i] = NEG_INIT; m1[i] = NEG_INIT;
  }
  m0[0] = 0;
  for (int t = 0; t < L; t++) {
    float t0 = *llr++, t1 = *llr++;
    float tab[4];
    tab[0] = 0; tab[1] = t1; tab[2] = t0; tab[3] = t1 + t0;
    for (int i = 0; i < 64; i++) {
      float a0 = pre[i] + tab[SV_OUT[i][0]];
      float a1 = pre[i] + tab[SV_OUT[i][1]];
      int n0 = SV_NEXT[i][0], n1 = SV_NEXT[i][1];
      if (a0 > cur[n0]) { cur[n0] = a0; his[n0][t + 1] = i; }
      if (a1 > cur[n1]) { cur[n1] = a1; his[n1][t + 1] = i; }
    }
    float* tmp = pre; pre = cur; cur = tmp;
    for (int i = 0; i < 64; i++) cur[i] = NEG_INIT;
  }
  seq[L] = 0;
  for (int j = L; j > 0; j--) seq[j - 1] = his[seq[j]][j];
  for (int j = 0; j < L; j++) bits[j] = (seq[j + 1] == SV_NEXT[seq[j]][1]) ? 1 : 0;
}

static const int PUNC_12[2] = {1, 1};
static const int PUNC_23[4] = {1, 1, 1, 0};
static const int PUNC_34[6] = {1, 1, 1, 0, 0, 1};
static const int PUNC_56[10] = {1, 1, 1, 0, 0, 1, 1, 0, 0, 1};

/* vstb_init :164-203, vstb_update :205-281, vstb_end :282-302 */
int c8po_viterbi(const float* llr, int n_llr, int trellis, int cr, uint8_t* sbits)
{
  c8po_init();
  const int* punc; int plen;
  switch (cr) {
    case C8P_CR_12: punc = PUNC_12; plen = 2; break;
    case C8P_CR_23: punc = PUNC_23; plen = 4; break;
    case C8P_CR_34: punc = PUNC_34; plen = 6; break;
    case C8P_CR_56: punc = PUNC_56; plen = 10; break;
    default: return -1;
  }
  /* survivor table: predecessor state per (step, state), 0-initialised like v_state_his */
  uint8_t* his = (uint8_t*)calloc((size_t)(trellis + 1) * 64, 1);
  int* seq = (int*)malloc(sizeof(int) * (size_t)(trellis + 1));
  float m0[64], m1[64];
  float *pre = m0, *cur = m1;
  for (int i = 0; i < 64; i++) { m0[i] = NEG_INIT; m1[i] = NEG_INIT; }
  m0[0] = 0;
  int used = 0, p = 0, t = 0;
  while (t < trellis && (used + punc[p] + punc[p + 1]) <= n_llr) {
    float t0, t1;
    if (punc[p]) { t0 = llr[used++]; } else { t0 = 0.0f; }
    if (punc[p + 1]) { t1 = llr[used++]; } else { t1 = 0.0f; }
    float tab[4];
    tab[0] = 0.0f; tab[1] = t1; tab[2] = t0; tab[3] = t1 + t0;
    uint8_t* h = his + (size_t)(t + 1) * 64;
    for (int i = 0; i < 64; i++) {
      float a0 = pre[i] + tab[SV_OUT[i][0]];
      float a1 = pre[i] + tab[SV_OUT[i][1]];
      int n0 = SV_NEXT[i][0], n1 = SV_NEXT[i][1];
      if (a0 > cur[n0]) { cur[n0] = a0; h[n0] = (uint8_t)i; }
      if (a1 > cur[n1]) { cur[n1] = a1; h[n1] = (uint8_t)i; }
    }
    float* tmp = pre; pre = cur; cur = tmp;
    for (int i = 0; i < 64; i++) cur[i] = NEG_INIT;
    p += 2; if (p >= plen) p = 0;
    t++;
  }
  /* the reference only runs vstb_end once v_t >= v_trellis (decode_impl.cc:131-137) */
  if (t >= trellis) {
    seq[trellis] = 0;
    for (int j = trellis; j > 0; j--) seq[j - 1] = his[(size_t)j * 64 + seq[j]];
    for (int j = 0; j < trellis; j++) sbits[j] = (seq[j + 1] == SV_NEXT[seq[j]][1]) ? 1 : 0;
  } else {
    used = -1;
  }
  free(his); free(seq);
  return used;
}

/* descramble decode_impl.cc:304-323 */
void c8po_descramble(const uint8_t* sbits, int trellis, uint8_t* bits)
{
  int state = 0;
  for (int i = 0; i < 7 && i < trellis; i++) if (sbits[i]) state |= 1 << (6 - i);
  for (int i = 0; i < 7 && i < trellis; i++) bits[i] = 0;
  for (int i = 7; i < trellis; i++) {
    int fb = (!!(state & 64)) ^ (!!(state & 8));
    bits[i] = (uint8_t)(fb ^ (sbits[i] & 1));
    state = ((state << 1) & 0x7e) | fb;
  }
}

/* ------------------------------------------------------------------ parsers */
typedef struct {
  int format, sumu, ampdu, nSym, nSymSamp, nSD, nSS, nLTF, mcs, len, mod, cr, nBPSCS, nDBPS, nCBPS, nCBPSS;
  int valid;   /* 0 when the reference would read stale fields (unknown MCS) */
} mod_t;

static void set_rate(mod_t* m)   /* tail of modParserVht/signalParserHt :1294-1322 / :947-998 */
{
  m->nSD = 52;
  m->nCBPSS = m->nBPSCS * m->nSD;
  m->nCBPS = m->nCBPSS * m->nSS;
  switch (m->cr) {
    case C8P_CR_12: m->nDBPS = m->nCBPS / 2; break;
    case C8P_CR_23: m->nDBPS = (m->nCBPS * 2) / 3; break;
    case C8P_CR_34: m->nDBPS = (m->nCBPS * 3) / 4; break;
    default:        m->nDBPS = (m->nCBPS * 5) / 6; break;
  }
  switch (m->nSS) { case 1: m->nLTF = 1; break; case 2: m->nLTF = 2; break; case 3: case 4: m->nLTF = 4; break; default: break; }
}

static void parse_legacy(int mcs, int len, mod_t* m)   /* signalParserL :773-850 */
{
  static const int nb[8] = {1, 1, 2, 2, 4, 4, 6, 6};
  static const int crs[8] = {C8P_CR_12, C8P_CR_34, C8P_CR_12, C8P_CR_34, C8P_CR_12, C8P_CR_34, C8P_CR_23, C8P_CR_34};
  static const int ndbps[8] = {24, 36, 48, 72, 96, 144, 192, 216};
  memset(m, 0, sizeof(*m));
  m->valid = (mcs >= 0 && mcs <= 7);
  if (!m->valid) return;
  m->mcs = mcs; m->nBPSCS = nb[mcs]; m->cr = crs[mcs]; m->nDBPS = ndbps[mcs]; m->nCBPS = 48 * nb[mcs];
  m->len = len; m->nCBPSS = m->nCBPS; m->nSD = 48; m->nSS = 1; m->sumu = 0; m->nLTF = 0;
  m->format = C8P_F_L; m->nSymSamp = 80;
  m->nSym = (len * 8 + 22) / m->nDBPS + (((len * 8 + 22) % m->nDBPS) != 0);
  m->ampdu = 0;
}

static void parse_ht(const uint8_t* b, mod_t* m)   /* signalParserHt :852-998 */
{
  static const int nb[8] = {1, 2, 2, 4, 4, 6, 6, 6};
  static const int crs[8] = {C8P_CR_12, C8P_CR_12, C8P_CR_34, C8P_CR_12, C8P_CR_34, C8P_CR_23, C8P_CR_34, C8P_CR_56};
  memset(m, 0, sizeof(*m));
  int mcs = 0, len = 0;
  for (int i = 0; i < 7; i++) mcs |= ((int)b[i]) << i;
  for (int i = 0; i < 16; i++) len |= ((int)b[i + 8]) << i;
  m->valid = 1;
  m->format = C8P_F_HT; m->sumu = 0;
  m->nSymSamp = b[31] ? 72 : 80;
  m->ampdu = b[27] ? 1 : 0;
  m->mcs = mcs; m->nBPSCS = nb[mcs % 8]; m->cr = crs[mcs % 8];
  m->len = len; m->nSS = mcs / 8 + 1;
  set_rate(m);
  m->nSym = ((len * 8 + 22) / m->nDBPS + (((len * 8 + 22) % m->nDBPS) != 0));
}

static int mod_vht(int mcs, mod_t* m)   /* modParserVht :1225-1323 */
{
  static const int nb[10] = {1, 2, 2, 4, 4, 6, 6, 6, 8, 8};
  static const int crs[10] = {C8P_CR_12, C8P_CR_12, C8P_CR_34, C8P_CR_12, C8P_CR_34, C8P_CR_23, C8P_CR_34, C8P_CR_56, C8P_CR_34, C8P_CR_56};
  m->mcs = mcs;
  if (mcs < 0 || mcs > 9) { m->valid = 0; return 0; }
  m->nBPSCS = nb[mcs]; m->cr = crs[mcs];
  set_rate(m);
  return 1;
}

static void parse_vht_a(const uint8_t* b, mod_t* m)   /* signalParserVhtA :1090-1178 */
{
  memset(m, 0, sizeof(*m));
  m->valid = 1;
  int gid = 0;
  for (int i = 0; i < 6; i++) gid |= ((int)b[i + 4]) << i;
  m->format = C8P_F_VHT;
  m->nSymSamp = b[24] ? 72 : 80;
  m->ampdu = 1;
  if (gid == 0 || gid == 63) {
    int nsts = 0, mcs = 0;
    for (int i = 0; i < 3; i++) nsts |= ((int)b[i + 10]) << i;
    for (int i = 0; i < 4; i++) mcs |= ((int)b[i + 28]) << i;
    m->sumu = 0; m->nSS = nsts + 1;
    if (m->nSS > 4) { m->nLTF = 0; m->valid = 0; }   /* nLTF would be stale in the reference */
    mod_vht(mcs, m);
  } else {
    m->sumu = 1; m->nLTF = 2; m->nSS = 1; m->nSD = 52;
  }
}

static void parse_vht_b(const uint8_t* b, mod_t* m)   /* signalParserVhtB :1180-1223 */
{
  int len = 0, mcs = 0;
  if (m->sumu) {
    for (int i = 0; i < 16; i++) len |= ((int)b[i]) << i;
    for (int i = 0; i < 4; i++) mcs |= ((int)b[i + 16]) << i;
    if (!mod_vht(mcs, m)) return;
    m->len = len * 4;
    m->nSym = (m->len * 8 + 16 + 6) / m->nDBPS + (((m->len * 8 + 16 + 6) % m->nDBPS) != 0);
    m->nLTF = 2;
  } else {
    if ((b[17] + b[18] + b[19]) == 3) {
      for (int i = 0; i < 17; i++) len |= ((int)b[i]) << i;
      m->len = len * 4;
      if (!m->valid) return;
      m->nSym = (m->len * 8 + 16 + 6) / m->nDBPS + (((m->len * 8 + 16 + 6) % m->nDBPS) != 0);
    } else {
      uint32_t pat = 0;
      for (int i = 0; i < 20; i++) pat |= ((uint32_t)b[i]) << i;
      if (pat == 0x422E0u) { m->len = 0; m->nSym = 0; }     /* 0b01000010001011100000 */
      else { m->len = -1; m->nSym = -1; }
    }
  }
}

static int check_ht(const uint8_t* b)   /* signalCheckHt :730-751 */
{
  if (b[26] != 1) return 0;
  if (!c8po_crc8_check(b, 34, &b[34])) return 0;
  if (b[5] + b[6] + b[7] + b[28] + b[29] + b[30] + b[32] + b[33]) return 0;
  return 1;
}
static int check_vht_a(const uint8_t* b)   /* signalCheckVhtA :753-771 */
{
  if ((b[2] != 1) || (b[23] != 1) || (b[33] != 1)) return 0;
  if (!c8po_crc8_check(b, 34, &b[34])) return 0;
  if (b[0] + b[1]) return 0;
  return 1;
}

/* The SIG checks and parsers on their own (test entry points: tests/test_oracle_rx_golden.py compares them with the reference
 * toolbox's procCheckSig* / procParserSig* / modulation on fuzzed fields).  out[10] = {format, mcs, len, nSS, nBPSCS, nCBPS, nDBPS, cr,
 * nSym, nLTF}; return value: the check (signalCheckLegacy :650-728, signalCheckHt :730-751, signalCheckVhtA :753-771). */
static void mod_out(const mod_t* m, int* out)
{
  out[0] = m->format; out[1] = m->mcs; out[2] = m->len; out[3] = m->nSS; out[4] = m->nBPSCS; out[5] = m->nCBPS; out[6] = m->nDBPS; out[7] = m->cr;
  out[8] = m->nSym; out[9] = m->nLTF;
}
int c8po_sig_legacy(const uint8_t* bits24, int* out)
{
  c8po_init();
  memset(out, 0, sizeof(int) * 10);
  const uint8_t* bits = bits24;
  if (!bits[3]) return 0;
  if (bits[4]) return 0;
  int par = 0; for (int i = 0; i < 17; i++) par += bits[i];
  if ((par & 1) ^ bits[17]) return 0;
  int rate = 0; for (int i = 0; i < 4; i++) rate |= ((int)bits[i]) << i;
  int mcs;
  switch (rate) {
    case 11: mcs = 0; break; case 15: mcs = 1; break; case 10: mcs = 2; break; case 14: mcs = 3; break;
    case 9:  mcs = 4; break; case 13: mcs = 5; break; case 8:  mcs = 6; break; case 12: mcs = 7; break;
    default: mcs = 0; break;
  }
  int len = 0; for (int i = 0; i < 12; i++) len |= ((int)bits[i + 5]) << i;
  if (len > 4095 || len < 14) return 0;
  mod_t m;
  parse_legacy(mcs, len, &m);
  mod_out(&m, out);
  return 1;
}
int c8po_sig_ht(const uint8_t* bits48, int* out)
{
  c8po_init();
  mod_t m;
  parse_ht(bits48, &m);
  mod_out(&m, out);
  return check_ht(bits48);
}
int c8po_sig_vht_a(const uint8_t* bits48, int* out)
{
  c8po_init();
  mod_t m;
  parse_vht_a(bits48, &m);
  mod_out(&m, out);
  out[2] = -1; out[8] = -1;                     /* length and symbol count come with SIG-B */
  return check_vht_a(bits48);
}

/* ------------------------------------------------------------------ LLR demapper  procSymQamToLlr cloud80211phy.cc:2090-2148 */
void c8po_qam_to_llr(cf* q, float* o, int nsd, int nbpscs)
{
  switch (nbpscs) {
    case 1:
      for (int i = 0; i < nsd; i++) o[i] = q[i].re;
      break;
    case 2:
      for (int i = 0; i < nsd; i++) { q[i] = cmulf(q[i], 1.4142135623730951f); o[i * 2] = q[i].re; o[i * 2 + 1] = q[i].im; }
      break;
    case 4:
      for (int i = 0; i < nsd; i++) {
        q[i] = cmulf(q[i], 3.1622776601683795f);
        o[i * 4] = q[i].re; o[i * 4 + 1] = 2.0f - fabsf(q[i].re);
        o[i * 4 + 2] = q[i].im; o[i * 4 + 3] = 2.0f - fabsf(q[i].im);
      }
      break;
    case 6:
      for (int i = 0; i < nsd; i++) {
        q[i] = cmulf(q[i], 6.48074069840786f);
        o[i * 6] = q[i].re; o[i * 6 + 1] = 4.0f - fabsf(q[i].re); o[i * 6 + 2] = 2 - fabsf(4.0f - fabsf(q[i].re));
        o[i * 6 + 3] = q[i].im; o[i * 6 + 4] = 4.0f - fabsf(q[i].im); o[i * 6 + 5] = 2 - fabsf(4.0f - fabsf(q[i].im));
      }
      break;
    case 8:
      for (int i = 0; i < nsd; i++) {
        q[i] = cmulf(q[i], 13.038404810405298f);
        float r = q[i].re, m = q[i].im;
        o[i * 8] = r; o[i * 8 + 1] = 8.0f - fabsf(r); o[i * 8 + 2] = 4 - fabsf(8.0f - fabsf(r)); o[i * 8 + 3] = 2 - fabsf(4 - fabsf(8.0f - fabsf(r)));
        o[i * 8 + 4] = m; o[i * 8 + 5] = 8.0f - fabsf(m); o[i * 8 + 6] = 4 - fabsf(8.0f - fabsf(m)); o[i * 8 + 7] = 2 - fabsf(4 - fabsf(8.0f - fabsf(m)));
      }
      break;
    default: break;
  }
}

static int nbpsc_index(int nb) { switch (nb) { case 1: return 0; case 2: return 1; case 4: return 2; case 6: return 3; case 8: return 4; default: return -1; } }

/* ------------------------------------------------------------------ signal front-end (cfo = 0)  signal_impl.cc:108-162, procLHSigDemodDeint :609-627 */
int c8po_signal(const cf* raw, cf* h, int* mcs, int* len, int* nsamp, float* llr48, uint8_t* bits24)
{
  c8po_init();
  cf f1[64], f2[64], fs[64];
  float llr[48]; uint8_t bits[24];
  c8po_fft64(raw + 8, f1); c8po_fft64(raw + 8 + 64, f2); c8po_fft64(raw + 8 + 144, fs);
  for (int i = 0; i < 64; i++) h[i] = C(0, 0);
  const int pil[4] = {7, 21, 43, 57};
  for (int k = 0; k < 4; k++) { int i = pil[k]; h[i] = cdivf(cadd(f1[i], f2[i]), LTF_L[i] * 2.0f); }
  cf ps = cconj(cadd(cadd(csub(cdiv(fs[7], h[7]), cdiv(fs[21], h[21])), cdiv(fs[43], h[43])), cdiv(fs[57], h[57])));
  float pa = cabsf_(ps);
  for (int i = 0; i < 64; i++) {
    if (SIG_DEMAP_NL[i] > -1) {
      h[i] = cdivf(cadd(f1[i], f2[i]), LTF_L[i] * 2.0f);
      llr[SIG_DEMAP_NL[i]] = cdivf(cmul(cdiv(fs[i], h[i]), ps), pa).re;
    }
  }
  c8po_sig_viterbi(llr, bits, 24);
  if (llr48) memcpy(llr48, llr, sizeof(llr));
  if (bits24) memcpy(bits24, bits, 24);
  /* signalCheckLegacy :650-728 */
  if (!bits[3]) return 0;
  if (bits[4]) return 0;
  int par = 0; for (int i = 0; i < 17; i++) par += bits[i];
  if ((par & 1) ^ bits[17]) return 0;
  int rate = 0; for (int i = 0; i < 4; i++) rate |= ((int)bits[i]) << i;
  int ndbps;
  switch (rate) {
    case 11: *mcs = 0; ndbps = 24; break; case 15: *mcs = 1; ndbps = 36; break;
    case 10: *mcs = 2; ndbps = 48; break; case 14: *mcs = 3; ndbps = 72; break;
    case 9:  *mcs = 4; ndbps = 96; break; case 13: *mcs = 5; ndbps = 144; break;
    case 8:  *mcs = 6; ndbps = 192; break; case 12: *mcs = 7; ndbps = 216; break;
    default: *mcs = 0; ndbps = 24; break;
  }
  *len = 0; for (int i = 0; i < 12; i++) *len |= ((int)bits[i + 5]) << i;
  if (*len > 4095 || *len < 14) return 0;
  *nsamp = ((*len * 8 + 22 + ndbps - 1) / ndbps) * 80;
  return 1;
}

static cf rot(cf x, float th) { return cmul(x, C(cosf(th), sinf(th))); }

int c8po_signal_event(const c8p_sig_in* ev, const cf* raw_arena, c8p_pkt_in* pkt, cf* h, cf* out0, cf* out1)
{
  c8po_init();
  const cf* in1 = raw_arena + ev->raw_off[0];
  const cf* in2 = ev->n_ant == 2 ? raw_arena + ev->raw_off[1] : NULL;
  memset(pkt, 0, sizeof(*pkt));
  pkt->n_ant = ev->n_ant; pkt->snr = ev->snr; pkt->rssi = ev->rssi;
  pkt->cfo = ev->cfo_rad * 3183098.8618379068f;       /* signal_impl.cc:136 */
  for (int i = 0; i < 64; i++) h[i] = C(0, 0);
  if (ev->n_avail < 224) return 0;
  cf t1[64], t2[64], ts[64], f1[64], f2[64], fs[64];
  for (int i = 0; i < 64; i++) {                       /* signal_impl.cc:115-120 */
    t1[i] = rot(in1[8 + i], (float)(i + 8) * ev->cfo_rad);
    t2[i] = rot(in1[8 + 64 + i], (float)(i + 8 + 64) * ev->cfo_rad);
    ts[i] = rot(in1[8 + 144 + i], (float)(i + 8 + 144) * ev->cfo_rad);
  }
  c8po_fft64(t1, f1); c8po_fft64(t2, f2); c8po_fft64(ts, fs);
  float llr[48]; uint8_t bits[24];
  const int pil[4] = {7, 21, 43, 57};
  for (int k = 0; k < 4; k++) { int i = pil[k]; h[i] = cdivf(cadd(f1[i], f2[i]), LTF_L[i] * 2.0f); }
  cf ps = cconj(cadd(cadd(csub(cdiv(fs[7], h[7]), cdiv(fs[21], h[21])), cdiv(fs[43], h[43])), cdiv(fs[57], h[57])));
  float pa = cabsf_(ps);
  for (int i = 0; i < 64; i++) {
    if (SIG_DEMAP_NL[i] > -1) {
      h[i] = cdivf(cadd(f1[i], f2[i]), LTF_L[i] * 2.0f);
      llr[SIG_DEMAP_NL[i]] = cdivf(cmul(cdiv(fs[i], h[i]), ps), pa).re;
    }
  }
  c8po_sig_viterbi(llr, bits, 24);
  if (!bits[3] || bits[4]) return 0;
  int par = 0; for (int i = 0; i < 17; i++) par += bits[i];
  if ((par & 1) ^ bits[17]) return 0;
  int rate = 0; for (int i = 0; i < 4; i++) rate |= ((int)bits[i]) << i;
  int mcs, ndbps;
  switch (rate) {
    case 11: mcs = 0; ndbps = 24; break; case 15: mcs = 1; ndbps = 36; break;
    case 10: mcs = 2; ndbps = 48; break; case 14: mcs = 3; ndbps = 72; break;
    case 9:  mcs = 4; ndbps = 96; break; case 13: mcs = 5; ndbps = 144; break;
    case 8:  mcs = 6; ndbps = 192; break; case 12: mcs = 7; ndbps = 216; break;
    default: mcs = 0; ndbps = 24; break;
  }
  int len = 0; for (int i = 0; i < 12; i++) len |= ((int)bits[i + 5]) << i;
  if (len > 4095 || len < 14) return 0;
  const int nsamp = ((len * 8 + 22 + ndbps - 1) / ndbps) * 80;
  if ((uint32_t)(224 + nsamp) > ev->n_avail) {                /* the stream would have to wait for more items */
    pkt->lsig_mcs = mcs; pkt->lsig_len = len; pkt->lsig_nsamp = nsamp; pkt->rsv[0] = C8P_SIG_NEED_MORE;
    return 0;
  }
  for (int k = 0; k < nsamp; k++) {                            /* S_COPY signal_impl.cc:164-192 */
    const float th = (float)(k + 224) * ev->cfo_rad;
    out0[k] = rot(in1[224 + k], th);
    if (in2) out1[k] = rot(in2[224 + k], th);
  }
  for (int k = nsamp; k < nsamp + 320; k++) { out0[k] = C(0, 0); if (in2) out1[k] = C(0, 0); }
  pkt->lsig_mcs = mcs; pkt->lsig_len = len; pkt->lsig_nsamp = nsamp; pkt->n_items = (uint32_t)nsamp + 320u;
  return 1;
}

/* ------------------------------------------------------------------ demod */
static int is_null_nl(int i) { return i == 0 || (i >= 29 && i <= 35); }
static int is_null_l(int i) { return i == 0 || (i >= 27 && i <= 37); }
static int is_pilot(int i) { return i == 7 || i == 21 || i == 43 || i == 57; }

/* procNLSigDemodDeint cloud80211phy.cc:629-648 */
static void nl_sig_demod(const cf* s1, const cf* s2, const cf* h, float* llrht, float* llrvht)
{
  cf p1 = cconj(cadd(cadd(csub(cdiv(s1[7], h[7]), cdiv(s1[21], h[21])), cdiv(s1[43], h[43])), cdiv(s1[57], h[57])));
  cf p2 = cconj(cadd(cadd(csub(cdiv(s2[7], h[7]), cdiv(s2[21], h[21])), cdiv(s2[43], h[43])), cdiv(s2[57], h[57])));
  float a1 = cabsf_(p1), a2 = cabsf_(p2);
  for (int i = 0; i < 64; i++) {
    if (SIG_DEMAP_NL[i] > -1) {
      cf m1 = cdivf(cmul(cdiv(s1[i], h[i]), p1), a1);
      cf m2 = cdivf(cmul(cdiv(s2[i], h[i]), p2), a2);
      llrht[SIG_DEMAP_NL[i]] = m1.im;
      llrht[SIG_DEMAP_NL[i + 64]] = m2.im;
      llrvht[SIG_DEMAP_NL[i]] = m1.re;
      llrvht[SIG_DEMAP_NL[i + 64]] = m2.im;
    }
  }
}

/* SIG-B decode + sssnr tail shared by demod_impl.cc:483-504 and demod2_impl.cc:710-757 */
static void sigb_finish(const float* inted, uint8_t* bits26, cf* q0, cf* q1, int nss, float* sssnr0, float* sssnr1)
{
  float coded[52]; uint8_t enc[52], intd[52];
  for (int i = 0; i < 52; i++) coded[DEINT_SIGB[i]] = inted[i];
  c8po_sig_viterbi(coded, bits26, 26);
  c8po_bcc_encode(bits26, enc, 26);
  for (int i = 0; i < 52; i++) intd[INTL_SIGB[i]] = enc[i];
  double n0 = 0.0, n1 = 0.0;
  for (int i = 0; i < 52; i++) {
    cf ref = intd[i] ? C(1.0f, 0.0f) : C(-1.0f, 0.0f);
    q0[i] = csub(q0[i], ref);
    n0 += (double)(q0[i].re * q0[i].re + q0[i].im * q0[i].im);
    if (nss == 2) {
      q1[i] = csub(q1[i], ref);
      n1 += (double)(q1[i].re * q1[i].re + q1[i].im * q1[i].im);
    }
  }
  *sssnr0 = (float)(log10(52.0 / n0) * 10.0);
  if (nss == 2) *sssnr1 = (float)(log10(52.0 / n1) * 10.0);
}

typedef struct {   /* 2x2 state of demod2_impl.h:95-105 */
  cf H[64][4], INV[64][4];
  cf pil1[4], pil2[4];
} mimo_t;

/* demod2_impl::nonLegacyChanEstimate nSS==2 branch  demod2_impl.cc:372-463 */
static void chan_est_2x2(const cf* f11, const cf* f21, const cf* f12, const cf* f22, int vht, mimo_t* M)
{
  for (int i = 0; i < 64; i++) {
    if (is_null_nl(i)) continue;
    float l2 = LTF_NL[i] * 0.5f;
    M->H[i][0] = cmulf(csub(f11[i], f12[i]), l2);
    M->H[i][1] = cmulf(csub(f21[i], f22[i]), l2);
    M->H[i][2] = cmulf(cadd(f11[i], f12[i]), l2);
    M->H[i][3] = cmulf(cadd(f21[i], f22[i]), l2);
  }
  if (vht) {
    const int pil[4] = {7, 21, 43, 57};
    for (int k = 0; k < 4; k++) for (int c = 0; c < 4; c++)
      M->H[pil[k]][c] = cdivf(cadd(M->H[pil[k] - 1][c], M->H[pil[k] + 1][c]), 2.0f);
  }
  for (int i = 0; i < 64; i++) {
    if (is_null_nl(i)) continue;
    cf* H = M->H[i];
    cf a = cadd(cmul(H[0], cconj(H[0])), cmul(H[1], cconj(H[1])));
    cf b = cadd(cmul(H[0], cconj(H[2])), cmul(H[1], cconj(H[3])));
    cf c = cadd(cmul(H[2], cconj(H[0])), cmul(H[3], cconj(H[1])));
    cf d = cadd(cmul(H[2], cconj(H[2])), cmul(H[3], cconj(H[3])));
    cf k = cdiv(C(1.0f, 0.0f), csub(cmul(a, d), cmul(b, c)));
    M->INV[i][0] = cmul(k, d);
    M->INV[i][1] = cmul(cneg(k), b);
    M->INV[i][2] = cmul(cneg(k), c);
    M->INV[i][3] = cmul(k, a);
  }
  const int pil[4] = {7, 21, 43, 57};
  const int slot[4] = {2, 3, 0, 1};
  for (int k = 0; k < 4; k++) {
    int i = pil[k];
    cf t1 = cadd(cmul(f11[i], cconj(M->H[i][0])), cmul(f21[i], cconj(M->H[i][1])));
    cf t2 = cadd(cmul(f11[i], cconj(M->H[i][2])), cmul(f21[i], cconj(M->H[i][3])));
    cf s1 = cadd(cmul(t1, M->INV[i][0]), cmul(t2, M->INV[i][2]));
    cf s2 = cadd(cmul(t1, M->INV[i][1]), cmul(t2, M->INV[i][3]));
    if (i == 57) { s1 = cneg(s1); s2 = cneg(s2); }
    M->pil1[slot[k]] = cconj(s1);
    M->pil2[slot[k]] = cconj(s2);
  }
}

static void zf_2x2(const cf* y1, const cf* y2, const mimo_t* M, cf* s1, cf* s2)   /* demod2_impl.cc:509-519 */
{
  for (int i = 0; i < 64; i++) {
    if (is_null_nl(i)) continue;
    cf t1 = cadd(cmul(y1[i], cconj(M->H[i][0])), cmul(y2[i], cconj(M->H[i][1])));
    cf t2 = cadd(cmul(y1[i], cconj(M->H[i][2])), cmul(y2[i], cconj(M->H[i][3])));
    s1[i] = cadd(cmul(t1, M->INV[i][0]), cmul(t2, M->INV[i][2]));
    s2[i] = cadd(cmul(t1, M->INV[i][1]), cmul(t2, M->INV[i][3]));
  }
}

int c8po_demod(const c8p_pkt_in* pkt, const cf* iq_arena, const cf* chan_arena,
               c8p_pkt_tags* tg, float* llr_out, cf* mu2x1chan)
{
  c8po_init();
  const cf* in1 = iq_arena + pkt->iq_off[0];
  const cf* in2 = pkt->n_ant == 2 ? iq_arena + pkt->iq_off[1] : NULL;
  const cf* HL = chan_arena + pkt->chan_off;
  const int two = pkt->n_ant == 2;
  const int nitems = (int)pkt->n_items;
  mod_t m; memset(&m, 0, sizeof(m));
  cf H[64]; memset(H, 0, sizeof(H));
  mimo_t* M = NULL;
  float pil[4], pil2[4];
  int pilotP = 0, data_off = 0, trellis = 0;
  float sssnr0 = 0.f, sssnr1 = 0.f;
  int is_ndp = 0;
  int fmt = C8P_F_L;

  memset(tg, 0, sizeof(*tg));
  tg->cfo = pkt->cfo; tg->snr = pkt->snr; tg->rssi = pkt->rssi; tg->seq = pkt->seq;
  tg->status = C8P_ST_REJ_SHORT;
  if (two) M = (mimo_t*)calloc(1, sizeof(mimo_t));

  /* DEMOD_S_RDTAG :93-100, DEMOD_S_FORMAT :106-148 */
  if (!(pkt->lsig_mcs > 0)) {
    if (nitems < 160) goto done;
    cf f1[64], f2[64];
    float llrht[96], llrvht[96];
    uint8_t bits[48];
    c8po_fft64(in1 + 8, f1); c8po_fft64(in1 + 88, f2);
    nl_sig_demod(f1, f2, HL, llrht, llrvht);
    c8po_sig_viterbi(llrvht, bits, 48);
    if (check_vht_a(bits)) { parse_vht_a(bits, &m); fmt = C8P_F_VHT; }
    else {
      c8po_sig_viterbi(llrht, bits, 48);
      if (check_ht(bits)) { parse_ht(bits, &m); fmt = C8P_F_HT; }
    }
  }

  if (fmt == C8P_F_VHT) {
    /* DEMOD_S_VHT demod_impl.cc:150-178 / demod2_impl.cc:150-180 */
    if (!m.valid && !m.sumu) { tg->status = C8P_ST_REJ_UNSUP; goto done; }
    if (m.nSS > 2) { tg->status = C8P_ST_REJ_VHT; goto done; }
    if (nitems < 160 + 80 + m.nLTF * 80 + 80) goto done;
    const cf* l1 = in1 + 240;
    cf f1[64], f2[64], fb[64], sig[64], q0[52], q1[52];
    float inted[52]; uint8_t bbits[26];
    int nss_eff = 1;
    if (!two) {
      /* demod_impl::nonLegacyChanEstimate :344-411 */
      if (m.sumu) {
        c8po_fft64(l1 + 8, f1); c8po_fft64(l1 + 88, f2);
        for (int i = 0; i < 64; i++) {
          if (is_null_nl(i)) continue;
          if (pkt->mupos == 0) H[i] = cdivf(csub(f1[i], f2[i]), LTF_NL[i] * 2.0f);
          else H[i] = cdivf(cadd(cdivf(f1[i], LTF_NL[i]), cdivf(f2[i], LTF_NL_VHT22[i])), 2.0f);
        }
      } else {
        if (!(m.nSS == 1 && m.nLTF == 1)) {
          memcpy(&mu2x1chan[0], l1 + 8, sizeof(cf) * 64);
          memcpy(&mu2x1chan[64], l1 + 88, sizeof(cf) * 64);
        }
        c8po_fft64(l1 + 8, f1);
        for (int i = 0; i < 64; i++) if (!is_null_nl(i)) H[i] = cdivf(f1[i], LTF_NL[i]);
      }
      /* vhtSigBDemod :449-505 */
      c8po_fft64(l1 + m.nLTF * 80 + 8, fb);
      for (int i = 0; i < 64; i++) if (!is_null_nl(i)) sig[i] = cdiv(fb[i], H[i]);
      cf ps = cconj(cadd(cadd(csub(sig[7], sig[21]), sig[43]), sig[57]));
      float pa = cabsf_(ps);
      int j = 26;
      for (int i = 0; i < 64; i++) {
        if (is_null_nl(i) || is_pilot(i)) continue;
        q0[j] = cdivf(cmul(sig[i], ps), pa);
        inted[j] = q0[j].re;
        j++; if (j >= 52) j = 0;
      }
    } else {
      /* demod2: MU and nSS==1 with nLTF!=1 leave H stale */
      if (m.sumu) { tg->status = C8P_ST_REJ_UNSUP; goto done; }
      const cf* l2 = in2 + 240;
      if (m.nSS == 1) {
        c8po_fft64(l1 + 8, f1);
        for (int i = 0; i < 64; i++) if (!is_null_nl(i)) H[i] = cdivf(f1[i], LTF_NL[i]);
        c8po_fft64(l1 + m.nLTF * 80 + 8, fb);
        for (int i = 0; i < 64; i++) if (!is_null_nl(i)) sig[i] = cdiv(fb[i], H[i]);
        cf ps = cconj(cadd(cadd(csub(sig[7], sig[21]), sig[43]), sig[57]));
        float pa = cabsf_(ps);
        int j = 26;
        for (int i = 0; i < 64; i++) {
          if (is_null_nl(i) || is_pilot(i)) continue;
          q0[j] = cdivf(cmul(sig[i], ps), pa);
          inted[j] = q0[j].re;
          j++; if (j >= 52) j = 0;
        }
      } else {
        cf f11[64], f21[64], f12[64], f22[64], y1[64], y2[64], s1[64], s2[64];
        c8po_fft64(l1 + 8, f11); c8po_fft64(l2 + 8, f21);
        c8po_fft64(l1 + 88, f12); c8po_fft64(l2 + 88, f22);
        chan_est_2x2(f11, f21, f12, f22, 1, M);
        /* vhtSigBDemod nSS==2 demod2_impl.cc:663-705 */
        c8po_fft64(l1 + m.nLTF * 80 + 8, y1); c8po_fft64(l2 + m.nLTF * 80 + 8, y2);
        zf_2x2(y1, y2, M, s1, s2);
        cf acc = cmul(s1[7], M->pil1[2]);
        acc = csub(acc, cmul(s1[21], M->pil1[3]));
        acc = cadd(acc, cmul(s1[43], M->pil1[0]));
        acc = cadd(acc, cmul(s1[57], M->pil1[1]));
        acc = cadd(acc, cmul(s2[7], M->pil2[2]));
        acc = csub(acc, cmul(s2[21], M->pil2[3]));
        acc = cadd(acc, cmul(s2[43], M->pil2[0]));
        acc = cadd(acc, cmul(s2[57], M->pil2[1]));
        cf ps = cconj(acc);
        float pa = cabsf_(ps);
        int j = 26;
        for (int i = 0; i < 64; i++) {
          if (is_null_nl(i) || is_pilot(i)) continue;
          q0[j] = cdivf(cmul(s1[i], ps), pa);
          q1[j] = cdivf(cmul(s2[i], ps), pa);
          inted[j] = (q0[j].re + q1[j].re) / 2.0f;
          j++; if (j >= 52) j = 0;
        }
        nss_eff = 2;
      }
    }
    sigb_finish(inted, bbits, q0, q1, nss_eff, &sssnr0, &sssnr1);
    parse_vht_b(bbits, &m);
    if (!m.valid) { tg->status = C8P_ST_REJ_UNSUP; goto done; }
    int nSymL = (pkt->lsig_len * 8 + 22 + 23) / 24;
    int len_ok = two ? (m.len > 0) : (m.len >= 0);
    if (!(len_ok && m.len <= 4095 && m.nSS <= 2 && (nSymL * 80) >= (m.nSym * m.nSymSamp + 160 + 80 + m.nLTF * 80 + 80))) {
      tg->status = C8P_ST_REJ_VHT; goto done;
    }
    trellis = m.nSym * m.nDBPS;
    memcpy(pil, PILOT_L4, sizeof(pil)); memcpy(pil2, PILOT_L4, sizeof(pil2));
    pilotP = 4;
    data_off = 160 + 80 + m.nLTF * 80 + 80;
    is_ndp = (m.nSym == 0);
    if (!two && m.nSS == 2 && !is_ndp) { tg->status = C8P_ST_REJ_UNSUP; goto done; }
  } else if (fmt == C8P_F_HT) {
    /* DEMOD_S_HT demod_impl.cc:180-205 / demod2_impl.cc:182-214 */
    if (m.nSS > 2) { tg->status = C8P_ST_REJ_HT; goto done; }
    if (nitems < 160 + 80 + m.nLTF * 80) goto done;
    const cf* l1 = in1 + 240;
    cf f1[64];
    if (!two || m.nSS == 1) {
      c8po_fft64(l1 + 8, f1);
      for (int i = 0; i < 64; i++) if (!is_null_nl(i)) H[i] = cdivf(f1[i], LTF_NL[i]);
    } else {
      const cf* l2 = in2 + 240;
      cf f11[64], f21[64], f12[64], f22[64];
      c8po_fft64(l1 + 8, f11); c8po_fft64(l2 + 8, f21);
      c8po_fft64(l1 + 88, f12); c8po_fft64(l2 + 88, f22);
      chan_est_2x2(f11, f21, f12, f22, 0, M);
    }
    int nSymL = (pkt->lsig_len * 8 + 22 + 23) / 24;
    if (!(m.len > 0 && m.len <= 4095 && m.nSS <= 2 && (nSymL * 80) >= (m.nSym * m.nSymSamp + 160 + 80 + m.nLTF * 80))) {
      tg->status = C8P_ST_REJ_HT; goto done;
    }
    if (!two && m.nSS == 2) { tg->status = C8P_ST_REJ_UNSUP; goto done; }
    trellis = m.len * 8 + 22;
    if (two && m.nSS == 2) memcpy(pil, PILOT_HT21, sizeof(pil)); else memcpy(pil, PILOT_L4, sizeof(pil));
    memcpy(pil2, PILOT_HT22, sizeof(pil2));
    pilotP = 3;
    data_off = 160 + 80 + m.nLTF * 80;
  } else {
    /* DEMOD_S_LEGACY :207-219 */
    parse_legacy(pkt->lsig_mcs, pkt->lsig_len, &m);
    if (!m.valid) { tg->status = C8P_ST_REJ_UNSUP; goto done; }
    trellis = m.len * 8 + 22;
    memcpy(pil, PILOT_L4, sizeof(pil));
    pilotP = 1;
    data_off = 0;
  }

  /* the DEMOD loop needs o1 + nSymSamp < d_nProc for every symbol (:283) */
  if (!is_ndp && data_off + m.nSym * m.nSymSamp >= nitems) { tg->status = C8P_ST_REJ_SHORT; goto done; }

  /* DEMOD_S_WRTAG :221-277 */
  tg->status = C8P_ST_OK;
  tg->format = m.format; tg->mcs = m.mcs; tg->len = m.len; tg->cr = m.cr; tg->ampdu = m.ampdu;
  tg->trellis = trellis; tg->total = is_ndp ? C8P_NDP_TOTAL : m.nSym * m.nCBPS;
  tg->nsym = m.nSym; tg->nss = m.nSS; tg->nltf = m.nLTF; tg->ncbps = m.nCBPS; tg->ncbpss = m.nCBPSS;
  tg->ndbps = m.nDBPS; tg->nbpscs = m.nBPSCS; tg->nsymsamp = m.nSymSamp; tg->sumu = m.sumu;
  tg->is_ndp = is_ndp; tg->data_off = data_off;
  tg->sssnr0 = sssnr0; tg->sssnr1 = sssnr1;
  if (is_ndp) {
    for (int i = 0; i < C8P_NDP_TOTAL; i++) llr_out[i] = 0.0f;   /* unwritten in the reference (:265-271) */
    goto done;
  }

  /* DEMOD_S_DEMOD :279-314 */
  {
    const int mi = nbpsc_index(m.nBPSCS);
    for (int s = 0; s < m.nSym; s++) {
      const cf* x1 = in1 + data_off + s * m.nSymSamp;
      float* out = llr_out + (size_t)s * m.nCBPS;
      cf f1[64], sig[64], q[2][52];
      float li[2][C8P_MAX_N_CBPSS], sp[2][C8P_MAX_N_CBPSS];
      float P = PILOT_P[(pilotP + s) % 127];
      c8po_fft64(x1 + 8, f1);
      if (m.format == C8P_F_L) {
        /* legacyChanUpdate :507-539 */
        for (int i = 0; i < 64; i++) if (!is_null_l(i)) sig[i] = cdiv(f1[i], HL[i]);
        cf acc = cmulf(cmulf(sig[7], pil[2]), P);
        acc = cadd(acc, cmulf(cmulf(sig[21], pil[3]), P));
        acc = cadd(acc, cmulf(cmulf(sig[43], pil[0]), P));
        acc = cadd(acc, cmulf(cmulf(sig[57], pil[1]), P));
        cf ps = cconj(acc);
        float pa = cabsf_(ps);
        int j = 24;
        for (int i = 0; i < 64; i++) {
          if (is_null_l(i) || is_pilot(i)) continue;
          q[0][j] = cdivf(cmul(sig[i], ps), pa);
          j++; if (j >= 48) j = 0;
        }
        c8po_qam_to_llr(q[0], li[0], 48, m.nBPSCS);
        for (int i = 0; i < m.nCBPS; i++) out[DEINT_L[mi][i]] = li[0][i];
      } else if (!(two && m.nSS == 2)) {
        /* nonLegacyChanUpdate :413-447 (== demod2 nSS==1 branches) */
        float p[4];
        for (int k = 0; k < 4; k++) p[k] = pil[(k + s) % 4];
        for (int i = 0; i < 64; i++) if (!is_null_nl(i)) sig[i] = cdiv(f1[i], H[i]);
        cf acc = cmulf(cmulf(sig[7], p[2]), P);
        acc = cadd(acc, cmulf(cmulf(sig[21], p[3]), P));
        acc = cadd(acc, cmulf(cmulf(sig[43], p[0]), P));
        acc = cadd(acc, cmulf(cmulf(sig[57], p[1]), P));
        cf ps = cconj(acc);
        float pa = cabsf_(ps);
        int j = 26;
        for (int i = 0; i < 64; i++) {
          if (is_null_nl(i) || is_pilot(i)) continue;
          q[0][j] = cdivf(cmul(sig[i], ps), pa);
          j++; if (j >= 52) j = 0;
        }
        c8po_qam_to_llr(q[0], li[0], 52, m.nBPSCS);
        for (int i = 0; i < m.nCBPSS; i++) out[DEINT_NL[mi][i]] = li[0][i];
      } else {
        /* htChanUpdate :471-551 / vhtChanUpdate :553-630, nSS==2 */
        const cf* x2 = in2 + data_off + s * m.nSymSamp;
        cf f2[64], s1[64], s2[64];
        float p[4], pb[4];
        for (int k = 0; k < 4; k++) { p[k] = pil[(k + s) % 4]; pb[k] = (m.format == C8P_F_VHT) ? p[k] : pil2[(k + s) % 4]; }
        c8po_fft64(x2 + 8, f2);
        zf_2x2(f1, f2, M, s1, s2);
        cf acc = cmul(cmulf(cmulf(s1[7], p[2]), P), M->pil1[2]);
        acc = cadd(acc, cmul(cmulf(cmulf(s1[21], p[3]), P), M->pil1[3]));
        acc = cadd(acc, cmul(cmulf(cmulf(s1[43], p[0]), P), M->pil1[0]));
        acc = cadd(acc, cmul(cmulf(cmulf(s1[57], p[1]), P), M->pil1[1]));
        acc = cadd(acc, cmul(cmulf(cmulf(s2[7], pb[2]), P), M->pil2[2]));
        acc = cadd(acc, cmul(cmulf(cmulf(s2[21], pb[3]), P), M->pil2[3]));
        acc = cadd(acc, cmul(cmulf(cmulf(s2[43], pb[0]), P), M->pil2[0]));
        acc = cadd(acc, cmul(cmulf(cmulf(s2[57], pb[1]), P), M->pil2[1]));
        cf ps = cconj(acc);
        float pa = cabsf_(ps);
        int j = 26;
        for (int i = 0; i < 64; i++) {
          if (is_null_nl(i) || is_pilot(i)) continue;
          q[0][j] = cdivf(cmul(s1[i], ps), pa);
          q[1][j] = cdivf(cmul(s2[i], ps), pa);
          j++; if (j >= 52) j = 0;
        }
        c8po_qam_to_llr(q[0], li[0], 52, m.nBPSCS);
        c8po_qam_to_llr(q[1], li[1], 52, m.nBPSCS);
        for (int i = 0; i < m.nCBPSS; i++) { sp[0][DEINT_NL[mi][i]] = li[0][i]; sp[1][DEINT_NL2[mi][i]] = li[1][i]; }
        /* procSymDepasNL cloud80211phy.cc:2442-2451 */
        int sz = m.nBPSCS / 2 > 1 ? m.nBPSCS / 2 : 1;
        for (int i = 0; i < m.nCBPSS / sz; i++) {
          memcpy(&out[i * 2 * sz], &sp[0][i * sz], sizeof(float) * (size_t)sz);
          memcpy(&out[(i * 2 + 1) * sz], &sp[1][i * sz], sizeof(float) * (size_t)sz);
        }
      }
    }
  }
done:
  if (M) free(M);
  return tg->status;
}

/* ------------------------------------------------------------------ decode */
static int pdu_put(uint8_t* pdu, uint32_t cap, uint32_t* used, const uint8_t* blob, int n)
{
  if (*used + 2u + (uint32_t)n > cap) return 0;
  pdu[*used] = (uint8_t)(n & 0xff); pdu[*used + 1] = (uint8_t)(n >> 8);
  memcpy(pdu + *used + 2, blob, (size_t)n);
  *used += 2u + (uint32_t)n;
  return 1;
}

int c8po_decode(const c8p_pkt_tags* tg, const float* llr, const cf* mu2x1chan,
                uint8_t* sbits_o, uint8_t* bits_o, uint8_t* pdu, uint32_t pdu_cap, c8p_pkt_out* out)
{
  c8po_init();
  memset(out, 0, sizeof(*out));
  if (tg->status != C8P_ST_OK) return 0;
  /* decode_impl.cc:93-97 */
  if (tg->len > C8P_DECODE_B_MAX || tg->trellis > C8P_DECODE_T_MAX) return 0;
  uint32_t used = 0;
  if (tg->trellis == 0) {
    /* NDP channel report :100-121 */
    uint8_t blob[C8P_NDP_BLOB_BYTES];
    int tmpLen = (int)sizeof(float) * 256;
    blob[0] = C8P_F_VHT_CHAN; blob[1] = (uint8_t)(tmpLen % 256); blob[2] = (uint8_t)(tmpLen / 256);
    for (int i = 0; i < 128; i++) {
      float re = mu2x1chan ? mu2x1chan[i].re : 0.f, im = mu2x1chan ? mu2x1chan[i].im : 0.f;
      memcpy(&blob[3 + 8 * i], &re, 4); memcpy(&blob[3 + 8 * i + 4], &im, 4);
    }
    if (pdu_put(pdu, pdu_cap, &used, blob, tmpLen + 3)) out->n_pdu++;
    out->pdu_bytes = used;
    return out->n_pdu;
  }
  const int T = tg->trellis;
  uint8_t* sb = sbits_o ? sbits_o : (uint8_t*)malloc((size_t)T);
  uint8_t* ub = bits_o ? bits_o : (uint8_t*)malloc((size_t)T + 8);
  int usedllr = c8po_viterbi(llr, tg->total, T, tg->cr, sb);
  if (usedllr < 0) goto fin;    /* stream ran dry: the reference would wait forever */
  c8po_descramble(sb, T, ub);
  /* packetAssemble :325-520 */
  {
    uint8_t* pkt = (uint8_t*)malloc(16384 + 8);
    int procd = 0;
    if (tg->format == C8P_F_VHT) {
      procd += 16;
      if (procd < T) {
        const uint8_t* bp = &ub[16];
        int eof, tmpLen = 0;
        while (1) {
          procd += 32;
          if (procd > T) break;
          eof = bp[0];
          tmpLen |= (((int)bp[2]) << 12);
          tmpLen |= (((int)bp[3]) << 13);
          for (int i = 0; i < 12; i++) tmpLen |= (((int)bp[4 + i]) << i);
          procd += ((tmpLen / 4 + ((tmpLen % 4) != 0)) * 4 * 8);
          if (procd > T) break;
          bp += 32;
          pkt[0] = (uint8_t)tg->format; pkt[1] = (uint8_t)(tmpLen % 256); pkt[2] = (uint8_t)(tmpLen / 256);
          for (int i = 0; i < tmpLen; i++) {
            pkt[i + 3] = 0;
            for (int j = 0; j < 8; j++) pkt[i + 3] |= (uint8_t)(bp[i * 8 + j] << j);
          }
          bp += ((tmpLen / 4 + ((tmpLen % 4) != 0)) * 4 * 8);
          if (c8po_crc32(pkt + 3, tmpLen) != 558161692u) out->n_crc_fail++;
          else {
            pkt[tmpLen + 3] = (uint8_t)tg->mcs;
            tmpLen += 4;                                 /* :414 — and the next delimiter's length is OR-ed into THIS value (:346-351) */
            if (pdu_put(pdu, pdu_cap, &used, pkt, tmpLen)) out->n_pdu++;
          }
          if (eof) break;
        }
      }
    } else if (!tg->ampdu) {
      const uint8_t* bp = &ub[16];
      int len = tg->len;
      pkt[0] = (uint8_t)tg->format; pkt[1] = (uint8_t)(len % 256); pkt[2] = (uint8_t)(len / 256);
      for (int i = 0; i < len; i++) {
        pkt[i + 3] = 0;
        for (int j = 0; j < 8; j++) pkt[i + 3] |= (uint8_t)(bp[i * 8 + j] << j);
      }
      if (c8po_crc32(pkt + 3, len) != 558161692u) out->n_crc_fail++;
      else {
        pkt[3 + len] = (uint8_t)tg->mcs;
        if (pdu_put(pdu, pdu_cap, &used, pkt, len + 4)) out->n_pdu++;
      }
    }
    free(pkt);
  }
fin:
  out->pdu_bytes = used;
  if (!sbits_o) free(sb);
  if (!bits_o) free(ub);
  return out->n_pdu;
}

long c8po_rx_batch(const c8p_pkt_in* pkts, int n, const cf* iq, const cf* chan,
                   c8p_pkt_tags* tags, c8p_pkt_out* outs, uint8_t* pdu, uint32_t pdu_stride)
{
  long npdu = 0;
  float* llr = (float*)malloc(sizeof(float) * 832 * 512);
  cf mu[128];
  for (int p = 0; p < n; p++) {
    memset(mu, 0, sizeof(mu));
    c8po_demod(&pkts[p], iq, chan, &tags[p], llr, mu);
    tags[p].llr_off = 0;
    c8po_decode(&tags[p], llr, mu, NULL, NULL, pdu + (size_t)p * pdu_stride, pdu_stride, &outs[p]);
    outs[p].pdu_off = (uint64_t)p * pdu_stride;
    npdu += outs[p].n_pdu;
  }
  free(llr);
  return npdu;
}
