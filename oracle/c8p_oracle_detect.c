/*
 * c8p_oracle_detect.c — CPU restatement of the packet-detection front end:
 *   examples/presiso.grc  (a hier block of stock GNU Radio blocks)  ->  lib/trigger_impl.cc  ->  lib/sync_impl.cc
 *
 * TEST INFRASTRUCTURE ONLY (see c8p_oracle.h).
 *
 * Third-party arithmetic that is NOT in /root/reference: presiso.grc wires GNU Radio 3.10 gr-blocks
 * (blocks_delay 16, blocks_multiply_conjugate_cc, blocks_moving_average_cc length 48 scale 1 max_iter 4000,
 * blocks_complex_to_mag, blocks_complex_to_mag_squared, blocks_moving_average_ff length 64, blocks_divide_ff;
 * examples/presiso.grc:37-229, version unpinned by the reference) over VOLK kernels.  Their published algorithm is
 * restated here:   p[n] = x[n-16] * conj(x[n])      (x[k] = 0 for k < 0: delay and history start from zeros)
 *                  P[n] = sum_{k=n-47..n} p[k]        E[n] = sum_{k=n-63..n} |x[k]|^2
 *                  ac[n] = sqrtf(P.re^2 + P.im^2) / E[n]          conj[n] = P[n]
 * GNU Radio's moving average keeps a running float sum that is re-seeded at every work() call (at most max_iter
 * items per call), so its low-order bits depend on how the scheduler chunks the stream: the reference itself is not
 * bit-reproducible at this boundary.  The oracle pins the windows to a fresh oldest-first float sum per sample.
 * "parity unpinned" for ac/conj (tolerance 2e-6 relative in the tests); the trigger / sync state machines below are
 * restated literally from the reference's own sources and are exact functions of (ac, conj, samples).
 */
#include <math.h>
#include <string.h>

#include "c8p_oracle.h"

typedef c8p_c32 cf;

void c8po_presiso(const cf* x, long n, float* ac, cf* conj)
{
  for (long i = 0; i < n; i++) {
    float pr = 0.0f, pi = 0.0f, e = 0.0f;
    for (long k = i - 47; k <= i; k++) {
      if (k < 16) continue;                       /* x[k-16] is a delay-line zero */
      const cf a = x[k - 16], b = x[k];           /* volk_32fc_x2_multiply_conjugate_32fc: a * conj(b) */
      pr += a.re * b.re + a.im * b.im;
      pi += a.im * b.re - a.re * b.im;
    }
    for (long k = i - 63; k <= i; k++) {
      if (k < 0) continue;
      e += x[k].re * x[k].re + x[k].im * x[k].im; /* volk_32fc_magnitude_squared_32f */
    }
    conj[i].re = pr; conj[i].im = pi;
    ac[i] = sqrtf(pr * pr + pi * pi) / e;         /* volk_32fc_magnitude_32f, blocks_divide_ff */
  }
}

/* trigger_impl::general_work (lib/trigger_impl.cc:59-117) over a whole stream from the constructor's state */
void c8po_trigger(const float* ac, long n, uint8_t* trig)
{
  int nPlateau = 0, fPlateau = 0, fPlateauEnd = 0, countDown = 0;
  float conjAc = 0.0f;
  for (long i = 0; i < n; i++) {
    trig[i] = 0;
    if (ac[i] > 0.3f) {
      nPlateau++;
      if (ac[i] > conjAc) { conjAc = ac[i]; trig[i] |= 0x02; }
      if (nPlateau > 20 && (fPlateau + fPlateauEnd) == 0) { fPlateau = 1; fPlateauEnd = 1; countDown = 80; }
    } else {
      nPlateau = 0; fPlateauEnd = 0; conjAc = 0.0f;
    }
    if (fPlateau) {
      countDown--;
      if (countDown == 0) { fPlateau = 0; trig[i] |= 0x01; }
    }
  }
}

#define SYNC_MAX_BUF_LEN 240
#define SYNC_MAX_RES_LEN (SYNC_MAX_BUF_LEN - 128 - 1)   /* lib/sync_impl.h:29-30 */

static float cabs_f(cf a) { return hypotf(a.re, a.im); }   /* std::abs(std::complex<float>) */

/* sync_impl::ltf_autoCorrelation lib/sync_impl.cc:150-175 */
static void ltf_autocorr(const cf* sig, float* ac, float* pwr)
{
  cf ms = {0.0f, 0.0f};
  float s1 = 0.0f, s2 = 0.0f;
  for (int i = 0; i < 64; i++) {
    ms.re += sig[i].re * sig[i + 64].re + sig[i].im * sig[i + 64].im;    /* sig[i] * conj(sig[i+64]) */
    ms.im += sig[i].im * sig[i + 64].re - sig[i].re * sig[i + 64].im;
    s1 += cabs_f(sig[i]) * cabs_f(sig[i]);
    s2 += cabs_f(sig[i + 64]) * cabs_f(sig[i + 64]);
  }
  for (int i = 0; i < SYNC_MAX_RES_LEN; i++) {
    ac[i] = cabs_f(ms) / sqrtf(s1) / sqrtf(s2);
    pwr[i] = s1;
    ms.re -= sig[i].re * sig[i + 64].re + sig[i].im * sig[i + 64].im;
    ms.im -= sig[i].im * sig[i + 64].re - sig[i].re * sig[i + 64].im;
    s1 -= cabs_f(sig[i]) * cabs_f(sig[i]);
    s2 -= cabs_f(sig[i + 64]) * cabs_f(sig[i + 64]);
    ms.re += sig[i + 64].re * sig[i + 128].re + sig[i + 64].im * sig[i + 128].im;
    ms.im += sig[i + 64].im * sig[i + 128].re - sig[i + 64].re * sig[i + 128].im;
    s1 += cabs_f(sig[i + 64]) * cabs_f(sig[i + 64]);
    s2 += cabs_f(sig[i + 128]) * cabs_f(sig[i + 128]);
  }
}

/* sync_impl::ltf_cfo lib/sync_impl.cc:177-192 */
static float ltf_cfo(const cf* sig, cf conjAvg)
{
  cf tmp[128], sum = {0.0f, 0.0f};
  const float stf = atan2f(conjAvg.im, conjAvg.re) / 16.0f;
  for (int i = 0; i < 128; i++) {
    const float c = cosf(i * stf), s = sinf(i * stf);
    tmp[i].re = sig[i].re * c - sig[i].im * s;
    tmp[i].im = sig[i].re * s + sig[i].im * c;
  }
  for (int i = 0; i < 64; i++) {
    sum.re += tmp[i].re * tmp[i + 64].re + tmp[i].im * tmp[i + 64].im;
    sum.im += tmp[i].im * tmp[i + 64].re - tmp[i].re * tmp[i + 64].im;
  }
  const float ltf = atan2f(sum.im / 64.0f, sum.re / 64.0f) / 64.0f;
  return stf + ltf;
}

/* sync_impl::general_work (lib/sync_impl.cc:62-148) over a whole stream: IDLE scans the trigger bytes (bit 0x02 latches
 * conj, bit 0x01 starts SYNC), SYNC needs 240 items from the trigger item on, tags one sync item and consumes 111.
 * d_conjMultiAvg is not initialised by the reference's constructor; the oracle starts it at 0.
 * Returns the number of events written (at most cap); sync_bytes (may be NULL) gets the block's output stream. */
long c8po_sync(const uint8_t* trig, const cf* conj, const cf* sig, long n, c8p_sync_evt* evts, long cap, uint8_t* sync_bytes)
{
  long pos = 0, nev = 0;
  cf conjAvg = {0.0f, 0.0f};
  if (sync_bytes) memset(sync_bytes, 0, (size_t)n);
  while (pos < n) {
    long i = pos;
    int hit = 0;
    for (; i < n; i++) {
      if (trig[i] & 0x01) { hit = 1; break; }
      else if (trig[i] & 0x02) conjAvg = conj[i];
    }
    pos = i;
    if (!hit) break;
    if (n - pos < SYNC_MAX_BUF_LEN) break;           /* the block would wait for more input */
    float ac[SYNC_MAX_RES_LEN], pwr[SYNC_MAX_RES_LEN];
    ltf_autocorr(sig + pos, ac, pwr);
    int mx = 0;
    for (int j = 1; j < SYNC_MAX_RES_LEN; j++) if (ac[j] > ac[mx]) mx = j;   /* std::max_element: first maximum */
    if (ac[mx] > 0.5) {
      const float thr = ac[mx] * 0.8;                 /* d_maxAc = *d_maxAcP * 0.8 (double product stored as float) */
      int l = mx, r = mx;
      for (int j = mx; j >= 0; j--) if (ac[j] < thr) { l = j; break; }
      for (int j = mx; j < SYNC_MAX_RES_LEN; j++) if (ac[j] < thr) { r = j; break; }
      const int m = (l + r) / 2;
      if (nev < cap) {
        const double d = (double)ac[mx];
        evts[nev].index = (uint64_t)(pos + m);
        evts[nev].rad = ltf_cfo(sig + pos + m, conjAvg);
        evts[nev].snr = (float)(10.0 * log10(d / (1.0 - d)));
        evts[nev].rssi = pwr[mx] / 64.0f;
        evts[nev].trig = (uint32_t)pos;
      }
      nev++;
      if (sync_bytes) sync_bytes[pos + m] = 0x01;
    }
    pos += SYNC_MAX_RES_LEN;
  }
  return nev;
}
