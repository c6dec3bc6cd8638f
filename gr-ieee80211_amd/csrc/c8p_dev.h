// c8p_dev.h — device-side structures and arithmetic helpers shared by the gfx950 kernels.
//
// Arithmetic contract (DESIGN.md "Arithmetic model"): float32 everywhere, no FMA contraction
// (-ffp-contract=off), complex*complex as (ac-bd, ad+bc), complex/complex evaluated in double and
// rounded once (what libgcc's __divsc3 does for the reference build), |z| = (float)sqrt((double)x*x+(double)y*y).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "c8p_rx.h"

#define C8P_NEG_INIT (-1000000000000000.0f)   // decode_impl.cc:174
#define C8P_WAVE 64

typedef float2 cf;

struct PktLayout {          // per-packet placement inside the engine arenas (written by k_layout)
  uint64_t llr_off;         // floats
  uint64_t surv_off;        // ulonglong2 entries
  uint64_t bits_off;        // bytes
  uint64_t pdu_off;         // bytes
  uint32_t sym_base;        // first symbol-wave of this packet in k_demod_sym's grid
  uint32_t nsym_bound;
  uint32_t pdu_cap;
  uint32_t trellis_cap;
};

struct Mimo {               // demod2_impl.h:95-105
  cf H[64][4];
  cf INV[64][4];
  cf pil1[4], pil2[4];
};

#define C8P_X8N 4104
struct DevTables {
  int16_t deintL[4][288];         // mapDeintLegacy*      cloud80211phy.cc:1413-1500
  int16_t deintNL[5][416];        // mapDeintNonlegacy*   :1711-1831
  int16_t deint2ss[2][5][416];    // de-interleave (stream 1 / stream 2 map) fused with procSymDepasNL :2442
  int8_t  sigDemapNL[128];        // FFT_NL_SIG_DEMAP :31
  int8_t  deintSigB[52];          // mapDeintVhtSigB20 :1503
  int8_t  intlSigB[52];           // mapIntelVhtSigB20
  float   pilotP[127];            // PILOT_P :160
  float   ltfL[64];               // LTF_L_26_F_FLOAT :77
  float   ltfNL[64];              // LTF_NL_28_F_FLOAT :113
  float   ltfNLvht22[64];         // LTF_NL_28_F_FLOAT_VHT22 :131
  double2 tw[32];                 // e^{-j 2 pi k / 64}, k = 0..31 (double: the FFT runs in fp64, see wave_fft64)
  uint8_t bitrev6[64];
  uint8_t scrSeq[127];            // x^7+x^4+1 sequence from the all-ones state
  uint8_t scrPhase[128];          // phase of a 7-bit state inside scrSeq
  uint64_t scrW[4];               // scrSeq repeated, packed: bit i = scrSeq[i % 127], i = 0..255
  uint32_t crc32[256];
  uint32_t x2n[32];               // x^(2^n) mod P, reflected CRC-32 domain
  uint32_t x8n[C8P_X8N];          // x^(8 n) mod P for every byte distance a frame can have (wave_crc32's segment shifts)
};

// ---------------------------------------------------------------------------------------------- complex helpers
__device__ __forceinline__ cf mk(float a, float b) { return make_float2(a, b); }
__device__ __forceinline__ cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cneg(cf a) { return mk(-a.x, -a.y); }
__device__ __forceinline__ cf cconj(cf a) { return mk(a.x, -a.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulf(cf a, float f) { return mk(a.x * f, a.y * f); }
__device__ __forceinline__ cf cdivf(cf a, float f) { return mk(a.x / f, a.y / f); }
__device__ __forceinline__ cf cdiv(cf a, cf b)
{
  double aa = a.x, bb = a.y, cc = b.x, dd = b.y;
  double denom = (cc * cc) + (dd * dd);
  double x = ((aa * cc) + (bb * dd)) / denom;
  double y = ((bb * cc) - (aa * dd)) / denom;
  return mk((float)x, (float)y);
}
__device__ __forceinline__ float cabs_(cf a)
{
  double x = a.x, y = a.y;
  return (float)sqrt(x * x + y * y);
}

// ---------------------------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// the lane index recomputed on the spot (two VALU instructions the compiler can neither hoist nor share): for kernels whose phases
// must not keep ANY per-lane value alive between them — a shared lane_id() is one more register held, or spilled, across a loop
__device__ __forceinline__ int lane_fresh()
{
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ cf shfl_c(cf v, int src) { return mk(__shfl(v.x, src, 64), __shfl(v.y, src, 64)); }
__device__ __forceinline__ float readlane_f(float v, int l)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l)
{
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
// lane l keeps `val` (wave-uniform), the others keep `old`; no writelane builtin in this toolchain
__device__ __forceinline__ unsigned long long writelane_u64(unsigned long long val, int l, unsigned long long old)
{
  return lane_id() == l ? val : old;
}

// 64-point forward DFT across the 64 lanes of a wave: lane n holds x[n] on entry, X[n] on return.
// Radix-2 decimation-in-frequency, 6 exchange stages, bit-reversal at the end.  The butterflies run in fp64 and
// the result is rounded to float once: the reference's FFT is FFTW3f (third party, plan-dependent rounding), the
// parity tests pin it to "exact DFT rounded to float", and fp64 butterflies reproduce that to the last float bit in all
// but ~1e-8 of the outputs, which keeps the whole LLR stream inside the 1e-5 parity bound even for |LLR| > 20.
__device__ __forceinline__ double shfl_xor_d(double v, int m)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_d(double v, int src)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double neg_if(double v, bool c)      // sign-bit flip: exact, one 32-bit op
{
  return __hiloint2double(__double2hiint(v) ^ (c ? (int)0x80000000 : 0), __double2loint(v));
}
// Decimation in frequency, lane = point, branch-free: the upper lane of a pair negates its own value before the exchange
// (p - x = p + (-x)), so both lanes do ONE add, and the lower lane's twiddle is 1.  The last two stages only see the
// twiddles 1 and -j (a swap), so they have no multiplication at all; the others use 2 mul + 2 fma.  (fp64 runs at half
// the fp32 rate on gfx950 and this kernel competes with the decoder for VALU issue: 28 fp64 ops per transform
// instead of 60.)
// The exchanges: lane ^ HALF for HALF <= 16 is a ds_swizzle in bit mode (and 0x1f, or 0, xor HALF: the pattern is an immediate, no
// address register and no VALU work; it swizzles inside each half of the wave, which is all these stages need), lane ^ 32 a
// ds_bpermute whose byte address is computed once per transform.  (__shfl_xor computes lane ^ m, clamps it to the width and shifts
// it — three VALU instructions per stage in a kernel whose time is VALU issue slots.)
template <int HALF> __device__ __forceinline__ double xchg_d(double v, int addr32)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (HALF == 32) {
    lo = __builtin_amdgcn_ds_bpermute(addr32, lo); hi = __builtin_amdgcn_ds_bpermute(addr32, hi);
  } else {
    lo = __builtin_amdgcn_ds_swizzle(lo, (HALF << 10) | 0x1f); hi = __builtin_amdgcn_ds_swizzle(hi, (HALF << 10) | 0x1f);
  }
  return __hiloint2double(hi, lo);
}
template <int S> __device__ __forceinline__ void fft64_stage(double& xr, double& xi, const DevTables* __restrict__ T, int lane, int addr32)
{
  constexpr int half = 32 >> S;
  const bool up = (lane & half) != 0;
  const double pr = xchg_d<half>(xr, addr32), pi = xchg_d<half>(xi, addr32);
  const double yr = pr + neg_if(xr, up), yi = pi + neg_if(xi, up);
  if constexpr (S == 5) { xr = yr; xi = yi; }
  else if constexpr (S == 4) {                          // twiddle index (lane & 1) << 4: 1 or -j;  (a + jb)(-j) = b - ja
    const bool rot = up && (lane & 1);
    xr = rot ? yi : yr;
    xi = rot ? -yr : yi;
  } else {
    const double2 w = T->tw[up ? ((lane & (half - 1)) << S) : 0];
    xr = __builtin_fma(yr, w.x, -(yi * w.y));
    xi = __builtin_fma(yr, w.y, yi * w.x);
  }
}
__device__ __forceinline__ double2 wave_fft64_d(double xr, double xi, const DevTables* __restrict__ T, int lane)
{
  const int addr32 = (lane ^ 32) << 2;
  fft64_stage<0>(xr, xi, T, lane, addr32);
  fft64_stage<1>(xr, xi, T, lane, addr32);
  fft64_stage<2>(xr, xi, T, lane, addr32);
  fft64_stage<3>(xr, xi, T, lane, addr32);
  fft64_stage<4>(xr, xi, T, lane, addr32);
  fft64_stage<5>(xr, xi, T, lane, addr32);
  // bit reversal: X[n] sits in lane bitrev6(n) (the table T->bitrev6 holds the same numbers; two VALU instructions instead of a load)
  const int src4 = (int)(__builtin_bitreverse32((unsigned)lane) >> 26) << 2;
  int rl = __double2loint(xr), rh = __double2hiint(xr), il = __double2loint(xi), ih = __double2hiint(xi);
  rl = __builtin_amdgcn_ds_bpermute(src4, rl); rh = __builtin_amdgcn_ds_bpermute(src4, rh);
  il = __builtin_amdgcn_ds_bpermute(src4, il); ih = __builtin_amdgcn_ds_bpermute(src4, ih);
  return make_double2(__hiloint2double(rh, rl), __hiloint2double(ih, il));
}
__device__ __forceinline__ double2 wave_fft64_d(double xr, double xi, const DevTables* __restrict__ T) { return wave_fft64_d(xr, xi, T, lane_id()); }
// (lane passed in: a kernel that transforms in a loop hands over an opaque copy of its lane index, otherwise the per-lane twiddle
// addresses of all four multiplying stages are hoisted out of the loop and held in — or spilled from — 20 VGPRs)
__device__ __forceinline__ cf wave_fft64(cf xin, const DevTables* __restrict__ T, int lane)
{
  const double2 r = wave_fft64_d((double)xin.x, (double)xin.y, T, lane);
  return mk((float)r.x, (float)r.y);
}
__device__ __forceinline__ cf wave_fft64(cf xin, const DevTables* __restrict__ T) { return wave_fft64(xin, T, lane_id()); }
// inverse DFT with the 1/64 of numpy.fft.ifft: ifft(X) = conj(fft(conj(X))) / 64, in double
__device__ __forceinline__ double2 wave_ifft64_d(double xr, double xi, const DevTables* __restrict__ T)
{
  const double2 r = wave_fft64_d(xr, -xi, T);
  return make_double2(r.x / 64.0, -r.y / 64.0);
}

__device__ __forceinline__ bool is_null_nl(int i) { return i == 0 || (i >= 29 && i <= 35); }
__device__ __forceinline__ bool is_null_l(int i) { return i == 0 || (i >= 27 && i <= 37); }
__device__ __forceinline__ bool is_pilot(int i) { return i == 7 || i == 21 || i == 43 || i == 57; }
// position j of FFT bin i in the q[] order of nonLegacyChanUpdate (demod_impl.cc:434-446): -28..-1 -> 0..25, +1..+28 -> 26..51
__device__ __forceinline__ int qidx_nl(int i)
{
  return i <= 28 ? 26 + (i - 1 - (i > 7) - (i > 21)) : (i - 36 - (i > 43) - (i > 57));
}
__device__ __forceinline__ int qidx_l(int i)
{
  return i <= 26 ? 24 + (i - 1 - (i > 7) - (i > 21)) : (i - 38 - (i > 43) - (i > 57));
}
__device__ __forceinline__ int nbpsc_index(int nb)
{
  return nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : nb == 6 ? 3 : 4;
}

// branch-metric class of the transition into next-state n from predecessor 2(n&31)+j  (SV_STATE_OUTPUT :1877)
__device__ __forceinline__ int sv_out(int i, int b)
{
  int x0 = b, x1 = (i >> 5) & 1, x2 = (i >> 4) & 1, x3 = (i >> 3) & 1, x5 = (i >> 1) & 1, x6 = i & 1;
  int A = x0 ^ x2 ^ x3 ^ x5 ^ x6;
  int B = x0 ^ x1 ^ x2 ^ x3 ^ x6;
  return 2 * A + B;
}
__device__ __forceinline__ float bm_sel(int c, float t0, float t1, float t01)
{
  return c == 0 ? 0.0f : (c == 1 ? t1 : (c == 2 ? t0 : t01));
}

// One add-compare-select step for next-state = lane.  Exactly the reference recurrence
// (decode_impl.cc:237-259) restated per next-state: candidates arrive in ascending predecessor order,
// strict '>' against a -1e15 pre-fill.  Returns the new metric; take1 / none are the survivor facts.
__device__ __forceinline__ float acs_step(float pre, int src0, int cA, int cB, float t0, float t1, bool& take1, bool& none)
{
  const float t01 = t1 + t0;
  const float p0 = __shfl(pre, src0, 64);
  const float p1 = __shfl(pre, src0 + 1, 64);
  const float a0 = p0 + bm_sel(cA, t0, t1, t01);
  const float a1 = p1 + bm_sel(cB, t0, t1, t01);
  const float base = a0 > C8P_NEG_INIT ? a0 : C8P_NEG_INIT;
  take1 = a1 > base;
  const float cur = take1 ? a1 : base;
  none = (cur == C8P_NEG_INIT);
  return cur;
}

// Short rate-1/2 decoder of the SIG fields (svSigDecoder::decode cloud80211phy.cc:2001-2088), L <= 48: the reference's recurrence
// literally (clamp, NaN behaviour).  The kernels call wave_sig_viterbi (c8p_viterbi_lean.h), which takes the lean step when the LLRs allow
// it and this function otherwise.
// llr in LDS (2L floats), bits out to LDS; w1/w0: 49-entry scratch in LDS.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)      // the same value in every lane -> SGPR pair
{
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// Returns the decoded bits as a wave-uniform word (bit t = bit t) and leaves them in bits[] as bytes.
__device__ __forceinline__ unsigned long long wave_sig_viterbi_exact(const float* llr, uint8_t* bits, int L,
                                                                     unsigned long long* w1, unsigned long long* w0)
{
  const int lane = lane_id();
  const int m = lane & 31, b = lane >> 5;
  const int cA = sv_out(2 * m, b), cB = sv_out(2 * m + 1, b);
  float pre = lane == 0 ? 0.0f : C8P_NEG_INIT;
  for (int t = 0; t < L; t++) {
    bool take1, none;
    pre = acs_step(pre, 2 * m, cA, cB, llr[2 * t], llr[2 * t + 1], take1, none);
    unsigned long long b1 = __ballot(take1), b0 = __ballot(none);
    if (lane == 0) { w1[t] = b1; w0[t] = b0; }
  }
  __syncthreads();
  // traceback from state 0 on the scalar unit: the survivor words come back through readfirstlane, the state is uniform
  int st = 0;
  unsigned long long word = 0;
  for (int t = L - 1; t >= 0; t--) {
    const unsigned long long a1 = uniform_u64(w1[t]), a0 = uniform_u64(w0[t]);
    const int prev = ((a0 >> st) & 1ull) ? 0 : (2 * (st & 31) + (int)((a1 >> st) & 1ull));
    if (st == ((prev >> 1) | 32)) word |= 1ull << t;
    st = prev;
  }
  if (lane < L) bits[lane] = (uint8_t)((word >> lane) & 1ull);
  __syncthreads();
  return word;
}
