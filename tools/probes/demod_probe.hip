// ISA accounting of one OFDM symbol's demodulation (tools/probes/count_demod_isa.py): each kernel isolates one piece so that its VALU
// instructions can be counted in the compiler's output.  Not part of the product.
#include "c8p_kernels.h"
extern "C" __global__ void p_fft(const cf* x, cf* y, const DevTables* T) { y[lane_id()] = wave_fft64(x[lane_id()], T); }
extern "C" __global__ void p_empty(const cf* x, cf* y, const DevTables* T) { y[lane_id()] = x[lane_id()]; }
extern "C" __global__ void p_cdiv(const cf* x, cf* y, const DevTables* T) { y[lane_id()] = cdiv(x[lane_id()], x[64 + lane_id()]); }
extern "C" __global__ void p_cabs(const cf* x, float* y) { y[lane_id()] = cabs_(x[lane_id()]); }
extern "C" __global__ void p_cdivf(const cf* x, cf* y) { y[lane_id()] = cdivf(x[lane_id()], x[64].x); }
extern "C" __global__ void p_pilots(const cf* x, cf* y, const DevTables* T, int s)
{
  const cf sig = x[lane_id()];
  const float Pp = T->pilotP[(4 + s) % 127];
  float pa_[4];
  for (int k = 0; k < 4; k++) { const int kk = (k + s) & 3; pa_[k] = __uint_as_float(kk == 3 ? 0xbf800000u : 0x3f800000u); }
  cf acc = cmulf(cmulf(shfl_c(sig, 7), pa_[2]), Pp);
  acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 21), pa_[3]), Pp));
  acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 43), pa_[0]), Pp));
  acc = cadd(acc, cmulf(cmulf(shfl_c(sig, 57), pa_[1]), Pp));
  y[lane_id()] = cconj(acc);
}
extern "C" __global__ void p_scatter6(const cf* x, float* out, const DevTables* T, float* negout)
{
  __shared__ float lb[688];
  float neg = 0.f;
  const int lane = lane_id();
  const int jq = (!is_null_nl(lane) && !is_pilot(lane)) ? qidx_nl(lane) : -1;
  bool bad = false;
  if (jq >= 0) bad = llr_scatter<6>(x[lane], jq, T->deintNL[3], lb, neg);
  neg = wave_sum_lane63(neg);
  negout[lane] = neg + (bad ? 1.f : 0.f);
  out[lane] = lb[lane];
}
extern "C" __global__ void p_sym(const cf* x, float* out, const DevTables* T, const c8p_pkt_tags* tg, int s, float* negout)
{
  __shared__ float lb[688];
  const c8p_pkt_tags t = tg[0];
  const int lane = lane_id();
  float neg = 0.f;
  const cf f1 = wave_fft64(x[lane], T, lane);
  const bool bad = demod_sym_1ss<true>(t, s, lane, f1, x[64 + lane], T, lb, neg, (!is_null_nl(lane) && !is_pilot(lane)) ? qidx_nl(lane) : -1);
  neg = wave_sum_lane63(neg);
  negout[lane] = neg + (bad ? 1.f : 0.f);
  out[lane] = lb[lane];
}
extern "C" __global__ void p_sigvit(const float* llr_g, uint8_t* out, int L)
{
  __shared__ float llr[96];
  __shared__ uint8_t bits[48];
  __shared__ unsigned long long w1[49], w0[49];
  for (int i = lane_id(); i < 96; i += 64) llr[i] = llr_g[i];
  __syncthreads();
  const unsigned long long w = wave_sig_viterbi(llr, bits, L, w1, w0);
  if (lane_id() == 0) out[0] = (uint8_t)w;
}
