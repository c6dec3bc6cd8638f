// c8p_k_signal.h — k_signal: the signal / signal2 front end
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_k_parse.h"

// ================================================================================================ signal front-end
// signal_impl::general_work (lib/signal_impl.cc:108-202) / signal2_impl: CFO de-rotation, L-LTF channel estimate,
// L-SIG demod + 24-step Viterbi + signalCheckLegacy, then the de-rotated packet items.  SURVEY.md §8f-1.
__device__ __forceinline__ cf rot(cf x, float th)
{
  float sn, cs;
  sincosf(th, &sn, &cs);                    // cosf(th), sinf(th) with one argument reduction (signal_impl.cc:117-119,172-173)
  return cmul(x, mk(cs, sn));
}

// S_DEMOD of one sync flag (signal_impl.cc:112-160): L-LTF channel estimate, L-SIG demod + 24-step Viterbi + signalCheckLegacy.
// Returns the boundary descriptor without its items (n_items = nsamp + 320 when the packet is complete); chan[64] is written.
__device__ __forceinline__ c8p_pkt_in sig_lsig(const c8p_sig_in& E, int e, const cf* __restrict__ raw, const DevTables* __restrict__ T,
                                               cf* __restrict__ chan_out, float* s_llr, uint8_t* s_bits, unsigned long long* s_w1, unsigned long long* s_w0)
{
  const int lane = lane_id();
  const cf* in1 = raw + E.raw_off[0];
  c8p_pkt_in P;
  memset(&P, 0, sizeof(P));
  P.n_ant = E.n_ant; P.snr = E.snr; P.rssi = E.rssi; P.cfo = E.cfo_rad * 3183098.8618379068f;   // signal_impl.cc:136
  P.chan_off = 64u * (uint32_t)e;
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
  if (ok) { P.lsig_mcs = mcs; P.lsig_len = len; P.lsig_nsamp = nsamp; P.n_items = (uint32_t)nsamp + 320u; }
  return P;
}
// S_COPY + S_PAD (signal_impl.cc:164-192): the packet's items, de-rotated, then 320 zero items
__device__ __forceinline__ void sig_copy(const c8p_sig_in& E, int nsamp, const cf* __restrict__ raw, cf* __restrict__ o1, cf* __restrict__ o2)
{
  const int lane = lane_id();
  const cf* in1 = raw + E.raw_off[0];
  const cf* in2 = raw + E.raw_off[1];
  for (int k = lane; k < nsamp; k += 64) {
    const float th = (float)(k + 224) * E.cfo_rad;
    o1[k] = rot(in1[224 + k], th);
    if (E.n_ant == 2) o2[k] = rot(in2[224 + k], th);
  }
  for (int k = nsamp + lane; k < nsamp + 320; k += 64) { o1[k] = mk(0.f, 0.f); if (E.n_ant == 2) o2[k] = mk(0.f, 0.f); }
}

// every flag on its own: the items go to a slot sized for what the flag has behind it (out_off, planned on the host from n_avail)
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
  const c8p_sig_in E = ev[e];
  const uint32_t cap = (E.n_avail >= 224u ? E.n_avail - 224u : 0u) + 320u;
  c8p_pkt_in P = sig_lsig(E, e, raw, T, chan_out, s_llr, s_bits, s_w1, s_w0);
  P.iq_off[0] = out_off[e]; P.iq_off[1] = E.n_ant == 2 ? out_off[e] + cap : 0;
  if (P.n_items) sig_copy(E, P.lsig_nsamp, raw, iq_out + out_off[e], iq_out + out_off[e] + cap);
  if (lane_id() == 0) pk_out[e] = P;
}
// the two halves as separate launches (c8p_signal_in.consume): the host applies the block's scan rule to the descriptors in
// between, so only the packets the block would copy are copied, back to back
__global__ __launch_bounds__(64) void k_signal_lsig(const c8p_sig_in* __restrict__ ev, const cf* __restrict__ raw, const DevTables* __restrict__ T,
                                                    c8p_pkt_in* __restrict__ pk_out, cf* __restrict__ chan_out, int n)
{
  __shared__ float s_llr[48];
  __shared__ uint8_t s_bits[24];
  __shared__ unsigned long long s_w1[49], s_w0[49];
  const int e = blockIdx.x;
  if (e >= n) return;
  const c8p_pkt_in P = sig_lsig(ev[e], e, raw, T, chan_out, s_llr, s_bits, s_w1, s_w0);
  if (lane_id() == 0) pk_out[e] = P;
}
__global__ __launch_bounds__(64) void k_signal_copy(const c8p_sig_in* __restrict__ ev, const cf* __restrict__ raw, const c8p_pkt_in* __restrict__ pk,
                                                    const int* __restrict__ kept, cf* __restrict__ iq_out, int nk)
{
  if ((int)blockIdx.x >= nk) return;
  const int e = kept[blockIdx.x];
  const c8p_pkt_in P = pk[e];
  sig_copy(ev[e], P.lsig_nsamp, raw, iq_out + P.iq_off[0], iq_out + P.iq_off[1]);
}
