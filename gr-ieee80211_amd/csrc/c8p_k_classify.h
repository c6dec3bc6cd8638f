// c8p_k_classify.h — k_classify: format detection, SIG decode, HT / VHT channel estimates, SIG-B (one wave per packet)
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_k_parse.h"

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
