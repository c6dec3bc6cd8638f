// c8p_k_finish.h — k_finish: traceback, descramble, A-MPDU walk, CRC-32, PDU records, NDP report (+ the forward pass of packets the lean kernels left)
// (one of the kernel-family headers c8p_kernels.h pulls in, in dependency order; all reference citations are relative to /root/reference/lib/)
#pragma once
#include "c8p_k_decode.h"

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
  constexpr int TS = 64;                                  // tile row stride in dwords.  (68 — rows shifted by 4 banks — until round 3: without
                                                          // the 1 KB of padding, tile + snapshots of a cfg-2 batch are 20.3 KB and EIGHT
                                                          // workgroups share a CU's LDS instead of seven: 0.135 -> 0.129 ms; the staging writes'
                                                          // bank conflicts cost less than the eighth wave brings)
  // tile[row][lane of the forward pass]: the current survivor block of every segment.  After the traceback the same
  // memory holds bw[], the decoded bits packed 64 per word.
  __shared__ __attribute__((aligned(16))) uint32_t tile[C8P_TB_SEGS * TS];
  // lane D at the entry of every 6-step group = 6 decoded bits; later the CRC-32 byte table.  Dynamic: sized by the launch for the
  // longest trellis of the batch (the launch passes it: 3.8 KB at cfg 2 instead of the 5.5 KB of the longest legal trellis), which lets
  // 7 instead of 6 of these one-wave workgroups share a CU's LDS
  extern __shared__ __attribute__((aligned(16))) uint8_t snaps[];
  __shared__ uint16_t s_btop[64];
  unsigned long long* bw = (unsigned long long*)tile;      // (C8P_DECODE_T_MAX + 63) / 64 + 2 words <= 64 * TS / 2
  static_assert(((C8P_DECODE_T_MAX + 63) / 64 + 2) * 8 <= C8P_TB_SEGS * TS * 4, "bw must fit in the tile");
  static_assert(C8P_TB_SEGS % 4 == 0 && C8P_TB_SEGS <= 64, "the tile is fetched four rows per instruction");
  const int p = blockIdx.x;
  if (p >= n) return;
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_FINCLOCK)   // experiment build (tools/probes/exp_finclock.py): shader cycles per phase of this wave instead of the results
  const unsigned long long exp_t0 = clock64();
  unsigned long long exp_t1 = exp_t0, exp_t2 = exp_t0, exp_t3 = exp_t0;
#endif
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
#if C8P_TB_SEGS == 64
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#elif C8P_TB_SEGS == 52
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#elif C8P_TB_SEGS == 56
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#elif C8P_TB_SEGS == 48
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#elif C8P_TB_SEGS == 44
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#elif C8P_TB_SEGS == 40
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9)
#elif C8P_TB_SEGS == 36
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#elif C8P_TB_SEGS == 32
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#elif C8P_TB_SEGS == 28
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#elif C8P_TB_SEGS == 24
#define C8P_ROWS16(X) X(0) X(1) X(2) X(3) X(4) X(5)
#endif
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
              // (Round 3 also walked TWO steps per LDS round trip — both candidate rows of the second step fetched with one
              // ds_read2_b32, 15 instead of 30 dependent reads per block.  Bit-exact, and not faster: the wave waits for its tile loads,
              // not for the walk (profiles/r3e/k_finish_wave_phases.txt) — while the selects cost 890 more VALU instructions per packet
              // (19.1 M instead of 15.4 M per launch) in a step that is bound by VALU issue.  Back to one step per read.)
#pragma unroll
              for (int g = 4; g >= 0; g--) {
                if (A & (63u << (6 * g))) snaps[blk * 5 + g] = (uint8_t)D;
#pragma unroll
                for (int q = 5; q >= 0; q--) {
                  const int k = 6 * g + q;                 // step b0 + k moves lane bit q  ((b0 + k) % 6 == q)
                  const uint32_t w = row[D];
                  const uint32_t sel = ((A >> k) & 1u) << q;             // 0 for steps outside [lo, hi]
                  const int sh = (C8P_SV_BLK - 1 - k) - q;               // decision bit (29 - k) -> bit q
                  const uint32_t wq = sh >= 0 ? (w >> sh) : (w << -sh);
                  D = (int)((wq & sel) | ((uint32_t)D & ~sel));
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
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_FINCLOCK)
      exp_t1 = clock64();
#endif
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
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_FINCLOCK)
    exp_t2 = clock64();
#endif
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
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_FINCLOCK)
    exp_t3 = clock64();
#endif
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
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_FINCLOCK)
  const unsigned long long exp_t4 = clock64();
#endif
  if (lane == 0) {
    c8p_pkt_out o;
    o.n_pdu = n_pdu; o.n_crc_fail = n_fail; o.pdu_bytes = used; o.rsv = (uint32_t)tb_passes | ((uint32_t)n_refetch << 8);   // diagnostics
    o.pdu_off = Lp.pdu_off; o.bits_off = Lp.bits_off;
#if defined(C8P_EXPERIMENTS) && defined(C8P_EXP_FINCLOCK)
    o.n_pdu = (int32_t)(exp_t1 - exp_t0); o.n_crc_fail = (int32_t)(exp_t2 - exp_t1); o.pdu_bytes = (uint32_t)(exp_t3 - exp_t2);
    o.rsv = (uint32_t)(exp_t4 - exp_t3); o.bits_off = exp_t0;
#endif
    outs[p] = o;
  }
}
