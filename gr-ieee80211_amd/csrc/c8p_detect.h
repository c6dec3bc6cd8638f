// c8p_detect.h — packet detection front end on gfx950: examples/presiso.grc (stock GNU Radio blocks)
// -> lib/trigger_impl.cc -> lib/sync_impl.cc, restructured as scans over the raw 20 MS/s stream.
//
//   k_detect_scan  ONE pass, thread = 8 consecutive samples.  p[n] = x[n-16] conj(x[n]), windows of 48 (complex) and 64
//                  (energy) assembled from per-thread block sums + suffix sums staged in LDS: 13 + 15 LDS reads per 8
//                  samples, no re-summation; per sample "ac > 0.3" and "new maximum of this plateau" (trigger bit 0x02)
//                  by a segmented max-scan seeded with the state the earlier tiles leave behind (published tile
//                  summaries, looked back over inside the launch); every plateau of >= 21 samples goes to the tile's
//                  candidate list.  HBM-bound: 8 B read + 1 B written per sample; ac never leaves the chip.
//   k_cand_offsets, k_cand_gather   the tiles' candidate lists concatenated in stream order
//   (host)         trigger_impl's countdown logic over the (few) candidates, sync_impl's IDLE/SYNC alternation
//   k_sync_mark    wave = accepted trigger: last "latch conj" mark the sync block saw, and the conj value there
//   k_sync_evt     wave = accepted trigger: LTF autocorrelation over 111 offsets with the reference's running sums in
//                  the reference's operation order (lane 0, bit-exact given the samples), plateau centre, CFO, tags
#pragma once
#include "c8p_dev.h"

#ifndef C8P_DET_THREADS
#define C8P_DET_THREADS 128         // threads of a scan workgroup = blocks of 8 samples it loads (the first 8 of them are history).
                                    // 256 until round 4; measured per 33.5 M items, packets / noise (tools/probes/ab_scan.sh, profiles/r4a):
                                    // 128: 81.0 / 67.0 us, 192: 82.5 / 66.7, 256: 84.0 / 66.6, 512: 112.7 / 80.4 — with two waves per
                                    // workgroup a plateau drags one other wave through the heavy path instead of three
#endif
#define C8P_DET_WAVES (C8P_DET_THREADS / 64)
#define C8P_DET_TILE ((C8P_DET_THREADS - 8) * 8)       // samples a workgroup produces: 1984 with 256 threads
#define C8P_AC_THR 0.3f             // trigger_impl.cc:77
#define C8P_SYNC_BUF 240            // sync_impl.h:29
#define C8P_SYNC_RES 111            // sync_impl.h:30

struct TrigSeg {                    // summary of a stretch of samples for the plateau scan
  int all;                          // every sample above the threshold
  float tmax;                       // maximum of ac over the trailing run (inside this stretch)
  long long tstart;                 // first sample of the trailing run of above-threshold samples (negative: it began
                                    // before the commit point), C8P_SEG_NONE: last sample below, C8P_SEG_IDENT: empty stretch
};
#define C8P_SEG_NONE  (-0x7fffffffffffffffLL - 1)
#define C8P_SEG_IDENT (-0x7fffffffffffffffLL)
__device__ __host__ __forceinline__ bool seg_has_run(const TrigSeg& v) { return v.tstart > C8P_SEG_IDENT; }
struct TrigCand {                   // a plateau of >= 21 samples: [s, e); s < 0: it began before the commit point
  long long s, e;
};

// One stream of a detection pass.  c8p_detect has one, c8p_detect_multi K of them: the tiles of a k_detect_scan_multi launch are the
// tiles of stream 0, then those of stream 1, ...; a tile never looks back past its own stream's first tile.
struct DetStreamDev {
  const cf* x;                  // the stream's first NEW item (x[-hist .. -1] = the history before its commit point)
  long long hist;
  unsigned long long n;         // new items
  TrigSeg init;                 // the plateau that is open at the commit point
  unsigned tile_base;           // first tile of this stream in the launch
  int x_al16;                   // x is 16-byte aligned
  unsigned long long out_off;   // where this stream's per-item outputs start: trig (bytes), ac / conj (items); a multiple of 8
  float2 conj0;                 // d_conjMultiAvg at the commit point
  unsigned long long idx_base;  // n_hist: event indices count from the first supplied item
};

__device__ __forceinline__ TrigSeg seg_identity() { TrigSeg r; r.all = 1; r.tstart = C8P_SEG_IDENT; r.tmax = 0.f; return r; }
__device__ __forceinline__ TrigSeg seg_combine(const TrigSeg& L, const TrigSeg& R)   // L then R
{
  if (R.tstart == C8P_SEG_IDENT) return L;
  if (L.tstart == C8P_SEG_IDENT) return R;
  TrigSeg o;
  o.all = L.all && R.all;
  if (!R.all || !seg_has_run(L)) { o.tstart = R.tstart; o.tmax = R.tmax; }
  else { o.tstart = L.tstart; o.tmax = fmaxf(L.tmax, R.tmax); }   // R entirely above and L ends inside a run: it continues
  return o;
}
// Scans across the 64 lanes of a wave by data-parallel primitives (DPP) instead of the LDS crossbar: row_shr:1,2,4,8 inside the rows
// of 16 lanes, then row_bcast:15 (lane 15 of a row -> the next row) and row_bcast:31; a lane with no source keeps `old` (the identity).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) { return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(old), __float_as_int(v))); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ TrigSeg seg_dpp(const TrigSeg& v)
{
  const TrigSeg id = seg_identity();
  TrigSeg r;
  r.all = dpp_i<CTRL, ROW_MASK>(id.all, v.all);
  r.tmax = dpp_f<CTRL, ROW_MASK>(id.tmax, v.tmax);
  const int lo = dpp_i<CTRL, ROW_MASK>((int)(unsigned)(unsigned long long)id.tstart, (int)(unsigned)(unsigned long long)v.tstart);
  const int hi = dpp_i<CTRL, ROW_MASK>((int)(unsigned)((unsigned long long)id.tstart >> 32), (int)(unsigned)((unsigned long long)v.tstart >> 32));
  r.tstart = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  return r;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ TrigSeg seg_wave_inclusive(TrigSeg v)
{
  v = seg_combine(seg_dpp<0x111, 0xf>(v), v);
  v = seg_combine(seg_dpp<0x112, 0xf>(v), v);
  v = seg_combine(seg_dpp<0x114, 0xf>(v), v);
  v = seg_combine(seg_dpp<0x118, 0xf>(v), v);
  v = seg_combine(seg_dpp<0x142, 0xa>(v), v);
  v = seg_combine(seg_dpp<0x143, 0xc>(v), v);
  return v;
}

// ------------------------------------------------------------------------------------------------ presiso + trigger scan
// ONE pass over the raw stream: 8 B read and 1 B (trigger byte) written per sample; the autocorrelation stream `ac` that
// presiso.grc hands to the trigger block stays in registers (it is stored only for tests, keep != 0).
//
// One workgroup = 256 blocks of 8 samples: blocks 0..7 only supply the 16-sample delay and the window history
// (48 products = 6 blocks back, 64 energies = 8 blocks back), blocks 8..255 (248 blocks = 1984 samples = one TILE) are
// produced.  Global traffic is fully coalesced: samples enter through LDS transposed ([sample % 8][block], row stride
// 264 = 8 mod 16 eight-byte banks: conflict-free both for the coalesced writes and for "thread t reads row j, column t - m").
//
// The plateau state machine of trigger_impl.cc:77-85 needs, at the start of a tile, the state everything before it leaves
// behind.  Tiles take their number from a ticket counter (a tile's predecessors have all started), publish their own
// summary as ONE 8-byte granule {valid, all-above, start of the trailing run inside the tile, its maximum} with an
// agent-scope store BEFORE they look at anybody else's (no chain, no deadlock), and then read back over their
// predecessors' granules, 64 per step, until a chunk holds a tile that is not entirely above the threshold — such a
// tile fixes the state by itself — or the start of the segment (`init`).  Plateaus are a few hundred samples long: one
// step, one or two granules awaited in practice; a 400 000-sample carrier walks 4 steps.
#define C8P_PRE_HALO 8
#define C8P_PRE_OUT_BLOCKS (C8P_DET_THREADS - C8P_PRE_HALO)
#ifndef C8P_PRE_RS
#define C8P_PRE_RS (C8P_DET_THREADS + 12)     // 268 (264 until round 4: rows four apart then share banks in the staging stores — SQ_LDS_BANK_CONFLICT = a third of the LDS cycles; 260 / 268 / 276 measured 1-2 % faster, profiles/r4a)
#endif
#define C8P_DET_TILE_CAND ((C8P_DET_TILE / 22 + 8) & ~7)   // plateaus of >= 21 samples that can END inside one tile: ends lie >= 22 apart (48 at 960 samples per tile, 96 at 1984)

__device__ __forceinline__ unsigned long long seg_pack(const TrigSeg& v, long long tile0)
{
  const unsigned rel = seg_has_run(v) ? (unsigned)(v.tstart - tile0) : 0xFFFu;     // a tile's own trailing run starts inside it
  return (unsigned long long)__float_as_uint(v.tmax) | ((unsigned long long)rel << 32) | ((unsigned long long)(v.all ? 1u : 0u) << 44) | (1ull << 63);
}
__device__ __forceinline__ TrigSeg seg_unpack(unsigned long long g, long long tile0)
{
  TrigSeg r;
  const unsigned rel = (unsigned)(g >> 32) & 0xFFFu;
  r.all = (int)((g >> 44) & 1u);
  r.tmax = __uint_as_float((unsigned)g);
  r.tstart = rel == 0xFFFu ? C8P_SEG_NONE : tile0 + (long long)rel;
  return r;
}
__device__ __forceinline__ TrigSeg seg_bcast(const TrigSeg& v, int src)
{
  TrigSeg r;
  r.all = __shfl(v.all, src, 64);
  r.tmax = __shfl(v.tmax, src, 64);
  const int lo = __shfl((int)(unsigned)(unsigned long long)v.tstart, src, 64);
  const int hi = __shfl((int)(unsigned)((unsigned long long)v.tstart >> 32), src, 64);
  r.tstart = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  return r;
}
// wave 0 of a workgroup: the state at the start of `tile` (lane L of a step looks at tile base - 64 + L).
// WAIT = false: every granule is known to be there (a later launch).  *gave_up is set when a granule did not arrive within
// C8P_DET_SPINS polls (~tens of ms: a tile's predecessor is normally some microseconds away); the result is then meaningless
// and the host repeats the launch with ticket numbering, under which a tile never waits for one that has not started.
// Under ticket numbering every predecessor HAS started, so the wait always ends: a co-tenant, a pre-empted CU or a debugger may
// stall it for a long time and the watchdog is only there to turn a lost GPU into an error (C8P_DET_SPINS_TICKET polls: many seconds).
#define C8P_DET_SPINS (1 << 15)
#define C8P_DET_SPINS_TICKET (1 << 24)
template <bool WAIT>
__device__ __forceinline__ TrigSeg trig_lookback(const unsigned long long* __restrict__ summ, long long tile, const TrigSeg& init,
                                                 unsigned int* __restrict__ gave_up, int spins = C8P_DET_SPINS)
{
  const int lane = lane_id();
  TrigSeg acc = seg_identity();                       // tiles [base, tile)
  for (long long base = tile; ; base -= 64) {
    if (base <= 0) return seg_combine(init, acc);
    const long long t = base - 64 + lane;
    unsigned long long g = 0;
    if (!WAIT) { if (t >= 0) g = summ[t]; }
    else for (int spin = 0; ; spin++) {
      // granules are polled with agent-scope loads (they bypass this CU's L1); a granule carries its own valid bit, so no
      // ordering against other data is needed.  Enough = every tile from the nearest one that is not entirely above the
      // threshold up to `tile` is there (or all 64, if this chunk has none).
      if (t >= 0 && !(g >> 63)) g = __hip_atomic_load(summ + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long have = __ballot(t < 0 || (g >> 63));
      const unsigned long long stop = __ballot(t >= 0 && (g >> 63) && !((g >> 44) & 1u));
      if (have == ~0ull) break;
      if (stop) {
        const int j = 63 - __builtin_clzll(stop);      // nearest such tile that has published
        if ((have >> j) == (~0ull >> j)) break;        // and everything nearer than it
      }
      if (spin >= spins) { if (lane == 0) atomicOr(gave_up, 1u); return seg_identity(); }
      __builtin_amdgcn_s_sleep(2);
    }
    const bool ok = t >= 0 && (g >> 63);               // (granules further back than the deciding tile may be missing: ignored by the combine)
    const TrigSeg chunk = seg_bcast(seg_wave_inclusive(ok ? seg_unpack(g, t * C8P_DET_TILE) : seg_identity()), 63);
    acc = seg_combine(chunk, acc);
    if (!chunk.all) return acc;                        // independent of anything earlier (seg_combine with R.all == 0)
  }
}

// x points at the first NEW item; items x[-hist .. -1] exist (history before the commit point), x[g] = 0 for g < -hist.
// One tile's window sums: thread tid (>= C8P_PRE_HALO) gets P (48 products) and E (64 energies) of its 8 samples
// base + 8 tid .. + 7; returns `base`.
__device__ __forceinline__ long long tile_windows(const cf* __restrict__ x, long long hist, unsigned long long n, long long tile, int x_aligned16,
                                                  cf* sX, float* sE, cf (&Pv)[8], float (&Ev)[8])
{
  const int tid = threadIdx.x;
  const long long blk0 = tile * C8P_PRE_OUT_BLOCKS - C8P_PRE_HALO;   // first block of this workgroup
  const long long base = blk0 * 8;
  if (x_aligned16 && base >= -hist && base + 8 * C8P_DET_THREADS <= (long long)n) {
    // interior workgroup: 16-byte loads, thread tid takes samples base + 2 tid + 512 k and the next one (base is a
    // multiple of 8, so these are 16-byte aligned whenever x is: the host passes that in)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = 2 * tid + 2 * C8P_DET_THREADS * k;
      const float4 v = *(const float4*)(x + base + i);
      sX[(i & 7) * C8P_PRE_RS + (i >> 3)] = mk(v.x, v.y);
      sX[((i + 1) & 7) * C8P_PRE_RS + ((i + 1) >> 3)] = mk(v.z, v.w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) {           // coalesced: thread tid loads sample base + tid + 256 k
      const int i = tid + C8P_DET_THREADS * k;
      const long long g = base + i;
      sX[(i & 7) * C8P_PRE_RS + (i >> 3)] = (g >= -hist && g < (long long)n) ? x[g] : mk(0.f, 0.f);
    }
  }
  __syncthreads();
  cf p[8];
  float e[8];
  const int tid2 = tid >= 2 ? tid - 2 : tid;                                  // (blocks 0 and 1 have no x[g - 16] here: their products are never used)
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const cf a = sX[j * C8P_PRE_RS + tid];
    const cf d = sX[j * C8P_PRE_RS + tid2];                                    // x[g - 16]: two blocks back
    p[j] = mk(d.x * a.x + d.y * a.y, d.y * a.x - d.x * a.y);                    // x[g-16] * conj(x[g])
    e[j] = a.x * a.x + a.y * a.y;
  }
  __syncthreads();
  {
    cf s = mk(0.f, 0.f);
    float se = 0.f;
#pragma unroll
    for (int j = 7; j >= 0; j--) { s = cadd(s, p[j]); se += e[j]; sX[j * C8P_PRE_RS + tid] = s; sE[j * C8P_PRE_RS + tid] = se; }   // (row 8 would be zeros: not stored)
  }
  __syncthreads();
  if (tid >= C8P_PRE_HALO) {
    cf cp = mk(0.f, 0.f);
    float ce = 0.f;
#pragma unroll
    for (int m = 1; m <= 5; m++) cp = cadd(cp, sX[tid - m]);         // five whole blocks of the 48-window
#pragma unroll
    for (int m = 1; m <= 7; m++) ce += sE[tid - m];                  // seven whole blocks of the 64-window
    cf pre = mk(0.f, 0.f);
    float pe = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      pre = cadd(pre, p[j]); pe += e[j];
      Pv[j] = cadd(cadd(cp, j < 7 ? sX[(j + 1) * C8P_PRE_RS + tid - 6] : mk(0.f, 0.f)), pre);   // tail of block b-6 + blocks b-5..b-1 + head of block b
      Ev[j] = (ce + (j < 7 ? sE[(j + 1) * C8P_PRE_RS + tid - 8] : 0.f)) + pe;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) { Pv[j] = mk(0.f, 0.f); Ev[j] = 0.f; }
    if (tid == C8P_PRE_HALO - 1) {
      // the item just before the tile, with the operations and the order in which the previous tile's last thread forms it
      // (its j = 7: whole blocks b-1 .. b-5 resp. b-7, an empty tail, the whole own block front to back)
      cf cp = mk(0.f, 0.f), pre = mk(0.f, 0.f);
      float ce = 0.f, pe = 0.f;
#pragma unroll
      for (int m = 1; m <= 5; m++) cp = cadd(cp, sX[tid - m]);
#pragma unroll
      for (int m = 1; m <= 7; m++) ce += sE[tid - m];
#pragma unroll
      for (int j = 0; j < 8; j++) { pre = cadd(pre, p[j]); pe += e[j]; }
      Pv[7] = cadd(cadd(cp, mk(0.f, 0.f)), pre);
      Ev[7] = (ce + 0.f) + pe;
    }
  }
  return base;
}
// presiso's output 0.  Correctly rounded sqrt and quotient (-fhip-fp32-correctly-rounded-divide-sqrt), i.e. IEEE sqrtf(.) / e as
// VOLK + blocks_divide_ff compute it: ac feeds hard comparisons (ac > 0.3, ac > running maximum, trigger_impl.cc:77-85), so given
// the same window sums the trigger bytes are those of the scalar CPU restatement bit for bit (tests/test_detect_frontend.py:
// integer-valued torture stream straddling 0.3)
__device__ __forceinline__ float ac_of(const cf P, const float E) { return sqrtf(P.x * P.x + P.y * P.y) / E; }
// "certainly not above the threshold" without the square root and the quotient: |P|^2 < 0.08 E^2 means ac < 0.283 whatever the
// roundings; anything else (NaN, a borderline value) takes the exact path.  Idle air and payload symbols
// fail this test for whole waves, which then skip the two correctly rounded operations per sample.
__device__ __forceinline__ bool ac_surely_below(const cf P, const float E)
{
  // E clamped to 1e18 (keeps the square finite; a smaller E only makes the test harder to pass).  E is a sum of squares: >= 0, or NaN /
  // inf — for those bit patterns the unsigned order is the float order and NaN / inf lie above every finite value, so one unsigned min
  // gives what fminf(E, 1e18f) gives (fminf costs a canonicalising max in front of the min)
  const float Ec = __uint_as_float(min(__float_as_uint(E), 0x5d5e0b6bu));
  return P.x * P.x + P.y * P.y < 0.08f * Ec * Ec;
}

// in-tile scan element over a stretch of samples: position (tile-relative) of its last sample that is NOT above the threshold
// (-1: none), and the maximum of ac over the samples after that one
struct RunSeg { int pos; float tmax; };
__device__ __forceinline__ RunSeg run_combine(const RunSeg& L, const RunSeg& R)
{
  RunSeg o;
  o.pos = max(L.pos, R.pos);
  o.tmax = R.pos >= 0 ? R.tmax : fmaxf(L.tmax, R.tmax);
  return o;
}

// `numbering`: 0 = tiles are numbered by blockIdx.x (the hardware hands workgroups out in that order: a tile's predecessors have
// started, which is all the look-back needs — HIP does not promise it, hence the watchdog in trig_lookback), 1 = by a ticket
// counter (a promise, at the price of 16 910 same-address atomics per 33.5 M samples: 0.09 ms)
#define C8P_CAND_OPEN_START C8P_SEG_IDENT      // TrigCand.s of "the run that was open when this (quiet) tile began": k_cand_gather fills it in
// one tile of one stream (`tile` counts from the stream's first tile; summ / tcand / tcount point at that tile's entries, fin at the stream's)
__device__ __forceinline__ void detect_scan_tile(const cf* __restrict__ x, long long hist, unsigned long long n, int x_aligned16,
                                                 unsigned long long* __restrict__ summ, unsigned int* __restrict__ gave_up,
                                                 const TrigSeg& init, TrigSeg* __restrict__ fin, uint8_t* __restrict__ trig,
                                                 TrigCand* __restrict__ tcand, uint8_t* __restrict__ tcount,
                                                 float* __restrict__ ac, cf* __restrict__ conj, int keep, int numbering, long long tile)
{
  __shared__ TrigSeg s_cin;
  __shared__ RunSeg s_w[C8P_DET_WAVES];
  __shared__ unsigned s_wc[C8P_DET_WAVES];
  __shared__ cf sX[8 * C8P_PRE_RS];         // samples, later sufP[j][t] = sum_{i=j..7} p_i of block t
  __shared__ float sE[8 * C8P_PRE_RS];      // sufE[j][t], later the ac values on their way out
  const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
  const long long tile0 = tile * C8P_DET_TILE;
  cf Pv[8];
  float acv[8];
  const long long base = tile_windows(x, hist, n, tile, x_aligned16, sX, sE, Pv, acv);      // acv = the energies for now
  const bool live = tid >= C8P_PRE_HALO;
  const long long n0 = base + tid * 8;      // this thread's first sample
  const int r0 = (tid - C8P_PRE_HALO) * 8;  // the same, relative to the tile
  const int nv = live ? (int)min(8ll, max(0ll, (long long)n - n0)) : 0;     // how many of the 8 exist
  const float e_prev = acv[7];               // (halo's last thread: the energy window of the item before the tile)
  bool maybe = false;                        // one of this thread's samples is not certainly below the threshold (a lane mask: no VALU)
#pragma unroll
  for (int j = 0; j < 8; j++) maybe |= j < nv && !ac_surely_below(Pv[j], acv[j]);
  unsigned m = 0;                            // bit j: sample j is above the threshold
  if (keep || __ballot(maybe)) {
#pragma unroll
    for (int j = 0; j < 8; j++) { acv[j] = ac_of(Pv[j], acv[j]); if (j < nv && acv[j] > C8P_AC_THR) m |= 1u << j; }   // NaN (0/0 in silence) compares false, as in the reference
  }
  if (keep) {                                // test path: ac leaves through sE, once every thread has read its window sums from it
    __syncthreads();
    if (live) {
#pragma unroll
      for (int j = 0; j < 8; j++) sE[j * C8P_PRE_RS + tid] = acv[j];
    }
  }
  const int tile_any = __syncthreads_or(m != 0);
  const int nexist = (int)min((long long)C8P_DET_TILE, (long long)n - tile0);
  if (!tile_any) {
    // nothing above the threshold in this tile (idle air, payload): its summary is known at once and its trigger bytes are 0.
    // It waits for nobody: the only plateau that can end here is one that lasted up to the item before the tile — the halo's last
    // thread has that item's ac — and where it began is filled in by k_cand_gather.
    if (nv == 8) *(unsigned long long*)(trig + n0) = 0ull;
    else for (int j = 0; j < nv; j++) trig[n0 + j] = 0;
    if (tid == C8P_PRE_HALO - 1) {
      TrigSeg total; total.all = 0; total.tstart = C8P_SEG_NONE; total.tmax = 0.f;
      __hip_atomic_store(summ + tile, seg_pack(total, tile0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool open = tile == 0 ? seg_has_run(init) : (ac_of(Pv[7], e_prev) > C8P_AC_THR);
      if (open) { TrigCand k; k.s = C8P_CAND_OPEN_START; k.e = tile0; tcand[tile * C8P_DET_TILE_CAND] = k; }
      tcount[tile] = open ? 1 : 0;
      if (tile0 + nexist >= (long long)n) *fin = total;
    }
  } else {
    // per-thread element, wave scan, then across the four waves
    RunSeg mine; mine.pos = -1; mine.tmax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (j < nv) { if ((m >> j) & 1u) mine.tmax = fmaxf(mine.tmax, acv[j]); else { mine.pos = r0 + j; mine.tmax = 0.f; } }
    }
    // A wave of a FULL tile with no sample above the threshold (the plateau lies in the tile's other waves): every lane's element is
    // {its own last sample, 0}, so the last lane's element is the wave's total without a scan, the trigger bytes are zeros, and the only
    // thing that can happen here is that a plateau of the previous wave ends at this wave's first sample.  (Round 4: these waves ran
    // the six scan steps and the eight-sample walk like the others — ~170 of the ~300 VALU instructions of the heavy path.)
    const bool wave_quiet = nexist == C8P_DET_TILE && __ballot(m != 0) == 0ull;
    RunSeg inc = mine;
    if (!wave_quiet) {
#define C8P_RUN_STEP(CTRL, ROWS) { RunSeg u; u.pos = dpp_i<CTRL, ROWS>(-1, inc.pos); u.tmax = dpp_f<CTRL, ROWS>(0.f, inc.tmax); inc = run_combine(u, inc); }
      C8P_RUN_STEP(0x111, 0xf) C8P_RUN_STEP(0x112, 0xf) C8P_RUN_STEP(0x114, 0xf) C8P_RUN_STEP(0x118, 0xf)
      C8P_RUN_STEP(0x142, 0xa) C8P_RUN_STEP(0x143, 0xc)
#undef C8P_RUN_STEP
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    RunSeg ex; ex.pos = -1; ex.tmax = 0.f;              // (a quiet wave: only its first live lane looks at it, and nothing lies before that lane)
    if (!wave_quiet) { ex.pos = dpp_i<0x138, 0xf>(-1, inc.pos); ex.tmax = dpp_f<0x138, 0xf>(0.f, inc.tmax); }     // wave_shr:1 (lane 0: the identity)
    RunSeg pre; pre.pos = -1; pre.tmax = 0.f;
    RunSeg tot = pre;
#pragma unroll
    for (int k = 0; k < C8P_DET_WAVES; k++) { if (k == wv) pre = tot; tot = run_combine(tot, s_w[k]); }
    ex = run_combine(pre, ex);
    // the tile's summary: entirely above?  its trailing run (after the last sample that is not above), if the last sample is above
    TrigSeg total;
    total.all = tot.pos < 0;
    total.tstart = tot.pos + 1 < nexist ? tile0 + tot.pos + 1 : C8P_SEG_NONE;
    total.tmax = tot.pos + 1 < nexist ? tot.tmax : 0.f;
    if (tid == 0) __hip_atomic_store(summ + tile, seg_pack(total, tile0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
      const TrigSeg c0 = trig_lookback<true>(summ, tile, init, gave_up, numbering ? C8P_DET_SPINS_TICKET : C8P_DET_SPINS);
      if (tid == 0) {
        s_cin = c0;
        if (tile0 + nexist >= (long long)n) *fin = seg_combine(c0, total);      // the state at the end of the segment
      }
    }
    __syncthreads();
    // walk the 8 samples with the plateau state: start of the run that is open (tile-relative, negative: it began before the
    // tile) and its running maximum (d_conjAc, which restarts at 0 with every plateau: trigger_impl.cc:80-85)
    const TrigSeg cin = s_cin;
    const bool from_cin = ex.pos < 0 && seg_has_run(cin);
    int rs = from_cin ? (int)max(cin.tstart - tile0, -(1ll << 30)) : ex.pos + 1;
    float rmax = from_cin ? fmaxf(cin.tmax, ex.tmax) : ex.tmax;
    unsigned long long packed = 0;
    bool hasc = false;
    TrigCand mc; mc.s = 0; mc.e = 0;
    if (wave_quiet) {
      if (tid == (wv ? wv * 64 : C8P_PRE_HALO) && r0 - rs >= 21) { hasc = true; mc.s = rs < 0 ? cin.tstart : tile0 + rs; mc.e = tile0 + r0; }
    } else
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (j < nv) {
        const int g = r0 + j;
        unsigned b = 0;
        if ((m >> j) & 1u) {
          b = 0x04;
          if (acv[j] > rmax) { rmax = acv[j]; b |= 0x02; }
          if (n0 + j == (long long)n - 1 && g + 1 - rs >= 21) {                 // still open at the end of the stream
            hasc = true; mc.s = rs < 0 ? cin.tstart : tile0 + rs; mc.e = (long long)n;
          }
        } else {
          if (g - rs >= 21) { hasc = true; mc.s = rs < 0 ? cin.tstart : tile0 + rs; mc.e = tile0 + g; }   // a plateau long enough to arm the countdown ended here
          rs = g + 1; rmax = 0.f;
        }
        packed |= (unsigned long long)b << (8 * j);
      }
    }
    if (nv == 8) *(unsigned long long*)(trig + n0) = packed;
    else for (int j = 0; j < nv; j++) trig[n0 + j] = (uint8_t)(packed >> (8 * j));
    // the tile's candidates in stream order: ends lie >= 22 samples apart, so a thread holds at most one
    const unsigned long long bm = __ballot(hasc);
    if (lane == 0) s_wc[wv] = (unsigned)__builtin_popcountll(bm);
    __syncthreads();
    unsigned rank = (unsigned)__builtin_popcountll(bm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; w++) rank += s_wc[w];
    if (hasc) tcand[tile * C8P_DET_TILE_CAND + rank] = mc;
    if (tid == 0) { unsigned tc = 0; for (int w = 0; w < C8P_DET_WAVES; w++) tc += s_wc[w]; tcount[tile] = (uint8_t)tc; }
  }
  if (!keep) return;
  // test / debug path only: presiso's two output streams
  __syncthreads();
  if (base >= 0 && base + 8 * C8P_DET_THREADS <= (long long)n) {
    // interior workgroup: 16-byte stores, thread tid stores samples base + 4 tid + 1024 k .. +3 (base is a multiple of 8)
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = 4 * tid + 4 * C8P_DET_THREADS * k;
      if ((i >> 3) >= C8P_PRE_HALO) {
        const int r = (i >> 3);
        const float4 v = make_float4(sE[(i & 7) * C8P_PRE_RS + r], sE[((i + 1) & 7) * C8P_PRE_RS + r],
                                     sE[((i + 2) & 7) * C8P_PRE_RS + r], sE[((i + 3) & 7) * C8P_PRE_RS + r]);
        *(float4*)(ac + base + i) = v;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) {           // coalesced: thread tid stores sample base + tid + 256 k
      const int i = tid + C8P_DET_THREADS * k;
      const long long g = base + i;
      if ((i >> 3) >= C8P_PRE_HALO && (unsigned long long)g < n && g >= 0) ac[g] = sE[(i & 7) * C8P_PRE_RS + (i >> 3)];
    }
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < 8; j++) if (j < nv) conj[n0 + j] = Pv[j];
  }
}

__global__ __launch_bounds__(C8P_DET_THREADS, 6) void k_detect_scan(const cf* __restrict__ x, long long hist, unsigned long long n, int x_aligned16,
                                                     unsigned long long* __restrict__ summ, unsigned int* __restrict__ ticket,
                                                     TrigSeg init, TrigSeg* __restrict__ fin, uint8_t* __restrict__ trig,
                                                     TrigCand* __restrict__ tcand, uint8_t* __restrict__ tcount,
                                                     float* __restrict__ ac, cf* __restrict__ conj, int keep, int numbering)
{
  __shared__ unsigned s_tile;
  long long tile = blockIdx.x;
  if (numbering) {
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    tile = (long long)s_tile;
  }
  detect_scan_tile(x, hist, n, x_aligned16, summ, ticket + 2, init, fin, trig, tcand, tcount, ac, conj, keep, numbering, tile);
}
// K streams in one launch: tile g belongs to stream tile_stream[g]; everything a tile touches is that stream's own (its items, its
// summaries, its outputs), so the streams do not see each other
__global__ __launch_bounds__(C8P_DET_THREADS, 6) void k_detect_scan_multi(const DetStreamDev* __restrict__ ds, const unsigned short* __restrict__ tile_stream,
                                                           unsigned long long* __restrict__ summ, unsigned int* __restrict__ ticket,
                                                           TrigSeg* __restrict__ fin, uint8_t* __restrict__ trig,
                                                           TrigCand* __restrict__ tcand, uint8_t* __restrict__ tcount,
                                                           float* __restrict__ ac, cf* __restrict__ conj, int keep, int numbering)
{
  __shared__ unsigned s_tile;
  unsigned g = blockIdx.x;
  if (numbering) {
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    g = s_tile;
  }
  const unsigned s = tile_stream[g];
  const DetStreamDev D = ds[s];
  const unsigned tb = D.tile_base;
  detect_scan_tile(D.x, D.hist, D.n, D.x_al16, summ + tb, ticket + 2, D.init, fin + s, trig + D.out_off,
                   tcand + (size_t)tb * C8P_DET_TILE_CAND, tcount + tb, ac + D.out_off, conj + D.out_off, keep, numbering, (long long)(g - tb));
}

// the per-tile candidate lists, concatenated in tile (= stream) order.  k_cand_offsets: one workgroup, a thread takes a
// contiguous range of tiles (a multiple of 16: the counts are bytes, read 16 at a time; tcount is zero beyond ntile up to the
// next multiple of 16) and writes where each tile's list goes; k_cand_gather: thread = tile, moves the list.
__global__ __launch_bounds__(1024) void k_cand_offsets(const uint8_t* __restrict__ tcount, int ntile, unsigned int* __restrict__ toff,
                                                       unsigned int* __restrict__ n_out)
{
  __shared__ unsigned wtot[16];
  const int tid = threadIdx.x, w = tid >> 6, lane = lane_id();
  const int per = (((ntile + 1023) / 1024) + 15) & ~15;
  const int t0 = min(ntile, tid * per), t1 = min(ntile, t0 + per);
  unsigned mine = 0;
  for (int t = t0; t < t1; t += 16) {
    const uint4 v = *(const uint4*)(tcount + t);
    // bytes <= 96: the byte lanes of a word add up without carries into each other
    const unsigned q = (v.x & 0x00ff00ffu) + ((v.x >> 8) & 0x00ff00ffu) + (v.y & 0x00ff00ffu) + ((v.y >> 8) & 0x00ff00ffu)
                     + (v.z & 0x00ff00ffu) + ((v.z >> 8) & 0x00ff00ffu) + (v.w & 0x00ff00ffu) + ((v.w >> 8) & 0x00ff00ffu);
    mine += (q & 0xffffu) + (q >> 16);
  }
  unsigned inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  unsigned off = inc - mine, all = 0;
  for (int k = 0; k < 16; k++) { if (k < w) off += wtot[k]; all += wtot[k]; }
  if (tid == 0) *n_out = all;
  for (int t = t0; t < t1; t += 16) {
    const uint4 v = *(const uint4*)(tcount + t);
    const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int b = 0; b < 16; b++) {
      if (t + b < t1) toff[t + b] = off;
      off += (wd[b >> 2] >> (8 * (b & 3))) & 0xffu;
    }
  }
}
__global__ __launch_bounds__(256) void k_cand_gather(const uint8_t* __restrict__ tcount, const unsigned int* __restrict__ toff,
                                                     const TrigCand* __restrict__ tcand, int ntile, const unsigned long long* __restrict__ summ,
                                                     TrigSeg init, TrigCand* __restrict__ out, unsigned int cap,
                                                     const DetStreamDev* __restrict__ ds, const unsigned short* __restrict__ tile_stream)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntile) return;
  const unsigned cnt = tcount[t];
  if (!cnt) return;
  unsigned off = toff[t];
  for (unsigned i = 0; i < cnt; i++, off++) {
    TrigCand k = tcand[(size_t)t * C8P_DET_TILE_CAND + i];
    if (k.s == C8P_CAND_OPEN_START) {
      // a quiet tile closed the run that was open at its first item: that run's start, from the summaries before the tile
      // (ds: the tile's own stream only; positions count from that stream's commit point)
      long long lo = 0;
      if (ds) { const DetStreamDev& D = ds[tile_stream[t]]; lo = D.tile_base; init = D.init; }
      TrigSeg acc = seg_identity();
      long long u = t - 1;
      for (; u >= lo; u--) { acc = seg_combine(seg_unpack(summ[u], (u - lo) * C8P_DET_TILE), acc); if (!acc.all) break; }
      if (u < lo) acc = seg_combine(init, acc);
      k.s = seg_has_run(acc) ? acc.tstart : k.e;     // (s = e: no plateau, the host's walk skips it)
    }
    if (off < cap) out[off] = k;
  }
}

// where each stream's candidates start in the concatenated list: soff[s] (soff[K] = their number)
__global__ void k_stream_offsets(const unsigned int* __restrict__ toff, const unsigned int* __restrict__ n_all, const DetStreamDev* __restrict__ ds,
                                 int K, int ntile, unsigned int* __restrict__ soff)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > K) return;
  // (a stream without tiles has the tile base of the next one: no candidates of its own)
  soff[s] = (s < K && (int)ds[s].tile_base < ntile) ? toff[ds[s].tile_base] : *n_all;
}

// the maximum of ac over the items [lo, hi) of one plateau (d_conjAc at a commit point that falls inside it), from the
// same arithmetic as k_detect_scan; ac > 0.3 > 0 there, so the float order is the order of the bit patterns
struct RmaxJob {            // one stream's plateau that is open at its commit point
  const cf* x;              // the stream's first new item
  long long hist;
  unsigned long long n;
  long long tile_lo, lo, hi;
  unsigned int* out_bits;
  unsigned first;           // index of the job's first workgroup in the launch
  int x_aligned16;
};
// (one launch for all streams: a launch per stream was 30-250 small kernels in a row per round of 256 live streams)
__global__ __launch_bounds__(C8P_DET_THREADS) void k_ac_range_max(const RmaxJob* __restrict__ jobs, int nj)
{
  __shared__ cf sX[8 * C8P_PRE_RS];
  __shared__ float sE[8 * C8P_PRE_RS];
  cf Pv[8];
  float Ev[8];
  int lo_j = 0, hi_j = nj - 1;
  while (lo_j < hi_j) { const int mid = (lo_j + hi_j + 1) >> 1; if (jobs[mid].first <= blockIdx.x) lo_j = mid; else hi_j = mid - 1; }
  const RmaxJob J = jobs[lo_j];
  const long long base = tile_windows(J.x, J.hist, J.n, J.tile_lo + (long long)(blockIdx.x - J.first), J.x_aligned16, sX, sE, Pv, Ev);
  if (threadIdx.x < C8P_PRE_HALO) return;
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const long long g = base + threadIdx.x * 8 + j;
    const float a = ac_of(Pv[j], Ev[j]);
    if (g >= J.lo && g < J.hi && a > m) m = a;
  }
  if (m > 0.f) atomicMax(J.out_bits, __float_as_uint(m));
}

// the host adds the trigger bits (0x01) it derived from the candidates
__global__ void k_trig_fire(uint8_t* __restrict__ trig, const unsigned long long* __restrict__ fire, int nf)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nf) trig[fire[i]] |= 0x01;
}

// ------------------------------------------------------------------------------------------------ sync
struct SyncJob {            // one accepted trigger
  unsigned long long t;     // trigger item (sync_impl switches to SYNC_S_SYNC here)
  unsigned long long seen;  // first item the IDLE state scanned before reaching t
  unsigned stream;          // whose (index into the DetStreamDev array)
  unsigned first;           // index of that stream's first job
  unsigned kind;            // 0: a trigger; 1: only the look-back (what IDLE has latched when it reaches the commit point)
  unsigned rsv;
};
struct SyncMark {
  long long j;              // last item in [seen, t) with trigger bit 0x02, -1: none
  cf conj;                  // presiso output 1 at j
};

__global__ __launch_bounds__(64) void k_sync_mark(const SyncJob* __restrict__ job, const uint8_t* __restrict__ trig_all,
                                                  const DetStreamDev* __restrict__ ds, SyncMark* __restrict__ out, int nj)
{
  const int k = blockIdx.x;
  if (k >= nj) return;
  const int lane = lane_id();
  const SyncJob J = job[k];
  const cf* __restrict__ x = ds[J.stream].x;
  const long long hist = ds[J.stream].hist;
  const uint8_t* __restrict__ trig = trig_all + ds[J.stream].out_off;
  long long j = -1;
  for (long long hi = (long long)J.t; hi > (long long)J.seen && j < 0; hi -= 64) {   // newest 64 items first
    const long long g = hi - 1 - lane;
    const bool m = g >= (long long)J.seen && (trig[g] & 0x02);
    const unsigned long long bm = __ballot(m);
    if (bm) j = hi - 1 - (long long)__builtin_ctzll(bm);
  }
  cf P = mk(0.f, 0.f);
  if (j >= 0) {
    // conj[j] = sum_{k=j-47..j} x[k-16] conj(x[k]): 48 products, lanes 0..47, folded oldest-first by lane 0
    __shared__ cf pr[48];
    if (lane < 48) {
      const long long g = j - 47 + lane;
      cf a = mk(0.f, 0.f), d = mk(0.f, 0.f);
      if (g >= -hist) a = x[g];
      if (g - 16 >= -hist) d = x[g - 16];
      pr[lane] = mk(d.x * a.x + d.y * a.y, d.y * a.x - d.x * a.y);
    }
    __syncthreads();
    if (lane == 0) for (int i = 0; i < 48; i++) P = cadd(P, pr[i]);
  }
  if (lane == 0) { out[k].j = j; out[k].conj = P; }
}

__device__ __forceinline__ float habs(cf a) { return cabs_(a); }    // std::abs(std::complex<float>) == hypotf

// Sequential float chains across the lanes of a wave.  The reference accumulates its running sums one item after the other, and the
// results feed hard comparisons (> 0.5, < 0.8 max), so the order of the additions is part of the result.  A chain "value of lane l =
// f(value of lane l - 1, operands of lane l)" is evaluated by applying f to the WHOLE wave 64 times through a one-lane DPP shift: lane 0
// is right after the first application, lane l after l + 1, and a lane that is right stays right — one VALU instruction per link for all
// links at once, instead of a lane-0 loop over LDS (a load, an add and a store per link, the other 63 lanes idle).
__device__ __forceinline__ float dpp_shr1(float v)      // lane l gets v of lane l - 1, lane 0 gets 0
{
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
// C chains side by side (their links are independent, so one chain's DPP wait states are the others' issue slots).
// lane l: ((0 + x[0]) + x[1]) + ... + x[l]
template <int C, int N>
__device__ __forceinline__ void seq_prefix(const float (&x)[C], float (&a)[C])
{
#pragma unroll
  for (int c = 0; c < C; c++) a[c] = 0.f;
#pragma unroll 4
  for (int t = 0; t < N; t++) {
#pragma unroll
    for (int c = 0; c < C; c++) a[c] = dpp_shr1(a[c]) + x[c];
  }
}
// lane 0: T;  lane l: (value of lane l - 1  -  p[l - 1]) + q[l - 1]      (N lanes are wanted)
template <int C, int N>
__device__ __forceinline__ void seq_slide(const float (&T)[C], const float (&p)[C], const float (&q)[C], int lane, float (&b)[C])
{
  float ps[C], qs[C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    ps[c] = dpp_shr1(p[c]); qs[c] = dpp_shr1(q[c]);
    if (lane == 0) { ps[c] = -T[c]; qs[c] = 0.f; }      // (0 - (-T)) + 0 = T exactly
    b[c] = 0.f;
  }
#pragma unroll 4
  for (int t = 0; t < N; t++) {
#pragma unroll
    for (int c = 0; c < C; c++) b[c] = (dpp_shr1(b[c]) - ps[c]) + qs[c];
  }
}

__global__ __launch_bounds__(64) void k_sync_evt(const SyncJob* __restrict__ job, const SyncMark* __restrict__ mark,
                                                 const DetStreamDev* __restrict__ ds, c8p_sync_evt* __restrict__ evt,
                                                 int* __restrict__ valid, int nj)
{
  __shared__ cf sig[C8P_SYNC_BUF + 16];
  const int k = blockIdx.x;
  if (k >= nj) return;
  const int lane = lane_id();
  const SyncJob J = job[k];
  if (J.kind) { if (lane == 0) valid[k] = 0; return; }
  const cf* __restrict__ x = ds[J.stream].x;
  const cf conj0 = mk(ds[J.stream].conj0.x, ds[J.stream].conj0.y);
  const unsigned long long idx_base = ds[J.stream].idx_base;
  // item i of the 240: lane i % 64, register i / 64 (the fourth register only in lanes 0..47)
  cf s[4];
  float pw[4];                                       // |sig[i]| * |sig[i]| as the reference forms it (hypotf squared)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = lane + 64 * r;
    s[r] = i < C8P_SYNC_BUF ? x[J.t + i] : mk(0.f, 0.f);
    pw[r] = habs(s[r]) * habs(s[r]);
    if (i < C8P_SYNC_BUF) sig[i] = s[r];
  }
  // d_conjMultiAvg: the last latch the IDLE state saw — this trigger's mark, or the one an earlier trigger left behind
  cf cavg = conj0;                                   // what the state at the commit point carried in
  for (int kk = k; kk >= (int)J.first; kk--) if (mark[kk].j >= 0) { cavg = mark[kk].conj; break; }
  cf xc[3];                                          // sig[i] * conj(sig[i + 64]), i = lane + 64 r
#pragma unroll
  for (int r = 0; r < 3; r++) xc[r] = cmul(s[r], cconj(s[r + 1]));
  // ltf_autoCorrelation sync_impl.cc:150-175: the three running sums in the reference's order.  Start: the sums over i = 0..63
  float T0[4];
  {
    const float in4[4] = {xc[0].x, xc[0].y, pw[0], pw[1]};
    float pre[4];
    seq_prefix<4, 64>(in4, pre);
#pragma unroll
    for (int c = 0; c < 4; c++) T0[c] = readlane_f(pre[c], 63);
  }
  // offsets 0..63 (lane = offset): value before the update of that offset; update = - item i, + item i + 64
  const float pA[4] = {xc[0].x, xc[0].y, pw[0], pw[1]}, qA[4] = {xc[1].x, xc[1].y, pw[1], pw[2]};
  float A[4];
  seq_slide<4, 64>(T0, pA, qA, lane, A);
  // offsets 64..110 (lane = offset - 64), seeded with the value after offset 63's update
  constexpr int NB = C8P_SYNC_RES - 64;              // 47
  float T1[4];
#pragma unroll
  for (int c = 0; c < 4; c++) T1[c] = (readlane_f(A[c], 63) - readlane_f(pA[c], 63)) + readlane_f(qA[c], 63);
  const float qB[4] = {xc[2].x, xc[2].y, pw[2], pw[3]};
  float B[4];
  seq_slide<4, NB>(T1, qA, qB, lane, B);
  const float ax = A[0], ay = A[1], a1 = A[2], a2 = A[3], bx = B[0], by = B[1], b1 = B[2], b2 = B[3];
  const float tacA = habs(mk(ax, ay)) / sqrtf(a1) / sqrtf(a2);                              // offset lane
  const float tacB = lane < NB ? habs(mk(bx, by)) / sqrtf(b1) / sqrtf(b2) : 0.f;            // offset 64 + lane
  // std::max_element: the first maximum; a NaN is never "greater", and nothing is greater than a NaN at offset 0
  const float NEG = -__builtin_inff();
  float bv = tacA != tacA ? NEG : tacA;
  int bi = lane;
  {
    const float v1 = (lane < NB && tacB == tacB) ? tacB : NEG;
    if (v1 > bv) { bv = v1; bi = 64 + lane; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float ov = __shfl_xor(bv, d, 64);
      const int oi = __shfl_xor(bi, d, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
  }
  const float tac0 = readlane_f(tacA, 0);
  const int mx = tac0 != tac0 ? 0 : bi;
  const float tmax = tac0 != tac0 ? tac0 : bv;
  if (!((double)tmax > 0.5)) { if (lane == 0) valid[k] = 0; return; }                        // sync_impl.cc:93
  const float thr = (float)((double)tmax * 0.8);
  int l = mx, r = mx;
  {
    // plateau around the maximum: nearest offsets below 0.8 max on either side (the maximum itself if there is none)
    const unsigned long long lo0 = __ballot(lane <= mx && tacA < thr), lo1 = __ballot(lane < NB && 64 + lane <= mx && tacB < thr);
    const unsigned long long hi0 = __ballot(lane >= mx && tacA < thr), hi1 = __ballot(lane < NB && 64 + lane >= mx && tacB < thr);
    if (lo1) l = 64 + 63 - __builtin_clzll(lo1); else if (lo0) l = 63 - __builtin_clzll(lo0);
    if (hi0) r = __builtin_ctzll(hi0); else if (hi1) r = 64 + __builtin_ctzll(hi1);
  }
  const int mIndex = (l + r) / 2;
  // ltf_cfo sync_impl.cc:177-192
  const float stf = atan2f(cavg.y, cavg.x) / 16.0f;
  __syncthreads();
  const cf r0 = cmul(sig[mIndex + lane], mk(cosf(lane * stf), sinf(lane * stf)));
  const cf r1 = cmul(sig[mIndex + lane + 64], mk(cosf((lane + 64) * stf), sinf((lane + 64) * stf)));
  const cf xr = cmul(r0, cconj(r1));
  const float xr2[2] = {xr.x, xr.y};
  float sr[2];
  seq_prefix<2, 64>(xr2, sr);
  const cf sum = mk(readlane_f(sr[0], 63), readlane_f(sr[1], 63));
  const float rs1mx = __shfl(mx < 64 ? a1 : b1, mx & 63, 64);
  if (lane == 0) {
    const cf q = cdivf(sum, 64.0f);
    const float ltf = atan2f(q.y, q.x) / 64.0f;
    const double d = (double)tmax;
    c8p_sync_evt e;
    e.index = idx_base + J.t + (unsigned long long)mIndex;
    e.rad = stf + ltf;
    e.snr = (float)(10.0 * log10(d / (1.0 - d)));
    e.rssi = rs1mx / 64.0f;
    e.trig = (uint32_t)(idx_base + J.t);
    evt[k] = e;
    valid[k] = 1;
  }
}
