// c8p_detect.h — packet detection front end on gfx950: examples/presiso.grc (stock GNU Radio blocks)
// -> lib/trigger_impl.cc -> lib/sync_impl.cc, restructured as scans over the raw 20 MS/s stream.
//
//   k_presiso      thread = 8 consecutive samples.  p[n] = x[n-16] conj(x[n]), windows of 48 (complex) and 64 (energy)
//                  assembled from per-thread block sums + suffix sums staged in LDS: 13 + 15 LDS reads per 8 samples,
//                  no re-summation.  HBM-bound: 8 B read + 4 B (ac) [+ 8 B conj when kept] written per sample.
//                  The same kernel reduces its 1984 ac values to the tile summary of the plateau state machine
//                  (whole tile above the threshold? trailing run: start, maximum).
//   (tile carry)   every k_trig_apply workgroup looks back over the tile summaries for the run open at its tile start
//   k_trig_apply   per sample: "ac > 0.3" and "new maximum of this plateau" (trigger bit 0x02) by a segmented
//                  max-scan seeded with the carry; every plateau of >= 21 samples is appended to a candidate list
//   (host)         trigger_impl's countdown logic over the (few) candidates, sync_impl's IDLE/SYNC alternation
//   k_sync_mark    wave = accepted trigger: last "latch conj" mark the sync block saw, and the conj value there
//   k_sync_evt     wave = accepted trigger: LTF autocorrelation over 111 offsets with the reference's running sums in
//                  the reference's operation order (lane 0, bit-exact given the samples), plateau centre, CFO, tags
#pragma once
#include "c8p_dev.h"

#define C8P_DET_TILE 1984           // samples per tile (248 blocks of 8: what one k_presiso workgroup writes)
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
__device__ __forceinline__ TrigSeg seg_of_sample(float a, long long idx)
{
  TrigSeg r;
  const bool ab = a > C8P_AC_THR;                   // NaN (0/0 in silence) compares false, as in the reference
  r.all = ab; r.tstart = ab ? idx : C8P_SEG_NONE; r.tmax = ab ? a : 0.f;
  return r;
}
__device__ __forceinline__ TrigSeg seg_shfl_up(const TrigSeg& v, int d)
{
  TrigSeg r;
  r.all = __shfl_up(v.all, d, 64);
  r.tmax = __shfl_up(v.tmax, d, 64);
  const int lo = __shfl_up((int)(unsigned)(unsigned long long)v.tstart, d, 64);
  const int hi = __shfl_up((int)(unsigned)((unsigned long long)v.tstart >> 32), d, 64);
  r.tstart = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  return r;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ TrigSeg seg_wave_inclusive(TrigSeg v)
{
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const TrigSeg u = seg_shfl_up(v, d);
    if (lane >= d) v = seg_combine(u, v);
  }
  return v;
}
// exclusive prefix of `mine` over the threads of a workgroup (blockDim.x / 64 waves <= 16); `total` = all of them
__device__ __forceinline__ TrigSeg seg_block_exclusive(TrigSeg mine, TrigSeg* sh /*[16]*/, TrigSeg& total)
{
  const int lane = lane_id(), w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const TrigSeg inc = seg_wave_inclusive(mine);
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  TrigSeg pre = seg_identity();
  total = seg_identity();
  for (int k = 0; k < nw; k++) { if (k == w) pre = total; total = seg_combine(total, sh[k]); }
  TrigSeg ex = seg_shfl_up(inc, 1);
  if (lane == 0) ex = seg_identity();
  __syncthreads();
  return seg_combine(pre, ex);
}

// ------------------------------------------------------------------------------------------------ presiso
// One workgroup = 256 blocks of 8 samples: blocks 0..7 only supply the 16-sample delay and the window history
// (48 products = 6 blocks back, 64 energies = 8 blocks back), blocks 8..255 (248 blocks = 1984 samples) are written.  Global traffic is fully coalesced: samples enter
// through LDS transposed ([sample % 8][block], row stride 264 = 8 mod 16 eight-byte banks: conflict-free both for the
// coalesced writes and for "thread t reads row j, column t - m"), results leave the same way.
#define C8P_PRE_HALO 8
#define C8P_PRE_OUT_BLOCKS (256 - C8P_PRE_HALO)
#define C8P_PRE_RS 264
// x points at the first NEW item; items x[-hist .. -1] exist (history before the commit point), x[g] = 0 for g < -hist
__global__ __launch_bounds__(256) void k_presiso(const cf* __restrict__ x, long long hist, unsigned long long n, float* __restrict__ ac,
                                                 cf* __restrict__ conj, int keep, TrigSeg* __restrict__ summ, int x_aligned16)
{
  __shared__ TrigSeg shs[16];
  __shared__ cf sX[9 * C8P_PRE_RS];         // samples, later sufP[j][t] = sum_{i=j..7} p_i of block t  ([8] = 0)
  __shared__ float sE[9 * C8P_PRE_RS];      // sufE[j][t], later the ac values on their way out
  const int tid = threadIdx.x;
  const long long blk0 = (long long)blockIdx.x * C8P_PRE_OUT_BLOCKS - C8P_PRE_HALO;   // first block of this workgroup
  const long long base = blk0 * 8;
  if (x_aligned16 && base >= -hist && base + 2048 <= (long long)n) {
    // interior workgroup: 16-byte loads, thread tid takes samples base + 2 tid + 512 k and the next one (base is a
    // multiple of 8, so these are 16-byte aligned whenever x is: the host passes that in)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = 2 * tid + 512 * k;
      const float4 v = *(const float4*)(x + base + i);
      sX[(i & 7) * C8P_PRE_RS + (i >> 3)] = mk(v.x, v.y);
      sX[((i + 1) & 7) * C8P_PRE_RS + ((i + 1) >> 3)] = mk(v.z, v.w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) {           // coalesced: thread tid loads sample base + tid + 256 k
      const int i = tid + 256 * k;
      const long long g = base + i;
      sX[(i & 7) * C8P_PRE_RS + (i >> 3)] = (g >= -hist && g < (long long)n) ? x[g] : mk(0.f, 0.f);
    }
  }
  __syncthreads();
  cf p[8];
  float e[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const cf a = sX[j * C8P_PRE_RS + tid];
    const cf d = tid >= 2 ? sX[j * C8P_PRE_RS + tid - 2] : mk(0.f, 0.f);      // x[g - 16]: two blocks back
    p[j] = mk(d.x * a.x + d.y * a.y, d.y * a.x - d.x * a.y);                    // x[g-16] * conj(x[g])
    e[j] = a.x * a.x + a.y * a.y;
  }
  __syncthreads();
  {
    cf s = mk(0.f, 0.f);
    float se = 0.f;
    sX[8 * C8P_PRE_RS + tid] = s; sE[8 * C8P_PRE_RS + tid] = 0.f;
#pragma unroll
    for (int j = 7; j >= 0; j--) { s = cadd(s, p[j]); se += e[j]; sX[j * C8P_PRE_RS + tid] = s; sE[j * C8P_PRE_RS + tid] = se; }
  }
  __syncthreads();
  float acv[8];
  cf Pv[8];
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
      const cf P = cadd(cadd(cp, sX[(j + 1) * C8P_PRE_RS + tid - 6]), pre);   // tail of block b-6 + blocks b-5..b-1 + head of block b
      const float E = (ce + sE[(j + 1) * C8P_PRE_RS + tid - 8]) + pe;
      // correctly rounded sqrt and quotient (-fhip-fp32-correctly-rounded-divide-sqrt), i.e. IEEE sqrtf(.) / e as VOLK + blocks_divide_ff compute it: ac feeds
      // hard comparisons (ac > 0.3, ac > running maximum, trigger_impl.cc:77-85), so given the same window sums the trigger
      // bytes are those of the scalar CPU restatement bit for bit (tests/test_detect_frontend.py: integer-valued torture stream straddling 0.3)
      acv[j] = sqrtf(P.x * P.x + P.y * P.y) / E;
      Pv[j] = P;
    }
  }
  {                                         // tile summary for the plateau scan: ordered reduction of the 1984 samples
    TrigSeg mine = seg_identity();
    if (tid >= C8P_PRE_HALO) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const long long g = base + tid * 8 + j;
        if (g >= 0 && (unsigned long long)g < n) mine = seg_combine(mine, seg_of_sample(acv[j], g));
      }
    }
    TrigSeg total;
    (void)seg_block_exclusive(mine, shs, total);     // (contains the barriers that also order the LDS reuse below)
    if (tid == 0) summ[blockIdx.x] = total;
  }
  if (tid >= C8P_PRE_HALO) {
#pragma unroll
    for (int j = 0; j < 8; j++) sE[j * C8P_PRE_RS + tid] = acv[j];
  }
  __syncthreads();
  if (base >= 0 && base + 2048 <= (long long)n) {
    // interior workgroup: 16-byte stores, thread tid stores samples base + 4 tid + 1024 k .. +3 (base is a multiple of 8)
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int i = 4 * tid + 1024 * k;
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
      const int i = tid + 256 * k;
      const long long g = base + i;
      if ((i >> 3) >= C8P_PRE_HALO && (unsigned long long)g < n && g >= 0) ac[g] = sE[(i & 7) * C8P_PRE_RS + (i >> 3)];
    }
  }
  if (keep && tid >= C8P_PRE_HALO) {        // test / debug path only
#pragma unroll
    for (int j = 0; j < 8; j++) { const long long g = base + tid * 8 + j; if ((unsigned long long)g < n) conj[g] = Pv[j]; }
  }
}

// ------------------------------------------------------------------------------------------------ trigger scan
// one workgroup of 1024 threads: carry[t] = combine(init, summ[0..t-1]); carry[ntile] = the state after the last tile
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
// tile = the 1984 samples one k_presiso workgroup wrote; thread = 8 consecutive samples (threads 248..255 idle)
// The plateau state at the start of a tile = the combination of everything before it.  A tile that is not entirely above the
// threshold fixes that state by itself, so instead of a global scan kernel every workgroup looks BACK over the tile
// summaries k_presiso wrote: 64 tiles per step (lane = tile, coalesced), until a chunk contains such a tile or the start of
// the segment (`init`) is reached.  Plateaus are a few hundred samples long: one step in practice.
__device__ __forceinline__ TrigSeg trig_lookback(const TrigSeg* __restrict__ summ, long long tile, const TrigSeg& init)
{
  const int lane = lane_id();
  TrigSeg acc = seg_identity();                       // tiles [base, tile)
  for (long long base = tile; ; base -= 64) {
    if (base <= 0) return seg_combine(init, acc);
    const long long t = base - 64 + lane;
    const TrigSeg chunk = seg_bcast(seg_wave_inclusive(t >= 0 ? summ[t] : seg_identity()), 63);
    acc = seg_combine(chunk, acc);
    if (!chunk.all) return acc;                        // independent of anything earlier (seg_combine with R.all == 0)
  }
}

__global__ __launch_bounds__(256) void k_trig_apply(const float* __restrict__ ac, unsigned long long n, const TrigSeg* __restrict__ summ,
                                                    TrigSeg init, TrigSeg* __restrict__ fin,
                                                    uint8_t* __restrict__ trig, TrigCand* __restrict__ cand,
                                                    unsigned int* __restrict__ n_cand, unsigned int cap)
{
  __shared__ TrigSeg sh[16];
  __shared__ TrigSeg s_cin;
  float a[8];
  const bool live = threadIdx.x < C8P_DET_TILE / 8;
  const long long n0 = (long long)blockIdx.x * C8P_DET_TILE + threadIdx.x * 8;
  TrigSeg mine = seg_identity();
  if (live) {
    if ((unsigned long long)(n0 + 7) < n) {            // n0 * 4 bytes is 32-byte aligned
      const float4 v0 = *(const float4*)(ac + n0), v1 = *(const float4*)(ac + n0 + 4);
      a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) a[j] = (unsigned long long)(n0 + j) < n ? ac[n0 + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) if ((unsigned long long)(n0 + j) < n) mine = seg_combine(mine, seg_of_sample(a[j], n0 + j));
  }
  if (threadIdx.x < 64) {
    const TrigSeg c0 = trig_lookback(summ, (long long)blockIdx.x, init);
    if (threadIdx.x == 0) {
      s_cin = c0;
      if (blockIdx.x == gridDim.x - 1) *fin = seg_combine(c0, summ[blockIdx.x]);      // the state at the end of the segment
    }
  }
  __syncthreads();
  const TrigSeg cin = s_cin;
  {
    // idle air: nothing in the tile is above the threshold and no plateau is open at its start -> all bytes 0, no scan
    bool anyab = false;
    if (live) {
#pragma unroll
      for (int j = 0; j < 8; j++) anyab = anyab || (a[j] > C8P_AC_THR);    // a[j] = 0 past the end of the stream
    }
    if (!__syncthreads_or(anyab) && !seg_has_run(cin)) {
      if (live) {
        if ((unsigned long long)(n0 + 7) < n) *(unsigned long long*)(trig + n0) = 0ull;
        else for (int j = 0; j < 8 && (unsigned long long)(n0 + j) < n; j++) trig[n0 + j] = 0;
      }
      return;
    }
  }
  TrigSeg total;
  const TrigSeg ex = seg_block_exclusive(mine, sh, total);
  if (!live) return;
  const TrigSeg st = seg_combine(cin, ex);                 // everything before this thread's first sample
  // walk the 8 samples with the plateau state: run start and running maximum
  long long rs = seg_has_run(st) ? st.tstart : C8P_SEG_NONE;
  float rmax = seg_has_run(st) ? st.tmax : 0.f;
  unsigned long long packed = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const long long g = n0 + j;
    if ((unsigned long long)g >= n) break;
    unsigned b = 0;
    if (a[j] > C8P_AC_THR) {
      b = 0x04;
      if (rs == C8P_SEG_NONE) { rs = g; rmax = 0.f; }
      if (a[j] > rmax) { rmax = a[j]; b |= 0x02; }     // trigger_impl.cc:80-85 (d_conjAc restarts at 0 with every plateau)
    } else {
      if (rs != C8P_SEG_NONE && g - rs >= 21) {        // a plateau long enough to arm the countdown ended here
        const unsigned k = atomicAdd(n_cand, 1u);
        if (k < cap) { cand[k].s = rs; cand[k].e = g; }
      }
      rs = C8P_SEG_NONE; rmax = 0.f;
    }
    packed |= (unsigned long long)b << (8 * j);
    if ((unsigned long long)g == n - 1 && rs != C8P_SEG_NONE && g + 1 - rs >= 21) {   // plateau still open at the end of the stream
      const unsigned k = atomicAdd(n_cand, 1u);
      if (k < cap) { cand[k].s = rs; cand[k].e = (long long)n; }
    }
  }
  if ((unsigned long long)(n0 + 7) < n) *(unsigned long long*)(trig + n0) = packed;
  else for (int j = 0; j < 8 && (unsigned long long)(n0 + j) < n; j++) trig[n0 + j] = (uint8_t)(packed >> (8 * j));
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
};
struct SyncMark {
  long long j;              // last item in [seen, t) with trigger bit 0x02, -1: none
  cf conj;                  // presiso output 1 at j
};

__global__ __launch_bounds__(64) void k_sync_mark(const SyncJob* __restrict__ job, const uint8_t* __restrict__ trig,
                                                  const cf* __restrict__ x, long long hist, SyncMark* __restrict__ out, int nj)
{
  const int k = blockIdx.x;
  if (k >= nj) return;
  const int lane = lane_id();
  const SyncJob J = job[k];
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

__global__ __launch_bounds__(64) void k_sync_evt(const SyncJob* __restrict__ job, const SyncMark* __restrict__ mark,
                                                 const cf* __restrict__ x, c8p_sync_evt* __restrict__ evt,
                                                 int* __restrict__ valid, int nj, cf conj0, unsigned long long idx_base)
{
  __shared__ cf sig[C8P_SYNC_BUF];
  __shared__ float pw[C8P_SYNC_BUF];                 // |sig[i]| * |sig[i]| as the reference forms it (hypotf squared)
  __shared__ cf xc[C8P_SYNC_BUF - 64];               // sig[i] * conj(sig[i + 64])
  __shared__ cf rms[C8P_SYNC_RES];                   // the three running sums before each update
  __shared__ float rs1[C8P_SYNC_RES], rs2[C8P_SYNC_RES];
  __shared__ float tac[C8P_SYNC_RES];
  __shared__ cf rot128[128];
  const int k = blockIdx.x;
  if (k >= nj) return;
  const int lane = lane_id();
  const SyncJob J = job[k];
  for (int i = lane; i < C8P_SYNC_BUF; i += 64) { const cf v = x[J.t + i]; sig[i] = v; pw[i] = habs(v) * habs(v); }
  // d_conjMultiAvg: the last latch the IDLE state saw — this trigger's mark, or the one an earlier trigger left behind
  cf cavg = conj0;                                   // what the state at the commit point carried in
  for (int kk = k; kk >= 0; kk--) if (mark[kk].j >= 0) { cavg = mark[kk].conj; break; }
  __syncthreads();
  for (int i = lane; i < C8P_SYNC_BUF - 64; i += 64) xc[i] = cmul(sig[i], cconj(sig[i + 64]));
  __syncthreads();
  if (lane == 0) {
    // ltf_autoCorrelation sync_impl.cc:150-175: the reference's running sums, updated in the reference's order; only
    // these additions are sequential, the terms above and the quotients below are computed by all lanes
    cf ms = mk(0.f, 0.f);
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < 64; i++) { ms = cadd(ms, xc[i]); s1 += pw[i]; s2 += pw[i + 64]; }
    for (int i = 0; i < C8P_SYNC_RES; i++) {
      rms[i] = ms; rs1[i] = s1; rs2[i] = s2;
      ms = csub(ms, xc[i]); s1 -= pw[i]; s2 -= pw[i + 64];
      ms = cadd(ms, xc[i + 64]); s1 += pw[i + 64]; s2 += pw[i + 128];
    }
  }
  __syncthreads();
  for (int i = lane; i < C8P_SYNC_RES; i += 64) tac[i] = habs(rms[i]) / sqrtf(rs1[i]) / sqrtf(rs2[i]);
  __syncthreads();
  int ok = 0, mIndex = 0, mx = 0;
  if (lane == 0) {
    for (int j = 1; j < C8P_SYNC_RES; j++) if (tac[j] > tac[mx]) mx = j;     // std::max_element: first maximum
    if ((double)tac[mx] > 0.5) {                                             // sync_impl.cc:93
      const float thr = (float)((double)tac[mx] * 0.8);
      int l = mx, r = mx;
      for (int j = mx; j >= 0; j--) if (tac[j] < thr) { l = j; break; }
      for (int j = mx; j < C8P_SYNC_RES; j++) if (tac[j] < thr) { r = j; break; }
      mIndex = (l + r) / 2;
      ok = 1;
    }
  }
  ok = __shfl(ok, 0, 64); mIndex = __shfl(mIndex, 0, 64); mx = __shfl(mx, 0, 64);
  if (!ok) { if (lane == 0) valid[k] = 0; return; }
  // ltf_cfo sync_impl.cc:177-192
  const float stf = atan2f(cavg.y, cavg.x) / 16.0f;
  for (int i = lane; i < 128; i += 64) rot128[i] = cmul(sig[mIndex + i], mk(cosf(i * stf), sinf(i * stf)));
  __syncthreads();
  if (lane < 64) xc[lane] = cmul(rot128[lane], cconj(rot128[lane + 64]));
  __syncthreads();
  if (lane == 0) {
    cf sum = mk(0.f, 0.f);
    for (int i = 0; i < 64; i++) sum = cadd(sum, xc[i]);
    const cf q = cdivf(sum, 64.0f);
    const float ltf = atan2f(q.y, q.x) / 64.0f;
    const double d = (double)tac[mx];
    c8p_sync_evt e;
    e.index = idx_base + J.t + (unsigned long long)mIndex;
    e.rad = stf + ltf;
    e.snr = (float)(10.0 * log10(d / (1.0 - d)));
    e.rssi = rs1[mx] / 64.0f;
    e.trig = (uint32_t)(idx_base + J.t);
    evt[k] = e;
    valid[k] = 1;
  }
}
