// c8p_engine.hip — host side of libc8p_rx.so: the C ABI of include/c8p_rx.h on top of the gfx950 kernels.
//
// One ctx owns a HIP stream (or borrows the caller's), the constant tables, and a set of device arenas that
// only ever grow.  A submit computes the per-packet arena layout on the host from the packet descriptors
// (sizes depend only on n_items / trellis, never on device results), uploads descriptors + layout in one pinned
// staging copy, and enqueues the kernels back to back — no host round trip between them.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>
#include <memory>

#include <algorithm>
#include <chrono>

#include "c8p_kernels.h"
#include "c8p_detect.h"
#include "c8p_tx.h"

#define HIPCHK(ctx, call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return C8P_EHIP;                                                                     \
    }                                                                                      \
  } while (0)

// ------------------------------------------------------------------------------------------------ uploads of the engine's own staging
// A submit's descriptors travel from PINNED staging memory (h_stage) to the device.  Through a DMA engine (hipMemcpyAsync) that is a
// handful of read requests in flight; beside a bulk ingest that keeps megabytes of reads outstanding on the same link (c8p_copy_gather)
// each of them waits its turn and a few hundred KB take milliseconds (NOTES/frontend.md).  k_pull has every 16 bytes requested at once
// by its own lane: one trip through the link's queue for the whole table.  C8P_UPLOAD=dma: the copy engines, as before round 4.
__global__ void k_pull(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, size_t tail_bytes)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
  else if (i - n16 < tail_bytes) ((unsigned char*)(dst + n16))[i - n16] = ((const unsigned char*)(src + n16))[i - n16];
}
static hipError_t upload(hipStream_t st, void* dst, const void* src_pinned, size_t bytes)
{
  static const bool dma = [] { const char* e = getenv("C8P_UPLOAD"); return e && !strcmp(e, "dma"); }();
  if (bytes == 0) return hipSuccess;
  if (dma || (((uintptr_t)dst | (uintptr_t)src_pinned) & 15u) || bytes < 1024)
    return hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, st);
  const size_t n16 = bytes / 16, tail = bytes % 16, thr = n16 + tail;
  hipLaunchKernelGGL(k_pull, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, (uint4*)dst, (const uint4*)src_pinned, n16, tail);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ tables
// Generated from the 802.11 formulas, not copied; tests compare them with SHA-256
// digests of the reference's literals and with the Python toolbox's interleavers.
static void build_deint(int16_t* map, int ncbps, int nbpsc, int ncol, bool second_stream)
{
  const int s = std::max(nbpsc / 2, 1);
  const int nrow = ncbps / ncol;
  for (int k = 0; k < ncbps; k++) {
    const int i = nrow * (k % ncol) + k / ncol;
    const int j = s * (i / s) + (i + ncbps - (ncol * i) / ncbps) % s;
    int r = j;
    if (second_stream) r = ((j - 2 * 11 * nbpsc) % ncbps + ncbps) % ncbps;
    map[r] = (int16_t)k;
  }
}

static uint32_t h_multmodp(uint32_t a, uint32_t b)
{
  uint32_t m = 1u << 31, p = 0;
  for (int i = 0; i < 32; i++) {
    if (a & m) p ^= b;
    m >>= 1;
    b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
  }
  return p;
}

static void build_host_tables(DevTables& T);
static const DevTables& host_tables()      // built once; contexts may be created from several threads (one per GNU Radio block)
{
  static DevTables T;
  static std::once_flag once;
  std::call_once(once, [] { build_host_tables(T); });
  return T;
}
static void build_host_tables(DevTables& T)
{
  memset(&T, 0, sizeof(T));
  static const int nbL[4] = {1, 2, 4, 6}, nbNL[5] = {1, 2, 4, 6, 8};
  for (int m = 0; m < 4; m++) build_deint(T.deintL[m], 48 * nbL[m], nbL[m], 16, false);
  int16_t ss2[416];
  for (int m = 0; m < 5; m++) {
    const int nb = nbNL[m], n = 52 * nb, sz = std::max(nb / 2, 1);
    build_deint(T.deintNL[m], n, nb, 13, false);
    build_deint(ss2, n, nb, 13, true);
    // fuse procSymDepasNL: position r of stream ss lands at ((r / sz) * 2 + ss) * sz + r % sz
    for (int i = 0; i < n; i++) {
      const int r0 = T.deintNL[m][i], r1 = ss2[i];
      T.deint2ss[0][m][i] = (int16_t)(((r0 / sz) * 2 + 0) * sz + r0 % sz);
      T.deint2ss[1][m][i] = (int16_t)(((r1 / sz) * 2 + 1) * sz + r1 % sz);
    }
  }
  for (int i = 0; i < 52; i++) { T.deintSigB[i] = (int8_t)T.deintNL[0][i]; }
  for (int i = 0; i < 52; i++) { T.intlSigB[T.deintSigB[i]] = (int8_t)i; }
  {
    int rank = 0, dsc[64];
    for (int i = 0; i < 64; i++) dsc[i] = -1;
    for (int sc = -26; sc <= 26; sc++) {
      if (sc == 0 || sc == -21 || sc == -7 || sc == 7 || sc == 21) continue;
      dsc[(sc + 64) % 64] = rank++;
    }
    for (int i = 0; i < 64; i++) {
      const int v = dsc[i] < 0 ? -1 : T.deintL[0][dsc[i]];
      T.sigDemapNL[i] = (int8_t)v;
      T.sigDemapNL[i + 64] = (int8_t)(v < 0 ? -1 : v + 48);
    }
  }
  {
    int st = 0x7f;
    for (int k = 0; k < 128; k++) T.scrPhase[k] = 0;
    for (int n = 0; n < 127; n++) {
      T.scrPhase[st] = (uint8_t)n;
      const int fb = ((st >> 6) & 1) ^ ((st >> 3) & 1);
      T.scrSeq[n] = (uint8_t)fb;
      T.pilotP[n] = fb ? -1.0f : 1.0f;
      st = ((st << 1) & 0x7e) | fb;
    }
    for (int i = 0; i < 256; i++) if (T.scrSeq[i % 127]) T.scrW[i >> 6] |= 1ull << (i & 63);
  }
  {
    static const signed char lltf[53] = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                                         1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};
    for (int sc = -26; sc <= 26; sc++) { T.ltfNL[(sc + 64) % 64] = (float)lltf[sc + 26]; T.ltfL[(sc + 64) % 64] = (float)lltf[sc + 26]; }
    T.ltfNL[36] = 1; T.ltfNL[37] = 1; T.ltfNL[27] = -1; T.ltfNL[28] = -1;
    for (int i = 0; i < 64; i++) T.ltfNLvht22[i] = T.ltfNL[i];
    const int pil[4] = {7, 21, 43, 57};
    for (int k = 0; k < 4; k++) T.ltfNLvht22[pil[k]] = -T.ltfNL[pil[k]];
  }
  for (int k = 0; k < 32; k++) {
    const double a = 2.0 * M_PI * (double)k / 64.0;
    T.tw[k] = make_double2(cos(a), -sin(a));
  }
  T.tw[0] = make_double2(1.0, 0.0);
  T.tw[16] = make_double2(0.0, -1.0);
  for (int i = 0; i < 64; i++) {
    int r = 0;
    for (int b = 0; b < 6; b++) if (i & (1 << b)) r |= 1 << (5 - b);
    T.bitrev6[i] = (uint8_t)r;
  }
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
    T.crc32[n] = c;
  }
  {
    uint32_t p = 1u << 30;
    T.x2n[0] = p;
    for (int n = 1; n < 32; n++) T.x2n[n] = p = h_multmodp(p, p);
    T.x8n[0] = 1u << 31;                                   // the polynomial 1
    for (int n = 1; n < C8P_X8N; n++) T.x8n[n] = h_multmodp(T.x8n[n - 1], T.x2n[3]);      // times x^8
  }
}

extern "C" int c8p_table(int which, int sub, int32_t* dst, int cap)
{
  const DevTables& T = host_tables();
  static const int nbL[4] = {1, 2, 4, 6}, nbNL[5] = {1, 2, 4, 6, 8};
  std::vector<int32_t> v;
  switch (which) {
    case C8P_TAB_DEINT_L: if (sub < 0 || sub > 3) return C8P_EINVAL; for (int i = 0; i < 48 * nbL[sub]; i++) v.push_back(T.deintL[sub][i]); break;
    case C8P_TAB_DEINT_NL: if (sub < 0 || sub > 4) return C8P_EINVAL; for (int i = 0; i < 52 * nbNL[sub]; i++) v.push_back(T.deintNL[sub][i]); break;
    case C8P_TAB_DEINT_NL2: {
      if (sub < 0 || sub > 4) return C8P_EINVAL;
      // undo the fused stream de-parse to expose the plain stream-2 de-interleaver
      const int nb = nbNL[sub], sz = std::max(nb / 2, 1);
      for (int i = 0; i < 52 * nb; i++) { const int pos = T.deint2ss[1][sub][i]; v.push_back(((pos / sz - 1) / 2) * sz + pos % sz); }
      break;
    }
    case C8P_TAB_SIG_DEMAP_NL: for (int i = 0; i < 128; i++) v.push_back(T.sigDemapNL[i]); break;
    case C8P_TAB_DEINT_SIGB: for (int i = 0; i < 52; i++) v.push_back(T.deintSigB[i]); break;
    case C8P_TAB_SV_OUT:
      for (int i = 0; i < 64; i++) for (int b = 0; b < 2; b++) {
        int x0 = b, x1 = (i >> 5) & 1, x2 = (i >> 4) & 1, x3 = (i >> 3) & 1, x5 = (i >> 1) & 1, x6 = i & 1;
        v.push_back(2 * (x0 ^ x2 ^ x3 ^ x5 ^ x6) + (x0 ^ x1 ^ x2 ^ x3 ^ x6));
      }
      break;
    case C8P_TAB_PILOT_P: for (int i = 0; i < 127; i++) v.push_back((int)T.pilotP[i]); break;
    case C8P_TAB_LTF_NL: for (int i = 0; i < 64; i++) v.push_back((int)T.ltfNL[i]); break;
    case C8P_TAB_LTF_NL_VHT22: for (int i = 0; i < 64; i++) v.push_back((int)T.ltfNLvht22[i]); break;
    default: return C8P_EINVAL;
  }
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) dst[i] = v[i];
  return (int)v.size();
}

// ------------------------------------------------------------------------------------------------ ctx
struct Buf {
  void* p = nullptr;
  size_t cap = 0;
};

// One in-flight batch: its own stream, staging and result arenas.  A ctx owns several slots and rotates through
// them, so that the latency-bound kernels of batch i+1 (upload, classify, symbol demod) overlap the VALU-bound
// decoder of batch i on the same GPU.
struct Slot {
  hipStream_t stream = nullptr;
  void* h_stage = nullptr; size_t h_stage_cap = 0;   // pinned
  Buf d_stage;                                       // descriptors + layout (+ sym_base)
  Buf d_iq, d_chan, d_llr_in, d_mu_in;               // staging of caller host arenas
  Buf d_tags, d_outs, d_H, d_mimo, d_muchan, d_llr, d_surv, d_bits, d_sbits, d_pdu, d_flags, d_sym2pkt, d_negsym, d_done4;
  Buf d_sig_raw, d_sig_pkt, d_sig_iq, d_sig_chan, d_sig_kept;    // signal front-end
  Buf d_det_raw, d_det_ac, d_det_trig, d_det_conj, d_det_zero, d_det_carry, d_det_cand, d_det_tcand, d_det_toff, d_det_job, d_det_mark,
      d_det_evt, d_det_valid, d_det_fire, d_det_ds, d_det_soff;            // detection front-end
  Buf d_tx_plan, d_tx_psdu, d_tx_coded, d_tx_sig, d_tx_u2p, d_tx_iq, d_tx_desc, d_tx_bfq;   // transmit chain
  std::vector<TrigCand> h_det_cand;                  // host-side work lists of c8p_detect (kept: no allocation per call)
  std::vector<unsigned long long> h_det_fire;
  std::vector<SyncJob> h_det_jobs;
  int n_pkt = 0;
  unsigned ticket = 0;                               // c8p_ticket() of the submit that filled this slot (0: never used)
  size_t used[C8P_BUF_COUNT] = {};
  const void* ptr[C8P_BUF_COUNT] = {};
};

struct c8p_ctx {
  int device = 0;
  bool own_stream = false;
  int profile = 0, keep_bits = 0, keep_llr = 0;
  unsigned seq = 0;                                  // submits so far (profile sampling)
  int exact_only = 0;                                // C8P_EXACT_VITERBI=1: skip the lean forward pass (debug / A-B)
  unsigned det_spec = 8192;                          // c8p_detect: candidates fetched with the first sync
  std::string err;
  DevTables* d_tab = nullptr;
  std::vector<Slot> slots;
  Slot* s = nullptr;                                 // slot of the most recent submit
  int cur = -1;
  // profiling
  double k_ms[C8P_K_COUNT] = {};
  int64_t k_cnt[C8P_K_COUNT] = {};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
  std::vector<hipEvent_t> ev_pool;
  // c8p_counter
  int64_t n_grow = 0, dev_bytes = 0, host_bytes = 0, det_giveups = 0, det_ticket_scans = 0;
  // c8p_copy_*: a stream of its own, created on first use
  hipStream_t copy_stream = nullptr;                 // c8p_copy_async, c8p_copy_gather: confined to a few CUs (copy_stream_of)
  hipStream_t span_stream = nullptr;                 // c8p_copy_spans: device-to-device, the whole chip
  hipEvent_t ev_before_spans = nullptr, ev_spans = nullptr;
  bool spans_pending = false;                        // a span copy the next mark / wait has to include
  int copy_cus = 0;                                  // CUs the copy stream may use (0: all)
  int opt_copy_cus = -1;                             // c8p_option(C8P_OPT_COPY_CUS): what the caller asked for (-1: the defaults)
  Buf d_spans; void* h_spans = nullptr; size_t h_spans_cap = 0;
  void* d_spans2 = nullptr; void* h_spans2 = nullptr; size_t spans2_cap = 0;      // c8p_copy_spans' own table
  std::mutex copy_mu;                                // marks against copies enqueued from another thread
  hipEvent_t copy_ev[64] = {};
  uint64_t copy_marks = 0;
  hipStream_t out_stream = nullptr;                  // c8p_copy_gather with C8P_PIECE_TO_HOST pieces: device -> host, the link's other direction
  // ... what a mark records there, when something was enqueued there since the last mark (out_dirty, set AFTER the launch); out_of[k]:
  // what mark k waits for on that stream — its own event, or the newest one an earlier mark recorded (out_last): a mark covers
  // everything enqueued before it on all three streams, whichever mark recorded it
  hipEvent_t out_ev[64] = {};
  hipEvent_t out_of[64] = {};
  hipEvent_t out_last = nullptr;
  bool out_dirty = false;
  // (every record there has a generation number; a waiter that has seen generation g complete never asks about g or older again: the
  // inherited events are usually long done, and a query costs a microsecond or two thousands of times a second)
  uint64_t out_gen = 0, out_done_gen = 0, b_gen = 0, b_done_gen = 0;
  uint64_t out_of_gen[64] = {}, b_of_gen[64] = {};
  hipEvent_t ev_fence = nullptr;                     // c8p_copy_fence: what the next submit's stream waits for
  bool fence_pending = false;
  // C8P_PIECE_UNORDERED gathers alternate between the copy stream and this twin of it (same CU mask): gather k + 1 is launched and
  // ramps up while gather k drains, instead of starting when it has finished.  Marks, fences and waits cover both.
  hipStream_t copy_stream_b = nullptr;
  hipEvent_t b_ev[64] = {};
  hipEvent_t b_of[64] = {};
  hipEvent_t b_last = nullptr;
  bool b_dirty = false, b_next = false;
  hipEvent_t ev_async_b = nullptr;                   // c8p_copy_async: the copy stream behind its twin
  hipEvent_t ev_fence_b = nullptr;
  bool fence_b_pending = false;
  uint32_t copy_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

static bool grow_log() { static const bool on = getenv("C8P_GROW_LOG") != nullptr; return on; }
#define ensure(c, b, bytes) ensure_named(c, b, bytes, #b)
static int ensure_named(c8p_ctx* c, Buf& b, size_t bytes, const char* name)
{
  if (bytes <= b.cap) return C8P_OK;
  if (grow_log()) fprintf(stderr, "[c8p grow] ctx %p submit %u: %s %zu -> %zu bytes\n", (void*)c, c->seq, name, b.cap, bytes);
  if (b.p) { HIPCHK(c, hipStreamSynchronize(c->s->stream)); HIPCHK(c, hipFree(b.p)); b.p = nullptr; c->dev_bytes -= (int64_t)b.cap; b.cap = 0; }
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { c->err = std::string("hipMalloc: ") + hipGetErrorString(e); return C8P_ENOMEM; }
  b.cap = want;
  c->n_grow++; c->dev_bytes += (int64_t)want;
  return C8P_OK;
}
static int ensure_host(c8p_ctx* c, size_t bytes)
{
  if (bytes <= c->s->h_stage_cap) return C8P_OK;
  if (grow_log()) fprintf(stderr, "[c8p grow] ctx %p submit %u: pinned staging %zu -> %zu bytes\n", (void*)c, c->seq, c->s->h_stage_cap, bytes);
  if (c->s->h_stage) {
    HIPCHK(c, hipStreamSynchronize(c->s->stream)); HIPCHK(c, hipHostFree(c->s->h_stage));
    c->s->h_stage = nullptr; c->host_bytes -= (int64_t)c->s->h_stage_cap; c->s->h_stage_cap = 0;
  }
  size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(c, hipHostMalloc(&c->s->h_stage, want, hipHostMallocDefault));
  c->s->h_stage_cap = want;
  c->n_grow++; c->host_bytes += (int64_t)want;
  return C8P_OK;
}

static hipEvent_t get_event(c8p_ctx* c)
{
  if (!c->ev_pool.empty()) { hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
struct KTimer {   // brackets one launch with events on the stream it is launched on when profiling is on
  c8p_ctx* c; int k; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on = false;
  KTimer(c8p_ctx* c_, int k_, hipStream_t st_ = nullptr) : c(c_), k(k_), st(st_ ? st_ : c_->s->stream)
  {
    // profile = N > 1 samples: kernel k is bracketed on every N-th submit only (staggered by k), which keeps the event
    // packets (~5 % of a 0.85 ms step when every launch is bracketed) out of throughput measurements
    on = c->profile > 0 && ((c->seq + (unsigned)k) % (unsigned)c->profile) == 0;
    if (on) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, st); }
  }
  ~KTimer() { if (on) { (void)hipEventRecord(b, st); c->pending.push_back({k, {a, b}}); } }
};
static void drain_events(c8p_ctx* c)
{
  // only finished launches are folded in; launches of batches still in flight on other slots stay pending
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> keep;
  for (auto& pe : c->pending) {
    if (hipEventQuery(pe.second.second) != hipSuccess) { keep.push_back(pe); continue; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.second.first, pe.second.second) == hipSuccess) { c->k_ms[pe.first] += ms; c->k_cnt[pe.first]++; }
    c->ev_pool.push_back(pe.second.first); c->ev_pool.push_back(pe.second.second);
  }
  c->pending.swap(keep);
}

extern "C" c8p_ctx* c8p_create(const c8p_cfg* cfg)
{
  if (!cfg || cfg->abi_version != C8P_ABI_VERSION) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return nullptr;
  if (hipSetDevice(cfg->device) != hipSuccess) return nullptr;
  c8p_ctx* c = new c8p_ctx();
  c->device = cfg->device; c->profile = cfg->profile < 0 ? 0 : cfg->profile; c->keep_bits = cfg->keep_bits;
  c->keep_llr = cfg->keep_llr == 2 ? 2 : cfg->keep_llr != 0;
  { const char* e = getenv("C8P_RX_LLR"); if (e && atoi(e) > 0 && !c->keep_llr) c->keep_llr = atoi(e) == 2 ? 2 : 1; }   // (A/B runs: the env switch can only turn the LLRs ON)
  { const char* e = getenv("C8P_EXACT_VITERBI"); c->exact_only = (e && e[0] == '1') ? 1 : 0; }
  {
    // a caller-provided stream means strict stream order: one slot.  Otherwise C8P_SLOTS (default 3) engine-owned streams.
    // Round 1 (three kernels per step, LLRs through HBM) measured 2 / 3 / 4 / 5 / 6 slots: 4.64 / 4.95 / 5.15 / 4.64 / 4.88 M packets/s;
    // with the fused on-chip kernel of round 2 three batches in flight fill the machine: 2 / 3 / 4 / 6 / 8 slots = 6.47 / 6.71 / 6.62 /
    // 6.56 / 6.69 M (3 against 4, alternated four times on one box: +1.4 % every time, profiles/r2g/ab_slots.txt)
    int nslot = 3;
    const char* e = getenv("C8P_SLOTS");
    if (e && atoi(e) >= 1 && atoi(e) <= 8) nslot = atoi(e);
    if (cfg->stream) nslot = 1;
    c->slots.resize(nslot);
    c->own_stream = cfg->stream == nullptr;
    for (int i = 0; i < nslot; i++) {
      if (cfg->stream) c->slots[i].stream = (hipStream_t)cfg->stream;
      else {
        // C8P_STREAM_PRIORITY (read at create, like C8P_SLOTS): -1 = the device's highest priority — for the contexts whose small kernels
        // sit in front of another context's long ones (detect / signal in front of rx on one GPU)
        const char* pe = getenv("C8P_STREAM_PRIORITY");
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const int pr = pe ? std::max(hi, std::min(lo, atoi(pe))) : 0;
        if (hipStreamCreateWithPriority(&c->slots[i].stream, hipStreamNonBlocking, pr) != hipSuccess) { delete c; return nullptr; }
      }
    }
    c->s = &c->slots[0];
    c->cur = 0;
  }
  if (hipMalloc((void**)&c->d_tab, sizeof(DevTables)) != hipSuccess) { delete c; return nullptr; }
  const DevTables& T = host_tables();
  if (hipMemcpy(c->d_tab, &T, sizeof(DevTables), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(c->d_tab); delete c; return nullptr; }
  return c;
}

extern "C" void c8p_destroy(c8p_ctx* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (Slot& sl : c->slots) (void)hipStreamSynchronize(sl.stream);
  drain_events(c);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (Slot& sl : c->slots) {
    Buf* bufs[] = {&sl.d_stage, &sl.d_iq, &sl.d_chan, &sl.d_llr_in, &sl.d_mu_in, &sl.d_tags, &sl.d_outs, &sl.d_H, &sl.d_mimo,
                   &sl.d_muchan, &sl.d_llr, &sl.d_surv, &sl.d_bits, &sl.d_sbits, &sl.d_pdu, &sl.d_flags, &sl.d_sym2pkt, &sl.d_negsym, &sl.d_done4,
                   &sl.d_sig_raw, &sl.d_sig_pkt, &sl.d_sig_iq, &sl.d_sig_chan, &sl.d_sig_kept, &sl.d_det_raw, &sl.d_det_ac, &sl.d_det_trig, &sl.d_det_tcand, &sl.d_det_toff,
                   &sl.d_det_conj, &sl.d_det_zero, &sl.d_det_carry, &sl.d_det_cand, &sl.d_det_job, &sl.d_det_mark,
                   &sl.d_det_evt, &sl.d_det_valid, &sl.d_det_fire, &sl.d_det_ds, &sl.d_det_soff, &sl.d_tx_plan, &sl.d_tx_psdu, &sl.d_tx_bfq, &sl.d_tx_coded, &sl.d_tx_sig,
                   &sl.d_tx_u2p, &sl.d_tx_iq, &sl.d_tx_desc};
    for (Buf* b : bufs) if (b->p) (void)hipFree(b->p);
    if (sl.h_stage) (void)hipHostFree(sl.h_stage);
    if (c->own_stream) (void)hipStreamDestroy(sl.stream);
  }
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->out_stream) { (void)hipStreamSynchronize(c->out_stream); (void)hipStreamDestroy(c->out_stream); }
  for (hipEvent_t e : c->out_ev) if (e) (void)hipEventDestroy(e);
  if (c->copy_stream_b) { (void)hipStreamSynchronize(c->copy_stream_b); (void)hipStreamDestroy(c->copy_stream_b); }
  for (hipEvent_t e : c->b_ev) if (e) (void)hipEventDestroy(e);
  if (c->ev_fence_b) (void)hipEventDestroy(c->ev_fence_b);
  if (c->ev_async_b) (void)hipEventDestroy(c->ev_async_b);
  if (c->span_stream) { (void)hipStreamSynchronize(c->span_stream); (void)hipStreamDestroy(c->span_stream); }
  if (c->ev_before_spans) (void)hipEventDestroy(c->ev_before_spans);
  if (c->ev_spans) (void)hipEventDestroy(c->ev_spans);
  if (c->d_spans2) (void)hipFree(c->d_spans2);
  if (c->h_spans2) (void)hipHostFree(c->h_spans2);
  for (hipEvent_t e : c->copy_ev) if (e) (void)hipEventDestroy(e);
  if (c->ev_fence) (void)hipEventDestroy(c->ev_fence);
  if (c->d_spans.p) (void)hipFree(c->d_spans.p);
  if (c->h_spans) (void)hipHostFree(c->h_spans);
  if (c->d_tab) (void)hipFree(c->d_tab);
  delete c;
}

extern "C" const char* c8p_last_error(const c8p_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

// ------------------------------------------------------------------------------------------------ submit helpers
struct Plan {
  std::vector<PktLayout> lay;
  std::vector<uint32_t> sym_base;
  size_t llr = 0, surv = 0, bits = 0, pdu = 0;
  uint32_t syms = 0;
};

// survivor slots (16 bytes each) of one packet: the clamp-exact kernel writes one slot per step, the lean kernel one
// 256-byte block per 30 steps (8.5 bytes per step + the partial last block)
// Rounded up to 16 slots = 256 bytes: every packet's survivor area then starts on a 256-byte boundary (the arena does), so a
// 256-byte survivor block is exactly two 128-byte lines.  With areas starting at any 16-byte multiple a block straddled three lines:
// the forward pass wrote partial lines and k_finish fetched up to half as many lines again (FETCH_SIZE 1.18-1.19 x the bytes written).
static uint32_t surv_slots(uint32_t tcap) { return (tcap + 17u + 15u) & ~15u; }

// PDU area of one packet.  A published (sub)frame of len bytes costs a record of len + 6 bytes ({u16 len}{format, len_lo, len_hi,
// MPDU, mcs}) and consumed at least len + 4 bytes of the decoded stream (delimiter + MPDU padded to 4), at least 8 of them: the
// records of an A-MPDU made of many tiny subframes exceed the stream by 2 bytes per subframe, at most tcap / 64 subframes.
static uint32_t pdu_cap_for(uint32_t tcap)
{
  uint32_t pcap = tcap / 8u + tcap / 32u + 16u;
  if (pcap < C8P_NDP_BLOB_BYTES + 5u) pcap = C8P_NDP_BLOB_BYTES + 5u;
  return (pcap + 15u) & ~15u;
}

// Worst-case sizes from n_items alone: at most n_items/72+1 symbols (72 = short-GI symbol), 416 LLRs per
// symbol and antenna, 347 trellis steps per symbol and antenna (416 * 5/6).
static int plan_demod(c8p_ctx* c, const c8p_batch_in* in, Plan& P)
{
  const int n = in->n_pkt;
  P.lay.resize(n); P.sym_base.resize(n + 1);
  for (int p = 0; p < n; p++) {
    const c8p_pkt_in& k = in->pkts[p];
    if (k.n_ant != 1 && k.n_ant != 2) { c->err = "n_ant must be 1 or 2"; return C8P_EINVAL; }
    for (int a = 0; a < k.n_ant; a++)
      if (k.iq_off[a] + k.n_items > in->n_iq) { c->err = "packet items outside the iq arena"; return C8P_ERANGE; }
    if ((uint64_t)k.chan_off + 64 > in->n_chan) { c->err = "chan outside the chan arena"; return C8P_ERANGE; }
    const uint32_t nsb = k.n_items / 72u + 1u, nant = k.n_ant;
    uint32_t llr_cap = nsb * 416u * nant;
    if (llr_cap < C8P_NDP_TOTAL) llr_cap = C8P_NDP_TOTAL;
    uint32_t tcap = nsb * 347u * nant;
    if (tcap > C8P_DECODE_T_MAX) tcap = C8P_DECODE_T_MAX;
    PktLayout& L = P.lay[p];
    L.llr_off = P.llr; L.surv_off = P.surv; L.bits_off = P.bits; L.pdu_off = P.pdu;
    L.sym_base = P.syms; L.nsym_bound = nsb; L.pdu_cap = pdu_cap_for(tcap); L.trellis_cap = tcap;
    P.sym_base[p] = P.syms;
    P.llr += llr_cap; P.surv += surv_slots(tcap); P.bits += (tcap + 15u) & ~15u; P.pdu += L.pdu_cap; P.syms += nsb;
  }
  P.sym_base[n] = P.syms;
  return C8P_OK;
}

static int plan_decode(c8p_ctx* c, const c8p_decode_in* in, Plan& P)
{
  const int n = in->n_pkt;
  P.lay.resize(n); P.sym_base.assign(n + 1, 0);
  for (int p = 0; p < n; p++) {
    const c8p_pkt_tags& t = in->tags[p];
    uint32_t tcap = 0;
    if (t.status == C8P_ST_OK && t.trellis > 0 && t.trellis <= C8P_DECODE_T_MAX) tcap = (uint32_t)t.trellis;
    if (t.status == C8P_ST_OK && t.total > 0 && t.llr_off + (uint64_t)t.total > in->n_llr) { c->err = "LLR items outside the llr arena"; return C8P_ERANGE; }
    PktLayout& L = P.lay[p];
    L.llr_off = t.llr_off; L.surv_off = P.surv; L.bits_off = P.bits; L.pdu_off = P.pdu;
    L.sym_base = 0; L.nsym_bound = 0; L.pdu_cap = pdu_cap_for(tcap); L.trellis_cap = tcap;
    P.surv += surv_slots(tcap); P.bits += (tcap + 15u) & ~15u; P.pdu += L.pdu_cap;
  }
  return C8P_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// c8p_rx hands over what the fused single-stream kernel needs (k_rx_1ss); c8p_decode passes nullptr
struct RxFuse { const c8p_pkt_in* d_pk; const cf* d_iq; const cf* d_chan; bool any_2ant; const uint32_t* d_order; };
// d_cold: the FinCold record of this submit (device memory)

// The arenas of one submit, by stage (also what c8p_reserve sizes in advance).
static size_t demod_stage_bytes(int n, size_t* o_lay = nullptr, size_t* o_cold = nullptr, size_t* o_ord = nullptr)
{
  const size_t ol = align256(sizeof(c8p_pkt_in) * (size_t)n), oc = align256(ol + sizeof(PktLayout) * (size_t)n);
  const size_t oo = align256(oc + sizeof(FinCold));          // the launch order of the packets (uint32[n]; used when their lengths differ)
  if (o_lay) *o_lay = ol;
  if (o_cold) *o_cold = oc;
  if (o_ord) *o_ord = oo;
  return align256(oo + sizeof(uint32_t) * (size_t)n);
}
static int ensure_demod_arenas(c8p_ctx* c, int n, const Plan& P, uint64_t host_iq, uint64_t host_chan)
{
  int rc;
  const size_t stage_bytes = demod_stage_bytes(n);
  if ((rc = ensure_host(c, stage_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_stage, stage_bytes)) != C8P_OK) return rc;
  if (host_iq && (rc = ensure(c, c->s->d_iq, sizeof(cf) * host_iq)) != C8P_OK) return rc;
  if (host_chan && (rc = ensure(c, c->s->d_chan, sizeof(cf) * host_chan)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_tags, sizeof(c8p_pkt_tags) * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_H, sizeof(cf) * 64 * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_mimo, sizeof(Mimo) * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_muchan, sizeof(cf) * 128 * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_llr, sizeof(float) * P.llr)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_flags, sizeof(uint32_t) * 3 * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sym2pkt, sizeof(uint32_t) * (size_t)(P.syms + 1))) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_negsym, sizeof(float) * (size_t)(P.syms + 1))) != C8P_OK) return rc;
  return C8P_OK;
}
static int ensure_decode_arenas(c8p_ctx* c, int n, const Plan& P)
{
  int rc;
  if ((rc = ensure(c, c->s->d_outs, sizeof(c8p_pkt_out) * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_surv, sizeof(ulonglong2) * P.surv)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_pdu, P.pdu)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_bits, c->keep_bits ? P.bits : 16)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sbits, c->keep_bits ? P.bits : 16)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_flags, sizeof(uint32_t) * 3 * (size_t)n)) != C8P_OK) return rc;
  return C8P_OK;
}
static int ensure_signal_arenas(c8p_ctx* c, size_t n, uint64_t iq_items, uint64_t host_raw)
{
  int rc;
  const size_t up = align256(align256(sizeof(c8p_sig_in) * n) + sizeof(uint64_t) * n);           // flags + arena offsets on the way up
  const size_t down = align256(sizeof(c8p_pkt_in) * n) + sizeof(int) * n;                         // descriptors + kept list on the way back
  if ((rc = ensure_host(c, std::max(up, down))) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_stage, up)) != C8P_OK) return rc;
  if (host_raw && (rc = ensure(c, c->s->d_sig_raw, sizeof(cf) * host_raw)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sig_pkt, sizeof(c8p_pkt_in) * n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sig_chan, sizeof(cf) * 64 * n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sig_iq, sizeof(cf) * std::max<uint64_t>(iq_items, 2))) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sig_kept, sizeof(int) * std::max<size_t>(n, 1))) != C8P_OK) return rc;
  return C8P_OK;
}

static int run_decode_kernels(c8p_ctx* c, int n, const c8p_pkt_tags* d_tags, const PktLayout* d_lay, const float* d_llr,
                              const cf* d_mu, const Plan& P, const RxFuse* rx, const FinCold* d_cold)
{
  const bool flags_ready = rx != nullptr;
  int rc;
  if ((rc = ensure_decode_arenas(c, n, P)) != C8P_OK) return rc;
  const uint32_t* d_flags = c->exact_only ? nullptr : (const uint32_t*)c->s->d_flags.p;
  hipStream_t dst = c->s->stream, fst = c->s->stream;
  {
    // forward pass: lean kernels for every packet whose LLRs passed the range check; the rest is left to k_finish (viterbi_fwd_exact)
    KTimer kt(c, C8P_K_VITERBI, dst);
    if (!c->exact_only) {
      if (!flags_ready)
        hipLaunchKernelGGL(k_llr_check, dim3(n), dim3(64), 0, dst, d_tags, d_llr, (uint32_t*)c->s->d_flags.p, n);
      // c8p_rx: a packet is demodulated AND decoded by one wave of k_rx_1ss_lds / k_rx_2ss_lds (LLRs through LDS).  keep_llr = 1, the
      // arena flavour: k_rx_1ss for one stream, k_demod_sym + k_viterbi_lean for two (as every packet of a c8p_decode batch)
      if (rx && c->keep_llr == 1) {
        hipLaunchKernelGGL(k_rx_1ss, dim3((n + 3) / 4), dim3(256), 0, dst, rx->d_pk, rx->d_iq, rx->d_chan, d_lay, (const DevTables*)c->d_tab,
                           d_tags, (const cf*)c->s->d_H.p, (float*)c->s->d_llr.p, (ulonglong2*)c->s->d_surv.p, (uint32_t*)c->s->d_flags.p, n);
      } else if (rx) {   // the LLR stream stays in LDS; what the lean decoder cannot take is flagged and left to k_finish
        float* tap = c->keep_llr == 2 ? (float*)c->s->d_llr.p : nullptr;     // the parity tests' view of the on-chip LLRs
        hipLaunchKernelGGL(k_rx_1ss_lds, dim3((n + 3) / 4), dim3(256), 0, dst, rx->d_pk, rx->d_iq, rx->d_chan, d_lay, (const DevTables*)c->d_tab,
                           d_tags, (const cf*)c->s->d_H.p, tap, (ulonglong2*)c->s->d_surv.p, (uint32_t*)c->s->d_flags.p, n, rx->d_order);
        if (rx->any_2ant)
          hipLaunchKernelGGL(k_rx_2ss_lds, dim3((n + 3) / 4), dim3(256), 0, dst, rx->d_pk, rx->d_iq, d_lay, (const DevTables*)c->d_tab,
                             d_tags, (const Mimo*)c->s->d_mimo.p, tap, (ulonglong2*)c->s->d_surv.p, (uint32_t*)c->s->d_flags.p, n, rx->d_order);
      }
      // c8p_decode: quads of packets with the same rate and length can go through the four-packets-per-wave kernel (17 % fewer VALU
      // instructions per packet, but 4 x fewer waves): it wins when one batch alone gives every SIMD several of its fat waves —
      // measured -12 % decode time at 16384 packets per batch, +6 % at 4096.  C8P_LEAN4 = 1 / 0 forces it on / off.
      const char* e4 = getenv("C8P_LEAN4");
      const bool lean4 = e4 ? atoi(e4) != 0 : n >= 12288;
      const uint8_t* d_done4 = nullptr;
      if (!rx && lean4 && n >= 4) {
        if ((rc = ensure(c, c->s->d_done4, (size_t)n + 4)) != C8P_OK) return rc;
        HIPCHK(c, hipMemsetAsync(c->s->d_done4.p, 0, (size_t)n, dst));
        hipLaunchKernelGGL(k_viterbi_lean4, dim3((n / 4 + 3) / 4), dim3(256), 0, dst, d_tags, d_lay, d_llr, (ulonglong2*)c->s->d_surv.p,
                           (uint32_t*)c->s->d_flags.p, (uint8_t*)c->s->d_done4.p, n);
        d_done4 = (const uint8_t*)c->s->d_done4.p;
      }
      const char* el = getenv("C8P_DECODE_LDS");     // tests: c8p_decode through the LDS-fed block loop of k_rx_1ss_lds
      if (!rx && el && atoi(el) != 0) {
        const uint8_t* d2 = nullptr;
#ifdef C8P_EXPERIMENTS
        if (atoi(el) == 2 && n >= 2) {                // ... and pairs of alike packets through the two-packets-per-wave layout
          if ((rc = ensure(c, c->s->d_done4, (size_t)n + 4)) != C8P_OK) return rc;
          HIPCHK(c, hipMemsetAsync(c->s->d_done4.p, 0, (size_t)n, dst));
          hipLaunchKernelGGL(k_viterbi_lean2_lds, dim3((n / 2 + 3) / 4), dim3(256), 0, dst, d_tags, d_lay, d_llr, (ulonglong2*)c->s->d_surv.p,
                             (uint32_t*)c->s->d_flags.p, (uint8_t*)c->s->d_done4.p, n);
          d2 = (const uint8_t*)c->s->d_done4.p;
        }
#else
        if (atoi(el) == 2) { c->err = "C8P_DECODE_LDS=2 (two packets per wave) needs a -DC8P_EXPERIMENTS build (make -C gr-ieee80211_amd/csrc exp)"; return C8P_EINVAL; }
#endif
        hipLaunchKernelGGL(k_viterbi_lean_lds, dim3((n + 3) / 4), dim3(256), 0, dst, d_tags, d_lay, d_llr, (ulonglong2*)c->s->d_surv.p,
                           (uint32_t*)c->s->d_flags.p, d2, n);
      }
      else if (!rx || (rx->any_2ant && c->keep_llr == 1))    // (c8p_rx, arena flavour: two-stream packets came through k_demod_sym)
        hipLaunchKernelGGL(k_viterbi_lean, dim3((n + 3) / 4), dim3(256), 0, dst, d_tags, d_lay, d_llr, (ulonglong2*)c->s->d_surv.p,
                           (uint32_t*)c->s->d_flags.p, rx ? (const float*)c->s->d_negsym.p : nullptr, rx ? rx->d_pk : nullptr, d_done4, n);
    }
  }
  {
    KTimer kt(c, C8P_K_FINISH, fst);
    uint32_t tmax = 0;
    for (const PktLayout& L : P.lay) tmax = L.trellis_cap > tmax ? L.trellis_cap : tmax;
    uint32_t snap_bytes = ((tmax + 5u) / 6u + 2u + 15u) & ~15u;      // one byte per 6-step group; afterwards the 1 KB CRC table
    if (snap_bytes < 1024u) snap_bytes = 1024u;
    hipLaunchKernelGGL(k_finish, dim3(n), dim3(64), snap_bytes, fst, d_tags, d_lay, (ulonglong2*)c->s->d_surv.p, d_mu,
                       (const DevTables*)c->d_tab, (c8p_pkt_out*)c->s->d_outs.p, (uint8_t*)c->s->d_pdu.p, (uint8_t*)c->s->d_bits.p,
                       (uint8_t*)c->s->d_sbits.p, d_flags, c->keep_bits, n,
                       // (packets the lean kernels left: forward pass inside k_finish; the on-chip kernels of c8p_rx left them undemodulated)
                       (rx && c->keep_llr != 1 && !c->exact_only ? C8P_FIN_REDEMOD : 0) | (rx && c->keep_llr == 2 && !c->exact_only ? C8P_FIN_TAP : 0), d_cold);
  }
  HIPCHK(c, hipGetLastError());
  c->s->used[C8P_BUF_OUT] = sizeof(c8p_pkt_out) * (size_t)n; c->s->ptr[C8P_BUF_OUT] = c->s->d_outs.p;
  c->s->used[C8P_BUF_PDU] = P.pdu; c->s->ptr[C8P_BUF_PDU] = c->s->d_pdu.p;
  c->s->used[C8P_BUF_BITS] = c->keep_bits ? P.bits : 0; c->s->ptr[C8P_BUF_BITS] = c->s->d_bits.p;
  c->s->used[C8P_BUF_SBITS] = c->keep_bits ? P.bits : 0; c->s->ptr[C8P_BUF_SBITS] = c->s->d_sbits.p;
  c->s->used[C8P_BUF_DECSTAT] = sizeof(uint32_t) * 3 * (size_t)n; c->s->ptr[C8P_BUF_DECSTAT] = c->s->d_flags.p;
  return C8P_OK;
}


// c8p_copy_fence: the slot's stream waits (on the device) for the copies that were on the copy stream when the fence was set
static int honour_fence(c8p_ctx* c)
{
  std::lock_guard<std::mutex> g(c->copy_mu);
  if (c->fence_b_pending) { HIPCHK(c, hipStreamWaitEvent(c->s->stream, c->ev_fence_b, 0)); c->fence_b_pending = false; }
  if (!c->fence_pending) return C8P_OK;
  HIPCHK(c, hipStreamWaitEvent(c->s->stream, c->ev_fence, 0));
  c->fence_pending = false;
  return C8P_OK;
}

// C8P_SUBMIT_TLOG=1: where the host time of submit_demod goes (microseconds, summed over the submits of the process; printed at exit)
struct SubmitLog {
  bool on = getenv("C8P_SUBMIT_TLOG") != nullptr;
  double us[7] = {0, 0, 0, 0, 0, 0, 0};
  long n = 0;
  std::chrono::steady_clock::time_point t;
  void start() { if (on) t = std::chrono::steady_clock::now(); }
  void lap(int k) { if (!on) return; const auto u = std::chrono::steady_clock::now(); us[k] += std::chrono::duration<double, std::micro>(u - t).count(); t = u; }
  ~SubmitLog()
  {
    if (on && n) fprintf(stderr, "[c8p submit] %ld submits; us per submit: slot's stream synchronised %.1f, events drained %.1f, fence %.1f, plan + arenas %.1f, staging %.1f, upload + classify (+ demod) launches %.1f, "
                                 "decode launches %.1f\n", n, us[5] / n, us[6] / n, us[0] / n, us[1] / n, us[2] / n, us[3] / n, us[4] / n);
  }
};
static SubmitLog g_submit_log;

static int submit_demod(c8p_ctx* c, const c8p_batch_in* in, bool with_decode)
{
  g_submit_log.start();
  if (!c) return C8P_EINVAL;
  if (!in || in->n_pkt < 0 || (in->n_pkt > 0 && (!in->pkts || !in->iq || !in->chan))) { c->err = "bad batch"; return C8P_EINVAL; }
  if (in->mem != C8P_MEM_HOST && in->mem != C8P_MEM_DEVICE) { c->err = "bad mem kind"; return C8P_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  // next slot; its previous batch (submitted slots.size() submits ago) must be done before its arenas are reused
  c->cur = (c->cur + 1) % (int)c->slots.size(); c->seq++; c->slots[c->cur].ticket = c->seq;
  c->s = &c->slots[c->cur];
  HIPCHK(c, hipStreamSynchronize(c->s->stream));
  g_submit_log.lap(5);
  drain_events(c);
  g_submit_log.lap(6);
  { const int rf = honour_fence(c); if (rf != C8P_OK) return rf; }
  g_submit_log.lap(0);
  const int n = in->n_pkt;
  c->s->n_pkt = n;
  for (int i = 0; i < C8P_BUF_COUNT; i++) { c->s->used[i] = 0; c->s->ptr[i] = nullptr; }
  if (n == 0) return C8P_OK;
  Plan P;
  int rc = plan_demod(c, in, P);
  if (rc != C8P_OK) return rc;

  // descriptors + layout (+ the record k_finish's cold path reads) in one pinned staging copy
  size_t o_lay = 0, o_cold = 0, o_ord = 0;
  const size_t o_pk = 0, stage_bytes = demod_stage_bytes(n, &o_lay, &o_cold, &o_ord);
  if ((rc = ensure_demod_arenas(c, n, P, in->mem == C8P_MEM_HOST ? in->n_iq : 0, in->mem == C8P_MEM_HOST ? in->n_chan : 0)) != C8P_OK) return rc;
  const c8p_pkt_in* d_pk = (const c8p_pkt_in*)((char*)c->s->d_stage.p + o_pk);
  const PktLayout* d_lay = (const PktLayout*)((char*)c->s->d_stage.p + o_lay);
  g_submit_log.lap(1);

  const cf* d_iq = (const cf*)in->iq;
  const cf* d_chan = (const cf*)in->chan;
  if (in->mem == C8P_MEM_HOST) {
    HIPCHK(c, hipMemcpyAsync(c->s->d_iq.p, in->iq, sizeof(cf) * in->n_iq, hipMemcpyHostToDevice, c->s->stream));
    HIPCHK(c, hipMemcpyAsync(c->s->d_chan.p, in->chan, sizeof(cf) * in->n_chan, hipMemcpyHostToDevice, c->s->stream));
    d_iq = (const cf*)c->s->d_iq.p; d_chan = (const cf*)c->s->d_chan.p;
  }
  // (the slot's stream was synchronised when the slot was taken, and nothing but the uploads above has been enqueued on it since: the
  // previous submit's asynchronous copy out of the pinned buffer has finished)
  memcpy((char*)c->s->h_stage + o_pk, in->pkts, sizeof(c8p_pkt_in) * (size_t)n);
  memcpy((char*)c->s->h_stage + o_lay, P.lay.data(), sizeof(PktLayout) * (size_t)n);
  {
    const FinCold fc = {d_pk, d_iq, d_chan, (const cf*)c->s->d_H.p, (const Mimo*)c->s->d_mimo.p, (float*)c->s->d_llr.p};
    memcpy((char*)c->s->h_stage + o_cold, &fc, sizeof(fc));
  }
  // A batch whose packets differ in length is decoded longest first (the on-chip kernels of c8p_rx: four packets share a workgroup and
  // its LDS until the longest of them is done; measured on 4096 packets of random format / rate / length: +13 % packets/s).  The
  // results stay where the caller's order puts them: only the launch order changes.
  const uint32_t* d_order = nullptr;
  if (with_decode && n > 4) {
    uint32_t lo = in->pkts[0].n_items, hi = lo;
    for (int p = 1; p < n; p++) { lo = std::min(lo, in->pkts[p].n_items); hi = std::max(hi, in->pkts[p].n_items); }
    if (hi > lo + lo / 8) {
      uint32_t* ord = (uint32_t*)((char*)c->s->h_stage + o_ord);
      for (int p = 0; p < n; p++) ord[p] = (uint32_t)p;
      const c8p_pkt_in* pkts = in->pkts;
      std::stable_sort(ord, ord + n, [pkts](uint32_t a, uint32_t b) { return (uint64_t)pkts[a].n_items * pkts[a].n_ant > (uint64_t)pkts[b].n_items * pkts[b].n_ant; });
      d_order = (const uint32_t*)((char*)c->s->d_stage.p + o_ord);
    }
  }
  g_submit_log.lap(2);
  {
    KTimer kt(c, C8P_K_LAYOUT);
    HIPCHK(c, upload(c->s->stream, c->s->d_stage.p, c->s->h_stage, d_order ? stage_bytes : o_ord));
  }

  {
    KTimer kt(c, C8P_K_CLASSIFY);
    hipLaunchKernelGGL(k_classify, dim3(n), dim3(64), 0, c->s->stream, d_pk, d_iq, d_chan, d_lay, (const DevTables*)c->d_tab,
                       (c8p_pkt_tags*)c->s->d_tags.p, (cf*)c->s->d_H.p, (Mimo*)c->s->d_mimo.p, (cf*)c->s->d_muchan.p, (uint32_t*)c->s->d_flags.p,
                       (uint32_t*)c->s->d_sym2pkt.p, n);
  }
  bool any_2ant = false;
  for (int p = 0; p < n; p++) any_2ant |= in->pkts[p].n_ant == 2;
  const bool fused = with_decode && !c->exact_only;      // C8P_EXACT_VITERBI: plain demodulation, the clamp-exact decoder for everything
  if (!fused || (any_2ant && c->keep_llr == 1)) {   // c8p_rx, arena flavour: only the two-stream packets (single-stream ones belong to k_rx_1ss)
    KTimer kt(c, C8P_K_DEMOD);
    hipLaunchKernelGGL(k_demod_sym, dim3((P.syms + 3) / 4), dim3(256), 0, c->s->stream, d_pk, d_iq, d_chan, (const uint32_t*)c->s->d_sym2pkt.p, d_lay,
                       (const DevTables*)c->d_tab, (const c8p_pkt_tags*)c->s->d_tags.p, (const cf*)c->s->d_H.p,
                       (const Mimo*)c->s->d_mimo.p, (float*)c->s->d_llr.p, (uint32_t*)c->s->d_flags.p, (float*)c->s->d_negsym.p, n, P.syms,
                       fused ? 1 : 0);
  }
  HIPCHK(c, hipGetLastError());
  c->s->used[C8P_BUF_TAGS] = sizeof(c8p_pkt_tags) * (size_t)n; c->s->ptr[C8P_BUF_TAGS] = c->s->d_tags.p;
  // (c8p_rx without keep_llr: the fused kernel keeps the LLRs in LDS, the arena only holds what the exceptions needed)
  c->s->used[C8P_BUF_LLR] = (fused && !c->keep_llr) ? 0 : sizeof(float) * P.llr; c->s->ptr[C8P_BUF_LLR] = c->s->d_llr.p;
  // (keep_llr = 2, the tap: the kernels do not write the padding between two packets' LLR areas, and the tests read only the areas)
  c->s->used[C8P_BUF_MUCHAN] = sizeof(cf) * 128 * (size_t)n; c->s->ptr[C8P_BUF_MUCHAN] = c->s->d_muchan.p;
  g_submit_log.lap(3);
  if (!with_decode) { g_submit_log.n++; return C8P_OK; }
  const RxFuse rx = {d_pk, d_iq, d_chan, any_2ant, d_order};
  struct Fin { ~Fin() { g_submit_log.lap(4); g_submit_log.n++; } } fin;
  return run_decode_kernels(c, n, (const c8p_pkt_tags*)c->s->d_tags.p, d_lay, (const float*)c->s->d_llr.p, (const cf*)c->s->d_muchan.p, P, &rx,
                            (const FinCold*)((char*)c->s->d_stage.p + o_cold));
}

extern "C" int c8p_demod(c8p_ctx* c, const c8p_batch_in* in) { return submit_demod(c, in, false); }
extern "C" int c8p_rx(c8p_ctx* c, const c8p_batch_in* in) { return submit_demod(c, in, true); }

extern "C" int c8p_decode(c8p_ctx* c, const c8p_decode_in* in)
{
  if (!c) return C8P_EINVAL;
  if (!in || in->n_pkt < 0 || (in->n_pkt > 0 && (!in->tags || !in->llr))) { c->err = "bad batch"; return C8P_EINVAL; }
  if (in->mem != C8P_MEM_HOST && in->mem != C8P_MEM_DEVICE) { c->err = "bad mem kind"; return C8P_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  // next slot; its previous batch (submitted slots.size() submits ago) must be done before its arenas are reused
  c->cur = (c->cur + 1) % (int)c->slots.size(); c->seq++; c->slots[c->cur].ticket = c->seq;
  c->s = &c->slots[c->cur];
  HIPCHK(c, hipStreamSynchronize(c->s->stream));
  drain_events(c);
  { const int rf = honour_fence(c); if (rf != C8P_OK) return rf; }
  const int n = in->n_pkt;
  c->s->n_pkt = n;
  for (int i = 0; i < C8P_BUF_COUNT; i++) { c->s->used[i] = 0; c->s->ptr[i] = nullptr; }
  if (n == 0) return C8P_OK;
  Plan P;
  int rc = plan_decode(c, in, P);
  if (rc != C8P_OK) return rc;
  const size_t o_tg = 0, o_lay = align256(sizeof(c8p_pkt_tags) * (size_t)n), o_cold = align256(o_lay + sizeof(PktLayout) * (size_t)n);
  const size_t stage_bytes = align256(o_cold + sizeof(FinCold));
  if ((rc = ensure_host(c, stage_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_stage, stage_bytes)) != C8P_OK) return rc;
  const c8p_pkt_tags* d_tags = (const c8p_pkt_tags*)((char*)c->s->d_stage.p + o_tg);
  const PktLayout* d_lay = (const PktLayout*)((char*)c->s->d_stage.p + o_lay);
  const float* d_llr = in->llr;
  const cf* d_mu = (const cf*)in->mu2x1chan;
  if (in->mem == C8P_MEM_HOST) {
    if ((rc = ensure(c, c->s->d_llr_in, sizeof(float) * in->n_llr)) != C8P_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->s->d_llr_in.p, in->llr, sizeof(float) * in->n_llr, hipMemcpyHostToDevice, c->s->stream));
    d_llr = (const float*)c->s->d_llr_in.p;
    if (in->mu2x1chan) {
      if ((rc = ensure(c, c->s->d_mu_in, sizeof(cf) * 128 * (size_t)n)) != C8P_OK) return rc;
      HIPCHK(c, hipMemcpyAsync(c->s->d_mu_in.p, in->mu2x1chan, sizeof(cf) * 128 * (size_t)n, hipMemcpyHostToDevice, c->s->stream));
      d_mu = (const cf*)c->s->d_mu_in.p;
    }
  }
  if (!d_mu) {   // NDP reports then carry zeros
    if ((rc = ensure(c, c->s->d_mu_in, sizeof(cf) * 128 * (size_t)n)) != C8P_OK) return rc;
    HIPCHK(c, hipMemsetAsync(c->s->d_mu_in.p, 0, sizeof(cf) * 128 * (size_t)n, c->s->stream));
    d_mu = (const cf*)c->s->d_mu_in.p;
  }
  memcpy((char*)c->s->h_stage + o_tg, in->tags, sizeof(c8p_pkt_tags) * (size_t)n);
  memcpy((char*)c->s->h_stage + o_lay, P.lay.data(), sizeof(PktLayout) * (size_t)n);
  {
    const FinCold fc = {nullptr, nullptr, nullptr, nullptr, nullptr, (float*)d_llr};      // (k_finish only READS the caller's LLRs)
    memcpy((char*)c->s->h_stage + o_cold, &fc, sizeof(fc));
  }
  {
    KTimer kt(c, C8P_K_LAYOUT);
    HIPCHK(c, upload(c->s->stream, c->s->d_stage.p, c->s->h_stage, stage_bytes));
  }
  c->s->used[C8P_BUF_TAGS] = sizeof(c8p_pkt_tags) * (size_t)n; c->s->ptr[C8P_BUF_TAGS] = d_tags;
  return run_decode_kernels(c, n, d_tags, d_lay, d_llr, d_mu, P, nullptr, (const FinCold*)((char*)c->s->d_stage.p + o_cold));
}

extern "C" int c8p_signal(c8p_ctx* c, const c8p_signal_in* in)
{
  if (!c) return C8P_EINVAL;
  if (!in || in->n_evt < 0 || (in->n_evt > 0 && (!in->evts || !in->raw))) { c->err = "bad batch"; return C8P_EINVAL; }
  if (in->mem != C8P_MEM_HOST && in->mem != C8P_MEM_DEVICE) { c->err = "bad mem kind"; return C8P_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  c->cur = (c->cur + 1) % (int)c->slots.size(); c->seq++; c->slots[c->cur].ticket = c->seq;
  c->s = &c->slots[c->cur];
  HIPCHK(c, hipStreamSynchronize(c->s->stream));
  drain_events(c);
  const int n = in->n_evt;
  c->s->n_pkt = n;
  for (int i = 0; i < C8P_BUF_COUNT; i++) { c->s->used[i] = 0; c->s->ptr[i] = nullptr; }
  if (n == 0) return C8P_OK;
  std::vector<uint64_t> off(n);
  uint64_t tot = 0;
  for (int e = 0; e < n; e++) {
    const c8p_sig_in& k = in->evts[e];
    if (k.n_ant != 1 && k.n_ant != 2) { c->err = "n_ant must be 1 or 2"; return C8P_EINVAL; }
    for (int a = 0; a < k.n_ant; a++)
      if (k.raw_off[a] + k.n_avail > in->n_raw) { c->err = "sync event items outside the raw arena"; return C8P_ERANGE; }
    const uint64_t cap = (k.n_avail >= 224u ? k.n_avail - 224u : 0u) + 320u;
    off[e] = tot;
    tot += cap * k.n_ant;
  }
  int rc;
  const size_t o_ev = 0, o_off = align256(sizeof(c8p_sig_in) * (size_t)n);
  const size_t stage_bytes = align256(o_off + sizeof(uint64_t) * (size_t)n);
  if ((rc = ensure_host(c, stage_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_stage, stage_bytes)) != C8P_OK) return rc;
  memcpy((char*)c->s->h_stage + o_ev, in->evts, sizeof(c8p_sig_in) * (size_t)n);
  memcpy((char*)c->s->h_stage + o_off, off.data(), sizeof(uint64_t) * (size_t)n);
  HIPCHK(c, upload(c->s->stream, c->s->d_stage.p, c->s->h_stage, stage_bytes));
  const cf* d_raw = (const cf*)in->raw;
  if (in->mem == C8P_MEM_HOST) {
    if ((rc = ensure(c, c->s->d_sig_raw, sizeof(cf) * in->n_raw)) != C8P_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(c->s->d_sig_raw.p, in->raw, sizeof(cf) * in->n_raw, hipMemcpyHostToDevice, c->s->stream));
    d_raw = (const cf*)c->s->d_sig_raw.p;
  }
  if ((rc = ensure(c, c->s->d_sig_pkt, sizeof(c8p_pkt_in) * (size_t)n)) != C8P_OK) return rc;
  if ((rc = ensure(c, c->s->d_sig_chan, sizeof(cf) * 64 * (size_t)n)) != C8P_OK) return rc;
  const c8p_sig_in* d_ev = (const c8p_sig_in*)((char*)c->s->d_stage.p + o_ev);
  if (!in->consume) {
    if ((rc = ensure(c, c->s->d_sig_iq, sizeof(cf) * tot)) != C8P_OK) return rc;
    KTimer kt(c, C8P_K_SIGNAL);
    hipLaunchKernelGGL(k_signal, dim3(n), dim3(64), 0, c->s->stream, d_ev, d_raw,
                       (const uint64_t*)((char*)c->s->d_stage.p + o_off), (const DevTables*)c->d_tab,
                       (c8p_pkt_in*)c->s->d_sig_pkt.p, (cf*)c->s->d_sig_iq.p, (cf*)c->s->d_sig_chan.p, n);
    HIPCHK(c, hipGetLastError());
  } else {
    // S_DEMOD of every flag, then the block's scan over the descriptors (host), then S_COPY of the packets it keeps
    Slot& S = *c->s;
    {
      KTimer kt(c, C8P_K_SIGNAL);
      hipLaunchKernelGGL(k_signal_lsig, dim3(n), dim3(64), 0, S.stream, d_ev, d_raw, (const DevTables*)c->d_tab,
                         (c8p_pkt_in*)S.d_sig_pkt.p, (cf*)S.d_sig_chan.p, n);
    }
    HIPCHK(c, hipGetLastError());
    // (the staging upload above has been consumed once the kernel has run: the pinned buffer is reused for the way back)
    const size_t o_pk = 0, o_kept = align256(sizeof(c8p_pkt_in) * (size_t)n);
    if ((rc = ensure_host(c, o_kept + sizeof(int) * (size_t)n)) != C8P_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(S.stream));      // (before the pinned buffer is written again, should ensure_host have kept it)
    c8p_pkt_in* hpk = (c8p_pkt_in*)((char*)S.h_stage + o_pk);
    int* kept = (int*)((char*)S.h_stage + o_kept);
    HIPCHK(c, hipMemcpyAsync(hpk, S.d_sig_pkt.p, sizeof(c8p_pkt_in) * (size_t)n, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipStreamSynchronize(S.stream));
    // (n_streams > 1: K signal blocks, one per stream; the flags come grouped by stream and the rule runs inside each group)
    const int KS = in->n_streams > 1 ? in->n_streams : 1;
    std::vector<uint64_t> posv((size_t)KS, 0);
    std::vector<char> waitv((size_t)KS, 0);
    if (KS == 1) posv[0] = in->scan_from;
    else if (in->scan_from_v) for (int k = 0; k < KS; k++) posv[k] = in->scan_from_v[k];
    uint64_t used = 0;
    int nk = 0;
    uint32_t last_stream = 0;
    for (int e = 0; e < n; e++) {
      const c8p_sig_in& k = in->evts[e];
      c8p_pkt_in& p = hpk[e];
      const uint32_t si = KS > 1 ? k.stream : 0u;
      if (si >= (uint32_t)KS || si < last_stream) { c->err = "c8p_signal: flags must be grouped by stream, in ascending stream order"; return C8P_EINVAL; }
      last_stream = si;
      uint64_t& pos = posv[si];
      char& waiting = waitv[si];
      const uint64_t idx = k.raw_off[0];
      if (waiting) { p.n_items = 0; p.rsv[0] = C8P_SIG_PENDING; continue; }
      if (idx < pos) { p.n_items = 0; p.rsv[0] = C8P_SIG_SKIPPED; continue; }
      if (k.n_avail < 224u) { waiting = 1; p.n_items = 0; p.rsv[0] = C8P_SIG_PENDING; continue; }     // S_DEMOD waits for 224 items
      if (p.n_items == 0) {
        if (p.rsv[0] == C8P_SIG_NEED_MORE) waiting = 1;                                                  // S_COPY waits for the rest of the packet
        else pos = idx + 80;                                                                              // signal_impl.cc:157-160
        continue;
      }
      p.iq_off[0] = used; p.iq_off[1] = k.n_ant == 2 ? used + p.n_items : 0;
      used += (uint64_t)p.n_items * k.n_ant;
      kept[nk++] = e;
      pos = idx + 224 + (uint64_t)p.lsig_nsamp;                                                           // S_COPY consumed the packet
    }
    tot = used;
    if (in->scan_next) for (int k = 0; k < KS; k++) in->scan_next[k] = posv[k];
    if ((rc = ensure(c, S.d_sig_iq, sizeof(cf) * std::max<uint64_t>(tot, 2))) != C8P_OK) return rc;
    if ((rc = ensure(c, S.d_sig_kept, sizeof(int) * (size_t)std::max(nk, 1))) != C8P_OK) return rc;
    if (in->pkt_host) memcpy(in->pkt_host, hpk, sizeof(c8p_pkt_in) * (size_t)n);
    HIPCHK(c, upload(S.stream, S.d_sig_pkt.p, hpk, sizeof(c8p_pkt_in) * (size_t)n));
    if (nk) {
      HIPCHK(c, upload(S.stream, S.d_sig_kept.p, kept, sizeof(int) * (size_t)nk));
      KTimer kt(c, C8P_K_SIGNAL);
      hipLaunchKernelGGL(k_signal_copy, dim3(nk), dim3(64), 0, S.stream, d_ev, d_raw, (const c8p_pkt_in*)S.d_sig_pkt.p,
                         (const int*)S.d_sig_kept.p, (cf*)S.d_sig_iq.p, nk);
      HIPCHK(c, hipGetLastError());
    }
  }
  c->s->used[C8P_BUF_SIG_PKT] = sizeof(c8p_pkt_in) * (size_t)n; c->s->ptr[C8P_BUF_SIG_PKT] = c->s->d_sig_pkt.p;
  c->s->used[C8P_BUF_SIG_IQ] = sizeof(cf) * tot; c->s->ptr[C8P_BUF_SIG_IQ] = c->s->d_sig_iq.p;
  c->s->used[C8P_BUF_SIG_CHAN] = sizeof(cf) * 64 * (size_t)n; c->s->ptr[C8P_BUF_SIG_CHAN] = c->s->d_sig_chan.p;
  return C8P_OK;
}


// ------------------------------------------------------------------------------------------------ detection front-end
// presiso.grc -> trigger_impl -> sync_impl over one stream segment, or over the segments of K independent
// streams at once (c8p_detect_multi).  The dense work (windowed autocorrelation, plateau / running-maximum scan) runs on the GPU;
// trigger_impl's countdown logic and sync_impl's IDLE / SYNC alternation are sequential but touch only the few plateaus of
// >= 21 samples, so they run here on the host over the compacted candidate lists (trigger_impl.cc:86-109, sync_impl.cc:73-147).
struct DetHost {                 // one stream of a detection pass, host side
  uint64_t off = 0, n_total = 0, n_hist = 0;     // in: where its items lie in the arena
  c8p_det_state st0;                             // in: state at the commit point
  // derived
  uint64_t n = 0;                                // new items
  int ntile = 0; unsigned tile_base = 0, cap = 0;
  uint64_t out_off = 0;
  TrigSeg init;
  // out
  c8p_det_state st1;
  uint64_t commit = 0;                           // relative to the first NEW item
  uint32_t n_evt = 0;
  // work
  int job0 = 0, nj = 0;                          // its jobs: [job0, job0 + nj) triggers, job0 + nj the look-back at the commit point
  long long pos = 0, prev_start = 0, commit_ll = 0, stop_run_start = 0, stop_prev = 0;
  bool have = false, open_fired = false, stopped = false, stop_have = false, want_rmax = false;
  float rmax0 = 0.f;
};

static int ensure_detect_arenas(c8p_ctx* c, Slot& S, uint64_t out_items, size_t ntile, size_t cap, int K, bool keep, size_t zero_bytes)
{
  int rc;
  if ((rc = ensure(c, S.d_det_ac, keep ? sizeof(float) * out_items : 16)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_trig, out_items + 8)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_conj, keep ? sizeof(cf) * out_items : 16)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_carry, sizeof(TrigSeg) * (size_t)K)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_tcand, sizeof(TrigCand) * ntile * C8P_DET_TILE_CAND)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_zero, zero_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_toff, sizeof(unsigned) * (ntile + 1))) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_cand, sizeof(TrigCand) * cap)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_ds, sizeof(DetStreamDev) * (size_t)K + sizeof(unsigned short) * (ntile + 8) + 256)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_soff, sizeof(unsigned) * (size_t)(K + 1))) != C8P_OK) return rc;
  return C8P_OK;
}
static size_t rmax_jobs_off(size_t njobs) { return (sizeof(SyncJob) * njobs + 63) & ~(size_t)63; }      // the range-maximum jobs follow the sync jobs
static int ensure_detect_jobs(c8p_ctx* c, Slot& S, size_t nf, size_t njobs, size_t K)
{
  int rc;
  if ((rc = ensure(c, S.d_det_fire, sizeof(unsigned long long) * std::max<size_t>(nf, 1))) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_job, rmax_jobs_off(njobs) + sizeof(RmaxJob) * std::max<size_t>(K, 1))) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_mark, sizeof(SyncMark) * njobs)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_evt, sizeof(c8p_sync_evt) * njobs)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_det_valid, sizeof(int) * njobs)) != C8P_OK) return rc;
  return C8P_OK;
}
struct DetZero { size_t o_cnt, o_rmax, o_tc, o_summ, bytes; };
static DetZero det_zero_layout(size_t ntile, int K)
{
  // what every launch needs zeroed lies in one block (one fill): [counters 64 B | per-stream range maxima | tile candidate counts | tile summaries]
  DetZero z;
  z.o_cnt = 0; z.o_rmax = 64;
  z.o_tc = (z.o_rmax + sizeof(unsigned) * (size_t)K + 63) & ~(size_t)63;
  z.o_summ = (z.o_tc + ((ntile + 15) & ~(size_t)15) + 63) & ~(size_t)63;
  z.bytes = z.o_summ + sizeof(unsigned long long) * ntile;
  return z;
}

// trigger_impl.cc:86-109: a plateau arms the 80-item countdown at its 21st sample, or as soon as the previous countdown has run
// out if that is later and the plateau still lasts; one trigger per plateau.  sync_impl.cc:73-147: IDLE scans up to a trigger,
// SYNC needs 240 items from it and consumes 111.  The first countdown / search window that runs past the end of the segment
// stops the walk: the commit point moves to the item where that countdown was armed, and the next call derives it again.
static void det_walk(DetHost& H, const TrigCand* cand, size_t ncand, int stream, std::vector<unsigned long long>& fire, std::vector<SyncJob>& jobs)
{
  const c8p_det_state& st0 = H.st0;
  const long long N = (long long)H.n;
  H.have = st0.have_prev != 0; H.prev_start = st0.prev_start; H.pos = st0.sync_pos; H.commit_ll = N;
  H.open_fired = false; H.stopped = false;
  H.job0 = (int)jobs.size();
  for (size_t i = 0; i < ncand; i++) {
    const TrigCand& k = cand[i];
    if (st0.in_run && st0.run_fired && k.s == st0.run_start) { if (k.e == N) H.open_fired = true; continue; }   // armed before the commit point
    long long start = k.s + 20;
    if (H.have && H.prev_start + 80 > start) start = H.prev_start + 80;
    if (start >= k.e) continue;                        // the plateau ended before the previous countdown ran out
    const long long t = start + 79;
    const bool unseen = t < H.pos;                     // would fall inside the 111 items a SYNC pass consumes unseen
    if (t >= N || (!unseen && N - t < C8P_SYNC_BUF)) { // countdown or search window runs past the end: not decidable yet
      H.stopped = true; H.commit_ll = start; H.stop_run_start = k.s; H.stop_have = H.have; H.stop_prev = H.prev_start;
      break;
    }
    H.have = true; H.prev_start = start;
    if (k.e == N) H.open_fired = true;
    fire.push_back(H.out_off + (unsigned long long)t);
    if (unseen) continue;
    SyncJob j; j.t = (unsigned long long)t; j.seen = (unsigned long long)std::max(H.pos, 0ll);
    j.stream = (unsigned)stream; j.first = (unsigned)H.job0; j.kind = 0; j.rsv = 0;
    jobs.push_back(j);
    H.pos = t + C8P_SYNC_RES;
  }
  H.nj = (int)jobs.size() - H.job0;
  // one more look-back job, not a trigger: what the IDLE state has latched when it reaches the commit point
  SyncJob j; j.t = (unsigned long long)H.commit_ll; j.seen = (unsigned long long)std::min(std::max(H.pos, 0ll), H.commit_ll);
  j.stream = (unsigned)stream; j.first = (unsigned)H.job0; j.kind = 1; j.rsv = 0;
  jobs.push_back(j);
}

static int detect_run(c8p_ctx* c, const cf* d_all, uint64_t n_all, std::vector<DetHost>& hs, bool keep, bool multi, c8p_sync_evt* evt_host,
                      uint64_t evt_cap)
{
  Slot& S = *c->s;
  const int K = (int)hs.size();
  static const bool tlog = getenv("C8P_DET_TLOG") != nullptr;
  auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!tlog) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[c8p_detect] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - T0).count());
    T0 = t;
  };
  // ---- layout
  size_t ntile = 0, cap = 0;
  uint64_t out_items = 0;
  for (DetHost& H : hs) {
    H.n = H.n_total - H.n_hist;
    H.ntile = (int)((H.n + C8P_DET_TILE - 1) / C8P_DET_TILE);
    H.tile_base = (unsigned)ntile; ntile += (size_t)H.ntile;
    H.cap = H.n ? (unsigned)(H.n / 22 + 16) : 0u;        // a plateau needs 21 samples above and one below
    cap += H.cap;
    H.out_off = out_items; out_items += (H.n + 7) & ~(uint64_t)7;
    H.init.all = 0; H.init.tmax = 0.f; H.init.tstart = C8P_SEG_NONE;       // the plateau open at the commit point
    if (H.st0.in_run) { H.init.all = 1; H.init.tstart = H.st0.run_start; H.init.tmax = H.st0.run_max; }
    H.st1 = H.st0; H.commit = 0; H.n_evt = 0;
  }
  S.used[C8P_BUF_DET_RAW] = sizeof(cf) * n_all; S.ptr[C8P_BUF_DET_RAW] = d_all;
  if (ntile == 0) return C8P_OK;                          // no stream has new items
  if (ntile > 0x7fffffffu / C8P_DET_TILE_CAND) { c->err = "detect: too many items in one call"; return C8P_ERANGE; }
  const DetZero Z = det_zero_layout(ntile, K);
  int rc;
  if ((rc = ensure_detect_arenas(c, S, out_items, ntile, cap, K, keep, Z.bytes)) != C8P_OK) return rc;
  unsigned* d_cnt = (unsigned*)((char*)S.d_det_zero.p + Z.o_cnt);          // [0] candidates, [1] tile tickets, [3] a look-back gave up
  unsigned* d_rmax = (unsigned*)((char*)S.d_det_zero.p + Z.o_rmax);
  uint8_t* d_tcount = (uint8_t*)S.d_det_zero.p + Z.o_tc;
  unsigned long long* d_summ = (unsigned long long*)((char*)S.d_det_zero.p + Z.o_summ);
  const DetStreamDev* d_ds = (const DetStreamDev*)S.d_det_ds.p;
  const size_t map_off = (sizeof(DetStreamDev) * (size_t)K + 255) & ~(size_t)255;
  const unsigned short* d_map = (const unsigned short*)((const char*)S.d_det_ds.p + map_off);
  // small device <-> host exchanges of this call go through the slot's pinned staging buffer (pageable copies cost a synchronous
  // staging hop each).  up: [stream descriptors | tile -> stream map]; back with one sync: [counters | stream offsets | final segs |
  // the first `spec` candidates] (how many: a quarter more than the previous call of this context found)
  const unsigned spec = (unsigned)std::min<size_t>(cap, c->det_spec);
  const size_t up_bytes = multi ? map_off + sizeof(unsigned short) * ntile : sizeof(DetStreamDev) * (size_t)K;
  const size_t o_up = 0, o_cnt = (up_bytes + 63) & ~(size_t)63, o_soff = o_cnt + 64;
  const size_t o_fin = (o_soff + sizeof(unsigned) * (size_t)(K + 1) + 63) & ~(size_t)63, o_cand = (o_fin + sizeof(TrigSeg) * (size_t)K + 63) & ~(size_t)63;
  if ((rc = ensure_host(c, o_cand + sizeof(TrigCand) * (size_t)spec)) != C8P_OK) return rc;
  char* hp = (char*)S.h_stage;
  {
    DetStreamDev* hd = (DetStreamDev*)(hp + o_up);
    for (int s = 0; s < K; s++) {
      const DetHost& H = hs[s];
      DetStreamDev& D = hd[s];
      memset(&D, 0, sizeof(D));
      D.x = d_all + H.off + H.n_hist; D.hist = (long long)H.n_hist; D.n = H.n; D.init = H.init; D.tile_base = H.tile_base;
      D.x_al16 = (int)(((uintptr_t)D.x & 15u) == 0); D.out_off = H.out_off;
      D.conj0 = make_float2(H.st0.conj_avg.re, H.st0.conj_avg.im); D.idx_base = H.n_hist;
    }
    if (multi) {
      unsigned short* hm = (unsigned short*)(hp + o_up + map_off);
      for (int s = 0; s < K; s++) for (int t = 0; t < hs[s].ntile; t++) hm[hs[s].tile_base + (unsigned)t] = (unsigned short)s;
    }
    HIPCHK(c, upload(S.stream, S.d_det_ds.p, hp + o_up, up_bytes));
  }
  const DetHost& H0 = hs[0];
  const char* ft_env = getenv("C8P_DET_TICKETS");
  const int force_tickets = ft_env ? atoi(ft_env) : 0;   // 1: always, 2: pretend the first launch gave up (tests)
  for (int numbering = force_tickets == 1 ? 1 : 0; ; numbering = 1) {
    HIPCHK(c, hipMemsetAsync(S.d_det_zero.p, 0, Z.bytes, S.stream));      // no tile summary published yet, no candidates, no tickets
    if (numbering) c->det_ticket_scans++;
    {
      KTimer kt(c, C8P_K_PRESISO);
      if (!multi)
        hipLaunchKernelGGL(k_detect_scan, dim3((unsigned)ntile), dim3(C8P_DET_THREADS), 0, S.stream, d_all + H0.off + H0.n_hist, (long long)H0.n_hist,
                           (unsigned long long)H0.n, (int)(((uintptr_t)(d_all + H0.off + H0.n_hist) & 15u) == 0),
                           d_summ, d_cnt + 1, H0.init, (TrigSeg*)S.d_det_carry.p,
                           (uint8_t*)S.d_det_trig.p, (TrigCand*)S.d_det_tcand.p, d_tcount,
                           (float*)S.d_det_ac.p, (cf*)S.d_det_conj.p, (int)keep, numbering);
      else
        hipLaunchKernelGGL(k_detect_scan_multi, dim3((unsigned)ntile), dim3(C8P_DET_THREADS), 0, S.stream, d_ds, d_map, d_summ, d_cnt + 1,
                           (TrigSeg*)S.d_det_carry.p, (uint8_t*)S.d_det_trig.p, (TrigCand*)S.d_det_tcand.p, d_tcount,
                           (float*)S.d_det_ac.p, (cf*)S.d_det_conj.p, (int)keep, numbering);
    }
    {
      KTimer kt(c, C8P_K_TRIGGER);
      hipLaunchKernelGGL(k_cand_offsets, dim3(1), dim3(1024), 0, S.stream, (const uint8_t*)d_tcount, (int)ntile, (unsigned*)S.d_det_toff.p, d_cnt);
      hipLaunchKernelGGL(k_cand_gather, dim3((unsigned)((ntile + 255) / 256)), dim3(256), 0, S.stream, (const uint8_t*)d_tcount, (const unsigned*)S.d_det_toff.p,
                         (const TrigCand*)S.d_det_tcand.p, (int)ntile, (const unsigned long long*)d_summ, H0.init, (TrigCand*)S.d_det_cand.p, (unsigned)cap,
                         multi ? d_ds : (const DetStreamDev*)nullptr, multi ? d_map : (const unsigned short*)nullptr);
      if (multi)
        hipLaunchKernelGGL(k_stream_offsets, dim3((unsigned)((K + 256) / 256)), dim3(256), 0, S.stream, (const unsigned*)S.d_det_toff.p, (const unsigned*)d_cnt,
                           d_ds, K, (int)ntile, (unsigned*)S.d_det_soff.p);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hp + o_cnt, d_cnt, 16, hipMemcpyDeviceToHost, S.stream));
    if (multi) HIPCHK(c, hipMemcpyAsync(hp + o_soff, S.d_det_soff.p, sizeof(unsigned) * (size_t)(K + 1), hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipMemcpyAsync(hp + o_fin, S.d_det_carry.p, sizeof(TrigSeg) * (size_t)K, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipMemcpyAsync(hp + o_cand, S.d_det_cand.p, sizeof(TrigCand) * (size_t)spec, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipStreamSynchronize(S.stream));
    const bool gave_up = ((const unsigned*)(hp + o_cnt))[3] != 0 || (force_tickets == 2 && numbering == 0);
    if (!gave_up) break;
    // a tile waited for a predecessor that had not started (blockIdx numbering promises nothing): once more with tickets
    if (numbering == 1) { c->err = "detect: tile look-back timed out"; return C8P_EHIP; }
    c->det_giveups++;
    if (tlog) fprintf(stderr, "[c8p_detect] a tile's look-back gave up waiting for its predecessor; repeating the scan with ticket numbering\n");
  }
  lap("scan kernels + first sync");
  const unsigned ncand = *(const unsigned*)(hp + o_cnt);
  if (ncand > cap) { c->err = "candidate list overflow"; return C8P_ERANGE; }
  c->det_spec = std::max(4096u, ncand + ncand / 4 + 64u);
  // the plateaus of >= 21 samples, grouped by stream, in stream order
  std::vector<TrigCand>& cand = S.h_det_cand;
  cand.assign((const TrigCand*)(hp + o_cand), (const TrigCand*)(hp + o_cand) + std::min(ncand, spec));
  if (ncand > spec) {
    cand.resize(ncand);
    HIPCHK(c, hipMemcpy(cand.data() + spec, (const TrigCand*)S.d_det_cand.p + spec, sizeof(TrigCand) * (size_t)(ncand - spec), hipMemcpyDeviceToHost));
  }
  std::vector<unsigned> soff((size_t)K + 1, 0u);
  if (multi) memcpy(soff.data(), hp + o_soff, sizeof(unsigned) * (size_t)(K + 1));
  else { soff[0] = 0; soff[1] = ncand; }
  std::vector<TrigSeg> fin((const TrigSeg*)(hp + o_fin), (const TrigSeg*)(hp + o_fin) + K);     // (the pinned block is reused below)
  std::vector<unsigned long long>& fire = S.h_det_fire;
  std::vector<SyncJob>& jobs = S.h_det_jobs;
  fire.clear(); jobs.clear();
  fire.reserve(cand.size() + 1); jobs.reserve(cand.size() + (size_t)K + 1);
  for (int s = 0; s < K; s++) {
    DetHost& H = hs[s];
    if (H.n == 0) { H.job0 = (int)jobs.size(); H.nj = -1; continue; }     // nothing new: no jobs, the state stays
    if (soff[s] > soff[s + 1] || soff[s + 1] > ncand) { c->err = "detect: stream candidate offsets out of order"; return C8P_EHIP; }
    det_walk(H, cand.data() + soff[s], soff[s + 1] - soff[s], s, fire, jobs);
  }
  lap("state machines");
  const int NJ = (int)jobs.size();                     // triggers and look-backs of every stream
  const int nf = (int)fire.size();
  std::vector<c8p_sync_evt> evts;
  // pinned layout for the second exchange: [fires | jobs] up, [marks | events | valid | range maxima] down, [compacted events] up
  const size_t o_fire = 0, o_job = (o_fire + sizeof(unsigned long long) * (size_t)(nf + 1) + 15) & ~(size_t)15;
  const size_t o_mark = o_job + sizeof(SyncJob) * (size_t)(NJ + 1), o_evt = o_mark + sizeof(SyncMark) * (size_t)(NJ + 1);
  const size_t o_val = o_evt + sizeof(c8p_sync_evt) * (size_t)(NJ + 1), o_rm = (o_val + sizeof(int) * (size_t)(NJ + 1) + 63) & ~(size_t)63;
  const size_t o_rmj = (o_rm + sizeof(unsigned) * (size_t)K + 63) & ~(size_t)63;
  const size_t o_out = (o_rmj + sizeof(RmaxJob) * (size_t)K + 63) & ~(size_t)63;
  if ((rc = ensure_host(c, o_out + sizeof(c8p_sync_evt) * (size_t)(NJ + 1) + 64)) != C8P_OK) return rc;   // (phase-1 data is consumed)
  hp = (char*)S.h_stage;
  if ((rc = ensure_detect_jobs(c, S, (size_t)nf, (size_t)NJ + 1, (size_t)K)) != C8P_OK) return rc;
  if (nf) {
    memcpy(hp + o_fire, fire.data(), sizeof(unsigned long long) * (size_t)nf);
    HIPCHK(c, upload(S.stream, S.d_det_fire.p, hp + o_fire, sizeof(unsigned long long) * (size_t)nf));
    hipLaunchKernelGGL(k_trig_fire, dim3((nf + 255) / 256), dim3(256), 0, S.stream, (uint8_t*)S.d_det_trig.p,
                       (const unsigned long long*)S.d_det_fire.p, nf);
  }
  const SyncMark* marks = (const SyncMark*)(hp + o_mark);
  bool any_rmax = false;
  if (NJ) {
    memcpy(hp + o_job, jobs.data(), sizeof(SyncJob) * (size_t)NJ);
    HIPCHK(c, upload(S.stream, S.d_det_job.p, hp + o_job, sizeof(SyncJob) * (size_t)NJ));
    {
      KTimer kt(c, C8P_K_SYNCEVT);
      hipLaunchKernelGGL(k_sync_mark, dim3(NJ), dim3(64), 0, S.stream, (const SyncJob*)S.d_det_job.p, (const uint8_t*)S.d_det_trig.p, d_ds,
                         (SyncMark*)S.d_det_mark.p, NJ);
      int n_trig = 0;
      for (const DetHost& H : hs) if (H.nj > 0) n_trig += H.nj;
      if (n_trig > 0 || !multi)  // (at least one trigger somewhere; kind = 1 jobs return at once)
        hipLaunchKernelGGL(k_sync_evt, dim3(NJ), dim3(64), 0, S.stream, (const SyncJob*)S.d_det_job.p, (const SyncMark*)S.d_det_mark.p, d_ds,
                           (c8p_sync_evt*)S.d_det_evt.p, (int*)S.d_det_valid.p, NJ);
      else
        HIPCHK(c, hipMemsetAsync(S.d_det_valid.p, 0, sizeof(int) * (size_t)NJ, S.stream));
    }
    // d_conjAc at a commit point inside a plateau: maximum of ac over the plateau so far.  ac is not stored: the same arithmetic
    // once more over the tiles that stretch of the plateau touches
    int n_rmj = 0;
    unsigned rm_blocks = 0;
    for (int s = 0; s < K; s++) {
      DetHost& H = hs[s];
      H.want_rmax = false;
      if (H.nj < 0 || !H.stopped) continue;
      H.rmax0 = (H.st0.in_run && H.stop_run_start == H.st0.run_start) ? H.st0.run_max : 0.f;
      const long long lo = std::max(H.stop_run_start, 0ll);
      if (H.commit_ll <= lo) continue;
      const long long tl = lo / C8P_DET_TILE, th = (H.commit_ll - 1) / C8P_DET_TILE;
      const cf* x = d_all + H.off + H.n_hist;
      RmaxJob& J = ((RmaxJob*)(hp + o_rmj))[n_rmj++];
      J.x = x; J.hist = (long long)H.n_hist; J.n = (unsigned long long)H.n; J.tile_lo = tl; J.lo = lo; J.hi = H.commit_ll; J.out_bits = d_rmax + s;
      J.first = rm_blocks; J.x_aligned16 = (int)(((uintptr_t)x & 15u) == 0);
      rm_blocks += (unsigned)(th - tl + 1);
      H.want_rmax = true; any_rmax = true;
    }
    if (n_rmj) {
      RmaxJob* d_rmj = (RmaxJob*)((char*)S.d_det_job.p + rmax_jobs_off((size_t)NJ + 1));
      HIPCHK(c, upload(S.stream, d_rmj, hp + o_rmj, sizeof(RmaxJob) * (size_t)n_rmj));
      hipLaunchKernelGGL(k_ac_range_max, dim3(rm_blocks), dim3(C8P_DET_THREADS), 0, S.stream, (const RmaxJob*)d_rmj, n_rmj);
    }
    HIPCHK(c, hipGetLastError());
    lap("second phase enqueue");
    HIPCHK(c, hipMemcpyAsync(hp + o_evt, S.d_det_evt.p, sizeof(c8p_sync_evt) * (size_t)NJ, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipMemcpyAsync(hp + o_val, S.d_det_valid.p, sizeof(int) * (size_t)NJ, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipMemcpyAsync(hp + o_mark, S.d_det_mark.p, sizeof(SyncMark) * (size_t)NJ, hipMemcpyDeviceToHost, S.stream));
    if (any_rmax) HIPCHK(c, hipMemcpyAsync(hp + o_rm, d_rmax, sizeof(unsigned) * (size_t)K, hipMemcpyDeviceToHost, S.stream));
    HIPCHK(c, hipStreamSynchronize(S.stream));
    lap("sync kernels + second sync");
    const c8p_sync_evt* all = (const c8p_sync_evt*)(hp + o_evt);
    const int* valid = (const int*)(hp + o_val);
    for (int s = 0; s < K; s++) {
      DetHost& H = hs[s];
      if (H.nj < 0) continue;
      for (int k = H.job0; k < H.job0 + H.nj; k++) if (valid[k]) { evts.push_back(all[k]); H.n_evt++; }
    }
    if (evt_host && !evts.empty()) memcpy(evt_host, evts.data(), sizeof(c8p_sync_evt) * (size_t)std::min<uint64_t>(evts.size(), evt_cap));
    if (!evts.empty()) {    // compacted, grouped by stream, in stream order: what the sync blocks' output streams carry
      memcpy(hp + o_out, evts.data(), sizeof(c8p_sync_evt) * evts.size());
      HIPCHK(c, upload(S.stream, S.d_det_evt.p, hp + o_out, sizeof(c8p_sync_evt) * evts.size()));
    }
  }
  // ---- state at the new commit points
  for (int s = 0; s < K; s++) {
    DetHost& H = hs[s];
    if (H.nj < 0) continue;
    const c8p_det_state& st0 = H.st0;
    c8p_det_state st1;
    memset(&st1, 0, sizeof(st1));
    const long long commit = H.commit_ll;
    st1.conj_avg = st0.conj_avg;
    for (int k = H.job0 + H.nj; k >= H.job0; k--) if (marks[k].j >= 0) { st1.conj_avg.re = marks[k].conj.x; st1.conj_avg.im = marks[k].conj.y; break; }
    st1.sync_pos = H.pos - commit;
    if (H.stopped) {
      // the commit point is the item where the undecidable countdown is armed: its plateau is open, has not armed yet
      st1.in_run = 1; st1.run_fired = 0; st1.run_start = H.stop_run_start - commit;
      st1.have_prev = H.stop_have; st1.prev_start = H.stop_prev - commit;
      float rmax = H.rmax0;
      if (H.want_rmax) { float got = 0.f; memcpy(&got, hp + o_rm + sizeof(unsigned) * (size_t)s, sizeof(float)); if (got > rmax) rmax = got; }
      st1.run_max = rmax;
    } else {
      const TrigSeg& f = fin[s];
      st1.in_run = seg_has_run(f) ? 1 : 0;
      st1.run_start = st1.in_run ? f.tstart - commit : 0;
      st1.run_max = st1.in_run ? f.tmax : 0.f;
      st1.run_fired = (st1.in_run && (H.open_fired || (st0.in_run && st0.run_fired && f.tstart == st0.run_start))) ? 1 : 0;
      st1.have_prev = H.have; st1.prev_start = H.prev_start - commit;
    }
    H.st1 = st1; H.commit = (uint64_t)commit;
  }
  lap("compaction + state");
  drain_events(c);     // (the compacted events' upload may still be in flight: c8p_fetch / the next submit order behind it)
  S.used[C8P_BUF_DET_EVT] = sizeof(c8p_sync_evt) * evts.size(); S.ptr[C8P_BUF_DET_EVT] = S.d_det_evt.p;
  const uint64_t out_used = multi ? out_items : hs[0].n;      // (one stream: exactly its new items)
  S.used[C8P_BUF_DET_AC] = keep ? sizeof(float) * out_used : 0; S.ptr[C8P_BUF_DET_AC] = S.d_det_ac.p;
  S.used[C8P_BUF_DET_TRIG] = out_used; S.ptr[C8P_BUF_DET_TRIG] = S.d_det_trig.p;
  S.used[C8P_BUF_DET_CONJ] = keep ? sizeof(cf) * out_used : 0; S.ptr[C8P_BUF_DET_CONJ] = S.d_det_conj.p;
  return C8P_OK;
}

static int detect_begin(c8p_ctx* c)
{
  HIPCHK(c, hipSetDevice(c->device));
  c->cur = (c->cur + 1) % (int)c->slots.size(); c->seq++; c->slots[c->cur].ticket = c->seq;
  c->s = &c->slots[c->cur];
  Slot& S = *c->s;
  HIPCHK(c, hipStreamSynchronize(S.stream));
  drain_events(c);
  S.n_pkt = 0;
  for (int i = 0; i < C8P_BUF_COUNT; i++) { S.used[i] = 0; S.ptr[i] = nullptr; }
  return C8P_OK;
}

extern "C" int c8p_detect(c8p_ctx* c, const c8p_detect_in* in)
{
  if (!c) return C8P_EINVAL;
  if (!in || (in->n > 0 && !in->raw) || in->n_hist > in->n) { c->err = "bad detect input"; return C8P_EINVAL; }
  if (in->mem != C8P_MEM_HOST && in->mem != C8P_MEM_DEVICE) { c->err = "bad mem kind"; return C8P_EINVAL; }
  if (in->n > (1ull << 31)) { c->err = "stream segment too long (2^31 items max)"; return C8P_ERANGE; }
  int rc;
  if ((rc = detect_begin(c)) != C8P_OK) return rc;
  Slot& S = *c->s;
  std::vector<DetHost> hs(1);
  DetHost& H = hs[0];
  memset(&H.st0, 0, sizeof(H.st0));
  if (in->state) H.st0 = *in->state;
  H.off = 0; H.n_total = in->n; H.n_hist = in->n_hist;
  if (in->commit) *in->commit = in->n_hist;
  if (in->state_out) *in->state_out = H.st0;
  if (in->n == in->n_hist) return C8P_OK;
  const cf* d_all = (const cf*)in->raw;
  if (in->mem == C8P_MEM_HOST) {
    if ((rc = ensure(c, S.d_det_raw, sizeof(cf) * in->n)) != C8P_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(S.d_det_raw.p, in->raw, sizeof(cf) * in->n, hipMemcpyHostToDevice, S.stream));
    d_all = (const cf*)S.d_det_raw.p;
  }
  if ((rc = detect_run(c, d_all, in->n, hs, in->keep != 0, false, in->evt_host, in->evt_cap)) != C8P_OK) return rc;
  if (in->commit) *in->commit = in->n_hist + H.commit;
  if (in->state_out) *in->state_out = H.st1;
  return C8P_OK;
}

extern "C" int c8p_detect_multi(c8p_ctx* c, const c8p_detect_multi_in* in)
{
  if (!c) return C8P_EINVAL;
  if (!in || in->n_streams < 0 || (in->n_streams > 0 && (!in->streams || !in->raw))) { c->err = "bad detect input"; return C8P_EINVAL; }
  if (in->mem != C8P_MEM_HOST && in->mem != C8P_MEM_DEVICE) { c->err = "bad mem kind"; return C8P_EINVAL; }
  if (in->n_streams > 65535) { c->err = "at most 65535 streams per call"; return C8P_ERANGE; }
  uint64_t total = 0;
  for (int s = 0; s < in->n_streams; s++) {
    const c8p_det_stream& k = in->streams[s];
    if (k.n_hist > k.n || k.off + k.n > in->n_raw) { c->err = "stream items outside the raw arena"; return C8P_ERANGE; }
    total += k.n;
  }
  if (total > (1ull << 31)) { c->err = "too many items in one call (2^31 max)"; return C8P_ERANGE; }
  int rc;
  if ((rc = detect_begin(c)) != C8P_OK) return rc;
  Slot& S = *c->s;
  const int K = in->n_streams;
  if (K == 0) return C8P_OK;
  std::vector<DetHost> hs((size_t)K);
  for (int s = 0; s < K; s++) {
    const c8p_det_stream& k = in->streams[s];
    memset(&hs[s].st0, 0, sizeof(c8p_det_state));
    if (k.has_state) hs[s].st0 = k.state;
    hs[s].off = k.off; hs[s].n_total = k.n; hs[s].n_hist = k.n_hist;
  }
  const cf* d_all = (const cf*)in->raw;
  if (in->mem == C8P_MEM_HOST) {
    if ((rc = ensure(c, S.d_det_raw, sizeof(cf) * in->n_raw)) != C8P_OK) return rc;
    for (int s = 0; s < K; s++)
      if (hs[s].n_total)
        HIPCHK(c, hipMemcpyAsync((cf*)S.d_det_raw.p + hs[s].off, in->raw + hs[s].off, sizeof(cf) * hs[s].n_total, hipMemcpyHostToDevice, S.stream));
    d_all = (const cf*)S.d_det_raw.p;
  }
  if ((rc = detect_run(c, d_all, in->n_raw, hs, false, true, in->evt_host, in->evt_cap)) != C8P_OK) return rc;
  for (int s = 0; s < K; s++) {
    c8p_det_stream& k = in->streams[s];
    k.commit = k.n_hist + hs[s].commit; k.trig_off = hs[s].out_off; k.n_evt = hs[s].n_evt;
    k.state = hs[s].st1; k.has_state = 1;
  }
  return C8P_OK;
}

// ------------------------------------------------------------------------------------------------ transmit chain
extern "C" int c8p_tx_plan_mu(int mcs0, int len0, int mcs1, int len1, c8p_tx_plan* plan2)
{
  c8p_tx_plan a, b;
  const int ok = tx_plan_mu(mcs0, len0, mcs1, len1, a, b);
  if (plan2) { plan2[0] = a; plan2[1] = b; }
  return ok;
}
extern "C" int c8p_tx_plan_of(int format, int mcs, int nss, int len, c8p_tx_plan* plan)
{
  c8p_tx_plan p;
  const int ok = tx_plan_of(format, mcs, nss < 1 ? 1 : nss, len, p);
  if (plan) *plan = p;
  return ok;
}

// pktgen / encode / modulation for one spatial stream: PSDUs in, one sample stream out (packets
// separated by the requested gaps of zeros), plus encode's punctured coded bits for parity tests.
extern "C" int c8p_tx(c8p_ctx* c, const c8p_tx_in* in)
{
  if (!c) return C8P_EINVAL;
  if (!in || in->n_pkt < 0 || (in->n_pkt > 0 && (!in->pkts || !in->psdu))) { c->err = "bad tx input"; return C8P_EINVAL; }
  if (in->mem != C8P_MEM_HOST && in->mem != C8P_MEM_DEVICE) { c->err = "bad mem kind"; return C8P_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  c->cur = (c->cur + 1) % (int)c->slots.size(); c->seq++; c->slots[c->cur].ticket = c->seq;
  c->s = &c->slots[c->cur];
  Slot& S = *c->s;
  HIPCHK(c, hipStreamSynchronize(S.stream));
  drain_events(c);
  S.n_pkt = in->n_pkt;
  for (int i = 0; i < C8P_BUF_COUNT; i++) { S.used[i] = 0; S.ptr[i] = nullptr; }
  const int n = in->n_pkt;
  if (n == 0) return C8P_OK;
  std::vector<TxPlan> plans;                           // one per bit job: a packet, or each user of an MU-MIMO packet
  plans.reserve((size_t)n + in->n_mu);
  std::vector<c8p_tx_desc> desc(n);
  uint64_t iq = 0, coded = 0;
  uint32_t units = 0, mu_used = 0;
  int max_nss = 1;
  for (int p = 0; p < n; p++) {
    const c8p_tx_pkt& k = in->pkts[p];
    TxPlan P = TxPlan{}, P1 = TxPlan{};
    const bool mu = k.format == C8P_F_VHT_MU;
    const int nss = mu ? 2 : (k.nss < 1 ? 1 : k.nss);
    max_nss = std::max(max_nss, nss);
    if (mu) {
      if (!in->mu || mu_used >= in->n_mu || !in->bfq) { c->err = "MU-MIMO packet without its c8p_tx_mu entry / steering matrices"; return C8P_EINVAL; }
      const c8p_tx_mu& k1 = in->mu[mu_used++];
      if (!tx_plan_mu(k.mcs, (int)k.len, k1.mcs, (int)k1.len, P.p, P1.p)) { c->err = "unsupported MU-MIMO mcs / length"; return C8P_EINVAL; }
      if (k1.psdu_off + k1.len > in->n_psdu) { c->err = "psdu outside the arena"; return C8P_ERANGE; }
      if (k1.q_off + 256 > in->n_bfq) { c->err = "steering matrix outside the arena"; return C8P_ERANGE; }
      if (k1.group_id < 1 || k1.group_id > 62) { c->err = "MU-MIMO group id must be 1..62"; return C8P_EINVAL; }
      P.mu = 1; P1.mu = 2; P.gid = P1.gid = k1.group_id; P.q_off = P1.q_off = k1.q_off;
      P1.psdu_off = k1.psdu_off; P1.len = k1.len; P1.mcs = k1.mcs;
    } else if (!tx_plan_of(k.format, k.mcs, nss, (int)k.len, P.p)) { c->err = "unsupported tx format / mcs / streams / length"; return C8P_EINVAL; }
    if (k.psdu_off + k.len > in->n_psdu) { c->err = "psdu outside the arena"; return C8P_ERANGE; }
    if (k.scr_seed == 0 || k.scr_seed > 127) { c->err = "scrambler seed must be 1..127"; return C8P_EINVAL; }
    iq += k.gap;
    P.iq_off = iq; P.coded_off = coded; P.psdu_off = k.psdu_off; P.len = k.len;
    P.format = mu ? C8P_F_VHT : k.format; P.mcs = k.mcs; P.seed = k.scr_seed; P.flags = k.flags; P.mult = k.mult; P.unit_base = units; P.nss = nss;
    desc[p] = c8p_tx_desc{};
    desc[p].iq_off = iq; desc[p].coded_off = coded; desc[p].n_samp = (uint32_t)P.p.n_samp; desc[p].n_coded = (uint32_t)(P.p.nsym * P.p.ncbps);
    iq += (uint64_t)P.p.n_samp; coded += ((uint64_t)desc[p].n_coded + 15u) & ~(uint64_t)15u; units += (uint32_t)(P.p.n_samp / 80 * nss);
    plans.push_back(P);
    if (mu) {
      P1.iq_off = P.iq_off; P1.coded_off = coded; P1.format = C8P_F_VHT; P1.seed = P.seed; P1.flags = P.flags; P1.mult = P.mult;
      P1.unit_base = P.unit_base; P1.nss = 2;
      desc[p].coded_off1 = coded; desc[p].n_coded1 = (uint32_t)(P1.p.nsym * P1.p.ncbps);
      coded += ((uint64_t)desc[p].n_coded1 + 15u) & ~(uint64_t)15u;
      plans.push_back(P1);
    }
  }
  const int nj = (int)plans.size();
  iq += in->tail_gap;
  const uint64_t stream_len = iq;                      // items per chain
  for (TxPlan& P : plans) P.stride = stream_len;
  iq *= (uint64_t)max_nss;
  int rc;
  const size_t plan_bytes = align256(sizeof(TxPlan) * (size_t)nj), desc_bytes = align256(sizeof(c8p_tx_desc) * (size_t)n);
  if ((rc = ensure_host(c, plan_bytes + desc_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_tx_plan, plan_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_tx_desc, desc_bytes)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_tx_coded, coded + 16)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_tx_sig, (size_t)C8P_TX_SIGBITS * (size_t)nj)) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_tx_u2p, sizeof(unsigned) * (size_t)(units + 1))) != C8P_OK) return rc;
  if ((rc = ensure(c, S.d_tx_iq, sizeof(cf) * (size_t)(iq + 1))) != C8P_OK) return rc;
  HIPCHK(c, hipStreamSynchronize(S.stream));
  memcpy(S.h_stage, plans.data(), sizeof(TxPlan) * (size_t)nj);
  memcpy((char*)S.h_stage + plan_bytes, desc.data(), sizeof(c8p_tx_desc) * (size_t)n);
  HIPCHK(c, hipMemcpyAsync(S.d_tx_plan.p, S.h_stage, sizeof(TxPlan) * (size_t)nj, hipMemcpyHostToDevice, S.stream));
  HIPCHK(c, hipMemcpyAsync(S.d_tx_desc.p, (char*)S.h_stage + plan_bytes, sizeof(c8p_tx_desc) * (size_t)n, hipMemcpyHostToDevice, S.stream));
  const uint8_t* d_psdu = in->psdu;
  if (in->mem == C8P_MEM_HOST) {
    if ((rc = ensure(c, S.d_tx_psdu, in->n_psdu + 16)) != C8P_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(S.d_tx_psdu.p, in->psdu, in->n_psdu, hipMemcpyHostToDevice, S.stream));
    d_psdu = (const uint8_t*)S.d_tx_psdu.p;
  }
  const cf* d_bfq = (const cf*)in->bfq;
  if (in->mem == C8P_MEM_HOST && mu_used) {
    if ((rc = ensure(c, S.d_tx_bfq, sizeof(cf) * in->n_bfq + 16)) != C8P_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(S.d_tx_bfq.p, in->bfq, sizeof(cf) * in->n_bfq, hipMemcpyHostToDevice, S.stream));
    d_bfq = (const cf*)S.d_tx_bfq.p;
  }
  HIPCHK(c, hipMemsetAsync(S.d_tx_iq.p, 0, sizeof(cf) * (size_t)iq, S.stream));      // the gaps
  {
    KTimer kt(c, C8P_K_TXBITS);
    hipLaunchKernelGGL(k_tx_bits, dim3(nj), dim3(64), 0, S.stream, (const TxPlan*)S.d_tx_plan.p, d_psdu, (const DevTables*)c->d_tab,
                       (uint8_t*)S.d_tx_coded.p, (uint8_t*)S.d_tx_sig.p, (unsigned*)S.d_tx_u2p.p, nj);
  }
  {
    KTimer kt(c, C8P_K_TXSYM);
    hipLaunchKernelGGL(k_tx_sym, dim3((units + 3) / 4), dim3(256), 0, S.stream, (const TxPlan*)S.d_tx_plan.p, (const unsigned*)S.d_tx_u2p.p,
                       (const uint8_t*)S.d_tx_coded.p, (const uint8_t*)S.d_tx_sig.p, d_bfq, (const DevTables*)c->d_tab, (cf*)S.d_tx_iq.p, units);
  }
  HIPCHK(c, hipGetLastError());
  S.used[C8P_BUF_TX_IQ] = sizeof(cf) * (size_t)iq; S.ptr[C8P_BUF_TX_IQ] = S.d_tx_iq.p;
  S.used[C8P_BUF_TX_DESC] = sizeof(c8p_tx_desc) * (size_t)n; S.ptr[C8P_BUF_TX_DESC] = S.d_tx_desc.p;
  S.used[C8P_BUF_TX_CODED] = coded; S.ptr[C8P_BUF_TX_CODED] = S.d_tx_coded.p;
  return C8P_OK;
}

extern "C" int c8p_sync(c8p_ctx* c)
{
  if (!c) return C8P_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  for (Slot& sl : c->slots) HIPCHK(c, hipStreamSynchronize(sl.stream));
  drain_events(c);
  return C8P_OK;
}

// Tickets: consecutive submits rotate through the slots, so the result arenas of the last slots.size() submits are all
// resident; c8p_select() points c8p_buffer / c8p_fetch at an earlier one (a caller that overlaps the upload of batch i+1
// with the download of batch i).  The next submit makes its own slot current again.
extern "C" int c8p_ticket(c8p_ctx* c) { return c ? (int)(c->seq & 0x7fffffffu) : C8P_EINVAL; }
extern "C" int c8p_select(c8p_ctx* c, int ticket)
{
  if (!c) return C8P_EINVAL;
  for (Slot& sl : c->slots)
    if (sl.ticket != 0 && (int)(sl.ticket & 0x7fffffffu) == ticket) { c->s = &sl; return C8P_OK; }
  c->err = "ticket is not resident any more (its slot has been reused) or was never issued";
  return C8P_ERANGE;
}

extern "C" const void* c8p_buffer(c8p_ctx* c, int which, uint64_t* bytes)
{
  if (!c || which < 0 || which >= C8P_BUF_COUNT) return nullptr;
  if (bytes) *bytes = c->s->used[which];
  return c->s->ptr[which];
}

extern "C" int c8p_fetch(c8p_ctx* c, int which, void* dst, uint64_t offset, uint64_t bytes)
{
  if (!c || which < 0 || which >= C8P_BUF_COUNT || !dst) return C8P_EINVAL;
  if (offset + bytes > c->s->used[which]) { c->err = "fetch outside the arena"; return C8P_ERANGE; }
  HIPCHK(c, hipSetDevice(c->device));
  if (bytes) HIPCHK(c, hipMemcpyAsync(dst, (const char*)c->s->ptr[which] + offset, bytes, hipMemcpyDeviceToHost, c->s->stream));
  HIPCHK(c, hipStreamSynchronize(c->s->stream));
  drain_events(c);
  return C8P_OK;
}

// c8p_fetch without the stream synchronisation: the copy is only enqueued behind the selected submit's kernels; dst should be
// pinned (c8p_host_alloc), otherwise the runtime stages the copy and the call blocks anyway.  Complete after c8p_wait(ticket)
// of that submit or c8p_sync().
extern "C" int c8p_fetch_async(c8p_ctx* c, int which, void* dst, uint64_t offset, uint64_t bytes)
{
  if (!c || which < 0 || which >= C8P_BUF_COUNT || !dst) return C8P_EINVAL;
  if (offset + bytes > c->s->used[which]) { c->err = "fetch outside the arena"; return C8P_ERANGE; }
  HIPCHK(c, hipSetDevice(c->device));
  // hipMemcpyDefault: dst may also be device memory (a caller gathering fixed-stride result arenas on the device, shard.py)
  if (bytes) HIPCHK(c, hipMemcpyAsync(dst, (const char*)c->s->ptr[which] + offset, bytes, hipMemcpyDefault, c->s->stream));
  return C8P_OK;
}

// Wait for ONE submit: its uploads, kernels and the c8p_fetch_async copies enqueued behind it (c8p_sync waits for all slots).
extern "C" int c8p_wait(c8p_ctx* c, int ticket)
{
  if (!c) return C8P_EINVAL;
  for (Slot& sl : c->slots)
    if (sl.ticket != 0 && (int)(sl.ticket & 0x7fffffffu) == ticket) {
      HIPCHK(c, hipSetDevice(c->device));
      HIPCHK(c, hipStreamSynchronize(sl.stream));
      drain_events(c);
      return C8P_OK;
    }
  c->err = "ticket is not resident any more (its slot has been reused) or was never issued";
  return C8P_ERANGE;
}

extern "C" int c8p_done(c8p_ctx* c, int ticket)
{
  if (!c) return C8P_EINVAL;
  for (Slot& sl : c->slots)
    if (sl.ticket != 0 && (int)(sl.ticket & 0x7fffffffu) == ticket) {
      HIPCHK(c, hipSetDevice(c->device));
      const hipError_t q = hipStreamQuery(sl.stream);
      if (q == hipSuccess) return 1;
      if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
      c->err = std::string("hipStreamQuery: ") + hipGetErrorString(q);
      return C8P_EHIP;
    }
  c->err = "ticket is not resident any more (its slot has been reused) or was never issued";
  return C8P_ERANGE;
}

// Pinned host memory for the caller's staging arenas: a C8P_MEM_HOST submit reads them with asynchronous copies (they must stay
// untouched until c8p_wait / c8p_sync), c8p_fetch_async writes into them.  Plain malloc memory works everywhere too, at
// pageable-copy speed and without overlap.
extern "C" void* c8p_host_alloc(uint64_t bytes)
{
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
extern "C" void c8p_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ------------------------------------------------------------------------------------------------ counters, reserve, caller arenas
extern "C" int64_t c8p_counter(c8p_ctx* c, int which)
{
  if (!c) return C8P_EINVAL;
  switch (which) {
    case C8P_CNT_ARENA_GROWTHS: return c->n_grow;
    case C8P_CNT_ARENA_BYTES: return c->dev_bytes;
    case C8P_CNT_HOST_BYTES: return c->host_bytes;
    case C8P_CNT_DET_GIVEUPS: return c->det_giveups;
    case C8P_CNT_DET_TICKET_SCANS: return c->det_ticket_scans;
    case C8P_CNT_SLOTS: return (int64_t)c->slots.size();
    case C8P_CNT_SUBMITS: return (int64_t)c->seq;
    case C8P_CNT_DET_TILE: return C8P_DET_TILE;
    default: return C8P_EINVAL;
  }
}

// Every slot's arenas grown to what submits of the given shape will ask for (the same size formulas: the submit paths call the same
// ensure_* helpers).  Slots that are in flight are synchronised by ensure() before an arena of theirs is replaced.
extern "C" int c8p_reserve(c8p_ctx* c, const c8p_reserve_in* in)
{
  if (!c) return C8P_EINVAL;
  if (!in || in->rx_packets < 0 || in->sig_events < 0 || in->det_events < 0 || in->det_streams < 0) { c->err = "bad reserve request"; return C8P_EINVAL; }
  if (in->rx_packets > 0 && in->rx_ant != 1 && in->rx_ant != 2) { c->err = "rx_ant must be 1 or 2"; return C8P_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  Slot* const keep_sel = c->s;
  int rc = C8P_OK;
  for (Slot& S : c->slots) {
    c->s = &S;
    if (in->rx_packets > 0) {
      const int n = in->rx_packets;
      std::vector<c8p_pkt_in> pk((size_t)n);
      memset(pk.data(), 0, sizeof(c8p_pkt_in) * (size_t)n);
      for (int p = 0; p < n; p++) { pk[p].n_items = in->rx_items; pk[p].n_ant = (uint8_t)in->rx_ant; }
      c8p_batch_in b;
      memset(&b, 0, sizeof(b));
      b.n_pkt = n; b.pkts = pk.data(); b.n_iq = (uint64_t)n * in->rx_items * (uint64_t)in->rx_ant; b.n_chan = 64ull * (uint64_t)n;
      Plan P;
      if ((rc = plan_demod(c, &b, P)) != C8P_OK) break;
      if ((rc = ensure_demod_arenas(c, n, P, in->rx_host ? b.n_iq : 0, in->rx_host ? b.n_chan : 0)) != C8P_OK) break;
      if ((rc = ensure_decode_arenas(c, n, P)) != C8P_OK) break;
    }
    if (in->sig_events > 0) {
      const size_t n = (size_t)in->sig_events;
      if ((rc = ensure_signal_arenas(c, n, in->sig_items, in->sig_host ? in->sig_raw : 0)) != C8P_OK) break;
    }
    if (in->det_items > 0) {
      const int K = std::max(1, in->det_streams);
      // (every stream may end inside a tile of its own and supplies C8P_DET_HIST items that are not new)
      const size_t ntile = (size_t)(in->det_items / C8P_DET_TILE) + (size_t)K + 1;
      const size_t cap = (size_t)(in->det_items / 22) + 16 * (size_t)K;
      const uint64_t out_items = in->det_items + 8ull * (uint64_t)K;
      const DetZero Z = det_zero_layout(ntile, K);
      if ((rc = ensure_detect_arenas(c, S, out_items, ntile, cap, K, in->det_keep != 0, Z.bytes)) != C8P_OK) break;
      if (in->det_host && (rc = ensure(c, S.d_det_raw, sizeof(cf) * in->det_items)) != C8P_OK) break;
      const size_t nj = (size_t)in->det_events + (size_t)K + 1;
      if ((rc = ensure_detect_jobs(c, S, nj, nj, (size_t)K)) != C8P_OK) break;
      // the pinned exchanges of one call: phase 1 [descriptors | tile map | counters | offsets | final segs | candidates], phase 2 [fires | jobs | marks | events | valid | maxima | events]
      const size_t ph1 = sizeof(DetStreamDev) * (size_t)K + 2 * ntile + 512 + 4 * (size_t)(K + 1) + sizeof(TrigSeg) * (size_t)K + 256 + sizeof(TrigCand) * std::min<size_t>(cap, 8192);
      const size_t ph2 = (8 + sizeof(SyncJob) + sizeof(SyncMark) + 2 * sizeof(c8p_sync_evt) + 4) * (nj + 1) + (4 + sizeof(RmaxJob)) * (size_t)K + 768;
      if ((rc = ensure_host(c, std::max(ph1, ph2))) != C8P_OK) break;
    }
  }
  c->s = keep_sel;
  return rc;
}

extern "C" void* c8p_dev_alloc(c8p_ctx* c, uint64_t bytes)
{
  if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { c->err = "c8p_dev_alloc: hipMalloc failed"; return nullptr; }
  return p;
}
extern "C" void c8p_dev_free(c8p_ctx* c, void* p)
{
  if (!p) return;
  if (c) {                                     // (NULL: the ctx that allocated it is gone, and with it anything it had in flight)
    (void)hipSetDevice(c->device);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->out_stream) (void)hipStreamSynchronize(c->out_stream);
    if (c->copy_stream_b) (void)hipStreamSynchronize(c->copy_stream_b);
  }
  (void)hipFree(p);
}
extern "C" int c8p_option(c8p_ctx* c, int which, int64_t value)
{
  if (!c) return C8P_EINVAL;
  if (which == C8P_OPT_COPY_CUS) {
    std::lock_guard<std::mutex> g(c->copy_mu);
    if (c->copy_stream || value < 0 || value > 256) { c->err = "c8p_option(C8P_OPT_COPY_CUS): 0..256, before the ctx's first copy"; return C8P_EINVAL; }
    c->opt_copy_cus = (int)value;
    return C8P_OK;
  }
  c->err = "c8p_option: unknown option";
  return C8P_EINVAL;
}
extern "C" int c8p_copy_wait(c8p_ctx* c);
static int copy_stream_of(c8p_ctx* c, int first_fmt = C8P_FMT_C32)
{
  HIPCHK(c, hipSetDevice(c->device));
  std::lock_guard<std::mutex> g(c->copy_mu);
  if (!c->copy_stream) {
    // The gather kernel's waves spend their lives waiting for host memory; on a CU they share with the waves of the chain's kernels they
    // hold its memory pipeline, and with a gather wave on EVERY CU (1024 workgroups over 256 CUs) every kernel beside it ran 2-10 x slower
    // (NOTES/frontend.md).  So the copy stream may only use C8P_COPY_CUS compute units (default 32 = 4 of the 32 of every XCD — the low
    // bits of a CU mask go round the XCDs — each filled to the brim with gather waves), and the other 224 never see one: 64 live streams
    // 6.0-6.3 -> 5.25 ms a round, sc16 3.7 -> 2.9.  0: no mask.  (hipExtStreamCreateWithCUMask has no flags argument: the stream is a
    // BLOCKING one, it orders itself against work on the null stream — which this library never uses.)
    // 32 / 48 / 64 CUs, ms per round: 64 streams of complex64 5.28-5.40 / 5.76 / 5.66; 128 of sc16 6.32 / 5.94 / 5.96; 128 of sc8 4.29 /
    // 4.23 / 3.76 — narrow items leave the link slack and the conversion wants lanes: the first gather's item format decides.
    const char* ce = getenv("C8P_COPY_CUS");
    int ncu = ce ? atoi(ce) : c->opt_copy_cus >= 0 ? c->opt_copy_cus : (first_fmt == C8P_FMT_C32 ? 32 : 64);
    int dev_cus = 0;
    if (hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) dev_cus = 0;
    if (ncu < 0 || ncu >= dev_cus || dev_cus > 256) ncu = 0;
    bool masked = false;
    if (ncu > 0) {
      uint32_t* mask = c->copy_mask;
      for (int i = 0; i < ncu; i++) mask[i >> 5] |= 1u << (i & 31);
      masked = hipExtStreamCreateWithCUMask(&c->copy_stream, 8, mask) == hipSuccess;
      if (!masked) { (void)hipGetLastError(); c->copy_stream = nullptr; }
    }
    if (!masked) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    c->copy_cus = masked ? ncu : 0;
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_before_spans, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_spans, hipEventDisableTiming));
    // (both side streams from the start.  Created at their first use instead — a process's streams share four hardware queues, and the
    // idea was to leave them to the submit streams — the device -> host stream lands on a queue a submit stream uses: block chain plain
    // 275-301 k against 327-335 k packets/s on one box, fused no different; profiles/r5q/submit_breakdown.txt)
    HIPCHK(c, hipStreamCreateWithFlags(&c->span_stream, hipStreamNonBlocking));
    // The device -> host stream on a hardware queue of its own: a process's ordinary streams share four of them (GPU_MAX_HW_QUEUES) and this
    // one would sit behind a submit stream's 0.3 ms kernels; a stream with a CU mask gets its own queue, and the mask names every CU.  Block
    // chain plain 345-353 k against 331-336 k packets/s, alternated on one box (C8P_OUT_OWN_QUEUE=0: the shared queue;
    // profiles/r5q/submit_breakdown.txt).  The same for the submit streams: no difference, chain or headline (measured, not kept).
    const char* oq = getenv("C8P_OUT_OWN_QUEUE");
    if (!(oq && oq[0] == '0') && dev_cus > 0 && dev_cus <= 256) {
      uint32_t all[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < dev_cus; i++) all[i >> 5] |= 1u << (i & 31);
      if (hipExtStreamCreateWithCUMask(&c->out_stream, 8, all) != hipSuccess) { (void)hipGetLastError(); c->out_stream = nullptr; }
    }
    if (!c->out_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
    // the copy stream's twin (same CU mask) with the others, not at the first unordered gather: c8p_dev_free on another thread reads the
    // handle without the lock, and marks / fences / c8p_copy_async record on it every time
    if (c->copy_cus) { if (hipExtStreamCreateWithCUMask(&c->copy_stream_b, 8, c->copy_mask) != hipSuccess) { (void)hipGetLastError(); c->copy_stream_b = nullptr; } }
    if (!c->copy_stream_b) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream_b, hipStreamNonBlocking));
  }
  return C8P_OK;
}
extern "C" uint64_t c8p_copy_mark(c8p_ctx* c)
{
  if (!c || copy_stream_of(c) != C8P_OK) return 0;
  std::lock_guard<std::mutex> g(c->copy_mu);
  hipEvent_t& e = c->copy_ev[c->copy_marks % 64];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 0;
  if (c->spans_pending) {                      // "every copy enqueued so far" includes the span copies, which run on their own stream
    if (hipStreamWaitEvent(c->copy_stream, c->ev_spans, 0) != hipSuccess) return 0;
    c->spans_pending = false;
  }
  if (hipEventRecord(e, c->copy_stream) != hipSuccess) return 0;
  const size_t slot = c->copy_marks % 64;
  // ... and what went the other way (its own stream), and the unordered gathers on the copy stream's twin: recorded when something was
  // enqueued there since the last mark; a mark that records nothing there still WAITS for the newest event an earlier mark recorded
  // (out_of / b_of) — a mark covers everything enqueued before it on any of the three streams, which is what the callers assume
  // (gather_lane::wait_range waits for the newest overlapping mark only).  Round 5 forgot the earlier events: waiting for mark M2 (a
  // gather on the copy stream) did not wait for M1's gather on the twin.  Recording on all three streams at every mark
  // instead costs the block chain 10-20 % of its rate (profiles/r6a/README.txt): three records and up to three queries per mark.
  if (c->out_dirty) {
    hipEvent_t& eo = c->out_ev[slot];
    if (!eo && hipEventCreateWithFlags(&eo, hipEventDisableTiming) != hipSuccess) return 0;
    if (hipEventRecord(eo, c->out_stream) != hipSuccess) return 0;
    c->out_last = eo;
    c->out_dirty = false;
    c->out_gen++;
  }
  c->out_of[slot] = c->out_last; c->out_of_gen[slot] = c->out_gen;
  if (c->b_dirty) {
    hipEvent_t& eb = c->b_ev[slot];
    if (!eb && hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) return 0;
    if (hipEventRecord(eb, c->copy_stream_b) != hipSuccess) return 0;
    c->b_last = eb;
    c->b_dirty = false;
    c->b_gen++;
  }
  c->b_of[slot] = c->b_last; c->b_of_gen[slot] = c->b_gen;
  return ++c->copy_marks;
}
// the events mark `mark` stands for: the copy stream's own, and — unless a waiter has already seen that generation complete — the newest
// ones of the device -> host stream and of the copy stream's twin at the time of the mark
struct MarkEvents { hipEvent_t e = nullptr, eo = nullptr, eb = nullptr; uint64_t go = 0, gb = 0; };
static bool mark_events(c8p_ctx* c, uint64_t mark, MarkEvents& m)
{
  std::lock_guard<std::mutex> g(c->copy_mu);
  if (mark > c->copy_marks) return false;
  const size_t k = (mark - 1) % 64;
  m.e = c->copy_ev[k];                          // (re-recorded by a later mark if 64 marks have passed: a later point of the same stream)
  m.go = c->out_of_gen[k]; m.gb = c->b_of_gen[k];
  if (m.go > c->out_done_gen) m.eo = c->out_of[k];
  if (m.gb > c->b_done_gen) m.eb = c->b_of[k];
  return true;
}
static void mark_seen_done(c8p_ctx* c, const MarkEvents& m, bool o_done, bool b_done)
{
  if (!(o_done && m.eo) && !(b_done && m.eb)) return;
  std::lock_guard<std::mutex> g(c->copy_mu);
  if (o_done && m.eo) c->out_done_gen = std::max(c->out_done_gen, m.go);
  if (b_done && m.eb) c->b_done_gen = std::max(c->b_done_gen, m.gb);
}
extern "C" int c8p_copy_wait_mark(c8p_ctx* c, uint64_t mark)
{
  if (!c) return C8P_EINVAL;
  if (mark == 0) return c8p_copy_wait(c);
  if (hipSetDevice(c->device) != hipSuccess) return C8P_EHIP;
  MarkEvents m;
  if (!mark_events(c, mark, m)) return C8P_EINVAL;
  // a short wait is spun out on the event's state before hipEventSynchronize (which may put the thread to sleep on the signal's
  // interrupt) is asked: the block fronts wait 20-50 us for a gather, thousands of times a second (profiles/r5q/gather_waits.txt: the
  // waits themselves did not get shorter — what is waited for is the gather)
  for (hipEvent_t ev : {m.e, m.eb, m.eo}) {
    if (!ev) continue;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t q;
    bool spun = false;
    while ((q = hipEventQuery(ev)) == hipErrorNotReady) {
      spun = true;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) { q = hipEventSynchronize(ev); break; }
      __builtin_ia32_pause();
    }
    if (spun) (void)hipGetLastError();            // ("not ready" must not be what the thread's next hipGetLastError finds)
    if (q != hipSuccess) { (void)hipGetLastError(); return C8P_EHIP; }
  }
  mark_seen_done(c, m, true, true);
  return C8P_OK;
}
extern "C" int c8p_copy_mark_done(c8p_ctx* c, uint64_t mark)
{
  if (!c || mark == 0) return C8P_EINVAL;
  if (hipSetDevice(c->device) != hipSuccess) return C8P_EHIP;
  MarkEvents m;
  if (!mark_events(c, mark, m)) return C8P_EINVAL;
  bool o_done = false, b_done = false;
  int res = 1;
  for (hipEvent_t ev : {m.eb, m.eo, m.e}) {
    if (!ev) continue;
    const hipError_t q = hipEventQuery(ev);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); res = 0; break; }
    if (q != hipSuccess) { (void)hipGetLastError(); return C8P_EHIP; }
    if (ev == m.eb) b_done = true;
    if (ev == m.eo) o_done = true;
  }
  mark_seen_done(c, m, o_done, b_done);
  return res;
}
extern "C" int c8p_copy_async(c8p_ctx* c, void* dst, const void* src, uint64_t bytes)
{
  if (!c || (bytes && (!dst || !src))) return C8P_EINVAL;
  int rc;
  if ((rc = copy_stream_of(c)) != C8P_OK) return rc;
  if (!bytes) return C8P_OK;
  // behind every gather enqueued so far, on EITHER copy stream: unordered gathers alternate between the copy stream and its twin, and what
  // this call moves (the block fronts: a partly received packet, device to device) may have been written by one on the twin
  std::lock_guard<std::mutex> g(c->copy_mu);
  if (!c->ev_async_b) HIPCHK(c, hipEventCreateWithFlags(&c->ev_async_b, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_async_b, c->copy_stream_b));
  HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_async_b, 0));
  HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->copy_stream));
  return C8P_OK;
}
// a wave = one chunk of 512 items of one span: the span table is searched once per wave (a uniform walk), then every lane has eight
// coalesced 8-byte loads in flight (source and destination offsets of a span may differ in parity, so 16-byte accesses cannot be
// aligned on both sides).  Round 4, before: one item per thread, the table searched per thread — 225 MB of carry in 0.27-0.45 ms.
#define C8P_SPAN_CHUNK 512
__global__ __launch_bounds__(256) void k_copy_spans(cf* __restrict__ dst, const cf* __restrict__ src, const c8p_span* __restrict__ sp,
                                                    const unsigned long long* __restrict__ first, int n, unsigned long long total)
{
  const unsigned long long w = __builtin_amdgcn_readfirstlane((unsigned)(((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  if (w >= total) return;
  int lo = 0, hi = n - 1;                              // last span whose first chunk is <= w
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first[mid] <= w) lo = mid; else hi = mid - 1; }
  const c8p_span S = sp[lo];
  const unsigned long long k0 = (w - first[lo]) * C8P_SPAN_CHUNK + (threadIdx.x & 63u);
  const cf* s = src + S.src_off;
  cf* d = dst + S.dst_off;
  cf v[C8P_SPAN_CHUNK / 64];
#pragma unroll
  for (int j = 0; j < C8P_SPAN_CHUNK / 64; j++) if (k0 + 64u * j < S.n) v[j] = s[k0 + 64u * j];
#pragma unroll
  for (int j = 0; j < C8P_SPAN_CHUNK / 64; j++) if (k0 + 64u * j < S.n) d[k0 + 64u * j] = v[j];
}
extern "C" int c8p_copy_spans(c8p_ctx* c, c8p_c32* dst, const c8p_c32* src, const c8p_span* spans, int32_t n)
{
  if (!c || n < 0 || (n > 0 && (!dst || !src || !spans))) return C8P_EINVAL;
  int rc;
  if ((rc = copy_stream_of(c)) != C8P_OK) return rc;
  if (n == 0) return C8P_OK;
  const size_t tab = (sizeof(c8p_span) + sizeof(unsigned long long)) * (size_t)n;
  HIPCHK(c, hipStreamSynchronize(c->span_stream));        // (the previous table may still be read: that copy is a round old)
  if (tab > c->spans2_cap) {
    if (c->h_spans2) HIPCHK(c, hipHostFree(c->h_spans2));
    if (c->d_spans2) HIPCHK(c, hipFree(c->d_spans2));
    c->h_spans2 = c->d_spans2 = nullptr; c->spans2_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->h_spans2, tab * 2 + 4096, hipHostMallocDefault));
    HIPCHK(c, hipMalloc(&c->d_spans2, tab * 2 + 4096));
    c->spans2_cap = tab * 2 + 4096;
  }
  c8p_span* hs = (c8p_span*)c->h_spans2;
  unsigned long long* hf = (unsigned long long*)((char*)c->h_spans2 + sizeof(c8p_span) * (size_t)n);
  unsigned long long total = 0;
  for (int i = 0; i < n; i++) { hs[i] = spans[i]; hf[i] = total; total += (spans[i].n + C8P_SPAN_CHUNK - 1) / C8P_SPAN_CHUNK; }      // chunks
  if (total == 0) return C8P_OK;
  // in stream order behind everything enqueued on the copy stream so far (the gather that filled the spans' source), but on the
  // span stream: a device-to-device copy wants the whole chip, the copy stream has 32 CUs
  std::lock_guard<std::mutex> g(c->copy_mu);
  HIPCHK(c, hipEventRecord(c->ev_before_spans, c->copy_stream));
  HIPCHK(c, hipStreamWaitEvent(c->span_stream, c->ev_before_spans, 0));
  HIPCHK(c, upload(c->span_stream, c->d_spans2, c->h_spans2, tab));
  hipLaunchKernelGGL(k_copy_spans, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, c->span_stream, (cf*)dst, (const cf*)src,
                     (const c8p_span*)c->d_spans2, (const unsigned long long*)((const char*)c->d_spans2 + sizeof(c8p_span) * (size_t)n), n, total);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev_spans, c->span_stream));
  c->spans_pending = true;
  return C8P_OK;
}
// work item = 16 bytes (two items) of one piece; pieces start at any item, so the first / last item of a piece may be moved alone.
// A bounded grid walks the work items with a stride (at most C8P_GATHER_WGS workgroups of 256 threads, four independent 16-byte loads per
// thread and step: a 55 GB/s link with ~2.5 us of latency needs ~140 KB in flight).  Measured with 64 / 256 / 1024 workgroups and with
// one thread per work item (65 000 workgroups): the same 54-56 GB/s, and the same slow-down of the pipeline's other kernels and
// copies beside it — while the host-to-device direction of the link is saturated the stages' own small transfers take milliseconds,
// whichever engine moves the bulk (NOTES/frontend.md).
#ifndef C8P_GATHER_WGS
#define C8P_GATHER_WGS 1024
#endif
struct PieceDev { const cf* src; unsigned long long dst_off, n, first; float scale; unsigned rsv; };      // first: index of the piece's first work item
// A short piece list travels in the launch's own arguments (no table upload, no table to protect from the next call: what a block
// mirror enqueues per general_work call is a handful of pieces); a long one (a live round's hundreds) through the table in device memory.
struct PieceTab { PieceDev p[C8P_GATHER_INLINE]; };
__device__ __forceinline__ void gather_item(cf* __restrict__ dst, const PieceDev* __restrict__ pc, int n, unsigned long long i, float4& v, cf*& d,
                                            int& kind)
{
  int lo = 0, hi = n - 1;                              // last piece whose first work item is <= i
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pc[mid].first <= i) lo = mid; else hi = mid - 1; }
  const PieceDev P = pc[lo];
  // work item k of a piece covers its items [2k - a, 2k - a + 2) where a = 1 if the source starts on an odd item (so that the
  // 16-byte loads are aligned), clipped to the piece
  const unsigned long long k = i - P.first;
  const unsigned a = (unsigned)(((uintptr_t)P.src >> 3) & 1u);
  const long long i0 = 2 * (long long)k - (long long)a;
  const cf* s = P.src + i0;
  d = dst + P.dst_off + i0;
  if (i0 >= 0 && (unsigned long long)i0 + 2 <= P.n) { v = *(const float4*)s; kind = 3; }
  else {
    kind = 0;
    v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 >= 0 && (unsigned long long)i0 < P.n) { v.x = s[0].x; v.y = s[0].y; kind |= 1; }
    if (i0 + 1 >= 0 && (unsigned long long)(i0 + 1) < P.n) { v.z = s[1].x; v.w = s[1].y; kind |= 2; }
  }
}
__device__ __forceinline__ void gather_store(const float4 v, cf* d, int kind)
{
  if (kind == 3 && (((uintptr_t)d) & 15u) == 0) *(float4*)d = v;
  else { if (kind & 1) d[0] = mk(v.x, v.y); if (kind & 2) d[1] = mk(v.z, v.w); }
}
__global__ __launch_bounds__(256) void k_copy_gather(cf* __restrict__ dst, const PieceDev* __restrict__ pc_mem, const PieceTab tab, int n,
                                                     unsigned long long total)
{
  const PieceDev* __restrict__ pc = pc_mem ? pc_mem : tab.p;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += 4 * stride) {
    float4 v[4];
    cf* d[4];
    int kind[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { kind[u] = 0; if (i + u * stride < total) gather_item(dst, pc, n, i + u * stride, v[u], d[u], kind[u]); }      // four loads in flight
#pragma unroll
    for (int u = 0; u < 4; u++) if (kind[u]) gather_store(v[u], d[u], kind[u]);
  }
}
// C8P_FMT_SC16 / C8P_FMT_SC8 pieces: an item is IB = 4 / 2 bytes at the source (int16 I, int16 Q / int8 I, int8 Q) and a complex64
// in the arena.  Work item k of a piece = the 16-byte aligned chunk k of its source = its NI = 16 / IB items [NI k - a, NI k - a + NI),
// a = the items between the aligned address below the source and the source; clipped to the piece.
template <int IB> __device__ __forceinline__ unsigned q_raw(const uint4& v, int u)      // item u of a chunk, as stored
{
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  return IB == 4 ? w[u] : (w[u >> 1] >> (16 * (u & 1))) & 0xffffu;
}
template <int IB> __device__ __forceinline__ cf q_item(unsigned r, float scale)
{
  return IB == 4 ? mk((float)(short)(r & 0xffffu) * scale, (float)(short)(r >> 16) * scale)
                 : mk((float)(signed char)(r & 0xffu) * scale, (float)(signed char)(r >> 8) * scale);
}
template <int IB> __device__ __forceinline__ void gather_item_q(cf* __restrict__ dst, const PieceDev* __restrict__ pc, int n, unsigned long long i, uint4& v,
                                                                cf*& d, unsigned& kind, float& scale)
{
  constexpr int NI = 16 / IB;
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pc[mid].first <= i) lo = mid; else hi = mid - 1; }
  const PieceDev P = pc[lo];
  const unsigned long long k = i - P.first;
  const unsigned char* src = (const unsigned char*)P.src;
  const unsigned a = (unsigned)(((uintptr_t)src / IB) & (NI - 1));
  const long long i0 = NI * (long long)k - (long long)a;
  const unsigned char* s = src + i0 * IB;
  d = dst + P.dst_off + i0;
  scale = P.scale;
  if (i0 >= 0 && (unsigned long long)i0 + NI <= P.n) { v = *(const uint4*)s; kind = (1u << NI) - 1u; }
  else {                                               // the piece's first or last chunk: item by item, nothing outside the piece is read
    const long long last = (long long)P.n;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    kind = 0;
#pragma unroll
    for (int u = 0; u < NI; u++)
      if (i0 + u >= 0 && i0 + u < last) {
        if (IB == 4) w[u] = ((const unsigned*)s)[u];
        else w[u >> 1] |= (unsigned)((const unsigned short*)s)[u] << (16 * (u & 1));
        kind |= 1u << u;
      }
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
}
template <int IB> __device__ __forceinline__ void gather_store_q(const uint4 v, cf* d, unsigned kind, float scale)
{
  constexpr int NI = 16 / IB;
  const bool whole = kind == (1u << NI) - 1u && (((uintptr_t)d) & 15u) == 0;
#pragma unroll
  for (int u = 0; u < NI; u += 2) {
    const cf a = q_item<IB>(q_raw<IB>(v, u), scale), b = q_item<IB>(q_raw<IB>(v, u + 1), scale);
    if (whole) *(float4*)(d + u) = make_float4(a.x, a.y, b.x, b.y);
    else { if (kind & (1u << u)) d[u] = a; if (kind & (2u << u)) d[u + 1] = b; }
  }
}
template <int IB> __global__ __launch_bounds__(256) void k_copy_gather_q(cf* __restrict__ dst, const PieceDev* __restrict__ pc_mem, const PieceTab tab, int n,
                                                                          unsigned long long total)
{
  const PieceDev* __restrict__ pc = pc_mem ? pc_mem : tab.p;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += 4 * stride) {
    uint4 v[4];
    cf* d[4];
    unsigned kind[4];
    float sc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { kind[u] = 0; if (i + u * stride < total) gather_item_q<IB>(dst, pc, n, i + u * stride, v[u], d[u], kind[u], sc[u]); }
#pragma unroll
    for (int u = 0; u < 4; u++) if (kind[u]) gather_store_q<IB>(v[u], d[u], kind[u], sc[u]);
  }
}
// C8P_FMT_F32 pieces: 4-byte items moved as they are.  Work item k of a piece = the 16-byte aligned chunk k of its source = its items
// [4k - a, 4k - a + 4), a = the items between the aligned address below the source and the source; clipped to the piece.
__global__ __launch_bounds__(256) void k_copy_gather_w(unsigned* __restrict__ dst, const PieceDev* __restrict__ pc_mem, const PieceTab tab, int n,
                                                       unsigned long long total)
{
  const PieceDev* __restrict__ pc = pc_mem ? pc_mem : tab.p;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += 4 * stride) {
    uint4 v[4];
    unsigned* d[4];
    unsigned kind[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      kind[u] = 0;
      const unsigned long long iu = i + u * stride;
      if (iu >= total) continue;
      int lo = 0, hi = n - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pc[mid].first <= iu) lo = mid; else hi = mid - 1; }
      const PieceDev P = pc[lo];
      const unsigned* src = (const unsigned*)P.src;
      const unsigned a = (unsigned)(((uintptr_t)src >> 2) & 3u);
      const long long i0 = 4 * (long long)(iu - P.first) - (long long)a;
      const unsigned* sp = src + i0;
      d[u] = dst + P.dst_off + i0;
      if (i0 >= 0 && (unsigned long long)i0 + 4 <= P.n) { v[u] = *(const uint4*)sp; kind[u] = 15u; }
      else {
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 4; k++) if (i0 + k >= 0 && i0 + k < (long long)P.n) { w[k] = sp[k]; kind[u] |= 1u << k; }
        v[u] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (!kind[u]) continue;
      if (kind[u] == 15u && (((uintptr_t)d[u]) & 15u) == 0) *(uint4*)d[u] = v[u];
      else {
        const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; k++) if (kind[u] & (1u << k)) d[u][k] = w[k];
      }
    }
  }
}
extern "C" int c8p_copy_gather(c8p_ctx* c, c8p_c32* dst, const c8p_piece* pieces, int32_t n)
{
  if (!c || n < 0 || (n > 0 && (!dst || !pieces))) return C8P_EINVAL;
  int rc;
  if ((rc = copy_stream_of(c, n > 0 && pieces[0].fmt <= C8P_FMT_SC8 ? (int)pieces[0].fmt : C8P_FMT_C32)) != C8P_OK) return rc;
  if (n == 0) return C8P_OK;
  bool any_f32 = false, any_other = false;
  for (int i = 0; i < n; i++) {
    if (!pieces[i].n) continue;
    if (pieces[i].fmt > C8P_FMT_F32) { c->err = "c8p_copy_gather: unknown c8p_piece.fmt"; return C8P_EINVAL; }
    (pieces[i].fmt == C8P_FMT_F32 ? any_f32 : any_other) = true;
    if (!pieces[i].src || ((uintptr_t)pieces[i].src & (pieces[i].fmt == C8P_FMT_SC16 || pieces[i].fmt == C8P_FMT_F32 ? 3u : pieces[i].fmt == C8P_FMT_SC8 ? 1u : 7u))) {
      c->err = "c8p_copy_gather: a piece without source or not aligned to its items (complex64: 8 bytes, sc16 / f32: 4, sc8: 2)";
      return C8P_EINVAL;
    }
    if (pieces[i].trusted) continue;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, pieces[i].src) != hipSuccess || (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeHost
                                                                       && at.type != hipMemoryTypeManaged)) {
      (void)hipGetLastError();
      c->err = "c8p_copy_gather: a piece lies in host memory the device cannot read (pageable): pin it (c8p_host_alloc) or use c8p_copy_async";
      return C8P_ENOTDEV;
    }
  }
  if (any_f32 && any_other) { c->err = "c8p_copy_gather: C8P_FMT_F32 pieces cannot share a call with 8-byte-item pieces (dst_off units differ)"; return C8P_EINVAL; }
  const bool inl = n <= C8P_GATHER_INLINE;
  // device -> host pieces (the caller says so) on the stream of that direction; a long list stays on the copy stream (its table lives there)
  bool to_host = inl;
  for (int i = 0; i < n && to_host; i++) to_host = (pieces[i].rsv & C8P_PIECE_TO_HOST) != 0;
  // host -> device pieces whose caller needs no order among its gathers (C8P_PIECE_UNORDERED on every piece, a short list): every other
  // one on the copy stream's twin
  bool unordered = inl && !to_host;
  for (int i = 0; i < n && unordered; i++) unordered = (pieces[i].rsv & C8P_PIECE_UNORDERED) != 0;
  bool on_b = false;
  if (unordered) {
    std::lock_guard<std::mutex> g(c->copy_mu);
    on_b = c->b_next;
    c->b_next = !c->b_next;
  }
  const hipStream_t gst = to_host ? c->out_stream : on_b ? c->copy_stream_b : c->copy_stream;
  PieceTab itab;
  PieceDev* hp = itab.p;
  if (!inl) {
    const size_t tab = sizeof(PieceDev) * (size_t)n;
    if (tab > c->h_spans_cap) {
      HIPCHK(c, hipStreamSynchronize(c->copy_stream));
      if (c->h_spans) HIPCHK(c, hipHostFree(c->h_spans));
      c->h_spans = nullptr; c->h_spans_cap = 0;
      HIPCHK(c, hipHostMalloc(&c->h_spans, tab * 2 + 4096, hipHostMallocDefault));
      c->h_spans_cap = tab * 2 + 4096;
      if (c->d_spans.p) { HIPCHK(c, hipFree(c->d_spans.p)); c->d_spans.p = nullptr; c->d_spans.cap = 0; }
      HIPCHK(c, hipMalloc(&c->d_spans.p, tab * 2 + 4096));
      c->d_spans.cap = tab * 2 + 4096;
    } else
      // (the previous table may still be read.  A ring of tables guarded by events, which leaves the caller's thread free and the copy
      // stream never idle, was measured in round 4: 64 / 256 / 1024 live streams 4.86 / 6.5 / 7.8-8.3 ms a round against 4.75 / 6.2 /
      // 7.45 with this wait — the gaps it leaves are when the stages' own small transfers get through the link, NOTES/frontend.md)
      HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    hp = (PieceDev*)c->h_spans;
  }
  // one table, grouped by format (complex64, sc16, sc8, f32): one upload (none for a short list), one kernel per format that occurs
  unsigned long long total[4] = {0, 0, 0, 0};
  int cnt[4] = {0, 0, 0, 0}, at = 0, first_of[4] = {0, 0, 0, 0};
  for (int f = 0; f < 4; f++) {
    first_of[f] = at;
    for (int i = 0; i < n; i++) {
      if ((int)pieces[i].fmt != f || !pieces[i].n) continue;
      PieceDev& P = hp[at++];
      cnt[f]++;
      P.src = (const cf*)pieces[i].src; P.dst_off = pieces[i].dst_off; P.n = pieces[i].n; P.first = total[f]; P.scale = pieces[i].scale; P.rsv = 0;
      const uintptr_t sa = (uintptr_t)pieces[i].src;
      total[f] += f == 0 ? (pieces[i].n + 1 + ((sa >> 3) & 1u)) / 2 : f == 2 ? (pieces[i].n + 7 + ((sa >> 1) & 7u)) / 8 : (pieces[i].n + 3 + ((sa >> 2) & 3u)) / 4;
    }
  }
  if (total[0] + total[1] + total[2] + total[3] == 0) return C8P_OK;
  if (!inl) HIPCHK(c, hipMemcpyAsync(c->d_spans.p, c->h_spans, sizeof(PieceDev) * (size_t)at, hipMemcpyHostToDevice, c->copy_stream));      // (small: the DMA queues hold no bulk copies)
  // grid: as many workgroups as keep the link full.  complex64 and sc16 items: 1024 (the link binds; 512 / 256 / 128 cost 2-6 % of a
  // 64-stream round).  sc8: 256 — there the chain's kernels bind and every resident gather wave takes a slot from them (64 / 128
  // streams 2.06-2.12 / 3.69-3.73 ms a round against 2.16-2.19 / 4.00-4.05 with 1024).  C8P_GATHER_WGS overrides all three
  static const unsigned env_wgs = getenv("C8P_GATHER_WGS") ? (unsigned)std::max(1, atoi(getenv("C8P_GATHER_WGS"))) : 0u;
  const PieceDev* dp = inl ? nullptr : (const PieceDev*)c->d_spans.p;
  // (on a masked copy stream: eight 256-thread workgroups per CU = every wave slot of the CUs the stream may use)
  auto grid = [&](unsigned long long t, unsigned wgs) {
    const unsigned g = env_wgs ? env_wgs : (to_host ? 256u : (c->copy_cus ? 8u * (unsigned)c->copy_cus : wgs));
    return dim3((unsigned)std::min<unsigned long long>((t + 1023) / 1024, g));
  };
  // a format's pieces are a contiguous run of the table: the inline table is passed shifted to that run
  auto sub = [&](int f) { PieceTab t; memset(&t, 0, sizeof(t)); if (inl) memcpy(t.p, itab.p + first_of[f], sizeof(PieceDev) * (size_t)cnt[f]); return t; };
  if (total[0]) hipLaunchKernelGGL(k_copy_gather, grid(total[0], C8P_GATHER_WGS), dim3(256), 0, gst, (cf*)dst, inl ? nullptr : dp + first_of[0], sub(0), cnt[0], total[0]);
  if (total[1]) hipLaunchKernelGGL(k_copy_gather_q<4>, grid(total[1], C8P_GATHER_WGS), dim3(256), 0, gst, (cf*)dst, inl ? nullptr : dp + first_of[1], sub(1), cnt[1], total[1]);
  if (total[2]) hipLaunchKernelGGL(k_copy_gather_q<2>, grid(total[2], C8P_GATHER_WGS / 4), dim3(256), 0, gst, (cf*)dst, inl ? nullptr : dp + first_of[2], sub(2), cnt[2], total[2]);
  if (total[3]) hipLaunchKernelGGL(k_copy_gather_w, grid(total[3], C8P_GATHER_WGS), dim3(256), 0, gst, (unsigned*)dst, inl ? nullptr : dp + first_of[3], sub(3), cnt[3], total[3]);
  // the dirty bits AFTER the launches: a mark or fence on another thread that slips in between cannot clear a bit for work that is not
  // on the stream yet (it would never be covered: ADVICE r5); at worst a later mark records once more than it had to
  if (to_host || on_b) {
    std::lock_guard<std::mutex> g(c->copy_mu);
    if (to_host) c->out_dirty = true; else c->b_dirty = true;
  }
  HIPCHK(c, hipGetLastError());
  return C8P_OK;
}
// c8p_fetch_gather's flag piece: a second, one-wave launch behind the gather on the same stream — stream order puts it after the
// gather's last store, the system-scope fence after the stores' visibility to the host
__global__ void k_set_flag(uint32_t* dst, uint32_t value)
{
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(dst, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// Result arenas to host (or device) memory with one launch on the selected submit's own stream, behind its kernels: the pieces' sources
// are the slot's arenas, the destination offsets are taken from address 0 (k_copy_gather_w adds word offsets to its base pointer).
extern "C" int c8p_fetch_gather(c8p_ctx* c, const c8p_fetch_piece* pieces, int32_t n)
{
  if (!c || n < 0 || (n > 0 && !pieces)) return C8P_EINVAL;
  if (n > C8P_GATHER_INLINE) { c->err = "c8p_fetch_gather: more than C8P_GATHER_INLINE pieces"; return C8P_EINVAL; }
  HIPCHK(c, hipSetDevice(c->device));
  PieceTab tab;
  memset(&tab, 0, sizeof(tab));
  unsigned long long total = 0;
  int m = 0;
  uint32_t* flag = nullptr;
  uint32_t flag_value = 0;
  for (int i = 0; i < n; i++) {
    const c8p_fetch_piece& f = pieces[i];
    if (f.which == C8P_BUF_FLAG) {
      if (!f.dst || ((uintptr_t)f.dst & 3u) || flag) { c->err = "c8p_fetch_gather: bad flag piece (one per call, 4-byte aligned)"; return C8P_EINVAL; }
      flag = (uint32_t*)f.dst; flag_value = (uint32_t)f.offset;
      continue;
    }
    if (f.which < 0 || f.which >= C8P_BUF_COUNT || (f.bytes && !f.dst)) { c->err = "c8p_fetch_gather: bad piece"; return C8P_EINVAL; }
    if (f.offset + f.bytes > c->s->used[f.which]) { c->err = "fetch outside the arena"; return C8P_ERANGE; }
    if ((f.offset | f.bytes | (uint64_t)(uintptr_t)f.dst) & 3u) { c->err = "c8p_fetch_gather: offset, bytes and dst must be multiples of 4"; return C8P_EINVAL; }
    if (!f.bytes) continue;
    PieceDev& P = tab.p[m++];
    P.src = (const cf*)((const char*)c->s->ptr[f.which] + f.offset);
    P.dst_off = (unsigned long long)(uintptr_t)f.dst / 4u; P.n = f.bytes / 4u; P.first = total; P.scale = 1.0f; P.rsv = 0;
    total += (P.n + 3 + (((uintptr_t)P.src >> 2) & 3u)) / 4;
  }
  if (total) {
    const dim3 grid((unsigned)std::min<unsigned long long>((total + 1023) / 1024, 256ull));
    hipLaunchKernelGGL(k_copy_gather_w, grid, dim3(256), 0, c->s->stream, (unsigned*)nullptr, (const PieceDev*)nullptr, tab, m, total);
  }
  if (flag) hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(64), 0, c->s->stream, flag, flag_value);
  HIPCHK(c, hipGetLastError());
  return C8P_OK;
}
extern "C" int c8p_copy_fence(c8p_ctx* c)
{
  if (!c) return C8P_EINVAL;
  int rc;
  if ((rc = copy_stream_of(c)) != C8P_OK) return rc;
  std::lock_guard<std::mutex> g(c->copy_mu);
  if (!c->ev_fence) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fence, hipEventDisableTiming));
  if (c->spans_pending) {                      // "every copy enqueued so far" includes the span copies, which run on their own stream
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_spans, 0));
    c->spans_pending = false;
  }
  HIPCHK(c, hipEventRecord(c->ev_fence, c->copy_stream));
  c->fence_pending = true;
  if (!c->ev_fence_b) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fence_b, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->ev_fence_b, c->copy_stream_b));      // (the twin as well, every time)
  c->fence_b_pending = true;
  return C8P_OK;
}
extern "C" int c8p_copy_wait(c8p_ctx* c)
{
  if (!c) return C8P_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->copy_stream) HIPCHK(c, hipStreamSynchronize(c->copy_stream));
  if (c->span_stream) HIPCHK(c, hipStreamSynchronize(c->span_stream));
  if (c->out_stream) HIPCHK(c, hipStreamSynchronize(c->out_stream));
  if (c->copy_stream_b) HIPCHK(c, hipStreamSynchronize(c->copy_stream_b));
  return C8P_OK;
}

// ------------------------------------------------------------------------------------------------ device self-test hooks
// The scalar arithmetic primitives of c8p_dev.h, exposed so that a test can compare them with the CPU's libm / libgcc forms
// on millions of operands ON the device (the kernels only exercise them on whatever the test packets contain).
__global__ void k_selftest(int which, const float* __restrict__ in, float* __restrict__ out, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (which == C8P_SELFTEST_CABS) out[i] = cabs_(mk(in[2 * i], in[2 * i + 1]));
  else if (which == C8P_SELFTEST_CDIV) { const cf q = cdiv(mk(in[4 * i], in[4 * i + 1]), mk(in[4 * i + 2], in[4 * i + 3])); out[2 * i] = q.x; out[2 * i + 1] = q.y; }
  else out[i] = sqrtf(in[3 * i] * in[3 * i] + in[3 * i + 1] * in[3 * i + 1]) / in[3 * i + 2];     // k_presiso's ac
}

extern "C" int c8p_selftest(c8p_ctx* c, int which, const float* in, float* out, uint64_t n)
{
  if (!c || !in || !out || which < 0 || which > C8P_SELFTEST_AC) return C8P_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t fin = which == C8P_SELFTEST_CABS ? 2 : which == C8P_SELFTEST_CDIV ? 4 : 3, fout = which == C8P_SELFTEST_CDIV ? 2 : 1;
  float *d_in = nullptr, *d_out = nullptr;
  if (n == 0) return C8P_OK;
  HIPCHK(c, hipMalloc((void**)&d_in, n * fin * sizeof(float)));
  if (hipMalloc((void**)&d_out, n * fout * sizeof(float)) != hipSuccess) { (void)hipFree(d_in); c->err = "hipMalloc"; return C8P_ENOMEM; }
  hipError_t e = hipMemcpy(d_in, in, n * fin * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->slots[0].stream, which, d_in, d_out, (size_t)n);
    e = hipStreamSynchronize(c->slots[0].stream);
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, n * fout * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d_in); (void)hipFree(d_out);
  if (e != hipSuccess) { c->err = std::string("selftest: ") + hipGetErrorString(e); return C8P_EHIP; }
  return C8P_OK;
}

extern "C" int c8p_kernel_time(c8p_ctx* c, int kernel, double* total_ms, int64_t* launches)
{
  if (!c || kernel < 0 || kernel >= C8P_K_COUNT) return C8P_EINVAL;
  if (total_ms) *total_ms = c->k_ms[kernel];
  if (launches) *launches = c->k_cnt[kernel];
  return C8P_OK;
}
extern "C" int c8p_profile_reset(c8p_ctx* c)
{
  if (!c) return C8P_EINVAL;
  for (int k = 0; k < C8P_K_COUNT; k++) { c->k_ms[k] = 0; c->k_cnt[k] = 0; }
  return C8P_OK;
}

// sizes of the ABI structs as compiled, for the Python binding's self-check
extern "C" int c8p_abi_sizes(int32_t* dst, int cap)
{
  const int32_t v[] = {C8P_ABI_VERSION, (int32_t)sizeof(c8p_pkt_in), (int32_t)sizeof(c8p_pkt_tags), (int32_t)sizeof(c8p_pkt_out),
                       (int32_t)sizeof(c8p_cfg), (int32_t)sizeof(c8p_batch_in), (int32_t)sizeof(c8p_decode_in),
                       (int32_t)sizeof(c8p_sig_in), (int32_t)sizeof(c8p_signal_in), (int32_t)sizeof(c8p_sync_evt),
                       (int32_t)sizeof(c8p_detect_in), (int32_t)sizeof(c8p_det_state), (int32_t)sizeof(c8p_tx_pkt),
                       (int32_t)sizeof(c8p_tx_desc), (int32_t)sizeof(c8p_tx_in), (int32_t)sizeof(c8p_tx_plan), (int32_t)sizeof(c8p_tx_mu),
                       (int32_t)sizeof(c8p_det_stream), (int32_t)sizeof(c8p_detect_multi_in), (int32_t)sizeof(c8p_reserve_in), (int32_t)sizeof(c8p_span),
                       (int32_t)sizeof(c8p_piece), (int32_t)sizeof(c8p_fetch_piece)};
  for (int i = 0; i < 23 && i < cap; i++) dst[i] = v[i];
  return 23;
}
