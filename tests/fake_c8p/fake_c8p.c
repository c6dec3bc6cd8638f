/* TEST-ONLY stand-in for libc8p_rx.so: the same C entry points on top of the CPU oracle, so the host-side block
 * mirror (batching, tag replay, PDU publishing) can be exercised without a GPU.  Never shipped, never used by the
 * product path, bench.py or smoke(). */
#define _POSIX_C_SOURCE 200809L   /* nanosleep (a test knob) under -std=c11 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "c8p_rx.h"
#include "c8p_oracle.h"

#define SLOTS 3     /* like the engine's default: the results of the last 3 demod / decode / rx submits stay resident (c8p_ticket / c8p_select);
                       C8P_SLOTS = 1 or 2 in the environment (read at create, like the engine) keeps fewer */
struct res { int n, ticket; c8p_pkt_tags* tags; c8p_pkt_out* outs; float* llr; uint64_t n_llr; uint8_t* pdu; uint64_t n_pdu; c8p_c32* mu; };
struct pend;
struct detst { c8p_c32* x; uint64_t n, commit; };      /* one stream of c8p_detect: every item it was ever given */
struct c8p_ctx {
  struct detst det; c8p_sync_evt* det_evt; int det_nevt;   /* c8p_detect: the whole stream so far */
  struct detst* dets; int n_dets;                           /* c8p_detect_multi: the same per stream */
  int64_t submits;
  const c8p_c32* det_raw; uint64_t det_raw_n;                                           /* C8P_BUF_DET_RAW: the caller's items */
  int n_sig; c8p_pkt_in* sig_pkt; c8p_c32* sig_iq; uint64_t n_sig_iq; c8p_c32* sig_chan;
  struct res r[SLOTS];
  uint64_t asked_mark;                                        /* c8p_copy_mark_done: the mark last asked about */
  struct pend *pend_head, *pend_tail;                       /* c8p_copy_gather calls not yet executed (see run_pending) */
  int cur, seq, slots;
  struct res* s;      /* selected submit */
};
#define PDU_STRIDE 8192

c8p_ctx* c8p_create(const c8p_cfg* cfg)
{
  (void)cfg;
  c8po_init();
  c8p_ctx* c = (c8p_ctx*)calloc(1, sizeof(c8p_ctx));
  const char* e = getenv("C8P_SLOTS");
  c->slots = (e && atoi(e) >= 1 && atoi(e) <= SLOTS) ? atoi(e) : SLOTS;
  c->s = &c->r[0];
  return c;
}
static void drop_res(struct res* r) { free(r->tags); free(r->outs); free(r->llr); free(r->pdu); free(r->mu); memset(r, 0, sizeof(*r)); }
static void next_slot(c8p_ctx* c) { c->submits++; c->cur = (c->cur + 1) % c->slots; c->s = &c->r[c->cur]; drop_res(c->s); c->s->ticket = ++c->seq; }
static void drop(c8p_ctx* c)
{
  free(c->det.x); free(c->det_evt); free(c->sig_pkt); free(c->sig_iq); free(c->sig_chan);
  for (int i = 0; i < c->n_dets; i++) free(c->dets[i].x);
  free(c->dets); c->dets = NULL; c->n_dets = 0;
  c->det.x = NULL; c->det_evt = NULL; c->sig_pkt = NULL; c->sig_iq = NULL; c->sig_chan = NULL;
  c->det.n = c->det.commit = 0; c->det_nevt = 0; c->n_sig = 0; c->n_sig_iq = 0;
}
static void drop_sig(c8p_ctx* c)
{
  free(c->sig_pkt); free(c->sig_iq); free(c->sig_chan);
  c->sig_pkt = NULL; c->sig_iq = NULL; c->sig_chan = NULL; c->n_sig = 0; c->n_sig_iq = 0;
}

int c8p_signal(c8p_ctx* c, const c8p_signal_in* in)      /* same arena layout as the engine: per event cap = n_avail-224+320 items per antenna */
{
  drop_sig(c);
  c->submits++;
  c->n_sig = in->n_evt;
  c->sig_pkt = (c8p_pkt_in*)calloc((size_t)in->n_evt + 1, sizeof(c8p_pkt_in));
  c->sig_chan = (c8p_c32*)calloc((size_t)in->n_evt * 64 + 1, sizeof(c8p_c32));
  uint64_t tot = 0;
  for (int e = 0; e < in->n_evt; e++) tot += ((in->evts[e].n_avail >= 224u ? in->evts[e].n_avail - 224u : 0u) + 320u) * (uint64_t)in->evts[e].n_ant;
  c->sig_iq = (c8p_c32*)calloc(tot + 1, sizeof(c8p_c32));
  c->n_sig_iq = tot;
  uint64_t off = 0;
  for (int e = 0; e < in->n_evt; e++) {
    const c8p_sig_in* ev = &in->evts[e];
    const uint64_t cap = (ev->n_avail >= 224u ? ev->n_avail - 224u : 0u) + 320u;
    const int ok = c8po_signal_event(ev, in->raw, &c->sig_pkt[e], c->sig_chan + 64 * e, c->sig_iq + off, c->sig_iq + off + cap);
    c->sig_pkt[e].iq_off[0] = off; c->sig_pkt[e].iq_off[1] = ev->n_ant == 2 ? off + cap : 0; c->sig_pkt[e].chan_off = 64u * (uint32_t)e;
    if (!ok) c->sig_pkt[e].n_items = 0;
    off += cap * ev->n_ant;
  }
  if (in->consume) {      /* the block's scan rule over the per-flag results (c8p_rx.h), per stream; the slots stay where they are */
    const int KS = in->n_streams > 1 ? in->n_streams : 1;
    uint64_t* posv = (uint64_t*)calloc((size_t)KS, sizeof(uint64_t));
    char* waitv = (char*)calloc((size_t)KS, 1);
    if (KS == 1) posv[0] = in->scan_from;
    else if (in->scan_from_v) memcpy(posv, in->scan_from_v, sizeof(uint64_t) * (size_t)KS);
    uint32_t last = 0;
    for (int e = 0; e < in->n_evt; e++) {
      c8p_pkt_in* p = &c->sig_pkt[e];
      const uint32_t si = KS > 1 ? in->evts[e].stream : 0u;
      if (si >= (uint32_t)KS || si < last) { free(posv); free(waitv); return C8P_EINVAL; }
      last = si;
      const uint64_t idx = in->evts[e].raw_off[0];
      if (waitv[si]) { p->n_items = 0; p->rsv[0] = C8P_SIG_PENDING; continue; }
      if (idx < posv[si]) { p->n_items = 0; p->rsv[0] = C8P_SIG_SKIPPED; continue; }
      if (in->evts[e].n_avail < 224u) { waitv[si] = 1; p->n_items = 0; p->rsv[0] = C8P_SIG_PENDING; continue; }
      if (p->n_items == 0) { if (p->rsv[0] == C8P_SIG_NEED_MORE) waitv[si] = 1; else posv[si] = idx + 80; continue; }
      posv[si] = idx + 224 + (uint64_t)p->lsig_nsamp;
    }
    if (in->scan_next) memcpy(in->scan_next, posv, sizeof(uint64_t) * (size_t)KS);
    if (in->pkt_host) memcpy(in->pkt_host, c->sig_pkt, sizeof(c8p_pkt_in) * (size_t)in->n_evt);
    free(posv); free(waitv);
  }
  return C8P_OK;
}
static void run_pending(c8p_ctx* c);
void c8p_destroy(c8p_ctx* c) { if (c) { run_pending(c); drop(c); for (int i = 0; i < SLOTS; i++) drop_res(&c->r[i]); free(c); } }
const char* c8p_last_error(const c8p_ctx* c) { (void)c; return "fake backend"; }

int c8p_demod(c8p_ctx* c, const c8p_batch_in* in)
{
  { const char* e = getenv("C8P_FAKE_SUBMIT_SLEEP_US"); if (e && atoi(e) > 0) { struct timespec ts = {atoi(e) / 1000000, (long)(atoi(e) % 1000000) * 1000L}; nanosleep(&ts, NULL); } }     /* (a test: a submit that takes long, under the front's lock) */
  next_slot(c);
  c->s->n = in->n_pkt;
  c->s->tags = (c8p_pkt_tags*)calloc((size_t)in->n_pkt + 1, sizeof(c8p_pkt_tags));
  c->s->mu = (c8p_c32*)calloc((size_t)in->n_pkt * 128 + 1, sizeof(c8p_c32));
  uint64_t cap = 0;
  for (int p = 0; p < in->n_pkt; p++) cap += (uint64_t)(in->pkts[p].n_items / 72 + 1) * 832 + 1024;
  c->s->llr = (float*)calloc(cap + 1, sizeof(float));
  uint64_t off = 0;
  for (int p = 0; p < in->n_pkt; p++) {
    c8po_demod(&in->pkts[p], in->iq, in->chan, &c->s->tags[p], c->s->llr + off, c->s->mu + 128 * p);
    c->s->tags[p].llr_off = off;
    off += (uint64_t)(in->pkts[p].n_items / 72 + 1) * 832 + 1024;
  }
  c->s->n_llr = cap;
  return C8P_OK;
}

static int decode_into(c8p_ctx* c, const c8p_decode_in* in)
{
  free(c->s->outs); free(c->s->pdu);
  c->s->outs = (c8p_pkt_out*)calloc((size_t)in->n_pkt + 1, sizeof(c8p_pkt_out));
  c->s->pdu = (uint8_t*)calloc((size_t)in->n_pkt * PDU_STRIDE + 1, 1);
  c->s->n_pdu = (uint64_t)in->n_pkt * PDU_STRIDE;
  c->s->n = in->n_pkt;
  for (int p = 0; p < in->n_pkt; p++) {
    c8po_decode(&in->tags[p], in->llr + in->tags[p].llr_off, in->mu2x1chan ? in->mu2x1chan + 128 * p : NULL, NULL, NULL,
                c->s->pdu + (size_t)p * PDU_STRIDE, PDU_STRIDE, &c->s->outs[p]);
    c->s->outs[p].pdu_off = (uint64_t)p * PDU_STRIDE;
  }
  return C8P_OK;
}
int c8p_decode(c8p_ctx* c, const c8p_decode_in* in) { next_slot(c); return decode_into(c, in); }
/* c8p_detect on the oracle: keeps every item a stream was ever given, re-runs the one-shot oracle over the whole stream and
 * reports the events between the old and the new commit point (new commit = 400 items before the end: every event
 * before it has its 240-item search window inside the stream, so later items cannot change it).  Appends to *out. */
static void detect_stream(struct detst* d, const c8p_c32* raw, uint64_t n, uint64_t n_hist, int fresh, uint64_t* commit_out,
                          c8p_sync_evt** out, int* n_out)
{
  if (fresh) { free(d->x); d->x = NULL; d->n = d->commit = 0; }   /* a segment from the constructor state */
  const uint64_t n_new = n - n_hist;
  d->x = (c8p_c32*)realloc(d->x, (size_t)(d->commit + n_new + 1) * sizeof(c8p_c32));
  memcpy(d->x + d->commit, raw + n_hist, (size_t)n_new * sizeof(c8p_c32));
  d->n = d->commit + n_new;
  const long N = (long)d->n;
  float* ac = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  c8p_c32* cj = (c8p_c32*)malloc(sizeof(c8p_c32) * (size_t)(N + 1));
  uint8_t* tr = (uint8_t*)malloc((size_t)N + 1);
  c8p_sync_evt* ev = (c8p_sync_evt*)malloc(sizeof(c8p_sync_evt) * (size_t)(N / 100 + 16));
  c8po_presiso(d->x, N, ac, cj);
  c8po_trigger(ac, N, tr);
  const long nev = c8po_sync(tr, cj, d->x, N, ev, N / 100 + 16, NULL);
  const uint64_t old = d->commit;
  const uint64_t neu = d->n > 400 ? (d->n - 400 > old ? d->n - 400 : old) : old;
  *out = (c8p_sync_evt*)realloc(*out, (size_t)(*n_out + nev + 1) * sizeof(c8p_sync_evt));
  const uint64_t origin = old - n_hist;            /* stream index of raw[0] */
  for (long k = 0; k < nev; k++)
    if (ev[k].index >= old && ev[k].index < neu) { (*out)[*n_out] = ev[k]; (*out)[*n_out].index -= origin; (*n_out)++; }
  d->commit = neu;
  if (commit_out) *commit_out = neu - origin;
  free(ac); free(cj); free(tr); free(ev);
}
int c8p_detect(c8p_ctx* c, const c8p_detect_in* in)
{
  c->submits++;
  c->det_raw = in->raw;
  c->det_raw_n = in->n;
  free(c->det_evt); c->det_evt = NULL; c->det_nevt = 0;
  detect_stream(&c->det, in->raw, in->n, in->n_hist, !in->state && in->n_hist == 0, in->commit, &c->det_evt, &c->det_nevt);
  if (in->state_out) memset(in->state_out, 0, sizeof(*in->state_out));
  if (in->evt_host && c->det_nevt) memcpy(in->evt_host, c->det_evt, sizeof(c8p_sync_evt) * (size_t)((uint64_t)c->det_nevt < in->evt_cap ? (uint64_t)c->det_nevt : in->evt_cap));
  return C8P_OK;
}
int c8p_detect_multi(c8p_ctx* c, const c8p_detect_multi_in* in)
{
  /* C8P_FAKE_FAIL_DETECT_MULTI_AT = k: the k-th call of a context fails (tests of the callers' error paths) */
  { const char* e = getenv("C8P_FAKE_FAIL_DETECT_MULTI_AT"); static int calls = 0; if (e && ++calls == atoi(e)) { calls = 0; return C8P_EHIP; } }
  c->submits++;
  c->det_raw = in->raw;
  c->det_raw_n = in->n_raw;
  free(c->det_evt); c->det_evt = NULL; c->det_nevt = 0;
  if (in->n_streams > c->n_dets) {
    c->dets = (struct detst*)realloc(c->dets, sizeof(struct detst) * (size_t)in->n_streams);
    memset(c->dets + c->n_dets, 0, sizeof(struct detst) * (size_t)(in->n_streams - c->n_dets));
    c->n_dets = in->n_streams;
  }
  uint64_t toff = 0;
  for (int s = 0; s < in->n_streams; s++) {
    c8p_det_stream* k = &in->streams[s];
    if (k->off + k->n > in->n_raw || k->n_hist > k->n) return C8P_ERANGE;
    const int before = c->det_nevt;
    k->trig_off = toff; toff += (k->n - k->n_hist + 7) & ~(uint64_t)7;
    if (k->n == k->n_hist) { k->commit = k->n_hist; k->n_evt = 0; continue; }
    detect_stream(&c->dets[s], in->raw + k->off, k->n, k->n_hist, !k->has_state && k->n_hist == 0, &k->commit, &c->det_evt, &c->det_nevt);
    k->n_evt = (uint32_t)(c->det_nevt - before);
    memset(&k->state, 0, sizeof(k->state)); k->has_state = 1;
  }
  if (in->evt_host && c->det_nevt) memcpy(in->evt_host, c->det_evt, sizeof(c8p_sync_evt) * (size_t)((uint64_t)c->det_nevt < in->evt_cap ? (uint64_t)c->det_nevt : in->evt_cap));
  return C8P_OK;
}
int c8p_reserve(c8p_ctx* c, const c8p_reserve_in* in) { (void)c; (void)in; return C8P_OK; }
int c8p_option(c8p_ctx* c, int which, int64_t value) { return (c && which == C8P_OPT_COPY_CUS && value >= 0 && value <= 256) ? C8P_OK : C8P_EINVAL; }
int64_t c8p_counter(c8p_ctx* c, int which)
{
  switch (which) {
    case C8P_CNT_SLOTS: return c->slots;
    case C8P_CNT_SUBMITS: return c->submits;
    case C8P_CNT_DET_TILE: return 960;
    case C8P_CNT_ARENA_GROWTHS: case C8P_CNT_ARENA_BYTES: case C8P_CNT_HOST_BYTES: case C8P_CNT_DET_GIVEUPS: case C8P_CNT_DET_TICKET_SCANS: return 0;
    default: return C8P_EINVAL;
  }
}
/* caller arenas: "device" memory is the heap here, the copies are immediate */
void* c8p_dev_alloc(c8p_ctx* c, uint64_t bytes) { (void)c; return calloc(bytes ? bytes : 1, 1); }
static void run_pending(c8p_ctx* c);
void c8p_dev_free(c8p_ctx* c, void* p) { if (c) run_pending(c); free(p); }
/* The copy stream is ASYNCHRONOUS here too, so that a caller that releases a gather's source too early (a block mirror consuming
 * scheduler items whose gather it has not waited for) or forgets c8p_copy_fence before a submit FAILS on this backend as it would race
 * on the device: a c8p_copy_gather only records its pieces; they are executed — in order — by whatever the header says completes or
 * orders them: c8p_copy_wait / _wait_mark, c8p_copy_fence, a later copy on the same stream (c8p_copy_async, c8p_copy_spans, a long
 * c8p_copy_gather, which waits for the stream), c8p_dev_free.  Submits do NOT run them. */
struct pend { c8p_c32* dst; c8p_piece* pc; int n, unordered; struct pend *next, *rnext; };
static int run_gather(c8p_c32* dst, const c8p_piece* pc, int32_t n)
{
  for (int i = 0; i < n; i++) {
    if (pc[i].fmt == C8P_FMT_SC16) {          /* ((float)i * scale, (float)q * scale): one float32 multiply each, as the header says */
      const int16_t* s = (const int16_t*)pc[i].src;
      for (uint64_t k = 0; k < pc[i].n; k++) {
        c8p_c32* d = dst + pc[i].dst_off + k;
        d->re = (float)s[2 * k] * pc[i].scale; d->im = (float)s[2 * k + 1] * pc[i].scale;
      }
    } else if (pc[i].fmt == C8P_FMT_SC8) {
      const int8_t* s = (const int8_t*)pc[i].src;
      for (uint64_t k = 0; k < pc[i].n; k++) {
        c8p_c32* d = dst + pc[i].dst_off + k;
        d->re = (float)s[2 * k] * pc[i].scale; d->im = (float)s[2 * k + 1] * pc[i].scale;
      }
    } else if (pc[i].fmt == C8P_FMT_C32) memmove(dst + pc[i].dst_off, pc[i].src, sizeof(c8p_c32) * (size_t)pc[i].n);
    else if (pc[i].fmt == C8P_FMT_F32) memmove((float*)dst + pc[i].dst_off, pc[i].src, sizeof(float) * (size_t)pc[i].n);
    else return C8P_EINVAL;
  }
  return C8P_OK;
}
static void run_pending(c8p_ctx* c)
{
  struct pend* p = c->pend_head;
  c->pend_head = c->pend_tail = NULL;
  /* C8P_PIECE_UNORDERED calls first, newest first (they promised not to depend on any order), then the others in the order they came */
  struct pend* rev = NULL;
  for (struct pend* q = p; q; q = q->next)
    if (q->n > 0 && q->n <= C8P_GATHER_INLINE && (q->pc[0].rsv & C8P_PIECE_UNORDERED)) { q->unordered = 1; q->rnext = rev; rev = q; }
  for (struct pend* q = rev; q; q = q->rnext) run_gather(q->dst, q->pc, q->n);
  while (p) { struct pend* nx = p->next; if (!p->unordered) run_gather(p->dst, p->pc, p->n); free(p->pc); free(p); p = nx; }
}
/* (round 6: c8p_copy_async is ordered behind the C8P_PIECE_UNORDERED gathers as well — include/c8p_rx.h) */
int c8p_copy_async(c8p_ctx* c, void* dst, const void* src, uint64_t bytes) { run_pending(c); memmove(dst, src, bytes); return C8P_OK; }
int c8p_copy_spans(c8p_ctx* c, c8p_c32* dst, const c8p_c32* src, const c8p_span* sp, int32_t n)
{
  run_pending(c);
  for (int i = 0; i < n; i++) memmove(dst + sp[i].dst_off, src + sp[i].src_off, sizeof(c8p_c32) * (size_t)sp[i].n);
  return C8P_OK;
}
int c8p_copy_gather(c8p_ctx* c, c8p_c32* dst, const c8p_piece* pc, int32_t n)
{
  int f32 = 0, other = 0;
  for (int i = 0; i < n; i++) {
    if (pc[i].fmt > C8P_FMT_F32) return C8P_EINVAL;
    if (pc[i].n) { if (pc[i].fmt == C8P_FMT_F32) f32 = 1; else other = 1; }
  }
  if (f32 && other) return C8P_EINVAL;
  if (n > C8P_GATHER_INLINE) run_pending(c);        /* the long-list path waits for the copy stream before it reuses its table */
  struct pend* p = (struct pend*)calloc(1, sizeof(struct pend));
  p->dst = dst; p->n = n; p->pc = (c8p_piece*)malloc(sizeof(c8p_piece) * (size_t)(n > 0 ? n : 1));
  if (n > 0) memcpy(p->pc, pc, sizeof(c8p_piece) * (size_t)n);
  if (c->pend_tail) c->pend_tail->next = p; else c->pend_head = p;
  c->pend_tail = p;
  return C8P_OK;
}
int c8p_copy_wait(c8p_ctx* c) { run_pending(c); return C8P_OK; }
uint64_t c8p_copy_mark(c8p_ctx* c) { return (uint64_t)++c->submits; }
int c8p_copy_wait_mark(c8p_ctx* c, uint64_t mark) { (void)mark; run_pending(c); return C8P_OK; }
/* the first question about a mark finds its copies still on their way (nothing has been moved), the second one finds them done */
int c8p_copy_mark_done(c8p_ctx* c, uint64_t mark)
{
  if (!mark || mark > (uint64_t)c->submits) return C8P_EINVAL;
  if (!c->pend_head) return 1;
  if (c->asked_mark != mark) { c->asked_mark = mark; return 0; }
  run_pending(c);
  return 1;
}
int c8p_copy_fence(c8p_ctx* c) { run_pending(c); return C8P_OK; }
int c8p_sync(c8p_ctx* c) { (void)c; return C8P_OK; }

const void* c8p_buffer(c8p_ctx* c, int which, uint64_t* bytes)
{
  const void* p = NULL; uint64_t b = 0;
  switch (which) {
    case C8P_BUF_TAGS: p = c->s->tags; b = (uint64_t)c->s->n * sizeof(c8p_pkt_tags); break;
    case C8P_BUF_LLR: p = c->s->llr; b = c->s->n_llr * sizeof(float); break;
    case C8P_BUF_OUT: p = c->s->outs; b = (uint64_t)c->s->n * sizeof(c8p_pkt_out); break;
    case C8P_BUF_PDU: p = c->s->pdu; b = c->s->n_pdu; break;
    case C8P_BUF_MUCHAN: p = c->s->mu; b = (uint64_t)c->s->n * 128 * sizeof(c8p_c32); break;
    case C8P_BUF_DET_EVT: p = c->det_evt; b = (uint64_t)c->det_nevt * sizeof(c8p_sync_evt); break;
    case C8P_BUF_DET_RAW: p = c->det_raw; b = c->det_raw_n * sizeof(c8p_c32); break;
    case C8P_BUF_SIG_PKT: p = c->sig_pkt; b = (uint64_t)c->n_sig * sizeof(c8p_pkt_in); break;
    case C8P_BUF_SIG_IQ: p = c->sig_iq; b = c->n_sig_iq * sizeof(c8p_c32); break;
    case C8P_BUF_SIG_CHAN: p = c->sig_chan; b = (uint64_t)c->n_sig * 64 * sizeof(c8p_c32); break;
    default: break;
  }
  if (!p) b = 0;
  if (bytes) *bytes = b;
  return p;
}
int c8p_fetch(c8p_ctx* c, int which, void* dst, uint64_t offset, uint64_t bytes)
{
  uint64_t b; const void* p = c8p_buffer(c, which, &b);
  if (offset + bytes > b) return C8P_ERANGE;
  memcpy(dst, (const char*)p + offset, bytes);
  return C8P_OK;
}

/* the asynchronous / pipelining entry points: everything is synchronous here, one resident submit */
int c8p_rx(c8p_ctx* c, const c8p_batch_in* in)
{
  int rc = c8p_demod(c, in);
  if (rc != C8P_OK) return rc;
  c8p_decode_in d;
  memset(&d, 0, sizeof(d));
  d.n_pkt = in->n_pkt; d.mem = C8P_MEM_HOST; d.tags = c->s->tags; d.llr = c->s->llr; d.n_llr = c->s->n_llr; d.mu2x1chan = c->s->mu;
  return decode_into(c, &d);
}
int c8p_ticket(c8p_ctx* c) { return c->seq; }
int c8p_select(c8p_ctx* c, int ticket)
{
  for (int i = 0; i < SLOTS; i++) if (c->r[i].ticket == ticket && ticket != 0) { c->s = &c->r[i]; return C8P_OK; }
  return C8P_ERANGE;
}
int c8p_wait(c8p_ctx* c, int ticket) { (void)c; (void)ticket; return C8P_OK; }
int c8p_done(c8p_ctx* c, int ticket) { (void)c; (void)ticket; return 1; }
int c8p_fetch_async(c8p_ctx* c, int which, void* dst, uint64_t offset, uint64_t bytes) { return c8p_fetch(c, which, dst, offset, bytes); }
int c8p_fetch_gather(c8p_ctx* c, const c8p_fetch_piece* pc, int32_t n)
{
  if (n > C8P_GATHER_INLINE) return C8P_EINVAL;
  uint32_t* flag = NULL; uint32_t fv = 0;
  for (int i = 0; i < n; i++) {
    if (pc[i].which == C8P_BUF_FLAG) { flag = (uint32_t*)pc[i].dst; fv = (uint32_t)pc[i].offset; continue; }
    if ((pc[i].offset | pc[i].bytes | (uint64_t)(uintptr_t)pc[i].dst) & 3u) return C8P_EINVAL;
    const int rc = pc[i].bytes ? c8p_fetch(c, pc[i].which, pc[i].dst, pc[i].offset, pc[i].bytes) : C8P_OK;
    if (rc != C8P_OK) return rc;
  }
  if (flag) __atomic_store_n(flag, fv, __ATOMIC_RELEASE);
  return C8P_OK;
}
void* c8p_host_alloc(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void c8p_host_free(void* p) { free(p); }
