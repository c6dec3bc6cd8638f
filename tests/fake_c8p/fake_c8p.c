/* TEST-ONLY stand-in for libc8p_rx.so: the same C entry points on top of the CPU oracle, so the host-side block
 * mirror (batching, tag replay, PDU publishing) can be exercised without a GPU.  Never shipped, never used by the
 * product path, bench.py or smoke(). */
#include <stdlib.h>
#include <string.h>
#include "c8p_rx.h"
#include "c8p_oracle.h"

#define SLOTS 3     /* like the engine's default: the results of the last 3 demod / decode / rx submits stay resident (c8p_ticket / c8p_select) */
struct res { int n, ticket; c8p_pkt_tags* tags; c8p_pkt_out* outs; float* llr; uint64_t n_llr; uint8_t* pdu; uint64_t n_pdu; c8p_c32* mu; };
struct c8p_ctx {
  c8p_c32* det_x; uint64_t det_n, det_commit; c8p_sync_evt* det_evt; int det_nevt;   /* c8p_detect: the whole stream so far */
  const c8p_c32* det_raw; uint64_t det_raw_n;                                           /* C8P_BUF_DET_RAW: the caller's items */
  int n_sig; c8p_pkt_in* sig_pkt; c8p_c32* sig_iq; uint64_t n_sig_iq; c8p_c32* sig_chan;
  struct res r[SLOTS];
  int cur, seq;
  struct res* s;      /* selected submit */
};
#define PDU_STRIDE 8192

c8p_ctx* c8p_create(const c8p_cfg* cfg) { (void)cfg; c8po_init(); c8p_ctx* c = (c8p_ctx*)calloc(1, sizeof(c8p_ctx)); c->s = &c->r[0]; return c; }
static void drop_res(struct res* r) { free(r->tags); free(r->outs); free(r->llr); free(r->pdu); free(r->mu); memset(r, 0, sizeof(*r)); }
static void next_slot(c8p_ctx* c) { c->cur = (c->cur + 1) % SLOTS; c->s = &c->r[c->cur]; drop_res(c->s); c->s->ticket = ++c->seq; }
static void drop(c8p_ctx* c)
{
  free(c->det_x); free(c->det_evt); free(c->sig_pkt); free(c->sig_iq); free(c->sig_chan);
  c->det_x = NULL; c->det_evt = NULL; c->sig_pkt = NULL; c->sig_iq = NULL; c->sig_chan = NULL;
  c->det_n = c->det_commit = 0; c->det_nevt = 0; c->n_sig = 0; c->n_sig_iq = 0;
}

int c8p_signal(c8p_ctx* c, const c8p_signal_in* in)      /* same arena layout as the engine: per event cap = n_avail-224+320 items per antenna */
{
  drop(c);
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
  if (in->consume) {      /* the block's scan rule over the per-flag results (c8p_rx.h); the slots stay where they are */
    uint64_t pos = in->scan_from;
    int waiting = 0;
    for (int e = 0; e < in->n_evt; e++) {
      c8p_pkt_in* p = &c->sig_pkt[e];
      const uint64_t idx = in->evts[e].raw_off[0];
      if (waiting) { p->n_items = 0; p->rsv[0] = C8P_SIG_PENDING; continue; }
      if (idx < pos) { p->n_items = 0; p->rsv[0] = C8P_SIG_SKIPPED; continue; }
      if (in->evts[e].n_avail < 224u) { waiting = 1; p->n_items = 0; p->rsv[0] = C8P_SIG_PENDING; continue; }
      if (p->n_items == 0) { if (p->rsv[0] == C8P_SIG_NEED_MORE) waiting = 1; else pos = idx + 80; continue; }
      pos = idx + 224 + (uint64_t)p->lsig_nsamp;
    }
    if (in->scan_next) *in->scan_next = pos;
    if (in->pkt_host) memcpy(in->pkt_host, c->sig_pkt, sizeof(c8p_pkt_in) * (size_t)in->n_evt);
  }
  return C8P_OK;
}
void c8p_destroy(c8p_ctx* c) { if (c) { drop(c); for (int i = 0; i < SLOTS; i++) drop_res(&c->r[i]); free(c); } }
const char* c8p_last_error(const c8p_ctx* c) { (void)c; return "fake backend"; }

int c8p_demod(c8p_ctx* c, const c8p_batch_in* in)
{
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
/* c8p_detect on the oracle: keeps every item it was ever given, re-runs the one-shot oracle over the whole stream and
 * reports the events between the old and the new commit point (new commit = 400 items before the end: every event
 * before it has its 240-item search window inside the stream, so later items cannot change it). */
int c8p_detect(c8p_ctx* c, const c8p_detect_in* in)
{
  if (!in->state && in->n_hist == 0) { free(c->det_x); c->det_x = NULL; c->det_n = c->det_commit = 0; }   /* a segment from the constructor state */
  c->det_raw = in->raw;
  c->det_raw_n = in->n;
  const uint64_t n_new = in->n - in->n_hist;
  c->det_x = (c8p_c32*)realloc(c->det_x, (size_t)(c->det_commit + n_new + 1) * sizeof(c8p_c32));
  memcpy(c->det_x + c->det_commit, in->raw + in->n_hist, (size_t)n_new * sizeof(c8p_c32));
  c->det_n = c->det_commit + n_new;
  const long N = (long)c->det_n;
  float* ac = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  c8p_c32* cj = (c8p_c32*)malloc(sizeof(c8p_c32) * (size_t)(N + 1));
  uint8_t* tr = (uint8_t*)malloc((size_t)N + 1);
  c8p_sync_evt* ev = (c8p_sync_evt*)malloc(sizeof(c8p_sync_evt) * (size_t)(N / 100 + 16));
  c8po_presiso(c->det_x, N, ac, cj);
  c8po_trigger(ac, N, tr);
  const long nev = c8po_sync(tr, cj, c->det_x, N, ev, N / 100 + 16, NULL);
  const uint64_t old = c->det_commit;
  const uint64_t neu = c->det_n > 400 ? (c->det_n - 400 > old ? c->det_n - 400 : old) : old;
  free(c->det_evt);
  c->det_evt = (c8p_sync_evt*)calloc((size_t)nev + 1, sizeof(c8p_sync_evt));
  c->det_nevt = 0;
  const uint64_t origin = old - in->n_hist;            /* stream index of raw[0] */
  for (long k = 0; k < nev; k++)
    if (ev[k].index >= old && ev[k].index < neu) { c->det_evt[c->det_nevt] = ev[k]; c->det_evt[c->det_nevt].index -= origin; c->det_nevt++; }
  c->det_commit = neu;
  if (in->commit) *in->commit = neu - origin;
  if (in->state_out) memset(in->state_out, 0, sizeof(*in->state_out));
  free(ac); free(cj); free(tr); free(ev);
  return C8P_OK;
}
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
int c8p_fetch_async(c8p_ctx* c, int which, void* dst, uint64_t offset, uint64_t bytes) { return c8p_fetch(c, which, dst, offset, bytes); }
void* c8p_host_alloc(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void c8p_host_free(void* p) { free(p); }
