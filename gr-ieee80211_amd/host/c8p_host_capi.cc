// c8p_host_capi.cc — C entry points around the block mirror, used by the Python tests and by INTEGRATION examples:
// an in-memory block_io (tagged item streams in, items/tags/messages out) plus a tiny run loop that calls
// general_work() the way the GNU Radio scheduler does (bounded chunks, zero-progress calls allowed).
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "c8p_blocks.h"

using namespace C8P_MIRROR_QUAL;

namespace {

struct harness : public block_io {
  std::shared_ptr<c8p_block> blk;
  // the block goes first: its threads (a front's latency timer, rx_chain's pipeline) call wake() / message_port_pub() on this object
  // until they are joined, and `blk` is the first member — it would be destroyed LAST, after the mutexes and the condition variable
  ~harness() override { if (blk) release_buffers(); blk.reset(); }
  int nports = 1, nout = 1;
  size_t item_size = sizeof(gr_complex);        // items of the stream ports
  size_t port0_size = 0;                        // != 0: port 0 carries items of another size (signal: sync bytes)
  size_t out_size = sizeof(float);
  size_t isz(int q) const { return (q == 0 && port0_size) ? port0_size : item_size; }
  // One input buffer per port; the unread part starts at `pos` bytes.  When the block's io_signature names its buffer type for the port
  // (c8p_block::wants_pinned_input) the memory comes from the block's allocator, as GNU Radio 3.10 allocates an edge's buffer through the
  // gr::buffer_type of the block behind it: pinned host memory, which is never moved or compacted while items of it may be on their way
  // to the device.
  struct ibuf {
    uint8_t* p = nullptr; size_t size = 0, cap = 0, pos = 0; bool pinned = false;
  };
  std::vector<ibuf> in;
  bool pinned_out = false;
  bool poison = false;                          // consumed input items are overwritten at once (a test: nothing may still be reading them)
  // != 0: every `relocate`-th call of c8ph_run finds the unread items of its pinned input buffers MOVED to other memory, as a single-mapped
  // GNU Radio buffer moves them when a reader is blocked at its end: buffer_will_move first, then the copy, then the old memory is overwritten
  int relocate = 0;
  // != 0: every `move_out`-th call of c8ph_run finds its OUTPUT position moved the way GNU Radio's single-mapped buffers move it from
  // their blocked callbacks — without draining output_ahead first: the move is bracketed by c8p::buffer_move_scope (what
  // gnuradio/lib/c8p_pinned_buffer.h does), everything the scope has not saved is overwritten with NaNs
  int move_out = 0;
  long moves_done = 0;
  std::vector<std::pair<void*, bool>> graveyard; // ... the old buffers (kept to the end: a front that ignores the announcement reads 0xff, not freed memory)
  void relocate_inputs()
  {
    for (ibuf& b : in) {
      if (!b.pinned || !b.p || b.size == b.pos) continue;
      buffer_will_move(b.p, b.cap);
      uint8_t* np = static_cast<uint8_t*>(balloc(b.cap, true));
      memcpy(np + b.pos, b.p + b.pos, b.size - b.pos);
      memset(b.p, 0xff, b.cap);
      graveyard.emplace_back(b.p, true);
      b.p = np;
    }
  }
  void* balloc(size_t bytes, bool pinned) { void* q = pinned ? blk->buffer_alloc(bytes) : malloc(bytes); if (!q) throw std::runtime_error("harness: no memory for a buffer"); return q; }
  void bfree(void* q, bool pinned) { if (!q) return; if (pinned) blk->buffer_free(q); else free(q); }
  void setup_buffers()
  {
    in.resize(nports);
    for (int q = 0; q < nports; q++) in[q].pinned = blk->wants_pinned_input((unsigned)q);
    pinned_out = blk->wants_pinned_output(0);
  }
  // the segments of an edge this harness allocated for c8ph_run_chain: kept from one run to the next (pinning 96 MB takes tens of ms)
  std::vector<float*> edge; size_t edge_floats = 0; bool edge_pinned = false;
  float** edge_segments(size_t floats, bool pinned)
  {
    if (edge.size() == 3 && edge_floats >= floats && edge_pinned == pinned) return edge.data();
    for (float* q : edge) bfree(q, edge_pinned);
    edge.clear();
    for (int i = 0; i < 3; i++) edge.push_back(static_cast<float*>(balloc(floats * sizeof(float), pinned)));
    edge_floats = floats; edge_pinned = pinned;
    return edge.data();
  }
  void release_buffers()
  {
    for (ibuf& b : in) { bfree(b.p, b.pinned); b.p = nullptr; }
    for (auto& g : graveyard) bfree(g.first, g.second);
    graveyard.clear();
    for (float* q : edge) bfree(q, edge_pinned);
    edge.clear();
  }
  void append(int q, const uint8_t* src, size_t bytes)
  {
    ibuf& b = in[q];
    if (b.size + bytes > b.cap) {
      const size_t nc = std::max<size_t>(std::max(b.size + bytes, 2 * b.cap), 1u << 16);
      uint8_t* np = static_cast<uint8_t*>(balloc(nc, b.pinned));
      if (b.size > b.pos) memcpy(np + b.pos, b.p + b.pos, b.size - b.pos);
      bfree(b.p, b.pinned);
      b.p = np; b.cap = nc;
    }
    memcpy(b.p + b.size, src, bytes);
    b.size += bytes;
  }
  size_t avail(int q) const { return (in[q].size - in[q].pos) / isz(q); }
  const uint8_t* rd(int q) const { return in[q].p + in[q].pos; }
  void advance(int q, size_t items)
  {
    ibuf& b = in[q];
    if (poison && items) memset(b.p + b.pos, 0xff, items * isz(q));
    b.pos += items * isz(q);
    // consumed items are dropped in bulk, rarely: a scheduler's circular buffer costs nothing per item, and a compaction here is a
    // memmove of everything still unread (tens of megabytes in a throughput run: 2-3 us per packet of the chain when done at 4 MB)
    if (b.pos == b.size) { b.size = 0; b.pos = 0; }
    else if (!b.pinned && b.pos > (1u << 28) && b.pos * 2 > b.size) { memmove(b.p, b.p + b.pos, b.size - b.pos); b.size -= b.pos; b.pos = 0; }
  }
  bool tags_sorted = true;
  std::deque<tag_t> in_tags;                    // port 0, absolute offsets; tags of consumed items are dropped (as a scheduler prunes them)
  void prune_tags() { if (tags_sorted) while (!in_tags.empty() && in_tags.front().offset < n_read) in_tags.pop_front(); }
  uint64_t n_read = 0, n_written = 0;
  int consumed_this_call = 0;
  std::vector<float> out_items;                 // output port 0 (as floats: a complex item is two of them)
  std::vector<float> out_items1;                // output port 1 (signal2)
  std::vector<tag_t> out_tags;
  std::vector<pdu_t> msgs;
  std::vector<std::string> dbg;
  std::string err;
  bool capture = true;                          // false: produced items are counted, not kept (throughput runs)
  uint64_t jitter = 0;                          // != 0: xorshift state; every call gets random item counts (scheduler-like)
  uint32_t rnd() { jitter ^= jitter << 13; jitter ^= jitter >> 7; jitter ^= jitter << 17; return (uint32_t)(jitter >> 11); }

  uint64_t nitems_read(unsigned) override { return n_read; }
  uint64_t nitems_written(unsigned) override { return n_written; }
  void get_tags_in_range(std::vector<tag_t>& v, unsigned, uint64_t a, uint64_t b) override
  {
    v.clear();
    if (!tags_sorted) { std::stable_sort(in_tags.begin(), in_tags.end(), [](const tag_t& x, const tag_t& y) { return x.offset < y.offset; }); tags_sorted = true; }
    prune_tags();
    auto it = std::lower_bound(in_tags.begin(), in_tags.end(), a, [](const tag_t& t, uint64_t off) { return t.offset < off; });
    for (; it != in_tags.end() && it->offset < b; ++it) v.push_back(*it);
  }
  void push_tag(const tag_t& t) { if (!in_tags.empty() && t.offset < in_tags.back().offset) tags_sorted = false; in_tags.push_back(t); }
  void push_tag(tag_t&& t) { if (!in_tags.empty() && t.offset < in_tags.back().offset) tags_sorted = false; in_tags.push_back(std::move(t)); }
  void add_item_tag(unsigned, uint64_t off, const std::string& key, const tag_value& val) override
  {
    tag_t t; t.offset = off; t.key = key; t.value = val; out_tags.push_back(t);
    n_tags_out++;
  }
  long n_tags_out = 0;                          // every add_item_tag call of the block's life (c8ph_tags_emitted)
  void consume_each(int n) override { consumed_this_call = n; }
  void message_port_pub(const std::string&, const pdu_t& m) override { std::lock_guard<std::mutex> g(msg_mu); msgs.push_back(m); }
  void debug_print(const std::string& s) override { std::lock_guard<std::mutex> g(msg_mu); dbg.push_back(s); }
  // the front's latency timer asks for a call (results became ready while no input was arriving)
  void wake() override { { std::lock_guard<std::mutex> g(wake_mu); woken = true; } wake_cv.notify_all(); }
  bool input_is_pinned(unsigned port) override { return port < in.size() && in[port].pinned; }
  // the buffer a scheduler would have given this port: twice what a call is offered (set by the run loops before their first call)
  size_t in_cap_bytes = 0;
  size_t input_buffer_bytes(unsigned) override { return in_cap_bytes; }
  bool output_is_pinned(unsigned) override { return pinned_out; }
  // c8ph_run_chain knows the edge it stands for: 1 when the block behind it is the decode mirror, 0 when it is anything else; a block run
  // by itself (c8ph_run) has no reader the harness could know of
  int feeds_decode = -1;
  int output_feeds_only_decode(unsigned) override { return feeds_decode; }
  // an output buffer of n floats, from the block's allocator when its io_signature says so
  struct obuf_t {
    harness* h; float* p; bool pinned;
    obuf_t(harness* h_, size_t n) : h(h_), pinned(h_->pinned_out) { p = static_cast<float*>(h->balloc(std::max<size_t>(n, 16) * sizeof(float), pinned)); }
    ~obuf_t() { h->bfree(p, pinned); }
    obuf_t(const obuf_t&) = delete;
    obuf_t& operator=(const obuf_t&) = delete;
  };
  std::mutex msg_mu, wake_mu;
  std::condition_variable wake_cv;
  bool woken = false;
  std::string phases;
};

} // namespace

extern "C" {

// fused: 0 / 1 = batching_cfg::fused (fused_auto off), -1 = the defaults (environment C8P_FUSED; unset: decided from the flowgraph —
// c8ph_run_chain tells a demod whose reader is the decode mirror so, and it then runs fused)
// pinned_io / defer_consume: 0 / 1 = batching_cfg's fields, -1 = their defaults (environment C8P_PINNED_IO, C8P_DEFER_CONSUME)
void* c8ph_create_ex2(int kind, int mupos, int ifdebug, const char* lib, int max_batch, int flush_idle, int fused, int pinned_io, int defer_consume,
                      char* errbuf, int errcap)
{
  try {
    harness* h = new harness();
    c8p_api api = load_c8p_api(lib);
    batching_cfg cfg; cfg.max_batch = max_batch; cfg.flush_when_idle = flush_idle != 0 && !batching_cfg::safe_mode();
    if (fused >= 0) { cfg.fused = fused != 0; cfg.fused_auto = false; }      // (said explicitly: nothing left to decide from the flowgraph)
    if (pinned_io >= 0) cfg.pinned_io = pinned_io != 0;
    if (defer_consume >= 0) cfg.defer_consume = defer_consume != 0;
    if (kind == 0) { h->blk = demod::make(mupos, 0, api, cfg); h->nports = 1; }
    else if (kind == 1) { h->blk = demod2::make(api, cfg); h->nports = 2; }
    else if (kind == 2) { h->blk = decode::make(ifdebug != 0, api, cfg); h->nports = 1; h->item_size = sizeof(float); }
    else if (kind == 3) { h->blk = signal::make(api, cfg); h->nports = 2; h->port0_size = 1; h->out_size = sizeof(gr_complex); }
    else if (kind == 4) { h->blk = signal2::make(api, cfg); h->nports = 3; h->port0_size = 1; h->out_size = sizeof(gr_complex); h->nout = 2; }
    else if (kind >= 100) {                                  // rx_chain with kind - 100 streams; max_batch = items per stream and round
      cfg.detect_items = max_batch;
      h->blk = rx_chain::make(kind - 100, lib, cfg); h->nports = kind - 100;
    }
    else { cfg.detect_items = max_batch; h->blk = detect::make(api, cfg); h->nports = 1; h->out_size = 1; }   // max_batch = items per call
    h->setup_buffers();
    h->blk->set_io(h);
    return h;
  } catch (const std::exception& e) {
    if (errbuf && errcap > 0) { strncpy(errbuf, e.what(), errcap - 1); errbuf[errcap - 1] = 0; }
    return nullptr;
  }
}
void* c8ph_create_ex(int kind, int mupos, int ifdebug, const char* lib, int max_batch, int flush_idle, int fused, char* errbuf, int errcap)
{
  return c8ph_create_ex2(kind, mupos, ifdebug, lib, max_batch, flush_idle, fused, -1, -1, errbuf, errcap);
}
// rx_chain with item_format C8P_FMT_* ports (gr_complex / std::complex<short> / std::complex<char> items) and its conversion scale
void* c8ph_create_rx_chain(int n_streams, int item_format, float scale, const char* lib, int items_per_round, int flush_idle, char* errbuf, int errcap)
{
  try {
    harness* h = new harness();
    batching_cfg cfg; cfg.detect_items = items_per_round; cfg.flush_when_idle = flush_idle != 0;
    h->blk = rx_chain::make(n_streams, lib, cfg, item_format, scale);
    h->nports = n_streams;
    h->item_size = item_format == C8P_FMT_SC16 ? 4 : item_format == C8P_FMT_SC8 ? 2 : sizeof(gr_complex);
    h->setup_buffers();
    h->blk->set_io(h);
    return h;
  } catch (const std::exception& e) {
    if (errbuf && errcap > 0) { strncpy(errbuf, e.what(), errcap - 1); errbuf[errcap - 1] = 0; }
    return nullptr;
  }
}
void* c8ph_create(int kind, int mupos, int ifdebug, const char* lib, int max_batch, int flush_idle, char* errbuf, int errcap)
{
  return c8ph_create_ex(kind, mupos, ifdebug, lib, max_batch, flush_idle, -1, errbuf, errcap);
}
void c8ph_destroy(void* p) { delete static_cast<harness*>(p); }

void c8ph_push_items(void* p, int port, const void* data, long n_items)
{
  harness* h = static_cast<harness*>(p);
  h->append(port, static_cast<const uint8_t*>(data), (size_t)n_items * h->isz(port));
}
void c8ph_push_tag_long(void* p, uint64_t off, const char* key, long v)
{
  tag_t t; t.offset = off; t.key = key; t.value = tag_value::from_long(v); static_cast<harness*>(p)->push_tag(t);
}
void c8ph_push_tag_float(void* p, uint64_t off, const char* key, float v)
{
  tag_t t; t.offset = off; t.key = key; t.value = tag_value::from_float(v); static_cast<harness*>(p)->push_tag(t);
}
void c8ph_push_tag_c32(void* p, uint64_t off, const char* key, const float* reim, int n)
{
  std::vector<gr_complex> v(n);
  for (int i = 0; i < n; i++) v[i] = gr_complex(reim[2 * i], reim[2 * i + 1]);
  tag_t t; t.offset = off; t.key = key; t.value = tag_value::from_c32vector(v); static_cast<harness*>(p)->push_tag(t);
}

// Run until the input is exhausted; `chunk` = max items offered per call, `nout` = noutput_items per call.
// Returns the number of general_work calls, or -1 on error (c8ph_error()).
// from now on c8ph_run() hands general_work a random number of input items (1..chunk, sometimes more on one port than on the
// others, sometimes none) and a random output space (1..nout) on every call, the way a busy scheduler does
void c8ph_set_capture(void* p, int on) { static_cast<harness*>(p)->capture = on != 0; }
void c8ph_set_jitter(void* p, unsigned long long seed) { static_cast<harness*>(p)->jitter = seed ? seed : 0; }
// from now on every consumed input item (and every recycled buffer of c8ph_run_chain's edge) is overwritten with 0xff bytes at once: a
// block that lets go of items something is still reading — a gather it has not waited for — produces wrong results instead of passing by luck
void c8ph_set_poison(void* p, int on) { static_cast<harness*>(p)->poison = on != 0; }
// every n-th call of c8ph_run finds its pinned input buffers relocated (harness::relocate); 0: never
void c8ph_set_relocate(void* p, int every) { static_cast<harness*>(p)->relocate = every > 0 ? every : 0; }
// every n-th call of c8ph_run finds its output position moved under a c8p::buffer_move_scope (harness::move_out); 0: never
void c8ph_set_move_out(void* p, int every) { static_cast<harness*>(p)->move_out = every > 0 ? every : 0; }
long c8ph_moves_done(void* p) { return static_cast<harness*>(p)->moves_done; }
// a buffer somewhere wants to move, from ANOTHER thread than the block's: can every front be got hold of within timeout_us?  (1 / 0;
// the range is nobody's, so nothing is saved or moved — the test is about the bounded wait of c8p::buffer_move_scope)
int c8ph_try_move(int timeout_us) { static char nobody[16]; buffer_move_scope mv(nobody, sizeof(nobody), timeout_us); return mv.ok() ? 1 : 0; }
int c8ph_input_pinned(void* p, int port) { return static_cast<harness*>(p)->input_is_pinned((unsigned)port) ? 1 : 0; }
int c8ph_output_pinned(void* p) { return static_cast<harness*>(p)->output_is_pinned(0) ? 1 : 0; }
// c8p_block::fused_state: 1 / 0 = the demod front runs fused / plain, -1 = not decided yet (no general_work call so far) or no demod
int c8ph_fused_state(void* p) { return static_cast<harness*>(p)->blk->fused_state(); }
// stream tags the block has written so far (add_item_tag calls), whoever took them
long c8ph_tags_emitted(void* p) { return static_cast<harness*>(p)->n_tags_out; }

long c8ph_run(void* p, int chunk, int nout, int flush_at_end)
{
  harness* h = static_cast<harness*>(p);
  h->in_cap_bytes = 2 * (size_t)std::max(chunk, 1) * h->item_size;
  try {
    long calls = 0;
    int idle = 0, starved = 0;
    const size_t fpi = std::max<size_t>(h->out_size / sizeof(float), 1);   // floats per output item (byte items: widened below)
    // The output buffer: the pointer a call is given continues where the last call's produce() stopped (a block may have placed items
    // beyond what it reported: c8p_block::output_ahead); at the end of the buffer the block is first asked for exactly those items
    const size_t ocap = (size_t)std::max(nout, 1) * fpi * 4;
    harness::obuf_t ob0(h, ocap), ob1(h, ocap);
    // (however this function is left — a block that throws has launches under way that write into ob0 — they are waited for before
    // the buffer is freed)
    struct out_guard { float* p; size_t bytes; ~out_guard() { try { buffer_will_move(p, bytes); } catch (...) {} } } og{ob0.p, ocap * sizeof(float)};
    (void)og;
    size_t wpos = 0;
    auto take = [&](int produced, const float* o0, const float* o1) {
      if (!h->capture) {
      } else if (h->out_size == 1) {                                // byte stream: one float per item in the capture
        const uint8_t* b = reinterpret_cast<const uint8_t*>(o0);
        for (int i = 0; i < produced; i++) h->out_items.push_back((float)b[i]);
      } else
        h->out_items.insert(h->out_items.end(), o0, o0 + (size_t)produced * fpi);
      if (h->nout == 2 && h->capture) h->out_items1.insert(h->out_items1.end(), o1, o1 + (size_t)produced * fpi);
      h->n_written += produced;
      wpos += (size_t)produced * fpi;
    };
    auto wrap_if_needed = [&](int want_items) {
      if (ocap - wpos >= (size_t)want_items * fpi) return;
      for (int ahead; (ahead = h->blk->output_ahead()) > 0;) {      // calls without input and with room for exactly the items under way
        gr_vector_int nin0(h->nports, 0);
        gr_vector_const_void_star ins0(h->nports);
        for (int q = 0; q < h->nports; q++) ins0[q] = h->rd(q);
        gr_vector_void_star outs0(1, ob0.p + wpos);
        if (h->nout == 2) outs0.push_back(ob1.p + wpos);
        h->consumed_this_call = 0;
        const int produced = h->blk->general_work(ahead, nin0, ins0, outs0);
        for (int q = 0; q < h->nports; q++) h->advance(q, (size_t)h->consumed_this_call);
        h->n_read += h->consumed_this_call;
        take(produced, ob0.p + wpos, ob1.p + wpos);
      }
      wpos = 0;
    };
    while (idle < 4) {
      size_t avail = h->avail(0);
      for (int q = 1; q < h->nports; q++) avail = std::min(avail, h->avail(q));
      int n = (int)std::min<size_t>(avail, (size_t)chunk);
      int nout_call = nout;
      const bool full = !h->jitter || starved > 32;                 // after a run of fruitless small calls: one full-size call
      if (!full) {
        if (n > 0) n = (h->rnd() % 16 == 0) ? 0 : 1 + (int)(h->rnd() % (uint32_t)n);
        nout_call = 1 + (int)(h->rnd() % (uint32_t)std::max(nout, 1));
      }
      gr_vector_int nin(h->nports, n);
      if (!full && h->nports > 1) {                                 // ports rarely hold the same number of items
        const int q = (int)(h->rnd() % (uint32_t)h->nports);
        const size_t aq = h->avail(q);
        nin[q] = (int)std::min<size_t>(aq, (size_t)n + h->rnd() % 97);
      }
      if (h->relocate && calls % h->relocate == h->relocate - 1) h->relocate_inputs();
      gr_vector_const_void_star ins(h->nports);
      for (int q = 0; q < h->nports; q++) ins[q] = h->rd(q);
      if (h->move_out && h->nout == 1 && calls % h->move_out == h->move_out - 1) {
        // the write position jumps (to the start of the buffer, or a quarter in when it is there already) with items possibly placed ahead
        const size_t ahead = (size_t)h->blk->output_ahead() * fpi;
        const size_t to = wpos ? 0 : ocap / 4;
        if (to + ahead + (size_t)std::max(nout, 1) * fpi <= ocap) {
          if (getenv("C8P_TEST_UNBRACKETED_MOVE")) wpos = to;      // (a test: a mover that does not follow the protocol must be caught)
          else {
            buffer_move_scope mv(ob0.p, ocap * sizeof(float));
            memset(ob0.p, 0xff, ocap * sizeof(float));              // what the scope has not saved is gone
            wpos = to;
            mv.moved_to(ob0.p + wpos);
          }
          h->moves_done++;
        }
      }
      wrap_if_needed(std::max(nout, 1));
      float* const obuf = ob0.p + wpos; float* const obuf1 = ob1.p + wpos;
      gr_vector_void_star outs(1, obuf);
      if (h->nout == 2) outs.push_back(obuf1);
      h->consumed_this_call = 0;
      const int produced = h->blk->general_work(nout_call, nin, ins, outs);
      calls++;
      for (int q = 0; q < h->nports; q++) h->advance(q, (size_t)h->consumed_this_call);
      h->n_read += h->consumed_this_call;
      take(produced, obuf, obuf1);
      if (!full) {                                                  // a starved call that got nowhere proves nothing
        if (produced || h->consumed_this_call || h->blk->output_ahead()) { idle = 0; starved = 0; } else starved++;
        continue;
      }
      starved = 0;
      // (a call that only PLACED items in the output buffer — reported by the next one — has made progress)
      idle = (produced == 0 && h->consumed_this_call == 0 && h->blk->output_ahead() == 0) ? idle + 1 : 0;
      if (idle == 2 && flush_at_end && avail == 0) h->blk->flush();
    }
    return calls;
  } catch (const std::exception& e) {
    h->err = e.what();
    return -1;
  }
}
// Two blocks the way a flowgraph runs them: each on its own thread (GNU Radio's thread-per-block scheduler), A's output
// port 0 (items + tags) feeding B's input port 0 through the buffer of that edge.  A's items are not captured (c8ph_out_count stays 0),
// B's messages / debug lines are.  Returns the number of general_work calls of both, -1 on error (c8ph_error of either).
long c8ph_run_chain(void* pa, void* pb, int chunk, int nout)
{
  harness* A = static_cast<harness*>(pa);
  harness* B = static_cast<harness*>(pb);
  // The edge's buffer, as between two GNU Radio blocks: an item crosses it with ONE write (A's general_work, straight into the buffer)
  // and ONE read (B's general_work, straight out of it), A's output pointer continues where its last call stopped, and B reads what A
  // has produced while A goes on producing.  A ring is stood in for by three long segments that are used in turn (A moves on when the
  // space left in one is less than a call's worth, B when it has consumed a closed one to its end): no item ever moves.  The memory
  // comes from the blocks' buffer type when either names one for this edge (pinned host memory), as flat_flowgraph::allocate_buffer
  // picks an edge's gr::buffer_type.  The segments are address space only: the edge HOLDS what a buffer of two calls' worth holds
  // (edge_cap: 16 MB when a call is offered 8 MB) — A is offered no more than the room B has left by consuming, and waits when there is none.
  const size_t fpi = std::max<size_t>(A->out_size / sizeof(float), 1);
  const bool pin_edge = A->pinned_out || B->in[0].pinned;
  harness* owner = A->pinned_out ? A : B;
  A->pinned_out = pin_edge; B->in[0].pinned = pin_edge;          // what the two blocks are told about the edge
  A->feeds_decode = B->blk->is_decode_front() ? 1 : 0;           // ... and who reads it (block_io::output_feeds_only_decode)
  A->in_cap_bytes = 2 * (size_t)std::max(chunk, 1) * A->item_size;
  B->in_cap_bytes = 2 * std::max((size_t)std::max(chunk, 1), (size_t)std::max(nout, 1) * fpi) * sizeof(float);
  const size_t seg_floats = std::max<size_t>((size_t)std::max(nout, 1) * fpi * 16, (size_t)1 << 23);
  const size_t edge_cap = 2 * (size_t)std::max(nout, 1) * fpi;   // floats
  size_t a_reported = 0, b_consumed = 0;                         // floats A has reported as produced / B has consumed (b_consumed: under mu)
  struct segment { float* items = nullptr; size_t n = 0; bool closed = false; };      // n, closed: under mu
  segment segs[3];
  {
    float** e = owner->edge_segments(seg_floats, pin_edge);
    for (int i = 0; i < 3; i++) segs[i].items = e[i];
  }
  std::mutex mu;
  std::condition_variable cv;
  std::deque<int> filled, spare = {0, 1, 2};   // segments A has written into (in order; the last one may still be open) / free ones
  std::deque<tag_t> tagq;                      // A's tags on their way to B's input (moved before the items they sit on are published)
  bool a_done = false;
  std::atomic<bool> failed{false};
  long calls_a = 0, calls_b = 0;
  // C8P_CHAIN_TLOG=1: where the two scheduler threads spent the run (stderr): inside general_work, waiting for the other, the rest
  static const bool tlog = getenv("C8P_CHAIN_TLOG") != nullptr;
  typedef std::chrono::steady_clock clk;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  double a_work = 0, a_wait = 0, a_all = 0, b_work = 0, b_wait = 0, b_all = 0;
  // C8P_CHAIN_PIN="a,b": the two scheduler threads on CPUs a and b (GNU Radio: block::set_processor_affinity) — on a host whose
  // logical CPUs are hyper-thread pairs, two busy threads that land on one core each run at ~60 %
  int pin_a = -1, pin_b = -1;
  if (const char* e = getenv("C8P_CHAIN_PIN")) sscanf(e, "%d,%d", &pin_a, &pin_b);
  auto pin_self = [](int cpu) {
    if (cpu < 0) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
  };
  std::thread ta([&] {
    pin_self(pin_a);
    const clk::time_point ta0 = clk::now();
    try {
      int idle = 0, cur = -1;
      size_t wpos = 0;                           // floats of the current segment handed to B so far
      while (idle < 4) {
        size_t avail = A->avail(0);
        for (int qq = 1; qq < A->nports; qq++) avail = std::min(avail, A->avail(qq));
        const int n = (int)std::min<size_t>(avail, (size_t)chunk);
        gr_vector_int nin(A->nports, n);
        gr_vector_const_void_star ins(A->nports);
        for (int qq = 0; qq < A->nports; qq++) ins[qq] = A->rd(qq);
        if (cur >= 0 && seg_floats - wpos < (size_t)std::max(nout, 1) * fpi) {     // not a call's worth of space left: the next segment
          for (int ahead; (ahead = A->blk->output_ahead()) > 0;) {                 // (items the block has placed beyond what it reported)
            gr_vector_int nin0(A->nports, 0);
            gr_vector_void_star outs0(1, segs[cur].items + wpos);
            A->consumed_this_call = 0;
            const size_t tb = A->out_tags.size();
            const int produced = A->blk->general_work(ahead, nin0, ins, outs0);
            a_reported += (size_t)produced * fpi;
            calls_a++;
            for (int qq = 0; qq < A->nports; qq++) A->advance(qq, (size_t)A->consumed_this_call);
            A->n_read += A->consumed_this_call;
            A->n_written += produced;
            std::lock_guard<std::mutex> g(mu);
            for (size_t k = tb; k < A->out_tags.size(); k++) tagq.push_back(std::move(A->out_tags[k]));
            A->out_tags.resize(tb);
            wpos += (size_t)produced * fpi;
            segs[cur].n = wpos;
          }
          { std::lock_guard<std::mutex> g(mu); segs[cur].closed = true; }
          cv.notify_all();
          cur = -1;
        }
        if (cur < 0) {
          const clk::time_point tw = clk::now();
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return !spare.empty() || failed; });                   // (back-pressure: B has not finished a segment yet)
          a_wait += secs(tw, clk::now());
          if (failed) break;
          cur = spare.front(); spare.pop_front();
          segs[cur].n = 0; segs[cur].closed = false;
          filled.push_back(cur);
          wpos = 0;
        }
        // the room the edge has: what A has reported and B has not consumed is in it (items A has placed ahead lie at the start of the
        // space it is offered and are part of that offer)
        int nout_call = nout;
        {
          std::unique_lock<std::mutex> lk(mu);
          for (;;) {
            const size_t held = a_reported - b_consumed;
            nout_call = (int)std::min<size_t>((size_t)std::max(nout, 1), held < edge_cap ? (edge_cap - held) / fpi : 0);
            if (nout_call > 0 || failed) break;
            const clk::time_point tw = clk::now();
            cv.wait(lk);
            a_wait += secs(tw, clk::now());
          }
          if (failed) break;
        }
        gr_vector_void_star outs(1, segs[cur].items + wpos);
        A->consumed_this_call = 0;
        const size_t tags_before = A->out_tags.size();
        const clk::time_point tg = clk::now();
        const int produced = A->blk->general_work(nout_call, nin, ins, outs);
        a_work += secs(tg, clk::now());
        a_reported += (size_t)produced * fpi;
        calls_a++;
        for (int qq = 0; qq < A->nports; qq++) A->advance(qq, (size_t)A->consumed_this_call);
        A->n_read += A->consumed_this_call;
        A->n_written += produced;
        if (produced > 0 || A->out_tags.size() > tags_before) {
          std::lock_guard<std::mutex> g(mu);
          for (size_t k = tags_before; k < A->out_tags.size(); k++) tagq.push_back(std::move(A->out_tags[k]));
          A->out_tags.resize(tags_before);
          wpos += (size_t)produced * fpi;
          segs[cur].n = wpos;
        }
        if (produced > 0) cv.notify_all();
        idle = (produced == 0 && A->consumed_this_call == 0 && A->blk->output_ahead() == 0) ? idle + 1 : 0;
        if (idle == 2 && avail == 0) A->blk->flush();
      }
      if (cur >= 0) { std::lock_guard<std::mutex> g(mu); segs[cur].closed = true; }
    } catch (const std::exception& e) { A->err = e.what(); failed = true; }
    a_all = secs(ta0, clk::now());
    { std::lock_guard<std::mutex> g(mu); a_done = true; }
    cv.notify_all();
  });
  const clk::time_point tb0 = clk::now();
  cpu_set_t b_before;
  const bool b_saved = pin_b >= 0 && pthread_getaffinity_np(pthread_self(), sizeof(b_before), &b_before) == 0;
  pin_self(pin_b);
  try {
    std::vector<float> dummy(16);
    gr_vector_void_star outs(1, dummy.data());
    int cur = -1;
    size_t rpos = 0;                             // floats of the current segment B has consumed
    int fruitless = 0;                           // calls in a row that consumed nothing although nothing new had arrived
    size_t seen = 0;
    for (;;) {
      size_t n_avail = 0;
      bool closed = false, fin = false;
      {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
          if (cur < 0 && !filled.empty()) { cur = filled.front(); rpos = 0; seen = 0; fruitless = 0; }
          if (cur >= 0) {
            n_avail = segs[cur].n - rpos; closed = segs[cur].closed;
            if (closed && n_avail == 0) {           // consumed to its end: back to A
              if (B->poison) memset(segs[cur].items, 0xff, seg_floats * sizeof(float));
              filled.pop_front(); spare.push_back(cur); cur = -1;
              cv.notify_all();
              continue;
            }
            // something to offer: new items, or items B has looked at but not let go of yet (a deferred consume: calls that bring new
            // items may consume nothing — the block lets go of items two calls late — and the first call without new ones settles it all)
            if (n_avail > 0 && (segs[cur].n > seen || fruitless < 4)) break;
          }
          if (a_done && filled.empty()) { fin = true; break; }
          if (a_done && cur >= 0 && fruitless >= 4) { fin = true; break; }          // (never: a block that sits on items it will not consume)
          const clk::time_point tw = clk::now();
          cv.wait(lk);
          b_wait += secs(tw, clk::now());
        }
        while (!tagq.empty()) { B->push_tag(std::move(tagq.front())); tagq.pop_front(); }
        if (cur >= 0) seen = segs[cur].n;
      }
      if (fin) break;
      // (what is in the buffer, up to what one call of A can have put there: a reader is offered all the items its buffer holds)
      const int k = (int)std::min<size_t>(n_avail, std::max((size_t)chunk, (size_t)std::max(nout, 1) * fpi));
      gr_vector_int nin(1, k);
      gr_vector_const_void_star ins(1, segs[cur].items + rpos);
      B->consumed_this_call = 0;
      const clk::time_point tg = clk::now();
      B->blk->general_work(0, nin, ins, outs);
      b_work += secs(tg, clk::now());
      calls_b++;
      if (B->poison && B->consumed_this_call) memset(segs[cur].items + rpos, 0xff, (size_t)B->consumed_this_call * sizeof(float));
      rpos += (size_t)B->consumed_this_call;
      B->n_read += B->consumed_this_call;
      if (B->consumed_this_call) { { std::lock_guard<std::mutex> g(mu); b_consumed += (size_t)B->consumed_this_call; } cv.notify_all(); }
      fruitless = B->consumed_this_call == 0 ? fruitless + 1 : 0;
    }
    // the producer is done and everything it made was offered: a few dry calls, then end of stream
    for (int idle = 0; idle < 4; idle++) {
      gr_vector_int nin(1, 0);
      gr_vector_const_void_star ins(1, dummy.data());
      B->consumed_this_call = 0;
      B->blk->general_work(0, nin, ins, outs);
      calls_b++;
      if (idle == 1) B->blk->flush();
    }
  } catch (const std::exception& e) { B->err = e.what(); { std::lock_guard<std::mutex> g(mu); failed = true; } cv.notify_all(); }
  b_all = secs(tb0, clk::now());
  if (b_saved) (void)pthread_setaffinity_np(pthread_self(), sizeof(b_before), &b_before);
  ta.join();
  if (tlog)
    fprintf(stderr, "[c8ph_run_chain] A: %ld calls, %.2f ms in general_work, %.2f ms waiting for a segment, %.2f ms all | B: %ld calls, %.2f ms in general_work, "
            "%.2f ms waiting for items, %.2f ms all | items written %llu read %llu\n", calls_a, 1e3 * a_work, 1e3 * a_wait, 1e3 * a_all, calls_b, 1e3 * b_work, 1e3 * b_wait, 1e3 * b_all,
            (unsigned long long)A->n_written, (unsigned long long)B->n_read);
  return failed ? -1 : calls_a + calls_b;
}

// A LIVE scheduler: general_work is only ever called when the block's forecast() is satisfied — i.e. never with zero input items
// unless the block itself asks for nothing (results waiting for output space) — and nothing flushes at the end.  After the input
// is used up the loop idles like a flowgraph whose source went quiet: it sleeps until the block's wake() or `linger_ms` have passed.
// Returns the number of general_work calls, -1 on error.
long c8ph_run_live(void* p, int chunk, int nout, int linger_ms)
{
  harness* h = static_cast<harness*>(p);
  h->in_cap_bytes = 2 * (size_t)std::max(chunk, 1) * h->item_size;
  try {
    long calls = 0;
    const size_t fpi = std::max<size_t>(h->out_size / sizeof(float), 1);
    const size_t ocap = (size_t)std::max(nout, 1) * fpi * 4;      // the output pointer continues where produce() stopped (see c8ph_run)
    harness::obuf_t ob0(h, ocap), ob1(h, ocap);
    size_t wpos = 0;
    const auto t_end_idle = std::chrono::milliseconds(linger_ms);
    auto idle_since = std::chrono::steady_clock::now();
    for (;;) {
      size_t avail = h->avail(0);
      for (int q = 1; q < h->nports; q++) avail = std::min(avail, h->avail(q));
      gr_vector_int need(h->nports, 0);
      h->blk->forecast(nout, need);
      int want = 0;
      for (int q = 0; q < h->nports; q++) want = std::max(want, need[q]);
      const int n = (int)std::min<size_t>(avail, (size_t)chunk);
      // (forecast(noutput) asks for noutput items like the reference; the scheduler halves noutput until the request fits what
      // the buffers hold, down to one item: the call is made as soon as there is ANY input)
      if (want > 0 && n == 0) {
        std::unique_lock<std::mutex> lk(h->wake_mu);
        if (!h->woken) {
          if (std::chrono::steady_clock::now() - idle_since >= t_end_idle) break;
          h->wake_cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(1));
        }
        h->woken = false;
        continue;
      }
      gr_vector_int nin(h->nports, n);
      gr_vector_const_void_star ins(h->nports);
      for (int q = 0; q < h->nports; q++) ins[q] = h->rd(q);
      int nout_call = nout;
      if (ocap - wpos < (size_t)std::max(nout, 1) * fpi) {        // the end of the buffer: first the items the block has placed ahead, alone
        const int ahead = h->blk->output_ahead();
        if (ahead > 0) { nout_call = ahead; for (int q = 0; q < h->nports; q++) nin[q] = 0; }
        else wpos = 0;
      }
      float* const obuf = ob0.p + wpos; float* const obuf1 = ob1.p + wpos;
      gr_vector_void_star outs(1, obuf);
      if (h->nout == 2) outs.push_back(obuf1);
      h->consumed_this_call = 0;
      const int produced = h->blk->general_work(nout_call, nin, ins, outs);
      calls++;
      for (int q = 0; q < h->nports; q++) h->advance(q, (size_t)h->consumed_this_call);
      h->n_read += h->consumed_this_call;
      if (h->capture) {
        h->out_items.insert(h->out_items.end(), obuf, obuf + (size_t)produced * fpi);
        if (h->nout == 2) h->out_items1.insert(h->out_items1.end(), obuf1, obuf1 + (size_t)produced * fpi);
      }
      h->n_written += produced;
      wpos += (size_t)produced * fpi;
      if (produced || h->consumed_this_call) idle_since = std::chrono::steady_clock::now();
      else if (want == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (asked for a call and did nothing: do not spin)
    }
    return calls;
  } catch (const std::exception& e) {
    h->err = e.what();
    return -1;
  }
}

void c8ph_host_phase_reset(void* p) { static_cast<harness*>(p)->blk->host_phase_reset(); }
const char* c8ph_host_phase_us(void* p) { harness* h = static_cast<harness*>(p); h->phases = h->blk->host_phase_us(); return h->phases.c_str(); }

const char* c8ph_error(void* p) { return static_cast<harness*>(p)->err.c_str(); }

long c8ph_out_count(void* p) { return (long)static_cast<harness*>(p)->out_items.size(); }     // in floats
void c8ph_out_copy(void* p, float* dst) { harness* h = static_cast<harness*>(p); memcpy(dst, h->out_items.data(), h->out_items.size() * 4); }
void c8ph_out1_copy(void* p, float* dst) { harness* h = static_cast<harness*>(p); memcpy(dst, h->out_items1.data(), h->out_items1.size() * 4); }
long c8ph_tag_count(void* p) { return (long)static_cast<harness*>(p)->out_tags.size(); }
// kind: 0 long, 1 float, 2 c32vector (vec_n = element count)
int c8ph_tag_get(void* p, long i, uint64_t* off, char* key, int keycap, long* l, float* f, int* vec_n)
{
  const tag_t& t = static_cast<harness*>(p)->out_tags[(size_t)i];
  *off = t.offset; strncpy(key, t.key.c_str(), keycap - 1); key[keycap - 1] = 0;
  *l = t.value.l; *f = t.value.f; *vec_n = (int)t.value.v.size();
  return (int)t.value.kind;
}
void c8ph_tag_vec(void* p, long i, float* reim)
{
  const tag_t& t = static_cast<harness*>(p)->out_tags[(size_t)i];
  for (size_t k = 0; k < t.value.v.size(); k++) { reim[2 * k] = t.value.v[k].real(); reim[2 * k + 1] = t.value.v[k].imag(); }
}
long c8ph_msg_count(void* p) { return (long)static_cast<harness*>(p)->msgs.size(); }
long c8ph_msg_len(void* p, long i) { return static_cast<harness*>(p)->msgs[(size_t)i].len; }
long c8ph_msg_stream(void* p, long i) { return static_cast<harness*>(p)->msgs[(size_t)i].stream; }
void c8ph_msg_copy(void* p, long i, uint8_t* dst)
{
  const pdu_t& m = static_cast<harness*>(p)->msgs[(size_t)i];
  memcpy(dst, m.blob.data(), m.blob.size());
}
long c8ph_dbg_count(void* p) { return (long)static_cast<harness*>(p)->dbg.size(); }
int c8ph_dbg_get(void* p, long i, char* dst, int cap)
{
  const std::string& s = static_cast<harness*>(p)->dbg[(size_t)i];
  strncpy(dst, s.c_str(), cap - 1); dst[cap - 1] = 0;
  return (int)s.size();
}

} // extern "C"
