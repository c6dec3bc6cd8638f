// c8p_rawchain.cc — see c8p_rawchain.h.  Three host threads around four engine contexts that live as long as the handle; bounded
// hand-over queues; every stage keeps its own clock (busy / waiting) and every segment its marks, so that a slow run says where.
#include "c8p_rawchain.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "c8p_blocks.h"

using namespace C8P_MIRROR_QUAL;

namespace {

const uint32_t MAX_PKT_ITEMS = 224u + 109280u;   // the longest packet from its sync flag on: 4095 bytes at 6 Mb/s (signal_impl.cc:125-131)
// items of a stream carried into the next round: MAX_PKT_ITEMS + detection history + slack.  (C8P_RAWCHAIN_KEEP: tests only — a smaller
// carry for streams whose packets are known to be short, so that small test streams exercise the carry; a flag or a detection history
// that falls out of the arena is an error, never a silent loss)
uint64_t live_keep()
{
  const char* e = getenv("C8P_RAWCHAIN_KEEP");
  return e && atoll(e) >= 1024 ? (uint64_t)atoll(e) : 110000u;
}
const int LIVE_ARENAS = 3;      // (4 and 6 measured in round 4: 5-10 % slower — more rounds in flight, more transfers contending for the link)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T> class handover {           // bounded queue, closed by the producer (or by a failing stage)
public:
  explicit handover(size_t cap) : d_cap(cap) {}
  bool put(T&& v)
  {
    std::unique_lock<std::mutex> g(d_mu);
    d_cv.wait(g, [&] { return d_q.size() < d_cap || d_closed; });
    if (d_closed) return false;
    d_q.push_back(std::move(v));
    d_cv.notify_all();
    return true;
  }
  bool get(T& v)
  {
    std::unique_lock<std::mutex> g(d_mu);
    d_cv.wait(g, [&] { return !d_q.empty() || d_closed; });
    if (d_q.empty()) return false;
    v = std::move(d_q.front());
    d_q.pop_front();
    d_cv.notify_all();
    return true;
  }
  // 0: got one; 1: nothing arrived within `us` microseconds; 2: closed and empty
  int get_for(T& v, int us)
  {
    std::unique_lock<std::mutex> g(d_mu);
    // (system_clock: pthread_cond_timedwait — the steady-clock wait is pthread_cond_clockwait, which this toolchain's ThreadSanitizer
    // does not intercept and then reports the mutex as held across the wait; a clock step can only stretch one 100 us poll)
    if (!d_cv.wait_until(g, std::chrono::system_clock::now() + std::chrono::microseconds(us), [&] { return !d_q.empty() || d_closed; })) return 1;
    if (d_q.empty()) return 2;
    v = std::move(d_q.front());
    d_q.pop_front();
    d_cv.notify_all();
    return 0;
  }
  void close() { std::lock_guard<std::mutex> g(d_mu); d_closed = true; d_cv.notify_all(); }
private:
  std::mutex d_mu;
  std::condition_variable d_cv;
  std::deque<T> d_q;
  size_t d_cap;
  bool d_closed = false;
};

class counter {                                // counting semaphore; a failing stage opens it for good
public:
  explicit counter(int n) : d_n(n) {}
  void acquire() { std::unique_lock<std::mutex> g(d_mu); d_cv.wait(g, [&] { return d_n > 0 || d_open; }); d_n--; }
  void release() { std::lock_guard<std::mutex> g(d_mu); d_n++; d_cv.notify_all(); }
  void open() { std::lock_guard<std::mutex> g(d_mu); d_open = true; d_cv.notify_all(); }
private:
  std::mutex d_mu;
  std::condition_variable d_cv;
  int d_n;
  bool d_open = false;
};

struct clocked {                               // a stage's three clocks
  c8p_rawchain_stage t = {0, 0, 0};
  template <class F> auto busy(F&& f) { const double a = now_s(); auto r = f(); t.busy += now_s() - a; return r; }
  template <class F> auto in(F&& f) { const double a = now_s(); auto r = f(); t.wait_in += now_s() - a; return r; }
  template <class F> auto out(F&& f) { const double a = now_s(); auto r = f(); t.wait_out += now_s() - a; return r; }
};

struct flag { uint32_t stream; uint64_t abs; float rad, snr, rssi; };      // a sync flag of a live stream, position counted from the stream's first item
struct stream_pos { uint64_t base_abs, fill; };                            // a live stream's region in a round's arena: the items [base_abs, base_abs + fill)

struct seg_events {
  int64_t seg = 0; uint64_t new_items = 0;
  const c8p_c32* raw_dev = nullptr; uint64_t n_raw = 0;
  std::vector<c8p_sig_in> in;
  bool det_raw_slot = false;                   // raw_dev is the detect context's copy of host items: its slot is held until c8p_signal is through
  // live rounds
  std::vector<flag> flags; std::vector<stream_pos> pos;
  double det_begin = 0, det_end = 0;
};
struct seg_packets {
  int64_t seg = 0; int n_events = 0; uint64_t new_items = 0;
  std::vector<c8p_pkt_in> pk;                  // the kept packets
  std::vector<uint32_t> pkt_stream; std::vector<uint64_t> pkt_index;
  const c8p_c32* iq = nullptr; uint64_t n_iq = 0; const c8p_c32* chan = nullptr; uint64_t n_chan = 0;
  c8p_rawchain_mark mark = {0, 0, 0, 0, 0, 0};
};

struct ctx_holder {
  const c8p_api& api; c8p_ctx* c = nullptr;
  ctx_holder(const c8p_api& a, int device) : api(a)
  {
    c8p_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.abi_version = C8P_ABI_VERSION; cfg.device = device;
    c = api.create(&cfg);
    if (!c) throw std::runtime_error("c8p_create failed: no usable HIP device (there is no CPU fallback)");
  }
  ~ctx_holder() { if (c) api.destroy(c); }
  void check(int rc, const char* what) const
  {
    if (rc != C8P_OK) throw std::runtime_error(std::string(what) + ": " + api.last_error(c) + " (" + std::to_string(rc) + ")");
  }
  int64_t cnt(int which) const { const int64_t v = api.counter(c, which); return v < 0 ? 0 : v; }
};

struct seg_record { double done; int64_t items, packets, pdus; };

// what the stages of one run (or live session) share
struct run_state {
  handover<seg_events> q1{1};
  handover<seg_packets> q2{1};
  counter sig_arenas{2};        // segments between "c8p_signal started" and "c8p_rx collected": their SIG_IQ / SIG_CHAN arenas are being read
  std::unique_ptr<counter> det_slots;      // host segments: segments between "c8p_detect started" and "c8p_signal through" (C8P_BUF_DET_RAW is being read)
  std::unique_ptr<counter> raw_arenas;     // live: arenas between "round closed" and "c8p_signal through"
  std::mutex err_mu;
  std::string first_err;
  c8p_rawchain_stats st;
  // one record and one mark per segment / round.  A run reserves them (n_seg is known); a LIVE session has no end, so it keeps a
  // bounded window: the last LIVE_HISTORY..2 LIVE_HISTORY rounds (80 B a round: unbounded, 0.25-1 GB a day and a whole-history copy
  // on the rx thread at every doubling).  The counters in `st` are running sums and do not depend on the window.
  std::vector<seg_record> rec;
  std::vector<c8p_rawchain_mark> marks;
  bool bounded = false;
  int64_t n_rec = 0;            // records ever made
  double first_done = 0;        // rec[0].done of the very first one
  void record(const c8p_rawchain_mark& m, const seg_record& r)
  {
    static const size_t LIVE_HISTORY = 4096;
    if (n_rec++ == 0) first_done = r.done;
    if (bounded && rec.size() >= 2 * LIVE_HISTORY) {
      rec.erase(rec.begin(), rec.begin() + (std::ptrdiff_t)LIVE_HISTORY);
      marks.erase(marks.begin(), marks.begin() + (std::ptrdiff_t)LIVE_HISTORY);
    }
    marks.push_back(m); rec.push_back(r);
  }
  double t0 = 0;
  c8p_rawchain_cb cb = nullptr; void* user = nullptr;
  int64_t c0[4] = {0, 0, 0, 0};           // the contexts' counters at the start: growths, giveups, ticket scans
  std::function<void()> also_close;       // live: the queue between the caller and the first stage
  run_state() { memset(&st, 0, sizeof(st)); }
  void fail(const std::string& m)
  {
    { std::lock_guard<std::mutex> g(err_mu); if (first_err.empty()) first_err = m; }
    if (also_close) also_close();
    q1.close(); q2.close(); sig_arenas.open();
    if (det_slots) det_slots->open();
    if (raw_arenas) raw_arenas->open();
  }
  template <class F> void stage(F&& body)
  {
    try { body(); }
    catch (const std::exception& ex) { fail(ex.what()); }
  }
};

} // namespace

struct live_session;

struct c8p_rawchain {
  c8p_api api;
  c8p_rawchain_cfg cfg;
  std::unique_ptr<ctx_holder> det, sig0, sig1, rx, ing;
  std::string err;
  std::vector<c8p_rawchain_mark> timeline;
  std::unique_ptr<live_session> live;
  c8p_rawchain_cb live_cb = nullptr; void* live_user = nullptr;
  c8p_c32* arena[LIVE_ARENAS] = {nullptr, nullptr, nullptr};
  uint64_t C = 0;                              // items of one stream's region in an arena
  uint64_t keep = 0;                           // live_keep() when the handle was made
  uint64_t A1 = 0;                             // two antennas per live stream: antenna 1's regions start A1 items into an arena (0: one antenna)
  ctx_holder* all[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t sum(int which) const { int64_t v = 0; for (ctx_holder* h : all) v += h->cnt(which); return v; }
};

namespace {

void start_counters(c8p_rawchain* H, run_state& R)
{
  R.c0[0] = H->sum(C8P_CNT_ARENA_GROWTHS); R.c0[1] = H->sum(C8P_CNT_DET_GIVEUPS); R.c0[2] = H->sum(C8P_CNT_DET_TICKET_SCANS);
  R.t0 = now_s();
}

void finish_stats(c8p_rawchain* H, run_state& R, const clocked& cd, const clocked& cs, const clocked& cr)
{
  c8p_rawchain_stats& st = R.st;
  st.seconds = now_s() - R.t0;
  st.detect = cd.t; st.signal = cs.t; st.rx = cr.t;
  st.arena_growths = H->sum(C8P_CNT_ARENA_GROWTHS) - R.c0[0];
  st.arena_bytes = H->sum(C8P_CNT_ARENA_BYTES);
  st.det_giveups = H->sum(C8P_CNT_DET_GIVEUPS) - R.c0[1];
  st.det_ticket_scans = H->sum(C8P_CNT_DET_TICKET_SCANS) - R.c0[2];
  const int64_t n = (int64_t)R.rec.size();
  if (R.n_rec > 0) st.first_segment_s = R.first_done;
  // steady state: everything but the pipeline's fill (the first segments' results wait for three stages, later ones for the slowest);
  // of a live session that outgrew its window: the window (its oldest record only lends its time stamp)
  const int64_t m = R.n_rec > n ? n - 1 : (n >= 8 ? n - 4 : (n >= 2 ? n - 1 : 0));
  if (m > 0) {
    st.steady_segments = m;
    st.steady_s = R.rec[(size_t)n - 1].done - R.rec[(size_t)(n - 1 - m)].done;
    for (int64_t i = n - m; i < n; i++) { st.steady_items += R.rec[(size_t)i].items; st.steady_packets += R.rec[(size_t)i].packets; st.steady_pdus += R.rec[(size_t)i].pdus; }
  }
  H->timeline = R.marks;
}

// the rx stage: c8p_rx of segment i + 1 is submitted before segment i is collected (one slot: collected at once)
void rx_stage(c8p_rawchain* H, run_state& R, clocked& ck)
{
  const c8p_api& api = H->api;
  ctx_holder& rx = *H->rx;
  const bool overlap = rx.cnt(C8P_CNT_SLOTS) >= 2;
  seg_packets cur, prev;
  int prev_ticket = 0;
  bool have_prev = false;
  std::vector<c8p_pkt_out> outs;
  auto collect = [&] {
    ck.busy([&] {
      rx.check(api.select(rx.c, prev_ticket), "c8p_select");
      outs.assign(prev.pk.size(), c8p_pkt_out());
      if (!outs.empty()) rx.check(api.fetch(rx.c, C8P_BUF_OUT, outs.data(), 0, sizeof(c8p_pkt_out) * outs.size()), "c8p_fetch(outs)");
      else rx.check(api.wait(rx.c, prev_ticket), "c8p_wait");
      return 0;
    });
    c8p_rawchain_stats& st = R.st;
    int64_t pdus = 0;
    for (const c8p_pkt_out& o : outs) pdus += o.n_pdu;
    st.segments++; st.items += (int64_t)prev.new_items; st.events += prev.n_events; st.packets += (int64_t)prev.pk.size(); st.pdus += pdus;
    if (R.cb) ck.out([&] {
      c8p_rawchain_out r;
      r.segment = prev.seg; r.n_events = prev.n_events; r.n_packets = (int32_t)prev.pk.size();
      r.pkts = prev.pk.data(); r.outs = outs.data(); r.rx = rx.c;
      r.pkt_stream = prev.pkt_stream.data(); r.pkt_index = prev.pkt_index.data();
      R.cb(R.user, &r);
      return 0;
    });
    prev.mark.rx_done = now_s() - R.t0;
    R.record(prev.mark, seg_record{prev.mark.rx_done, (int64_t)prev.new_items, (int64_t)prev.pk.size(), pdus});
    R.sig_arenas.release();
  };
  for (;;) {
    // With a segment in flight: whichever comes first, the next segment (submitted BEFORE the one in flight is collected, so that the
    // GPU is never idle) or the end of the one in flight (c8p_done, polled every 100 us) — a pipeline that is kept full sees the
    // former, a live one that runs at the radios' pace the latter: its results must not wait for the next round to come by (64 streams
    // x 26 ms rounds: PDUs 32 ms after the round was closed instead of 7)
    int got = have_prev ? 1 : ck.in([&] { return R.q2.get(cur) ? 0 : 2; });
    while (got == 1) {
      got = ck.in([&] { return R.q2.get_for(cur, 100); });
      if (got == 1) {
        const int d = api.done(rx.c, prev_ticket);
        if (d < 0) rx.check(d, "c8p_done");
        if (d == 1) break;
      }
    }
    if (got == 1) { collect(); have_prev = false; continue; }
    const bool more = got == 0;
    if (more) {
      cur.mark.rx_submit = now_s() - R.t0;
      ck.busy([&] {
        c8p_batch_in b;
        memset(&b, 0, sizeof(b));
        b.n_pkt = (int32_t)cur.pk.size(); b.mem = C8P_MEM_DEVICE; b.pkts = cur.pk.data();
        b.iq = cur.iq; b.n_iq = cur.n_iq; b.chan = cur.chan; b.n_chan = cur.n_chan;
        rx.check(api.rx(rx.c, &b), "c8p_rx");
        return 0;
      });
    }
    const int t = more ? api.ticket(rx.c) : 0;
    if (have_prev) collect();                                  // (segment i + 1 has been submitted: the GPU is not idle meanwhile)
    if (!more) break;
    prev = std::move(cur); prev_ticket = t; have_prev = true;
    if (!overlap) { collect(); have_prev = false; }
  }
}

// stream = true: segs[0] is the whole capture, cut into n_seg pieces of seg_items with the state carried across the cuts
int run_impl(c8p_rawchain* H, const c8p_rawchain_seg* segs, int64_t n_seg, bool stream, uint64_t seg_items,
             c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap)
{
  auto fail = [&](const std::string& m, int rc) { H->err = m; if (err && errcap > 0) snprintf(err, (size_t)errcap, "%s", m.c_str()); return rc; };
  if (stats) memset(stats, 0, sizeof(*stats));
  if (H->live) return fail("a live session is open on this handle (c8p_rawchain_drain ends it)", C8P_EINVAL);
  if (n_seg < 0 || (n_seg > 0 && !segs)) return fail("bad segment list", C8P_EINVAL);
  bool any_host = false;
  for (int64_t s = 0; s < (stream ? 1 : n_seg); s++) {
    if (segs[s].mem == C8P_MEM_HOST && (stream || segs[s].ant1_off)) return fail("host segments: c8p_rawchain_run, one antenna", C8P_EINVAL);
    any_host |= segs[s].mem == C8P_MEM_HOST;
  }
  try {
    const c8p_api& api = H->api;
    ctx_holder& det = *H->det;
    ctx_holder* sigs[2] = {H->sig0.get(), H->sig1.get()};
    run_state R;
    R.cb = cb; R.user = user;
    if (any_host) R.det_slots.reset(new counter((int)std::max<int64_t>(1, det.cnt(C8P_CNT_SLOTS))));
    R.marks.reserve((size_t)n_seg); R.rec.reserve((size_t)n_seg);
    clocked cd, cs, cr;
    start_counters(H, R);

    std::thread t_detect([&] { R.stage([&] {
      std::vector<c8p_sync_evt> ev;
      c8p_det_state state, state2;
      memset(&state, 0, sizeof(state));
      uint64_t commit = 0;                            // stream: first item the next call decides about
      for (int64_t s = 0; s < n_seg; s++) {
        const c8p_rawchain_seg& G = stream ? segs[0] : segs[s];
        c8p_detect_in d;
        memset(&d, 0, sizeof(d));
        uint64_t lo = 0, c2 = 0;                      // stream: the call sees items [lo, cut), the first n_hist of them are history
        d.mem = G.mem == C8P_MEM_HOST ? C8P_MEM_HOST : C8P_MEM_DEVICE; d.raw = G.raw; d.n = G.n;
        if (stream) {
          const uint64_t cut = std::min<uint64_t>(G.n, (uint64_t)(s + 1) * seg_items);
          lo = commit > C8P_DET_HIST ? commit - C8P_DET_HIST : 0;
          d.raw = G.raw + lo; d.n = cut - lo; d.n_hist = commit - lo;
          d.state = s ? &state : nullptr; d.state_out = &state2; d.commit = &c2;
        }
        seg_events o;
        o.det_raw_slot = d.mem == C8P_MEM_HOST;
        if (o.det_raw_slot) cd.out([&] { R.det_slots->acquire(); return 0; });
        o.det_begin = now_s() - R.t0;
        cd.busy([&] {
          // the events come back with the call (evt_host); only if there are more than the buffer holds are they fetched
          if (ev.size() < 4096) ev.resize(4096);
          ev.resize(ev.capacity());
          d.evt_host = ev.data(); d.evt_cap = ev.size();
          det.check(api.detect(det.c, &d), "c8p_detect");
          uint64_t bytes = 0;
          (void)api.buffer(det.c, C8P_BUF_DET_EVT, &bytes);
          const size_t ne = (size_t)(bytes / sizeof(c8p_sync_evt));
          if (ne > ev.size()) {
            ev.resize(ne + ne / 4);
            det.check(api.fetch(det.c, C8P_BUF_DET_EVT, ev.data(), 0, bytes), "c8p_fetch(events)");
          }
          ev.resize(ne);
          return 0;
        });
        o.det_end = now_s() - R.t0;
        uint64_t bytes = 0;
        o.seg = s; o.in.resize(ev.size());
        o.raw_dev = o.det_raw_slot ? (const c8p_c32*)api.buffer(det.c, C8P_BUF_DET_RAW, &bytes) : G.raw;     // (the engine's copy of host items, or the caller's pointer)
        o.n_raw = G.ant1_off ? G.ant1_off + G.n : G.n;
        o.new_items = stream ? (lo + c2) - commit : G.n;
        if (stream) { commit = lo + c2; state = state2; }
        for (size_t i = 0; i < ev.size(); i++) {
          c8p_sig_in& k = o.in[i];
          memset(&k, 0, sizeof(k));
          k.raw_off[0] = lo + ev[i].index;
          k.n_avail = (uint32_t)std::min<uint64_t>(G.n - k.raw_off[0], MAX_PKT_ITEMS);
          k.cfo_rad = ev[i].rad; k.snr = ev[i].snr; k.rssi = ev[i].rssi; k.n_ant = 1;
          if (G.ant1_off) { k.n_ant = 2; k.raw_off[1] = G.ant1_off + k.raw_off[0]; }
        }
        if (!cd.out([&] { return R.q1.put(std::move(o)); })) return;
      }
      R.q1.close();
    }); });

    std::thread t_signal([&] { R.stage([&] {
      seg_events in;
      std::vector<c8p_pkt_in> all;
      uint64_t scan = 0;                              // stream: where the block's S_TRIGGER scan stands
      for (int64_t k = 0; cs.in([&] { return R.q1.get(in); }); k++) {
        ctx_holder& sg = *sigs[k & 1];
        cs.out([&] { R.sig_arenas.acquire(); return 0; });
        seg_packets o;
        o.mark.det_begin = in.det_begin; o.mark.det_end = in.det_end; o.mark.sig_begin = now_s() - R.t0;
        all.assign(in.in.size(), c8p_pkt_in());
        cs.busy([&] {
          c8p_signal_in b;
          memset(&b, 0, sizeof(b));
          b.n_evt = (int32_t)in.in.size(); b.mem = C8P_MEM_DEVICE; b.evts = in.in.data();
          b.raw = in.raw_dev; b.n_raw = in.n_raw;
          b.consume = 1; b.pkt_host = all.data();
          if (stream) { b.scan_from = scan; b.scan_next = &scan; }
          sg.check(api.signal(sg.c, &b), "c8p_signal");
          sg.check(api.sync(sg.c), "c8p_sync");                     // c8p_rx reads these arenas on ITS streams
          return 0;
        });
        if (in.det_raw_slot) R.det_slots->release();                // the detect context's copy of the items has been read
        o.mark.sig_end = now_s() - R.t0;
        o.seg = in.seg; o.n_events = (int)in.in.size(); o.new_items = in.new_items;
        for (size_t i = 0; i < all.size(); i++)
          if (all[i].n_items) { o.pk.push_back(all[i]); o.pkt_stream.push_back(0u); o.pkt_index.push_back(in.in[i].raw_off[0]); }
        uint64_t bi = 0, bc = 0;
        o.iq = (const c8p_c32*)api.buffer(sg.c, C8P_BUF_SIG_IQ, &bi);
        o.chan = (const c8p_c32*)api.buffer(sg.c, C8P_BUF_SIG_CHAN, &bc);
        o.n_iq = bi / sizeof(c8p_c32); o.n_chan = bc / sizeof(c8p_c32);
        if (!cs.out([&] { return R.q2.put(std::move(o)); })) return;
      }
      R.q2.close();
    }); });

    std::thread t_rx([&] { R.stage([&] { rx_stage(H, R, cr); }); });

    t_detect.join(); t_signal.join(); t_rx.join();
    finish_stats(H, R, cd, cs, cr);
    if (stats) *stats = R.st;
    if (!R.first_err.empty()) return fail(R.first_err, C8P_EHIP);
    return C8P_OK;
  } catch (const std::exception& ex) {
    return fail(ex.what(), C8P_EHIP);
  }
}

} // namespace

// ------------------------------------------------------------------------------------------------ live ingest
struct live_round { int64_t r = 0; int arena = 0; std::vector<stream_pos> pos; uint64_t new_items = 0; uint64_t mark = 0; };

struct live_session {
  run_state R;
  handover<live_round> q0{LIVE_ARENAS};
  std::thread t_detect, t_signal, t_rx;
  clocked cd, cs, cr;
  // the caller's side
  int open_arena = 0;
  int64_t round_no = 0;
  std::vector<stream_pos> open;
  uint64_t open_new = 0;
  uint64_t last_mark = 0;                      // copy mark of the last round handed over
  double t_last_round = 0.0;                   // C8P_RAWCHAIN_TLOG
  std::vector<c8p_piece> pieces;               // the open round's pushes that a kernel can gather (device / pinned host memory)
};

namespace {

void live_detect(c8p_rawchain* H, live_session& L)
{
  const c8p_api& api = H->api;
  ctx_holder& det = *H->det;
  run_state& R = L.R;
  const uint32_t K = H->cfg.streams;
  const uint64_t C = H->C;
  std::vector<c8p_det_state> state(K);
  std::vector<char> started(K, 0);
  std::vector<uint64_t> commit(K, 0), lo(K, 0);
  std::vector<c8p_det_stream> ks(K);
  std::vector<c8p_sync_evt> ev;
  live_round rd;
  while (L.cd.in([&] { return L.q0.get(rd); })) {
    seg_events o;
    o.seg = rd.r; o.new_items = rd.new_items; o.raw_dev = H->arena[rd.arena]; o.n_raw = (uint64_t)K * C * (H->A1 ? 2 : 1); o.pos = rd.pos;
    o.det_begin = now_s() - R.t0;
    for (uint32_t s = 0; s < K; s++) {
      const stream_pos& P = rd.pos[s];
      c8p_det_stream& k = ks[s];
      memset(&k, 0, sizeof(k));
      // the call sees the stream's items [lo, end): the first commit - lo of them are history
      lo[s] = commit[s] > C8P_DET_HIST ? commit[s] - C8P_DET_HIST : 0;
      if (lo[s] < P.base_abs) {
        if (started[s]) throw std::runtime_error("live ingest: a stream's detection history is no longer in its arena");
        lo[s] = P.base_abs;
      }
      k.off = (uint64_t)s * C + (lo[s] - P.base_abs); k.n = P.base_abs + P.fill - lo[s]; k.n_hist = commit[s] - lo[s];
      k.has_state = started[s]; if (started[s]) k.state = state[s];
    }
    if (rd.mark) {                                 // c8p_rawchain_round_async: the round's items may still be on their way
      const int rcw = L.cd.in([&] { return api.copy_wait_mark(H->ing->c, rd.mark); });
      if (rcw != C8P_OK) throw std::runtime_error("c8p_copy_wait_mark failed (" + std::to_string(rcw) + ")");
    }
    L.cd.busy([&] {
      c8p_detect_multi_in d;
      memset(&d, 0, sizeof(d));
      d.mem = C8P_MEM_DEVICE; d.n_streams = (int32_t)K; d.raw = o.raw_dev; d.n_raw = (uint64_t)K * C; d.streams = ks.data();   // (antenna 0)
      if (ev.size() < 4096) ev.resize(4096);
      ev.resize(ev.capacity());
      d.evt_host = ev.data(); d.evt_cap = ev.size();
      det.check(api.detect_multi(det.c, &d), "c8p_detect_multi");
      uint64_t bytes = 0;
      (void)api.buffer(det.c, C8P_BUF_DET_EVT, &bytes);
      const size_t ne = (size_t)(bytes / sizeof(c8p_sync_evt));
      if (ne > ev.size()) {                            // more events than the buffer held: fetch them (and keep the larger buffer)
        ev.resize(ne + ne / 4);
        det.check(api.fetch(det.c, C8P_BUF_DET_EVT, ev.data(), 0, bytes), "c8p_fetch(events)");
      }
      ev.resize(ne);
      return 0;
    });
    size_t e = 0;
    for (uint32_t s = 0; s < K; s++) {
      const c8p_det_stream& k = ks[s];
      if (k.n > k.n_hist) { started[s] = 1; state[s] = k.state; commit[s] = lo[s] + k.commit; }
      for (uint32_t i = 0; i < k.n_evt && e < ev.size(); i++, e++)
        o.flags.push_back(flag{s, lo[s] + ev[e].index, ev[e].rad, ev[e].snr, ev[e].rssi});
    }
    if (e != ev.size()) throw std::runtime_error("c8p_detect_multi: event counts do not add up");
    o.det_end = now_s() - R.t0;
    if (!L.cd.out([&] { return R.q1.put(std::move(o)); })) return;
  }
  R.q1.close();
}

void live_signal(c8p_rawchain* H, live_session& L)
{
  const c8p_api& api = H->api;
  ctx_holder* sigs[2] = {H->sig0.get(), H->sig1.get()};
  run_state& R = L.R;
  const uint32_t K = H->cfg.streams;
  const uint64_t C = H->C;
  std::vector<std::vector<flag>> pend(K);             // flags that wait for more items (S_DEMOD / S_COPY), per stream
  std::vector<uint64_t> scan_abs(K, 0), scan_from(K), scan_next(K);
  std::vector<flag> fl;
  std::vector<c8p_sig_in> evts;
  std::vector<c8p_pkt_in> all;
  seg_events in;
  for (int64_t k = 0; L.cs.in([&] { return R.q1.get(in); }); k++) {
    ctx_holder& sg = *sigs[k & 1];
    L.cs.out([&] { R.sig_arenas.acquire(); return 0; });
    seg_packets o;
    o.mark.det_begin = in.det_begin; o.mark.det_end = in.det_end; o.mark.sig_begin = now_s() - R.t0;
    // every stream's waiting flags, then its new ones (those lie behind: the commit point has moved past the waiting ones)
    fl.clear();
    size_t e = 0;
    for (uint32_t s = 0; s < K; s++) {
      for (const flag& f : pend[s]) fl.push_back(f);
      pend[s].clear();
      for (; e < in.flags.size() && in.flags[e].stream == s; e++) fl.push_back(in.flags[e]);
    }
    evts.assign(fl.size(), c8p_sig_in());
    for (size_t i = 0; i < fl.size(); i++) {
      const flag& f = fl[i];
      const stream_pos& P = in.pos[f.stream];
      if (f.abs < P.base_abs) throw std::runtime_error("live ingest: a waiting flag is no longer in its stream's arena");
      c8p_sig_in& q = evts[i];
      memset(&q, 0, sizeof(q));
      q.raw_off[0] = (uint64_t)f.stream * C + (f.abs - P.base_abs);
      q.n_avail = (uint32_t)std::min<uint64_t>(P.base_abs + P.fill - f.abs, MAX_PKT_ITEMS);
      q.cfo_rad = f.rad; q.snr = f.snr; q.rssi = f.rssi; q.n_ant = 1; q.stream = f.stream;
      if (H->A1) { q.n_ant = 2; q.raw_off[1] = H->A1 + q.raw_off[0]; }
    }
    for (uint32_t s = 0; s < K; s++) {
      const stream_pos& P = in.pos[s];
      scan_from[s] = (uint64_t)s * C + (scan_abs[s] > P.base_abs ? scan_abs[s] - P.base_abs : 0);
      scan_next[s] = scan_from[s];
    }
    all.assign(fl.size(), c8p_pkt_in());
    L.cs.busy([&] {
      c8p_signal_in b;
      memset(&b, 0, sizeof(b));
      b.n_evt = (int32_t)evts.size(); b.mem = C8P_MEM_DEVICE; b.evts = evts.data();
      b.raw = in.raw_dev; b.n_raw = in.n_raw;
      b.consume = 1; b.n_streams = (int32_t)K; b.pkt_host = all.data();
      b.scan_from = scan_from[0]; b.scan_from_v = scan_from.data(); b.scan_next = scan_next.data();
      sg.check(api.signal(sg.c, &b), "c8p_signal");
      sg.check(api.sync(sg.c), "c8p_sync");                     // c8p_rx reads these arenas on ITS streams
      return 0;
    });
    R.raw_arenas->release();                                    // the round's stream arena has been read
    o.mark.sig_end = now_s() - R.t0;
    for (uint32_t s = 0; s < K; s++) {
      const stream_pos& P = in.pos[s];
      const uint64_t rel = scan_next[s] - (uint64_t)s * C;
      if (P.base_abs + rel > scan_abs[s]) scan_abs[s] = P.base_abs + rel;
    }
    o.seg = in.seg; o.n_events = (int)in.flags.size(); o.new_items = in.new_items;
    for (size_t i = 0; i < all.size(); i++) {
      const c8p_pkt_in& p = all[i];
      if (p.n_items) { o.pk.push_back(p); o.pkt_stream.push_back(fl[i].stream); o.pkt_index.push_back(fl[i].abs); }
      else if (p.rsv[0] == C8P_SIG_NEED_MORE || p.rsv[0] == C8P_SIG_PENDING) pend[fl[i].stream].push_back(fl[i]);
    }
    uint64_t bi = 0, bc = 0;
    o.iq = (const c8p_c32*)api.buffer(sg.c, C8P_BUF_SIG_IQ, &bi);
    o.chan = (const c8p_c32*)api.buffer(sg.c, C8P_BUF_SIG_CHAN, &bc);
    o.n_iq = bi / sizeof(c8p_c32); o.n_chan = bc / sizeof(c8p_c32);
    if (!L.cs.out([&] { return R.q2.put(std::move(o)); })) return;
  }
  R.q2.close();
}

int live_start(c8p_rawchain* H)
{
  if (H->live) return C8P_OK;
  if (!H->cfg.streams) { H->err = "this handle was created without live streams (c8p_rawchain_cfg.streams = 0)"; return C8P_EINVAL; }
  std::unique_ptr<live_session> Lp(new live_session());
  live_session& L = *Lp;
  L.R.cb = H->live_cb; L.R.user = H->live_user;
  L.R.raw_arenas.reset(new counter(LIVE_ARENAS - 1));
  L.R.bounded = true;                                          // no end in sight: the per-round history is a window
  L.open.assign(H->cfg.streams, stream_pos{0, 0});
  start_counters(H, L.R);
  live_session* lp = Lp.get();
  L.R.also_close = [lp] { lp->q0.close(); };      // a failing stage also ends the caller's hand-over (c8p_rawchain_round then reports the error)
  L.t_detect = std::thread([H, lp] { lp->R.stage([&] { live_detect(H, *lp); }); });
  L.t_signal = std::thread([H, lp] { lp->R.stage([&] { live_signal(H, *lp); }); });
  L.t_rx = std::thread([H, lp] { lp->R.stage([&] { rx_stage(H, lp->R, lp->cr); }); });
  H->live = std::move(Lp);
  return C8P_OK;
}

int live_failed(c8p_rawchain* H)
{
  live_session& L = *H->live;
  std::lock_guard<std::mutex> g(L.R.err_mu);
  if (L.R.first_err.empty()) return C8P_OK;
  H->err = L.R.first_err;
  return C8P_EHIP;
}

} // namespace

extern "C" int c8p_rawchain_live(c8p_rawchain* h, c8p_rawchain_cb cb, void* user)
{
  if (!h) return C8P_EINVAL;
  if (h->live) { h->err = "a live session is open (c8p_rawchain_drain ends it)"; return C8P_EINVAL; }
  h->live_cb = cb; h->live_user = user;
  return C8P_OK;
}

static int flush_pieces(c8p_rawchain* h)
{
  live_session& L = *h->live;
  if (L.pieces.empty()) return C8P_OK;
  const c8p_api& api = h->api;
  c8p_c32* dst = h->arena[L.open_arena];
  // Pieces the caller vouched for (C8P_MEM_PINNED / _DEVICE) and pieces of unknown host memory (C8P_MEM_HOST) go in separate gathers:
  // if the engine finds that the device cannot read the latter (C8P_ENOTDEV — the only refusal answered this way; any other error is
  // an argument error and propagates), THEY, and only they, go through the DMA path piece by piece.
  std::vector<c8p_piece> vouched, unknown;
  for (const c8p_piece& pc : L.pieces) (pc.trusted ? vouched : unknown).push_back(pc);
  int rc = vouched.empty() ? C8P_OK : api.copy_gather(h->ing->c, dst, vouched.data(), (int32_t)vouched.size());
  if (rc == C8P_OK && !unknown.empty()) {
    rc = api.copy_gather(h->ing->c, dst, unknown.data(), (int32_t)unknown.size());
    if (rc == C8P_ENOTDEV) {
      rc = C8P_OK;
      for (const c8p_piece& pc : unknown) {
        if (pc.fmt != C8P_FMT_C32) { h->err = "live ingest: sc16 / sc8 items must lie in device or pinned host memory (c8p_host_alloc)"; L.pieces.clear(); return C8P_EINVAL; }
        if ((rc = api.copy_async(h->ing->c, dst + pc.dst_off, pc.src, sizeof(c8p_c32) * pc.n)) != C8P_OK) break;
      }
    }
  }
  if (rc != C8P_OK) h->err = std::string("live ingest copy: ") + api.last_error(h->ing->c);
  L.pieces.clear();
  return rc;
}

static int push_impl(c8p_rawchain* h, uint32_t stream, const void* items, const void* ant1, uint64_t n, int32_t mem_and_fmt)
{
  if (!h) return C8P_EINVAL;
  const bool sc16 = (mem_and_fmt & C8P_MEM_SC16) != 0, sc8 = (mem_and_fmt & C8P_MEM_SC8) != 0;
  const int32_t mem = mem_and_fmt & ~(C8P_MEM_SC16 | C8P_MEM_SC8);
  if (sc16 && sc8) { h->err = "C8P_MEM_SC16 and C8P_MEM_SC8 together"; return C8P_EINVAL; }
  if (mem != C8P_MEM_HOST && mem != C8P_MEM_DEVICE && mem != C8P_MEM_PINNED) { h->err = "bad mem kind"; return C8P_EINVAL; }
  int rc = live_start(h);
  if (rc != C8P_OK) return rc;
  if ((rc = live_failed(h)) != C8P_OK) return rc;
  if (stream >= h->cfg.streams || (n && !items)) { h->err = "c8p_rawchain_push: no such stream / no items"; return C8P_EINVAL; }
  if ((h->A1 != 0) != (ant1 != nullptr || n == 0) && n) {
    h->err = h->A1 ? "this handle's streams have two antennas: c8p_rawchain_push2" : "this handle's streams have one antenna: c8p_rawchain_push";
    return C8P_EINVAL;
  }
  live_session& L = *h->live;
  stream_pos& P = L.open[stream];
  if (P.fill + n > h->C) { h->err = "c8p_rawchain_push: more than stream_round_items new items of one stream in a round"; return C8P_ERANGE; }
  if (n) {
    // the items move when the round is closed (or when 4096 pieces have gathered): all pieces with ONE kernel that reads the host memory
    // itself — no DMA engine, so the bulk does not queue in front of the stages' small descriptor and result copies.  A source the
    // device cannot read (pageable host memory) goes through c8p_copy_async when the pieces are flushed (flush_pieces)
    c8p_piece pc;
    pc.src = items; pc.dst_off = (uint64_t)stream * h->C + P.fill; pc.n = n; pc.trusted = mem != C8P_MEM_HOST; pc.rsv = 0;
    pc.fmt = sc16 ? C8P_FMT_SC16 : sc8 ? C8P_FMT_SC8 : C8P_FMT_C32;
    pc.scale = sc8 ? (h->cfg.sc8_scale != 0.0f ? h->cfg.sc8_scale : 1.0f / 127.0f) : (h->cfg.sc16_scale != 0.0f ? h->cfg.sc16_scale : 1.0f / 32767.0f);
    L.pieces.push_back(pc);
    if (ant1) { pc.src = ant1; pc.dst_off += h->A1; L.pieces.push_back(pc); }
    if (L.pieces.size() >= 4096 && (rc = flush_pieces(h)) != C8P_OK) return rc;
  }
  P.fill += n; L.open_new += n;
  return C8P_OK;
}
extern "C" int c8p_rawchain_push(c8p_rawchain* h, uint32_t stream, const void* items, uint64_t n, int32_t mem)
{
  return push_impl(h, stream, items, nullptr, n, mem);
}
extern "C" int c8p_rawchain_pushv(c8p_rawchain* h, int32_t m, const uint32_t* streams, const void* const* items, const uint64_t* n, int32_t mem)
{
  if (!h || m < 0 || (m > 0 && (!streams || !items || !n))) return C8P_EINVAL;
  for (int32_t i = 0; i < m; i++) {
    const int rc = push_impl(h, streams[i], items[i], nullptr, n[i], mem);
    if (rc != C8P_OK) return rc;
  }
  return C8P_OK;
}
extern "C" int c8p_rawchain_push2(c8p_rawchain* h, uint32_t stream, const void* ant0, const void* ant1, uint64_t n, int32_t mem)
{
  if (h && n && !ant1) { h->err = "c8p_rawchain_push2: no items for antenna 1"; return C8P_EINVAL; }
  return push_impl(h, stream, ant0, ant1, n, mem);
}

static int round_impl(c8p_rawchain* h, bool wait)
{
  if (!h) return C8P_EINVAL;
  int rc = live_start(h);
  if (rc != C8P_OK) return rc;
  if ((rc = live_failed(h)) != C8P_OK) return rc;
  live_session& L = *h->live;
  const c8p_api& api = h->api;
  live_round rd;
  // C8P_RAWCHAIN_TLOG=1: where the caller's thread spends a round (stderr, one line per round)
  static const bool tlog = getenv("C8P_RAWCHAIN_TLOG") != nullptr;
  const double t_in = tlog ? now_s() : 0.0;
  if ((rc = flush_pieces(h)) != C8P_OK) return rc;
  const double t_flush = tlog ? now_s() : 0.0;
  if (wait) {
    if ((rc = api.copy_wait(h->ing->c)) != C8P_OK) { h->err = std::string("c8p_copy_wait: ") + api.last_error(h->ing->c); return rc; }
  } else {
    rd.mark = api.copy_mark(h->ing->c);
    if (!rd.mark) { h->err = "c8p_copy_mark failed"; return C8P_EHIP; }
    L.last_mark = rd.mark;
  }
  rd.r = L.round_no++; rd.arena = L.open_arena; rd.pos = L.open; rd.new_items = L.open_new;
  if (!L.q0.put(std::move(rd))) { (void)live_failed(h); return C8P_EHIP; }
  const double t_put = tlog ? now_s() : 0.0;
  L.R.raw_arenas->acquire();                                   // the next arena: free once the round that used it is through c8p_signal
  const double t_acq = tlog ? now_s() : 0.0;
  if ((rc = live_failed(h)) != C8P_OK) return rc;
  // every stream's last `keep` (110 000) items go with it into the next round: the detection history behind its commit point and the
  // items of packets that have begun (a flag that waits for more items is at most MAX_PKT_ITEMS back)
  const int next = (L.open_arena + 1) % LIVE_ARENAS;
  const uint32_t K = h->cfg.streams;
  std::vector<c8p_span> sp((size_t)K * (h->A1 ? 2 : 1));
  for (uint32_t s = 0; s < K; s++) {
    stream_pos& P = L.open[s];
    const uint64_t T = std::min<uint64_t>(P.fill, h->keep);
    sp[s].dst_off = (uint64_t)s * h->C; sp[s].src_off = (uint64_t)s * h->C + P.fill - T; sp[s].n = T;
    if (h->A1) { sp[K + s] = sp[s]; sp[K + s].dst_off += h->A1; sp[K + s].src_off += h->A1; }
    P.base_abs += P.fill - T; P.fill = T;
  }
  rc = api.copy_spans(h->ing->c, h->arena[next], h->arena[L.open_arena], sp.data(), (int32_t)sp.size());
  if (rc != C8P_OK) { h->err = std::string("c8p_copy_spans: ") + api.last_error(h->ing->c); return rc; }
  L.open_arena = next; L.open_new = 0;
  if (tlog) {
    const double t_out = now_s();
    fprintf(stderr, "[c8p_rawchain] round %llu: since last %.0f us | gather call %.0f, mark+put %.0f, arena wait %.0f, carry call %.0f us\n",
            (unsigned long long)L.round_no - 1, 1e6 * (t_in - L.t_last_round), 1e6 * (t_flush - t_in), 1e6 * (t_put - t_flush), 1e6 * (t_acq - t_put),
            1e6 * (t_out - t_acq));
    L.t_last_round = t_out;
  }
  return C8P_OK;
}
extern "C" int c8p_rawchain_round(c8p_rawchain* h) { return round_impl(h, true); }
extern "C" int c8p_rawchain_round_async(c8p_rawchain* h) { return round_impl(h, false); }
extern "C" int c8p_rawchain_ingested(c8p_rawchain* h)
{
  if (!h) return C8P_EINVAL;
  if (!h->live || !h->live->last_mark) return C8P_OK;
  const int rc = h->api.copy_wait_mark(h->ing->c, h->live->last_mark);
  if (rc != C8P_OK) h->err = "c8p_copy_wait_mark failed";
  return rc;
}

extern "C" int c8p_rawchain_drain(c8p_rawchain* h, c8p_rawchain_stats* stats)
{
  if (!h) return C8P_EINVAL;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (!h->live) return C8P_OK;
  live_session& L = *h->live;
  L.q0.close();
  L.t_detect.join(); L.t_signal.join(); L.t_rx.join();
  (void)h->api.copy_wait(h->ing->c);
  finish_stats(h, L.R, L.cd, L.cs, L.cr);
  if (stats) *stats = L.R.st;
  const std::string e = L.R.first_err;
  h->live.reset();
  if (!e.empty()) { h->err = e; return C8P_EHIP; }
  return C8P_OK;
}

extern "C" const char* c8p_rawchain_error(c8p_rawchain* h) { return h ? h->err.c_str() : "null handle"; }

// ------------------------------------------------------------------------------------------------ handle
extern "C" c8p_rawchain* c8p_rawchain_create(const c8p_rawchain_cfg* cfg, char* err, int errcap)
{
  auto fail = [&](const std::string& m) { if (err && errcap > 0) snprintf(err, (size_t)errcap, "%s", m.c_str()); return (c8p_rawchain*)nullptr; };
  if (!cfg || cfg->struct_size != sizeof(c8p_rawchain_cfg)) return fail("c8p_rawchain_cfg.struct_size does not match this library");
  std::unique_ptr<c8p_rawchain> H(new c8p_rawchain());
  try {
    const char* e = getenv("C8P_RX_LIB");
    H->api = load_c8p_api(cfg->lib ? cfg->lib : (e ? e : "libc8p_rx.so"));
    H->cfg = *cfg;
    H->cfg.lib = nullptr;
    H->det.reset(new ctx_holder(H->api, cfg->device)); H->sig0.reset(new ctx_holder(H->api, cfg->device));
    H->sig1.reset(new ctx_holder(H->api, cfg->device)); H->rx.reset(new ctx_holder(H->api, cfg->device));
    H->all[0] = H->det.get(); H->all[1] = H->sig0.get(); H->all[2] = H->sig1.get(); H->all[3] = H->rx.get();
    const int ant = (cfg->reserve_ant == 2 || cfg->stream_ant == 2) ? 2 : 1;
    if (cfg->streams) {
      if (cfg->streams > 65535u) return fail("at most 65535 live streams");
      const uint64_t per = cfg->stream_round_items ? cfg->stream_round_items : (1u << 20);
      H->cfg.stream_round_items = per;
      H->keep = live_keep();
      H->C = (H->keep + per + 7) & ~(uint64_t)7;
      H->ing.reset(new ctx_holder(H->api, cfg->device));
      H->A1 = cfg->stream_ant == 2 ? H->C * cfg->streams : 0;
      for (int a = 0; a < LIVE_ARENAS; a++) {
        H->arena[a] = (c8p_c32*)H->api.dev_alloc(H->ing->c, sizeof(c8p_c32) * H->C * cfg->streams * (H->A1 ? 2 : 1));
        if (!H->arena[a]) throw std::runtime_error("c8p_dev_alloc: no memory for the live streams' arenas");
      }
    }
    // arenas of every context for the announced shape of a segment / round
    uint64_t items = cfg->reserve_seg_items;
    if (cfg->streams && !items) items = (uint64_t)cfg->streams * H->cfg.stream_round_items;
    if (items) {
      c8p_reserve_in r;
      memset(&r, 0, sizeof(r));
      r.det_items = items + (uint64_t)(cfg->streams ? cfg->streams : 1) * 512u;      // (history + the items behind a commit point that lags the end)
      r.det_host = cfg->reserve_host; r.det_streams = (int32_t)cfg->streams;
      r.det_events = (int32_t)(cfg->reserve_seg_events ? cfg->reserve_seg_events : cfg->reserve_seg_packets);
      H->det->check(H->api.reserve(H->det->c, &r), "c8p_reserve(detect)");
    }
    if (cfg->reserve_seg_packets) {
      const uint32_t ev = cfg->reserve_seg_events ? cfg->reserve_seg_events : cfg->reserve_seg_packets;
      c8p_reserve_in r;
      memset(&r, 0, sizeof(r));
      r.sig_events = (int32_t)ev;
      r.sig_items = (uint64_t)cfg->reserve_seg_packets * cfg->reserve_pkt_items * (uint64_t)ant;
      H->sig0->check(H->api.reserve(H->sig0->c, &r), "c8p_reserve(signal)");
      H->sig1->check(H->api.reserve(H->sig1->c, &r), "c8p_reserve(signal)");
      memset(&r, 0, sizeof(r));
      r.rx_packets = (int32_t)cfg->reserve_seg_packets; r.rx_ant = ant; r.rx_items = cfg->reserve_pkt_items;
      H->rx->check(H->api.reserve(H->rx->c, &r), "c8p_reserve(rx)");
    }
  } catch (const std::exception& ex) {
    if (H->ing) for (c8p_c32* a : H->arena) if (a) H->api.dev_free(H->ing->c, a);
    return fail(ex.what());
  }
  return H.release();
}

extern "C" void c8p_rawchain_destroy(c8p_rawchain* h)
{
  if (!h) return;
  if (h->live) (void)c8p_rawchain_drain(h, nullptr);
  if (h->ing) for (c8p_c32* a : h->arena) if (a) h->api.dev_free(h->ing->c, a);
  delete h;
}

extern "C" int c8p_rawchain_run(c8p_rawchain* h, const c8p_rawchain_seg* segs, int64_t n_seg, c8p_rawchain_cb cb, void* user,
                                c8p_rawchain_stats* stats, char* err, int errcap)
{
  if (!h) return C8P_EINVAL;
  return run_impl(h, segs, n_seg, false, 0, cb, user, stats, err, errcap);
}

extern "C" int c8p_rawchain_stream(c8p_rawchain* h, const c8p_c32* raw, uint64_t n, uint64_t ant1_off, uint64_t seg_items,
                                   c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap)
{
  if (!h) return C8P_EINVAL;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (!raw || !n || seg_items < 4 * C8P_DET_HIST) {
    h->err = "bad capture or segment length";
    if (err && errcap > 0) snprintf(err, (size_t)errcap, "%s", h->err.c_str());
    return C8P_EINVAL;
  }
  c8p_rawchain_seg whole;
  memset(&whole, 0, sizeof(whole));
  whole.raw = raw; whole.n = n; whole.ant1_off = ant1_off; whole.mem = C8P_MEM_DEVICE;
  return run_impl(h, &whole, (int64_t)((n + seg_items - 1) / seg_items), true, seg_items, cb, user, stats, err, errcap);
}

extern "C" int64_t c8p_rawchain_timeline(c8p_rawchain* h, c8p_rawchain_mark* dst, int64_t cap)
{
  if (!h) return C8P_EINVAL;
  const int64_t n = (int64_t)h->timeline.size();
  for (int64_t i = 0; i < n && i < cap && dst; i++) dst[i] = h->timeline[(size_t)i];
  return n;
}

static c8p_rawchain* once(const char* lib, int device, char* err, int errcap)
{
  c8p_rawchain_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.struct_size = sizeof(cfg); cfg.device = device; cfg.lib = lib;
  return c8p_rawchain_create(&cfg, err, errcap);
}
extern "C" int c8p_rawchain_run_once(const char* lib, int device, const c8p_rawchain_seg* segs, int64_t n_seg, c8p_rawchain_cb cb, void* user,
                                     c8p_rawchain_stats* stats, char* err, int errcap)
{
  if (stats) memset(stats, 0, sizeof(*stats));
  c8p_rawchain* h = once(lib, device, err, errcap);
  if (!h) return C8P_ENODEV;
  const int rc = c8p_rawchain_run(h, segs, n_seg, cb, user, stats, err, errcap);
  c8p_rawchain_destroy(h);
  return rc;
}
extern "C" int c8p_rawchain_stream_once(const char* lib, int device, const c8p_c32* raw, uint64_t n, uint64_t ant1_off, uint64_t seg_items,
                                        c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap)
{
  if (stats) memset(stats, 0, sizeof(*stats));
  c8p_rawchain* h = once(lib, device, err, errcap);
  if (!h) return C8P_ENODEV;
  const int rc = c8p_rawchain_stream(h, raw, n, ant1_off, seg_items, cb, user, stats, err, errcap);
  c8p_rawchain_destroy(h);
  return rc;
}
