// c8p_rawchain.cc — see c8p_rawchain.h.  Three host threads around three (four) engine contexts; bounded hand-over queues.
#include "c8p_rawchain.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "c8p_blocks.h"

using namespace C8P_MIRROR_QUAL;

namespace {

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

struct seg_events { int64_t seg; uint64_t new_items; const c8p_c32* raw_dev; std::vector<c8p_sig_in> in; };
struct seg_packets {
  int64_t seg; int n_events; uint64_t new_items;
  std::vector<c8p_pkt_in> pk;                  // the kept packets
  const c8p_c32* iq; uint64_t n_iq; const c8p_c32* chan; uint64_t n_chan;
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
};

} // namespace

// stream = true: segs[0] is the whole capture, cut into n_seg pieces of seg_items with the state carried across the cuts
static int run_impl(const char* lib, int device, const c8p_rawchain_seg* segs, int64_t n_seg, bool stream, uint64_t seg_items,
                    c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap)
{
  auto fail = [&](const std::string& m, int rc) { if (err && errcap > 0) snprintf(err, (size_t)errcap, "%s", m.c_str()); return rc; };
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_seg < 0 || (n_seg > 0 && !segs)) return fail("bad segment list", C8P_EINVAL);
  for (int64_t s = 0; s < (stream ? 1 : n_seg); s++)
    if (segs[s].mem == C8P_MEM_HOST && (stream || segs[s].ant1_off)) return fail("host segments: c8p_rawchain_run, one antenna", C8P_EINVAL);
  try {
    const char* e = getenv("C8P_RX_LIB");
    const c8p_api api = load_c8p_api(lib ? lib : (e ? e : "libc8p_rx.so"));
    ctx_holder det(api, device), sig0(api, device), sig1(api, device), rx(api, device);
    ctx_holder* sigs[2] = {&sig0, &sig1};
    handover<seg_events> q1(1);
    handover<seg_packets> q2(1);
    counter sig_arenas(2);        // segments between "c8p_signal started" and "c8p_rx collected": their SIG_IQ / SIG_CHAN arenas are being read
    std::mutex err_mu;
    std::string first_err;
    auto stage = [&](auto&& body) {
      try { body(); }
      catch (const std::exception& ex) {
        { std::lock_guard<std::mutex> g(err_mu); if (first_err.empty()) first_err = ex.what(); }
        q1.close(); q2.close(); sig_arenas.open();
      }
    };
    c8p_rawchain_stats st;
    memset(&st, 0, sizeof(st));
    const auto t0 = std::chrono::steady_clock::now();

    std::thread t_detect([&] { stage([&] {
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
        det.check(api.detect(det.c, &d), "c8p_detect");
        uint64_t bytes = 0;
        (void)api.buffer(det.c, C8P_BUF_DET_EVT, &bytes);
        ev.resize((size_t)(bytes / sizeof(c8p_sync_evt)));
        if (bytes) det.check(api.fetch(det.c, C8P_BUF_DET_EVT, ev.data(), 0, bytes), "c8p_fetch(events)");
        seg_events o;
        o.seg = s; o.in.resize(ev.size());
        o.raw_dev = stream ? G.raw : (const c8p_c32*)api.buffer(det.c, C8P_BUF_DET_RAW, &bytes);     // (the caller's pointer, or the engine's copy of host items)
        o.new_items = stream ? (lo + c2) - commit : G.n;
        if (stream) { commit = lo + c2; state = state2; }
        for (size_t i = 0; i < ev.size(); i++) {
          c8p_sig_in& k = o.in[i];
          memset(&k, 0, sizeof(k));
          k.raw_off[0] = lo + ev[i].index;
          k.n_avail = (uint32_t)std::min<uint64_t>(G.n - k.raw_off[0], 224u + 109280u);       // the longest packet: 4095 bytes at 6 Mb/s
          k.cfo_rad = ev[i].rad; k.snr = ev[i].snr; k.rssi = ev[i].rssi; k.n_ant = 1;
          if (G.ant1_off) { k.n_ant = 2; k.raw_off[1] = G.ant1_off + k.raw_off[0]; }
        }
        if (!q1.put(std::move(o))) return;
      }
      q1.close();
    }); });

    std::thread t_signal([&] { stage([&] {
      seg_events in;
      std::vector<c8p_pkt_in> all;
      uint64_t scan = 0;                              // stream: where the block's S_TRIGGER scan stands
      for (int64_t k = 0; q1.get(in); k++) {
        ctx_holder& sg = *sigs[k & 1];
        const c8p_rawchain_seg& G = stream ? segs[0] : segs[in.seg];
        sig_arenas.acquire();
        all.assign(in.in.size(), c8p_pkt_in());
        c8p_signal_in b;
        memset(&b, 0, sizeof(b));
        b.n_evt = (int32_t)in.in.size(); b.mem = C8P_MEM_DEVICE; b.evts = in.in.data();
        b.raw = in.raw_dev; b.n_raw = G.ant1_off ? G.ant1_off + G.n : G.n;
        b.consume = 1; b.pkt_host = all.data();
        if (stream) { b.scan_from = scan; b.scan_next = &scan; }
        sg.check(api.signal(sg.c, &b), "c8p_signal");
        sg.check(api.sync(sg.c), "c8p_sync");                     // c8p_rx reads these arenas on ITS streams
        seg_packets o;
        o.seg = in.seg; o.n_events = (int)in.in.size(); o.new_items = in.new_items;
        for (const c8p_pkt_in& p : all) if (p.n_items) o.pk.push_back(p);
        uint64_t bi = 0, bc = 0;
        o.iq = (const c8p_c32*)api.buffer(sg.c, C8P_BUF_SIG_IQ, &bi);
        o.chan = (const c8p_c32*)api.buffer(sg.c, C8P_BUF_SIG_CHAN, &bc);
        o.n_iq = bi / sizeof(c8p_c32); o.n_chan = bc / sizeof(c8p_c32);
        if (!q2.put(std::move(o))) return;
      }
      q2.close();
    }); });

    std::thread t_rx([&] { stage([&] {
      seg_packets cur, prev;
      int prev_ticket = 0;
      bool have_prev = false;
      std::vector<c8p_pkt_out> outs;
      auto collect = [&] {
        rx.check(api.select(rx.c, prev_ticket), "c8p_select");
        outs.assign(prev.pk.size(), c8p_pkt_out());
        if (!outs.empty()) rx.check(api.fetch(rx.c, C8P_BUF_OUT, outs.data(), 0, sizeof(c8p_pkt_out) * outs.size()), "c8p_fetch(outs)");
        st.segments++; st.items += (int64_t)prev.new_items; st.events += prev.n_events; st.packets += (int64_t)prev.pk.size();
        for (const c8p_pkt_out& o : outs) st.pdus += o.n_pdu;
        if (cb) {
          c8p_rawchain_out r;
          r.segment = prev.seg; r.n_events = prev.n_events; r.n_packets = (int32_t)prev.pk.size();
          r.pkts = prev.pk.data(); r.outs = outs.data(); r.rx = rx.c;
          cb(user, &r);
        }
        sig_arenas.release();
      };
      for (;;) {
        const bool more = q2.get(cur);
        if (more) {
          c8p_batch_in b;
          memset(&b, 0, sizeof(b));
          b.n_pkt = (int32_t)cur.pk.size(); b.mem = C8P_MEM_DEVICE; b.pkts = cur.pk.data();
          b.iq = cur.iq; b.n_iq = cur.n_iq; b.chan = cur.chan; b.n_chan = cur.n_chan;
          rx.check(api.rx(rx.c, &b), "c8p_rx");
        }
        const int t = more ? api.ticket(rx.c) : 0;
        if (have_prev) collect();                                  // (segment i + 1 has been submitted: the GPU is not idle meanwhile)
        if (!more) break;
        prev = std::move(cur); prev_ticket = t; have_prev = true;
      }
    }); });

    t_detect.join(); t_signal.join(); t_rx.join();
    st.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    if (!first_err.empty()) return fail(first_err, C8P_EHIP);
    return C8P_OK;
  } catch (const std::exception& ex) {
    return fail(ex.what(), C8P_EHIP);
  }
}

extern "C" int c8p_rawchain_run(const char* lib, int device, const c8p_rawchain_seg* segs, int64_t n_seg, c8p_rawchain_cb cb, void* user,
                                c8p_rawchain_stats* stats, char* err, int errcap)
{
  return run_impl(lib, device, segs, n_seg, false, 0, cb, user, stats, err, errcap);
}

extern "C" int c8p_rawchain_stream(const char* lib, int device, const c8p_c32* raw, uint64_t n, uint64_t ant1_off, uint64_t seg_items,
                                   c8p_rawchain_cb cb, void* user, c8p_rawchain_stats* stats, char* err, int errcap)
{
  if (stats) memset(stats, 0, sizeof(*stats));
  if (!raw || !n || seg_items < 4 * C8P_DET_HIST) {
    if (err && errcap > 0) snprintf(err, (size_t)errcap, "bad capture or segment length");
    return C8P_EINVAL;
  }
  c8p_rawchain_seg whole;
  memset(&whole, 0, sizeof(whole));
  whole.raw = raw; whole.n = n; whole.ant1_off = ant1_off; whole.mem = C8P_MEM_DEVICE;
  return run_impl(lib, device, &whole, (int64_t)((n + seg_items - 1) / seg_items), true, seg_items, cb, user, stats, err, errcap);
}
