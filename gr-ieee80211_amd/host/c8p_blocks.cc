// c8p_blocks.cc — batching fronts that give the C ABI the reference's block interface (see c8p_blocks.h).
//
// Stream contract kept from the reference (lib/demod_impl.cc, lib/demod2_impl.cc, lib/decode_impl.cc):
//   * a packet starts at an item that carries tags (read with get_tags_in_range on port 0 at the current read
//     position, demod_impl.cc:75 / decode_impl.cc:70) and spans nsamp+320 items per port (demod) or `total`
//     items (decode);
//   * demod emits, per accepted packet, `total` float items with the tags of demod_impl.cc:224-253 on the first
//     one (an NDP: 1024 zero items + "mu2x1chan"); rejected packets emit nothing;
//   * decode publishes pmt::cons(dict{"len"}, blob) on port "out" per FCS-correct (sub)frame and prints the
//     reference's debug lines when ifdebug (decode_impl.cc:377-411, 456-509 — scraped by tools/performance).
// What differs is *when*: packets are processed a batch at a time, so outputs trail inputs by up to one batch.
#include "c8p_blocks.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace gr {
namespace ieee80211 {

c8p_api load_c8p_api(const std::string& so_path)
{
  void* h = dlopen(so_path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) throw std::runtime_error(std::string("cannot load ") + so_path + ": " + dlerror() + " (there is no CPU fallback)");
  c8p_api a;
  auto sym = [&](const char* n) {
    void* p = dlsym(h, n);
    if (!p) throw std::runtime_error(std::string("missing symbol ") + n + " in " + so_path);
    return p;
  };
  a.create = (c8p_ctx * (*)(const c8p_cfg*)) sym("c8p_create");
  a.destroy = (void (*)(c8p_ctx*))sym("c8p_destroy");
  a.last_error = (const char* (*)(const c8p_ctx*))sym("c8p_last_error");
  a.demod = (int (*)(c8p_ctx*, const c8p_batch_in*))sym("c8p_demod");
  a.decode = (int (*)(c8p_ctx*, const c8p_decode_in*))sym("c8p_decode");
  a.signal = (int (*)(c8p_ctx*, const c8p_signal_in*))sym("c8p_signal");
  a.detect = (int (*)(c8p_ctx*, const c8p_detect_in*))sym("c8p_detect");
  a.sync = (int (*)(c8p_ctx*))sym("c8p_sync");
  a.buffer = (const void* (*)(c8p_ctx*, int, uint64_t*))sym("c8p_buffer");
  a.fetch = (int (*)(c8p_ctx*, int, void*, uint64_t, uint64_t))sym("c8p_fetch");
  return a;
}

static c8p_api default_api()
{
  const char* e = getenv("C8P_RX_LIB");
  return load_c8p_api(e ? e : "libc8p_rx.so");
}

static const tag_value* find_tag(const std::vector<tag_t>& tags, const char* key)
{
  for (const auto& t : tags) if (t.key == key) return &t.value;
  return nullptr;
}
static long tag_long(const std::vector<tag_t>& tags, const char* key, long dflt)
{
  const tag_value* v = find_tag(tags, key);
  return v ? (v->kind == tag_value::FLOAT ? (long)v->f : v->l) : dflt;
}
static float tag_float(const std::vector<tag_t>& tags, const char* key, float dflt)
{
  const tag_value* v = find_tag(tags, key);
  return v ? (v->kind == tag_value::LONG ? (float)v->l : v->f) : dflt;
}

class engine_holder
{
public:
  engine_holder(const c8p_api& api, const batching_cfg& cfg) : d_api(api), d_cfg(cfg)
  {
    c8p_cfg c;
    memset(&c, 0, sizeof(c));
    c.abi_version = C8P_ABI_VERSION; c.device = cfg.device;
    d_ctx = d_api.create(&c);
    if (!d_ctx) throw std::runtime_error("c8p_create failed: no usable HIP device (there is no CPU fallback)");
  }
  ~engine_holder() { if (d_ctx) d_api.destroy(d_ctx); }
  void check(int rc)
  {
    if (rc != C8P_OK) throw std::runtime_error(std::string("c8p error: ") + d_api.last_error(d_ctx));
  }
  template <typename T> std::vector<T> fetch_all(int which)
  {
    uint64_t bytes = 0;
    d_api.buffer(d_ctx, which, &bytes);
    std::vector<T> v(bytes / sizeof(T));
    if (bytes) check(d_api.fetch(d_ctx, which, v.data(), 0, v.size() * sizeof(T)));
    return v;
  }
  c8p_api d_api;
  batching_cfg d_cfg;
  c8p_ctx* d_ctx = nullptr;
};

// ================================================================================================ demod / demod2
class demod_front : public demod, public demod2
{
public:
  demod_front(int nports, int mupos, int mugid, const c8p_api& api, const batching_cfg& cfg)
      : d_eng(api, cfg), d_nports(nports), d_mupos(mupos), d_mugid(mugid)
  {
  }

  void forecast(int noutput_items, gr_vector_int& req) override   // demod_impl.cc:53-57
  {
    for (auto& r : req) r = noutput_items;
  }

  int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& output_items) override
  {
    int nproc = ninput_items[0];
    for (int p = 1; p < d_nports; p++) nproc = std::min(nproc, ninput_items[p]);
    int used = 0;
    while (used < nproc) {
      if (d_state == S_RDTAG) {
        std::vector<tag_t> tags;
        const uint64_t pos = d_io->nitems_read(0) + used;
        d_io->get_tags_in_range(tags, 0, pos, pos + 1);
        if (tags.empty()) { used++; continue; }     // the reference would wait here forever; we resynchronise
        memset(&d_cur, 0, sizeof(d_cur));
        d_cur.cfo = tag_float(tags, "cfo", 0.f); d_cur.snr = tag_float(tags, "snr", 0.f); d_cur.rssi = tag_float(tags, "rssi", 0.f);
        d_cur.seq = (int32_t)tag_long(tags, "seq", -1);
        d_cur.lsig_mcs = (int32_t)tag_long(tags, "mcs", -1);
        d_cur.lsig_len = (int32_t)tag_long(tags, "len", -1);
        d_cur.lsig_nsamp = (int32_t)tag_long(tags, "nsamp", -1);
        d_cur.n_ant = (uint8_t)d_nports; d_cur.mupos = (uint8_t)(d_mupos ? 1 : 0);
        d_cur.n_items = (uint32_t)std::max(d_cur.lsig_nsamp, 0) + 320u;     // demod_impl.cc:92
        d_cur.chan_off = (uint32_t)d_chan.size();
        const tag_value* ch = find_tag(tags, "chan");
        for (int i = 0; i < 64; i++) d_chan.push_back(ch && (int)ch->v.size() > i ? ch->v[i] : gr_complex(0.f, 0.f));
        for (int p = 0; p < d_nports; p++) {
          d_cur.iq_off[p] = d_iq.size();
          d_iq.resize(d_iq.size() + d_cur.n_items);
        }
        d_have = 0;
        d_state = S_COLLECT;
      }
      const int take = (int)std::min<int64_t>(nproc - used, (int64_t)d_cur.n_items - d_have);
      for (int p = 0; p < d_nports; p++)
        memcpy(&d_iq[d_cur.iq_off[p] + d_have], static_cast<const gr_complex*>(input_items[p]) + used, sizeof(gr_complex) * (size_t)take);
      d_have += take; used += take;
      if (d_have == (int)d_cur.n_items) {
        d_pk.push_back(d_cur);
        d_state = S_RDTAG;
        if ((int)d_pk.size() >= d_eng.d_cfg.max_batch) submit();
      }
    }
    if (d_eng.d_cfg.flush_when_idle && d_state == S_RDTAG && !d_pk.empty()) submit();
    d_io->consume_each(used);
    return drain(static_cast<float*>(output_items[0]), noutput_items);
  }

  void flush() override
  {
    if (!d_pk.empty()) submit();
  }

private:
  struct out_pkt {
    std::vector<std::pair<std::string, tag_value>> tags;
    std::vector<float> llr;
    size_t sent = 0;
  };

  void submit()
  {
    c8p_batch_in b;
    memset(&b, 0, sizeof(b));
    b.n_pkt = (int32_t)d_pk.size(); b.mem = C8P_MEM_HOST; b.pkts = d_pk.data();
    b.iq = reinterpret_cast<const c8p_c32*>(d_iq.data()); b.n_iq = d_iq.size();
    b.chan = reinterpret_cast<const c8p_c32*>(d_chan.data()); b.n_chan = d_chan.size();
    d_eng.check(d_eng.d_api.demod(d_eng.d_ctx, &b));
    d_eng.check(d_eng.d_api.sync(d_eng.d_ctx));
    std::vector<c8p_pkt_tags> tags = d_eng.fetch_all<c8p_pkt_tags>(C8P_BUF_TAGS);
    std::vector<float> llr = d_eng.fetch_all<float>(C8P_BUF_LLR);
    std::vector<gr_complex> mu = d_eng.fetch_all<gr_complex>(C8P_BUF_MUCHAN);
    for (size_t p = 0; p < d_pk.size(); p++) {
      const c8p_pkt_tags& t = tags[p];
      if (t.status != C8P_ST_OK) continue;          // DEMOD_S_CLEAN without tags: nothing reaches decode
      out_pkt o;
      auto L = [&](const char* k, long v) { o.tags.push_back({k, tag_value::from_long(v)}); };
      auto F = [&](const char* k, float v) { o.tags.push_back({k, tag_value::from_float(v)}); };
      F("cfo", t.cfo); F("snr", t.snr);
      if (t.format == C8P_F_VHT) { F("sssnr0", t.sssnr0); if (d_nports == 2 && t.nss == 2) F("sssnr1", t.sssnr1); }
      F("rssi", t.rssi);
      L("format", t.format); L("mcs", t.mcs); L("len", t.len); L("cr", t.cr); L("ampdu", t.ampdu); L("trellis", t.trellis);
      if (t.is_ndp) {
        std::vector<gr_complex> c(mu.begin() + 128 * p, mu.begin() + 128 * p + 128);
        o.tags.push_back({"mu2x1chan", tag_value::from_c32vector(c)});
      }
      L("total", t.total);
      o.llr.assign(llr.begin() + t.llr_off, llr.begin() + t.llr_off + t.total);
      d_outq.push_back(std::move(o));
    }
    d_pk.clear(); d_iq.clear(); d_chan.clear();
  }

  int drain(float* out, int noutput_items)
  {
    int produced = 0;
    while (!d_outq.empty() && produced < noutput_items) {
      out_pkt& o = d_outq.front();
      if (o.sent == 0)
        for (auto& kv : o.tags) d_io->add_item_tag(0, d_io->nitems_written(0) + produced, kv.first, kv.second);
      const size_t k = std::min(o.llr.size() - o.sent, (size_t)(noutput_items - produced));
      memcpy(out + produced, o.llr.data() + o.sent, sizeof(float) * k);
      o.sent += k; produced += (int)k;
      if (o.sent == o.llr.size()) d_outq.pop_front();
    }
    return produced;
  }

  enum { S_RDTAG, S_COLLECT } d_state = S_RDTAG;
  engine_holder d_eng;
  int d_nports, d_mupos, d_mugid;
  c8p_pkt_in d_cur;
  int d_have = 0;
  std::vector<c8p_pkt_in> d_pk;
  std::vector<gr_complex> d_iq, d_chan;
  std::deque<out_pkt> d_outq;
};

demod::sptr demod::make(int mupos, int mugid) { return make(mupos, mugid, default_api(), batching_cfg()); }
demod::sptr demod::make(int mupos, int mugid, const c8p_api& api, const batching_cfg& cfg)
{
  return std::shared_ptr<demod>(new demod_front(1, mupos, mugid, api, cfg));
}
demod2::sptr demod2::make() { return make(default_api(), batching_cfg()); }
demod2::sptr demod2::make(const c8p_api& api, const batching_cfg& cfg)
{
  return std::shared_ptr<demod2>(new demod_front(2, 0, 0, api, cfg));
}

// ================================================================================================ signal / signal2
// Streaming front of c8p_signal().  The reference block walks S_TRIGGER -> S_DEMOD -> S_COPY -> S_PAD per packet
// (signal_impl.cc:76-201); which sync flags it honours depends on the L-SIG of the previous ones (flags inside a
// packet that is being copied are never looked at; a failed L-SIG check skips 80 items).  This front buffers the raw
// stream, hands every pending flag to the engine in one batch, and then applies that scan rule to the results in order.
class signal_front : public signal, public signal2
{
public:
  signal_front(int nant, const c8p_api& api, const batching_cfg& cfg) : d_eng(api, cfg), d_nant(nant), d_raw(nant) {}

  void forecast(int noutput_items, gr_vector_int& req) override   // signal_impl.cc:57-62
  {
    for (auto& r : req) r = noutput_items;
  }

  int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& output_items) override
  {
    int nproc = ninput_items[0];
    for (int p = 1; p <= d_nant; p++) nproc = std::min(nproc, ninput_items[p]);
    const uint8_t* sync = static_cast<const uint8_t*>(input_items[0]);
    const uint64_t rd = d_io->nitems_read(0);
    for (int i = 0; i < nproc; i++) {
      if (!sync[i]) continue;
      std::vector<tag_t> tags;
      d_io->get_tags_in_range(tags, 0, rd + i, rd + i + 1);
      if (tags.empty()) continue;                      // "input sync with no tag": the reference skips the item
      evt e;
      e.idx = rd + i; e.rad = tag_float(tags, "rad", 0.f); e.snr = tag_float(tags, "snr", 0.f); e.rssi = tag_float(tags, "rssi", 0.f);
      d_evts.push_back(e);
    }
    for (int a = 0; a < d_nant; a++) {
      const gr_complex* x = static_cast<const gr_complex*>(input_items[1 + a]);
      d_raw[a].insert(d_raw[a].end(), x, x + nproc);
    }
    d_io->consume_each(nproc);
    const bool dry = nproc == 0;
    if (!d_evts.empty() && avail_end() != d_last_end &&
        ((int)d_evts.size() >= d_eng.d_cfg.max_batch || (d_eng.d_cfg.flush_when_idle && dry))) {
      d_last_end = avail_end();                      // nothing new to learn until more items arrive
      process();
    }
    trim();
    return drain(output_items, noutput_items);
  }

  void flush() override { if (!d_evts.empty()) process(); trim(); }

private:
  struct evt { uint64_t idx; float rad, snr, rssi; };
  struct out_pkt {
    std::vector<std::pair<std::string, tag_value>> tags;
    std::vector<gr_complex> items[2];
    size_t sent = 0;
  };

  uint64_t avail_end() const { return d_base + d_raw[0].size(); }

  void process()
  {
    // every pending flag with at least 224 items behind it, against the items buffered so far
    std::vector<c8p_sig_in> in;
    std::vector<size_t> which;
    std::vector<gr_complex> arena;
    const uint64_t end = avail_end();
    for (int a = 0; a < d_nant; a++) arena.insert(arena.end(), d_raw[a].begin(), d_raw[a].end());
    const uint64_t per = d_raw[0].size();
    for (size_t k = 0; k < d_evts.size(); k++) {
      const evt& e = d_evts[k];
      if (e.idx < d_scan || e.idx < d_base || e.idx + 224 > end) continue;
      c8p_sig_in s;
      memset(&s, 0, sizeof(s));
      s.raw_off[0] = e.idx - d_base; s.raw_off[1] = d_nant == 2 ? per + (e.idx - d_base) : 0;
      s.n_avail = (uint32_t)std::min<uint64_t>(end - e.idx, 0xffffffffull);
      s.cfo_rad = e.rad; s.snr = e.snr; s.rssi = e.rssi; s.n_ant = (uint8_t)d_nant;
      in.push_back(s); which.push_back(k);
    }
    std::vector<c8p_pkt_in> pk;
    std::vector<gr_complex> iq, chan;
    if (!in.empty()) {
      c8p_signal_in b;
      memset(&b, 0, sizeof(b));
      b.n_evt = (int32_t)in.size(); b.mem = C8P_MEM_HOST; b.evts = in.data();
      b.raw = reinterpret_cast<const c8p_c32*>(arena.data()); b.n_raw = arena.size();
      d_eng.check(d_eng.d_api.signal(d_eng.d_ctx, &b));
      d_eng.check(d_eng.d_api.sync(d_eng.d_ctx));
      pk = d_eng.fetch_all<c8p_pkt_in>(C8P_BUF_SIG_PKT);
      iq = d_eng.fetch_all<gr_complex>(C8P_BUF_SIG_IQ);
      chan = d_eng.fetch_all<gr_complex>(C8P_BUF_SIG_CHAN);
    }
    // signal_impl's scan, in stream order
    size_t done = 0, r = 0;
    for (size_t k = 0; k < d_evts.size(); k++, done = k) {
      const evt& e = d_evts[k];
      const bool submitted = r < which.size() && which[r] == k;
      const size_t ri = r;
      if (submitted) r++;
      if (e.idx < d_scan || e.idx < d_base) continue;                     // inside a packet already copied: never seen
      if (!submitted) { done = k; break; }                               // not enough items yet: S_DEMOD waits here
      const c8p_pkt_in& p = pk[ri];
      if (p.n_items == 0) {
        if (p.rsv[0] == C8P_SIG_NEED_MORE) { done = k; break; }          // S_COPY would wait for the rest of the packet
        d_scan = e.idx + 80;                                              // signal_impl.cc:157-160
        continue;
      }
      out_pkt o;
      if (++d_seq >= 1000000000) d_seq = 0;                               // signal_impl.cc:133-134
      o.tags.push_back({"cfo", tag_value::from_float(p.cfo)});
      o.tags.push_back({"snr", tag_value::from_float(p.snr)});
      o.tags.push_back({"rssi", tag_value::from_float(p.rssi)});
      o.tags.push_back({"seq", tag_value::from_long(d_seq)});
      o.tags.push_back({"mcs", tag_value::from_long(p.lsig_mcs)});
      o.tags.push_back({"len", tag_value::from_long(p.lsig_len)});
      o.tags.push_back({"nsamp", tag_value::from_long(p.lsig_nsamp)});
      o.tags.push_back({"chan", tag_value::from_c32vector(std::vector<gr_complex>(chan.begin() + 64 * ri, chan.begin() + 64 * ri + 64))});
      for (int a = 0; a < d_nant; a++) o.items[a].assign(iq.begin() + p.iq_off[a], iq.begin() + p.iq_off[a] + p.n_items);
      d_outq.push_back(std::move(o));
      d_scan = e.idx + 224 + (uint64_t)p.lsig_nsamp;                      // S_COPY consumed the packet
    }
    d_evts.erase(d_evts.begin(), d_evts.begin() + (std::ptrdiff_t)done);
  }

  void trim()   // raw items before the oldest pending flag are no longer needed (later flags only point forward)
  {
    const uint64_t keep_from = d_evts.empty() ? avail_end() : std::max(d_base, std::min(d_evts.front().idx, avail_end()));
    const size_t drop = (size_t)(keep_from - d_base);
    if (drop) { for (int a = 0; a < d_nant; a++) d_raw[a].erase(d_raw[a].begin(), d_raw[a].begin() + (std::ptrdiff_t)drop); d_base += drop; }
  }

  int drain(gr_vector_void_star& output_items, int noutput_items)
  {
    int produced = 0;
    while (!d_outq.empty() && produced < noutput_items) {
      out_pkt& o = d_outq.front();
      if (o.sent == 0)
        for (auto& kv : o.tags) d_io->add_item_tag(0, d_io->nitems_written(0) + produced, kv.first, kv.second);
      const size_t k = std::min(o.items[0].size() - o.sent, (size_t)(noutput_items - produced));
      for (int a = 0; a < d_nant; a++)
        memcpy(static_cast<gr_complex*>(output_items[a]) + produced, o.items[a].data() + o.sent, sizeof(gr_complex) * k);
      o.sent += k; produced += (int)k;
      if (o.sent == o.items[0].size()) d_outq.pop_front();
    }
    return produced;
  }

  engine_holder d_eng;
  int d_nant;
  std::vector<std::vector<gr_complex>> d_raw;    // buffered raw items per antenna, d_raw[a][0] = stream item d_base
  uint64_t d_base = 0, d_scan = 0, d_last_end = ~0ull;   // d_scan: where S_TRIGGER resumes looking at sync flags
  std::vector<evt> d_evts;                       // pending flags, in stream order
  long d_seq = 0;
  std::deque<out_pkt> d_outq;
};

signal::sptr signal::make() { return make(default_api(), batching_cfg()); }
signal::sptr signal::make(const c8p_api& api, const batching_cfg& cfg) { return std::shared_ptr<signal>(new signal_front(1, api, cfg)); }
signal2::sptr signal2::make() { return make(default_api(), batching_cfg()); }
signal2::sptr signal2::make(const c8p_api& api, const batching_cfg& cfg) { return std::shared_ptr<signal2>(new signal_front(2, api, cfg)); }

// ================================================================================================ detect
// Streaming front of c8p_detect(): gathers raw items, continues from the state / commit point of the previous call with
// C8P_DET_HIST items of history in front, and emits sync's byte stream up to the new commit point (everything before
// it is final).  Events whose flag item lies beyond the commit point wait in a queue until the output gets there.
class detect_front : public detect
{
public:
  detect_front(const c8p_api& api, const batching_cfg& cfg) : d_eng(api, cfg) { memset(&d_state, 0, sizeof(d_state)); }

  void forecast(int noutput_items, gr_vector_int& req) override { for (auto& r : req) r = noutput_items; }

  int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& output_items) override
  {
    const int nproc = ninput_items[0];
    const gr_complex* x = static_cast<const gr_complex*>(input_items[0]);
    d_raw.insert(d_raw.end(), x, x + nproc);
    d_io->consume_each(nproc);
    const uint64_t fresh = buffered_end() - d_seen_end;
    if (fresh > 0 && ((int64_t)(buffered_end() - d_commit) >= d_eng.d_cfg.detect_items || (d_eng.d_cfg.flush_when_idle && nproc == 0))) process();
    return drain(static_cast<uint8_t*>(output_items[0]), noutput_items);
  }

  void flush() override { if (buffered_end() > d_seen_end) process(); }

private:
  uint64_t buffered_end() const { return d_base + d_raw.size(); }

  void process()
  {
    d_seen_end = buffered_end();
    c8p_detect_in in;
    memset(&in, 0, sizeof(in));
    uint64_t commit_rel = 0;
    c8p_det_state out;
    in.mem = C8P_MEM_HOST; in.keep = 0;
    in.raw = reinterpret_cast<const c8p_c32*>(d_raw.data()); in.n = d_raw.size(); in.n_hist = d_commit - d_base;
    in.state = d_have_state ? &d_state : nullptr; in.state_out = &out; in.commit = &commit_rel;
    d_eng.check(d_eng.d_api.detect(d_eng.d_ctx, &in));
    std::vector<c8p_sync_evt> ev = d_eng.fetch_all<c8p_sync_evt>(C8P_BUF_DET_EVT);
    for (c8p_sync_evt& e : ev) { e.index += d_base; d_evq.push_back(e); }
    d_state = out; d_have_state = true;
    d_commit = d_base + commit_rel;
    const uint64_t keep_from = d_commit > C8P_DET_HIST ? d_commit - C8P_DET_HIST : 0;   // history for the next call
    if (keep_from > d_base) { d_raw.erase(d_raw.begin(), d_raw.begin() + (std::ptrdiff_t)(keep_from - d_base)); d_base = keep_from; }
  }

  int drain(uint8_t* out, int noutput_items)
  {
    const int k = (int)std::min<uint64_t>((uint64_t)noutput_items, d_commit - d_out);
    if (k <= 0) return 0;
    memset(out, 0, (size_t)k);
    while (!d_evq.empty() && d_evq.front().index < d_out + (uint64_t)k) {
      const c8p_sync_evt& e = d_evq.front();
      if (e.index >= d_out) {
        out[e.index - d_out] = 0x01;                                             // sync_impl.cc:123
        const uint64_t off = d_io->nitems_written(0) + (e.index - d_out);
        d_io->add_item_tag(0, off, "rad", tag_value::from_float(e.rad));
        d_io->add_item_tag(0, off, "snr", tag_value::from_float(e.snr));
        d_io->add_item_tag(0, off, "rssi", tag_value::from_float(e.rssi));
      }
      d_evq.pop_front();
    }
    d_out += (uint64_t)k;
    return k;
  }

  engine_holder d_eng;
  std::vector<gr_complex> d_raw;           // d_raw[0] = stream item d_base
  uint64_t d_base = 0, d_commit = 0, d_out = 0, d_seen_end = 0;
  c8p_det_state d_state;
  bool d_have_state = false;
  std::deque<c8p_sync_evt> d_evq;          // events not yet written to the output stream (absolute indices)
};

detect::sptr detect::make() { return make(default_api(), batching_cfg()); }
detect::sptr detect::make(const c8p_api& api, const batching_cfg& cfg) { return std::shared_ptr<detect>(new detect_front(api, cfg)); }

// ================================================================================================ decode
class decode_front : public decode
{
public:
  decode_front(bool ifdebug, const c8p_api& api, const batching_cfg& cfg) : d_eng(api, cfg), d_debug(ifdebug)
  {
    memset(d_vhtMcsCount, 0, sizeof(d_vhtMcsCount));
    memset(d_legacyMcsCount, 0, sizeof(d_legacyMcsCount));
    memset(d_htMcsCount, 0, sizeof(d_htMcsCount));
  }

  void forecast(int noutput_items, gr_vector_int& req) override   // decode_impl.cc:54-58
  {
    req[0] = noutput_items + 160;
  }

  int general_work(int /*noutput_items*/, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& /*output_items*/) override
  {
    const float* in = static_cast<const float*>(input_items[0]);
    const int nproc = ninput_items[0];
    int used = 0;
    while (used < nproc) {
      if (d_state == S_IDLE) {
        std::vector<tag_t> tags;
        const uint64_t pos = d_io->nitems_read(0) + used;
        d_io->get_tags_in_range(tags, 0, pos, pos + 1);
        if (tags.empty()) { used++; continue; }
        memset(&d_cur, 0, sizeof(d_cur));
        d_cur.status = C8P_ST_OK;
        d_cur.format = (int32_t)tag_long(tags, "format", 99999); d_cur.len = (int32_t)tag_long(tags, "len", 99999);
        d_cur.total = (int32_t)tag_long(tags, "total", 99999); d_cur.cr = (int32_t)tag_long(tags, "cr", 99999);
        d_cur.mcs = (int32_t)tag_long(tags, "mcs", 99999); d_cur.ampdu = (int32_t)tag_long(tags, "ampdu", 99999);
        d_cur.trellis = (int32_t)tag_long(tags, "trellis", 99999);
        d_cur.cfo = tag_float(tags, "cfo", 0.f); d_cur.snr = tag_float(tags, "snr", 0.f); d_cur.rssi = tag_float(tags, "rssi", 0.f);
        d_cur.sssnr0 = tag_float(tags, "sssnr0", 0.f); d_cur.sssnr1 = tag_float(tags, "sssnr1", 0.f);
        d_cur.is_ndp = d_cur.trellis == 0;
        d_cur.llr_off = d_llr.size();
        const tag_value* mu = find_tag(tags, "mu2x1chan");
        for (int i = 0; i < 128; i++) d_mu.push_back(mu && (int)mu->v.size() > i ? mu->v[i] : gr_complex(0.f, 0.f));
        d_llr.resize(d_llr.size() + (size_t)std::max(d_cur.total, 0));
        d_have = 0;
        d_state = S_COLLECT;
      }
      const int take = (int)std::min<int64_t>(nproc - used, (int64_t)d_cur.total - d_have);
      if (take > 0) memcpy(&d_llr[d_cur.llr_off + d_have], in + used, sizeof(float) * (size_t)take);
      d_have += std::max(take, 0); used += std::max(take, 0);
      if (d_have >= d_cur.total) {
        d_tags.push_back(d_cur);
        d_state = S_IDLE;
        if ((int)d_tags.size() >= d_eng.d_cfg.max_batch) submit();
      }
    }
    if (d_eng.d_cfg.flush_when_idle && d_state == S_IDLE && !d_tags.empty()) submit();
    d_io->consume_each(used);
    return 0;
  }

  void flush() override
  {
    if (!d_tags.empty()) submit();
  }

private:
  void submit()
  {
    c8p_decode_in b;
    memset(&b, 0, sizeof(b));
    b.n_pkt = (int32_t)d_tags.size(); b.mem = C8P_MEM_HOST; b.tags = d_tags.data();
    if (d_llr.empty()) d_llr.push_back(0.f);
    b.llr = d_llr.data(); b.n_llr = d_llr.size();
    b.mu2x1chan = reinterpret_cast<const c8p_c32*>(d_mu.data());
    d_eng.check(d_eng.d_api.decode(d_eng.d_ctx, &b));
    d_eng.check(d_eng.d_api.sync(d_eng.d_ctx));
    std::vector<c8p_pkt_out> outs = d_eng.fetch_all<c8p_pkt_out>(C8P_BUF_OUT);
    std::vector<uint8_t> pdu = d_eng.fetch_all<uint8_t>(C8P_BUF_PDU);
    for (size_t p = 0; p < d_tags.size(); p++) {
      const c8p_pkt_tags& t = d_tags[p];
      const c8p_pkt_out& o = outs[p];
      size_t off = o.pdu_off;
      for (int k = 0; k < o.n_pdu; k++) {
        const int n = pdu[off] | (pdu[off + 1] << 8);
        pdu_t m;
        m.len = n;
        m.blob.assign(pdu.begin() + off + 2, pdu.begin() + off + 2 + n);
        off += 2 + (size_t)n;
        if (t.trellis != 0) debug_line(t, true);
        d_io->message_port_pub("out", m);
      }
      for (int k = 0; k < o.n_crc_fail; k++) debug_line(t, false);
    }
    d_tags.clear(); d_llr.clear(); d_mu.clear();
  }

  // decode_impl.cc:375-411 (vht) and :454-509 (legacy / ht)
  void debug_line(const c8p_pkt_tags& t, bool ok)
  {
    if (!d_debug) return;
    std::string s;
    if (t.format == C8P_F_VHT) {
      if (ok) { d_nPktCorrect++; if (t.mcs >= 0 && t.mcs < 10) d_vhtMcsCount[t.mcs]++; }
      s = std::string("ieee80211 decode, vht crc32 ") + (ok ? "correct" : "wrong") + ", total:" + std::to_string(d_nPktCorrect);
      for (int i = 0; i < 10; i++) s += "," + std::to_string(i) + ":" + std::to_string(d_vhtMcsCount[i]);
      s += ",cfo:" + std::to_string(t.cfo) + ",snr:" + std::to_string(t.snr) + ",rssi:" + std::to_string(t.rssi);
      s += ",sssnr0:" + std::to_string(t.sssnr0) + ",sssnr1:" + std::to_string(t.sssnr1);
    } else {
      const bool leg = t.format == C8P_F_L;
      if (ok) { d_nPktCorrect++; if (leg) d_legacyMcsCount[t.mcs & 7]++; else d_htMcsCount[t.mcs % 8]++; }
      s = std::string("ieee80211 decode, ") + (leg ? "legacy" : "ht") + " crc32 " + (ok ? "correct" : "wrong") + ", total:" +
          std::to_string(d_nPktCorrect);
      for (int i = 0; i < 8; i++) s += "," + std::to_string(i) + ":" + std::to_string(leg ? d_legacyMcsCount[i] : d_htMcsCount[i]);
      s += ",cfo:" + std::to_string(t.cfo) + ",snr:" + std::to_string(t.snr) + ",rssi:" + std::to_string(t.rssi);
    }
    d_io->debug_print(s);
  }

  enum { S_IDLE, S_COLLECT } d_state = S_IDLE;
  engine_holder d_eng;
  bool d_debug;
  c8p_pkt_tags d_cur;
  int d_have = 0;
  std::vector<c8p_pkt_tags> d_tags;
  std::vector<float> d_llr;
  std::vector<gr_complex> d_mu;
  uint64_t d_nPktCorrect = 0, d_legacyMcsCount[8], d_vhtMcsCount[10], d_htMcsCount[8];
};

decode::sptr decode::make(bool ifdebug) { return make(ifdebug, default_api(), batching_cfg()); }
decode::sptr decode::make(bool ifdebug, const c8p_api& api, const batching_cfg& cfg)
{
  return std::shared_ptr<decode>(new decode_front(ifdebug, api, cfg));
}

} // namespace ieee80211
} // namespace gr
