// c8p_rx_chain.cc — rx_chain: rx.grc's receive chain for K radios as one block, on host/c8p_rawchain's live ingest (see c8p_blocks.h).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "c8p_blocks.h"
#include "c8p_rawchain.h"

C8P_MIRROR_NS_BEGIN

namespace {

class rx_chain_front : public rx_chain
{
public:
  rx_chain_front(int n_streams, const std::string& engine_lib, const batching_cfg& cfg, int fmt, float scale)
    : d_k(n_streams), d_per(std::max(cfg.detect_items, 1024)), d_idle_flush(cfg.flush_when_idle), d_max_wait_us(cfg.max_wait_us),
      d_ib(fmt == C8P_FMT_SC16 ? 4 : fmt == C8P_FMT_SC8 ? 2 : 8), d_memflag(fmt == C8P_FMT_SC16 ? C8P_MEM_SC16 : fmt == C8P_FMT_SC8 ? C8P_MEM_SC8 : 0),
      d_api(load_c8p_api(engine_lib))
  {
    if (n_streams < 1 || n_streams > 65535) throw std::runtime_error("rx_chain: 1 .. 65535 streams");
    if (fmt != C8P_FMT_C32 && fmt != C8P_FMT_SC16 && fmt != C8P_FMT_SC8) throw std::runtime_error("rx_chain: unknown item format");
    c8p_rawchain_cfg rc;
    memset(&rc, 0, sizeof(rc));
    rc.struct_size = sizeof(rc); rc.device = cfg.device; rc.lib = engine_lib.c_str();
    rc.streams = (uint32_t)d_k; rc.stream_round_items = (uint64_t)d_per;
    rc.sc16_scale = rc.sc8_scale = scale;
    char err[512] = "";
    d_h = c8p_rawchain_create(&rc, err, (int)sizeof(err));
    if (!d_h) throw std::runtime_error(std::string("rx_chain: c8p_rawchain_create: ") + err);
    // one staging area of pinned memory: stream s's items of the open round at d_stage + s * d_per
    d_stage = static_cast<uint8_t*>(d_api.host_alloc((uint64_t)d_ib * (uint64_t)d_k * (uint64_t)d_per));
    if (!d_stage) { c8p_rawchain_destroy(d_h); throw std::runtime_error("rx_chain: no pinned memory for the staging area"); }
    if (c8p_rawchain_live(d_h, &rx_chain_front::on_round, this) != C8P_OK) {
      const std::string m = c8p_rawchain_error(d_h);
      c8p_rawchain_destroy(d_h); d_api.host_free(d_stage);
      throw std::runtime_error("rx_chain: c8p_rawchain_live: " + m);
    }
  }
  ~rx_chain_front() override
  {
    if (d_h) c8p_rawchain_destroy(d_h);          // (waits for the pipeline's threads: no callback after this)
    if (d_stage) d_api.host_free(d_stage);
  }

  // PDUs that wait to be published ask for a call WITHOUT input (the rawchain's callback wakes the scheduler): a source that pauses must
  // not hold back the results of the rounds it has already delivered
  void forecast(int noutput_items, gr_vector_int& req) override
  {
    bool pending;
    { std::lock_guard<std::mutex> g(d_mu); pending = !d_q.empty(); }
    for (auto& r : req) r = pending ? 0 : noutput_items;
  }

  int general_work(int, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star&) override
  {
    bool publish_only;                       // forecast asked for no input: this call is here for the PDUs, it says nothing about the sources
    { std::lock_guard<std::mutex> g(d_mu); publish_only = !d_q.empty(); }
    int n = ninput_items.empty() ? 0 : ninput_items[0];
    for (int v : ninput_items) n = std::min(n, v);
    n = std::min<int>(n, d_per - d_fill);                                      // the streams advance together (consume_each)
    if (n > 0 && d_fill == 0) d_t_open = std::chrono::steady_clock::now();
    for (int s = 0; s < d_k && n > 0; s++)
      memcpy(d_stage + ((size_t)s * (size_t)d_per + (size_t)d_fill) * d_ib, input_items[(size_t)s], d_ib * (size_t)n);
    d_fill += n;
    d_io->consume_each(n);
    // An open round is closed early only when the input has REALLY dried up: a call that came for input and found none on some port,
    // and the round either an eighth full or open for max_wait_us.  (Round 4 closed it on any call without items — one port momentarily
    // empty, or a wake-up to publish PDUs — and a busy scheduler cut the stream into many tiny, blocking rounds.)
    const bool dried_up = d_idle_flush && d_fill > 0 && n == 0 && !publish_only &&
                          (d_fill >= d_per / 8 || std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - d_t_open).count() >= d_max_wait_us);
    if (d_fill == d_per || dried_up) close_round();
    publish();
    return 0;
  }

  // end of stream: the open round, then everything that is still in the pipeline (flags that wait for more items yield nothing)
  void flush() override
  {
    if (d_fill > 0) close_round();
    c8p_rawchain_stats st;
    if (c8p_rawchain_drain(d_h, &st) != C8P_OK) fail("c8p_rawchain_drain");
    publish();
    if (c8p_rawchain_live(d_h, &rx_chain_front::on_round, this) != C8P_OK) fail("c8p_rawchain_live");       // the next session, should items follow
  }

private:
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("rx_chain: ") + what + ": " + c8p_rawchain_error(d_h)); }

  void close_round()
  {
    for (int s = 0; s < d_k; s++)
      if (c8p_rawchain_push(d_h, (uint32_t)s, d_stage + (size_t)s * (size_t)d_per * d_ib, (uint64_t)d_fill, C8P_MEM_PINNED | d_memflag) != C8P_OK) fail("c8p_rawchain_push");
    if (c8p_rawchain_round(d_h) != C8P_OK) fail("c8p_rawchain_round");           // returns when the items are on the device: the staging area is free
    d_fill = 0;
  }

  // on the pipeline's rx thread, round by round: the PDU records of all streams -> the queue general_work publishes from
  static void on_round(void* user, const c8p_rawchain_out* o)
  {
    rx_chain_front* self = static_cast<rx_chain_front*>(user);
    uint64_t nb = 0;
    (void)self->d_api.buffer(o->rx, C8P_BUF_PDU, &nb);
    std::vector<uint8_t> pdu((size_t)nb);
    if (nb && self->d_api.fetch(o->rx, C8P_BUF_PDU, pdu.data(), 0, nb) != C8P_OK) {
      std::lock_guard<std::mutex> g(self->d_mu);
      self->d_failed = true;
      return;
    }
    std::vector<pdu_t> out;
    for (int p = 0; p < o->n_packets; p++) {
      const c8p_pkt_out& r = o->outs[p];
      size_t off = (size_t)r.pdu_off;
      for (int k = 0; k < r.n_pdu; k++) {                   // records {u16 len, blob = format, len lo, len hi, MPDU.., mcs}
        const int len = pdu[off] | (pdu[off + 1] << 8);
        pdu_t m;
        m.len = len; m.stream = (int)o->pkt_stream[p];
        m.blob.assign(pdu.begin() + (std::ptrdiff_t)off + 2, pdu.begin() + (std::ptrdiff_t)off + 2 + len);
        off += 2 + (size_t)len;
        out.push_back(std::move(m));
      }
    }
    if (out.empty()) return;
    { std::lock_guard<std::mutex> g(self->d_mu); for (pdu_t& m : out) self->d_q.push_back(std::move(m)); }
    if (self->d_io) self->d_io->wake();                     // results while no input arrives: get general_work called
  }

  void publish()
  {
    std::deque<pdu_t> q;
    {
      std::lock_guard<std::mutex> g(d_mu);
      q.swap(d_q);
      if (d_failed) throw std::runtime_error("rx_chain: fetching a round's PDUs failed");
    }
    for (const pdu_t& m : q) d_io->message_port_pub("out", m);
  }

  const int d_k, d_per;
  const bool d_idle_flush;
  const int d_max_wait_us;
  std::chrono::steady_clock::time_point d_t_open;      // when the open round received its first items
  const size_t d_ib;                  // bytes per input item
  const int d_memflag;
  c8p_api d_api;
  c8p_rawchain* d_h = nullptr;
  uint8_t* d_stage = nullptr;
  int d_fill = 0;
  std::mutex d_mu;
  std::deque<pdu_t> d_q;
  bool d_failed = false;
};

} // namespace

rx_chain::sptr rx_chain::make(int n_streams)
{
  const char* e = getenv("C8P_RX_LIB");
  return make(n_streams, e ? e : "libc8p_rx.so", batching_cfg());
}
rx_chain::sptr rx_chain::make(int n_streams, const std::string& engine_lib, const batching_cfg& cfg)
{
  return make(n_streams, engine_lib, cfg, C8P_FMT_C32, 0.0f);
}
rx_chain::sptr rx_chain::make(int n_streams, const std::string& engine_lib, const batching_cfg& cfg, int item_format, float scale)
{
  return std::shared_ptr<rx_chain>(new rx_chain_front(n_streams, engine_lib, cfg, item_format, scale));
}

C8P_MIRROR_NS_END
