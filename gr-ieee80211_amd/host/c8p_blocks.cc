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
// What differs is *when*: packets are processed a batch at a time, so outputs trail inputs by up to one batch (two while
// full batches follow each other back to back: batch i+1 is submitted before batch i is collected, so that its upload and
// kernels overlap the download of batch i; a partial batch submitted because the input ran dry is collected at once).
// Staging arenas are pinned (c8p_host_alloc): the scheduler's buffers are pageable and only valid during general_work.
#include "c8p_blocks.h"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>

C8P_MIRROR_NS_BEGIN

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
  a.rx = (int (*)(c8p_ctx*, const c8p_batch_in*))sym("c8p_rx");
  a.ticket = (int (*)(c8p_ctx*))sym("c8p_ticket");
  a.select = (int (*)(c8p_ctx*, int))sym("c8p_select");
  a.wait = (int (*)(c8p_ctx*, int))sym("c8p_wait");
  a.done = (int (*)(c8p_ctx*, int))sym("c8p_done");
  a.fetch_async = (int (*)(c8p_ctx*, int, void*, uint64_t, uint64_t))sym("c8p_fetch_async");
  a.host_alloc = (void* (*)(uint64_t))sym("c8p_host_alloc");
  a.host_free = (void (*)(void*))sym("c8p_host_free");
  a.detect_multi = (int (*)(c8p_ctx*, const c8p_detect_multi_in*))sym("c8p_detect_multi");
  a.reserve = (int (*)(c8p_ctx*, const c8p_reserve_in*))sym("c8p_reserve");
  a.counter = (int64_t (*)(c8p_ctx*, int))sym("c8p_counter");
  a.dev_alloc = (void* (*)(c8p_ctx*, uint64_t))sym("c8p_dev_alloc");
  a.dev_free = (void (*)(c8p_ctx*, void*))sym("c8p_dev_free");
  a.copy_async = (int (*)(c8p_ctx*, void*, const void*, uint64_t))sym("c8p_copy_async");
  a.copy_spans = (int (*)(c8p_ctx*, c8p_c32*, const c8p_c32*, const c8p_span*, int32_t))sym("c8p_copy_spans");
  a.copy_wait = (int (*)(c8p_ctx*))sym("c8p_copy_wait");
  a.copy_gather = (int (*)(c8p_ctx*, c8p_c32*, const c8p_piece*, int32_t))sym("c8p_copy_gather");
  a.copy_mark = (uint64_t (*)(c8p_ctx*))sym("c8p_copy_mark");
  a.copy_wait_mark = (int (*)(c8p_ctx*, uint64_t))sym("c8p_copy_wait_mark");
  a.copy_mark_done = (int (*)(c8p_ctx*, uint64_t))sym("c8p_copy_mark_done");
  a.copy_fence = (int (*)(c8p_ctx*))sym("c8p_copy_fence");
  a.fetch_gather = (int (*)(c8p_ctx*, const c8p_fetch_piece*, int32_t))sym("c8p_fetch_gather");
  a.option = (int (*)(c8p_ctx*, int, int64_t))dlsym(h, "c8p_option");      // (optional)
  return a;
}

c8p_api default_c8p_api()
{
  const char* e = getenv("C8P_RX_LIB");
  return load_c8p_api(e ? e : "libc8p_rx.so");
}
static c8p_api default_api() { return default_c8p_api(); }

int batching_cfg::default_max_batch()
{
  const char* e = getenv("C8P_MAX_BATCH");
  return e && atoi(e) > 0 ? atoi(e) : 512;
}
int batching_cfg::default_max_wait_us()
{
  const char* e = getenv("C8P_MAX_WAIT_US");
  return e && atoi(e) > 0 ? atoi(e) : 1000;
}

bool batching_cfg::safe_mode()
{
  const char* e = getenv("C8P_GR_SAFE");
  return e && e[0] == '1';
}
bool batching_cfg::default_flush_when_idle() { return !safe_mode(); }
bool batching_cfg::default_fused()
{
  const char* e = getenv("C8P_FUSED");
  return !safe_mode() && e && e[0] == '1';
}
bool batching_cfg::default_fused_auto()
{
  const char* e = getenv("C8P_FUSED");
  return !safe_mode() && (!e || !*e || !strcmp(e, "auto"));
}
bool batching_cfg::default_fused_lean_tags()
{
  const char* e = getenv("C8P_FUSED_TAGS");
  return !(e && !strcmp(e, "full"));
}
bool batching_cfg::default_pinned_io()
{
  const char* e = getenv("C8P_PINNED_IO");
  return !safe_mode() && !(e && e[0] == '0');
}
// C8P_GATHER_MIN_BUFFER in the environment overrides the header's threshold (bytes; 0: always gather out of a pinned buffer — the tests)
static size_t gather_min_buffer()
{
  const char* e = getenv("C8P_GATHER_MIN_BUFFER");
  return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)C8P_GATHER_MIN_BUFFER;
}
static int env_int(const char* name, int dflt)
{
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
// the fronts of this process, for buffer_will_move
static std::mutex g_fronts_mu;
static std::vector<c8p_block*> g_fronts;
struct front_registration {
  c8p_block* b;
  explicit front_registration(c8p_block* b_) : b(b_) { std::lock_guard<std::mutex> g(g_fronts_mu); g_fronts.push_back(b); }
  void leave()
  {
    if (!b) return;
    std::lock_guard<std::mutex> g(g_fronts_mu);
    g_fronts.erase(std::remove(g_fronts.begin(), g_fronts.end(), b), g_fronts.end());
    b = nullptr;
  }
  ~front_registration() { leave(); }
};
void buffer_will_move(const void* base, size_t bytes)
{
  std::lock_guard<std::mutex> g(g_fronts_mu);           // (a front leaves the list before anything of it is destroyed)
  for (c8p_block* b : g_fronts) b->quiesce(base, bytes);
}
// (g_fronts_mu stays held from the constructor to the destructor: no front is created or destroyed while a buffer moves; a front's
// own mutex is taken inside move_begin — the order is always g_fronts_mu, then the front's)
buffer_move_scope::buffer_move_scope(const void* base, size_t bytes, int64_t timeout_us)
{
  g_fronts_mu.lock();
  for (c8p_block* b : g_fronts) {
    const int r = b->move_begin(base, bytes, timeout_us);
    if (r > 0) d_held.push_back(b);
    if (r < 0) {                                 // one front could not be held: let go of the others, the buffer stays where it is
      for (c8p_block* h : d_held) h->move_end(nullptr);
      d_held.clear();
      d_ok = false; d_done = true;
      break;
    }
  }
}
// a front's own mutex, tried for at most timeout_us (see buffer_move_scope: never waited for without a bound)
static bool try_lock_for(std::mutex& m, int64_t timeout_us)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (m.try_lock()) return true;
    if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= timeout_us) return false;
    std::this_thread::yield();
  }
}
void buffer_move_scope::moved_to(void* new_write_ptr)
{
  if (d_done) return;
  d_done = true;
  for (c8p_block* b : d_held) b->move_end(new_write_ptr);
}
buffer_move_scope::~buffer_move_scope()
{
  if (!d_done) for (c8p_block* b : d_held) b->move_end(nullptr);
  g_fronts_mu.unlock();
}
bool batching_cfg::default_llr_handle()
{
  const char* e = getenv("C8P_LLR_HANDLE");
  return !safe_mode() && !(e && e[0] == '0');
}
bool batching_cfg::default_defer_consume()
{
  const char* e = getenv("C8P_DEFER_CONSUME");
  return !safe_mode() && !(e && e[0] == '0');
}

// growable array in pinned host memory (uninitialised growth: no zero fill of tens of megabytes per batch).  Holds the allocator's two
// entry points, not the table they came from: a result arena handed to another block (fused mode) may outlive the front that made it.
template <typename T> class pinned_vec
{
public:
  explicit pinned_vec(const c8p_api* a) : d_alloc(a->host_alloc), d_free(a->host_free) {}
  pinned_vec(const pinned_vec&) = delete;
  pinned_vec& operator=(const pinned_vec&) = delete;
  ~pinned_vec() { if (d_p) d_free(d_p); }
  T* data() { return d_p; }
  const T* data() const { return d_p; }
  size_t size() const { return d_n; }
  bool empty() const { return d_n == 0; }
  void clear() { d_n = 0; }
  void reserve(size_t want)
  {
    if (want <= d_cap) return;
    const size_t nc = std::max<size_t>(std::max(want, d_cap * 2), 1u << 14);
    T* q = static_cast<T*>(d_alloc(nc * sizeof(T)));
    if (!q) throw std::runtime_error("c8p_host_alloc failed");
    if (d_n) memcpy(static_cast<void*>(q), static_cast<const void*>(d_p), d_n * sizeof(T));
    if (d_p) d_free(d_p);
    d_p = q; d_cap = nc;
  }
  T* grow(size_t extra) { reserve(d_n + extra); T* r = d_p + d_n; d_n += extra; return r; }
  void resize_uninit(size_t n) { reserve(n); d_n = n; }

private:
  void* (*d_alloc)(uint64_t);
  void (*d_free)(void*);
  T* d_p = nullptr;
  size_t d_n = 0, d_cap = 0;
};

// Fused mode hand-over (batching_cfg::fused): what `demod` learned about a BATCH from c8p_rx — the per-packet results and the PDU arena,
// as downloaded, nothing unpacked — parked until `decode` meets the "c8p_fused" tags that name its packets (value = batch id << 20 |
// packet index).  Both mirrors live in the flowgraph's process; the tags travel through the scheduler.  One registry operation per
// batch on either side (rounds 2-4: one map insert, one PDU unpacking and one lookup under the mutex per PACKET).
struct fused_batch {
  explicit fused_batch(const c8p_api* a) : outs(a), pdu(a) {}
  pinned_vec<c8p_pkt_out> outs;       // [packets of the batch]
  pinned_vec<uint8_t> pdu;            // the PDU arena (whole: its layout is fixed-stride)
  std::vector<c8p_pkt_tags> tags;     // lean tags (batching_cfg::fused_lean_tags): the packets' tag records, which then do not travel as stream tags
};
static const int C8P_FUSED_IDX_BITS = 20;
static std::mutex g_fused_mu;
static std::map<long, std::shared_ptr<fused_batch>> g_fused;
static std::atomic<long> g_fused_next{1};
// A batch is erased when `decode` moves on to the next one.  If no decode is downstream (or tags are dropped, or the flowgraph stops
// first) nobody does: the map is therefore bounded — the oldest batches go once more than C8P_FUSED_MAX are parked — and a demod front
// takes the batches it issued with it when it is destroyed.
static const size_t C8P_FUSED_MAX = 256;
static void fused_park(long id, std::shared_ptr<fused_batch> f)
{
  std::lock_guard<std::mutex> g(g_fused_mu);
  g_fused[id] = std::move(f);
  while (g_fused.size() > C8P_FUSED_MAX) g_fused.erase(g_fused.begin());     // (ids grow: begin() is the oldest)
}
static std::shared_ptr<fused_batch> fused_find(long id)
{
  std::lock_guard<std::mutex> g(g_fused_mu);
  auto it = g_fused.find(id);
  return it == g_fused.end() ? std::shared_ptr<fused_batch>() : it->second;
}
static void fused_purge(const std::vector<long>& ids)
{
  std::lock_guard<std::mutex> g(g_fused_mu);
  for (long id : ids) g_fused.erase(id);
}

// the tags of ONE item: [first, last) of a vector sorted by offset
struct tag_span {
  const tag_t* first; const tag_t* last;
  const tag_value* find(const char* key) const
  {
    for (const tag_t* t = first; t != last; ++t) if (t->key == key) return &t->value;
    return nullptr;
  }
  long get_long(const char* key, long dflt) const
  {
    const tag_value* v = find(key);
    return v ? (v->kind == tag_value::FLOAT ? (long)v->f : v->l) : dflt;
  }
  float get_float(const char* key, float dflt) const
  {
    const tag_value* v = find(key);
    return v ? (v->kind == tag_value::LONG ? (float)v->l : v->f) : dflt;
  }
};
// The tags of the items one general_work call may look at, fetched ONCE per call (rounds 1-4 asked the scheduler per packet:
// get_tags_in_range locks the buffer and walks its tag list every time) and then walked in step with the items.
struct tag_cursor {
  std::vector<tag_t> tags;
  size_t i = 0;
  void load(block_io* io, uint64_t a, uint64_t b)
  {
    tags.clear(); i = 0;
    if (b <= a) return;
    io->get_tags_in_range(tags, 0, a, b);
    auto by_off = [](const tag_t& x, const tag_t& y) { return x.offset < y.offset; };
    if (!std::is_sorted(tags.begin(), tags.end(), by_off)) std::stable_sort(tags.begin(), tags.end(), by_off);
  }
  // the tags on item `pos`; false: it carries none, and `next` = the next item that does (or `end`)
  bool at(uint64_t pos, uint64_t end, tag_span& sp, uint64_t& next)
  {
    while (i < tags.size() && tags[i].offset < pos) i++;
    if (i < tags.size() && tags[i].offset == pos) {
      sp.first = &tags[i];
      while (i < tags.size() && tags[i].offset == pos) i++;
      sp.last = tags.data() + i;
      return true;
    }
    next = i < tags.size() ? std::min(tags[i].offset, end) : end;
    return false;
  }
};
// Plain mode's hand-over of a batch's LLR arena on the device (batching_cfg::llr_handle): what `decode` needs to decode a packet of the
// batch where it lies — base address, arena size, every packet's offset — and something that keeps the memory alive.
struct llr_batch {
  std::shared_ptr<void> keep;         // demod's result object (owns the device buffer)
  const float* base = nullptr;        // device address of the arena copy
  uint64_t n_floats = 0;
  std::vector<uint64_t> llr_off;      // per packet of the batch (c8p_pkt_tags.llr_off)
};
static std::mutex g_llr_mu;
static std::map<long, std::shared_ptr<llr_batch>> g_llr;
static void llr_park(long id, std::shared_ptr<llr_batch> b)
{
  std::lock_guard<std::mutex> g(g_llr_mu);
  g_llr[id] = std::move(b);
  while (g_llr.size() > 64) g_llr.erase(g_llr.begin());       // (a decode that never comes: the oldest go; each holds tens of MB of device memory)
}
static std::shared_ptr<llr_batch> llr_find(long id)
{
  std::lock_guard<std::mutex> g(g_llr_mu);
  auto it = g_llr.find(id);
  return it == g_llr.end() ? std::shared_ptr<llr_batch>() : it->second;
}
static void llr_purge(const std::vector<long>& ids)
{
  std::lock_guard<std::mutex> g(g_llr_mu);
  for (long id : ids) g_llr.erase(id);
}

static const tag_value* find_tag(const std::vector<tag_t>& tags, const char* key)
{
  for (const auto& t : tags) if (t.key == key) return &t.value;
  return nullptr;
}
static float tag_float(const std::vector<tag_t>& tags, const char* key, float dflt)
{
  const tag_value* v = find_tag(tags, key);
  return v ? (v->kind == tag_value::LONG ? (float)v->l : v->f) : dflt;
}

typedef std::chrono::steady_clock::time_point tick_t;
static tick_t now_tick() { return std::chrono::steady_clock::now(); }
static long us_since(tick_t t) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(now_tick() - t).count(); }

// host-side time per phase of a front (microseconds, accumulated, EXCLUSIVE: entering a phase stops the clock of the one around
// it): what tools/bench_blocks.py reports as host_phase_us
struct phase_clock {
  enum { CUT, SUBMIT, ENQ, ALLOC, WAIT, FETCH, REPLAY, DRAIN, GATHER, GWAIT, N };
  double us[N] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int cur = -1;
  tick_t t_last;
  void charge()
  {
    const tick_t t = now_tick();
    if (cur >= 0) us[cur] += std::chrono::duration<double, std::micro>(t - t_last).count();
    t_last = t;
  }
  struct scope {
    phase_clock& c; int prev; bool open = true;
    scope(phase_clock& c_, int k) : c(c_) { c.charge(); prev = c.cur; c.cur = k; }
    void end() { if (open) { open = false; c.charge(); c.cur = prev; } }
    ~scope() { end(); }
  };
  std::string str() const
  {
    static const char* nm[N] = {"cut_and_stage", "submit", "enqueue_downloads", "staging_alloc", "wait", "fetch", "replay", "drain", "gather_enqueue", "gather_wait"};
    std::string s;
    for (int i = 0; i < N; i++) s += std::string(nm[i]) + "=" + std::to_string((long)us[i]) + ";";
    return s;
  }
};

typedef phase_clock::scope sub_scope;

// The latency timer of a front.  general_work() only runs while items arrive; `due()` tells when the oldest gathered packet or the
// batch in flight reaches max_wait_us, and `fire()` does what general_work's own age check would do (under the front's mutex).
class latency_timer
{
public:
  latency_timer(std::mutex& mu, std::function<bool(tick_t&)> due, std::function<void()> fire) : d_mu(mu), d_due(std::move(due)), d_fire(std::move(fire)) {}
  ~latency_timer() { stop(); }
  void start() { d_th = std::thread([this] { loop(); }); }
  void stop()
  {
    { std::lock_guard<std::mutex> g(d_mu); d_quit = true; }
    d_cv.notify_all();
    if (d_th.joinable()) d_th.join();
  }
  // the front's state changed and a deadline may have appeared: wake the timer if it sleeps WITHOUT one.  The caller holds the
  // front's mutex, as the timer does whenever it looks at d_idle: no wake-up is lost (the round-3 version remembered "was pending"
  // on the scheduler's side only — the timer drained a burst, slept without a deadline, and the next burst did not wake it)
  void poke() { if (d_idle) d_cv.notify_one(); }
  std::string error;                          // an engine error met on the timer thread is reported by the next general_work
private:
  void loop()
  {
    std::unique_lock<std::mutex> lk(d_mu);
    while (!d_quit) {
      tick_t when;
      if (!d_due(when)) { d_idle = true; d_cv.wait(lk); d_idle = false; continue; }
      // (a system_clock deadline = pthread_cond_timedwait; a steady-clock one is pthread_cond_clockwait, which this toolchain's
      // ThreadSanitizer does not intercept — it then reports the mutex as held across the wait)
      if (now_tick() < when) { d_cv.wait_until(lk, std::chrono::system_clock::now() + (when - now_tick())); continue; }
      try { d_fire(); } catch (const std::exception& e) { error = e.what(); return; }
    }
  }
  std::mutex& d_mu;
  std::function<bool(tick_t&)> d_due;
  std::function<void()> d_fire;
  std::condition_variable d_cv;
  std::thread d_th;
  bool d_quit = false;
  bool d_idle = false;                        // asleep with no deadline to wait for
};

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
    d_cell->ctx = d_ctx; d_cell->dev_free = d_api.dev_free;
    // a front's gathers are 2-3 MB a call: on 16 CUs they leave more of the chip to the chain's kernels than on the engine's default 32
    // (include/c8p_rx.h, C8P_OPT_COPY_CUS; the environment's C8P_COPY_CUS overrides)
    if (d_api.option) (void)d_api.option(d_ctx, C8P_OPT_COPY_CUS, 16);
    // (The fronts keep several batches in flight; everything they need of a batch is downloaded behind its kernels, so the engine's
    // result arenas need not stay resident: any C8P_SLOTS works — with one slot a submit waits inside the engine for its predecessor.)
  }
  ~engine_holder()
  {
    { std::lock_guard<std::mutex> g(d_cell->mu); d_cell->ctx = nullptr; }      // device buffers that outlive this engine are freed without it
    if (d_ctx) d_api.destroy(d_ctx);
  }
  // A device buffer of this engine that something else may still hold when the engine goes (a batch's LLR arena in the hands of
  // another block): freed through the cell — with the ctx while it lives (c8p_dev_free waits for the copy streams), without it after.
  struct ctx_cell { std::mutex mu; c8p_ctx* ctx = nullptr; void (*dev_free)(c8p_ctx*, void*) = nullptr; };
  std::shared_ptr<ctx_cell> d_cell = std::make_shared<ctx_cell>();
  void check(int rc, const char* what = "")
  {
    if (rc != C8P_OK) throw std::runtime_error(std::string("c8p error ") + std::to_string(rc) + " " + what + ": " + d_api.last_error(d_ctx));
  }
  template <typename T> std::vector<T> fetch_all(int which)
  {
    uint64_t bytes = 0;
    d_api.buffer(d_ctx, which, &bytes);
    std::vector<T> v(bytes / sizeof(T));
    if (bytes) check(d_api.fetch(d_ctx, which, v.data(), 0, v.size() * sizeof(T)));
    return v;
  }
  // device memory a front keeps across submits (the arenas its gathers fill): grown x 2, the first keep_bytes moved along by a copy
  // on the copy stream (behind every gather enqueued so far); c8p_dev_free waits for the copy streams before it frees
  struct dev_buf { void* p = nullptr; size_t cap = 0; };
  void dev_reserve(dev_buf& b, size_t bytes, size_t keep_bytes)
  {
    if (bytes <= b.cap) return;
    // (C8P_ARENA_MIN: the tests make the arenas start tiny, so that they grow while gathers into them are under way)
    const size_t arena_min = (size_t)std::max(env_int("C8P_ARENA_MIN", 1 << 20), 64);      // (read per call: arenas grow a handful of times in a front's life)
    const size_t nc = std::max<size_t>(std::max(bytes, 2 * b.cap), arena_min);
    void* q = d_api.dev_alloc(d_ctx, nc);
    if (!q) throw std::runtime_error(std::string("c8p_dev_alloc failed: ") + d_api.last_error(d_ctx));
    if (b.p) {
      // (the gathers that fill the old arena may be on either of two copy streams — C8P_PIECE_UNORDERED — and the move runs on one of
      // them: everything enqueued so far is waited for first.  Arenas grow in a front's first batches and then no more)
      if (keep_bytes) { check(d_api.copy_wait(d_ctx), "c8p_copy_wait"); check(d_api.copy_async(d_ctx, q, b.p, keep_bytes), "arena move"); }
      d_api.dev_free(d_ctx, b.p);
    }
    b.p = q; b.cap = nc;
  }
  void dev_release(dev_buf& b) { if (b.p) d_api.dev_free(d_ctx, b.p); b.p = nullptr; b.cap = 0; }
  c8p_api d_api;
  batching_cfg d_cfg;
  c8p_ctx* d_ctx = nullptr;
};
typedef engine_holder::dev_buf dev_buf;

// The pieces one general_work call cuts out of a pinned scheduler buffer, and when their items may be consumed.  All pieces of a call
// go to the device with ONE c8p_copy_gather (`flush`); with batching_cfg::defer_consume the items are consumed one call later — by then
// the gather has had a scheduler round trip to finish and `settle` returns at once — otherwise at the end of the same call, after a wait.
struct gather_lane {
  std::vector<c8p_piece> pieces;
  uint64_t scan_abs = 0;                 // port 0: items cut so far (absolute); items before it are never looked at again
  // the gathers that have been enqueued and not waited for, oldest first: the copy-stream mark behind each and the scan position it
  // covers.  A call consumes what the gathers of TWO calls ago moved (DEPTH = 2: a gather takes ~20 us from enqueue to done whatever
  // its size — launch, PCIe round trips, the event — and a call of a busy block comes round faster than that)
  int DEPTH = env_int("C8P_GATHER_DEPTH", 2);
  bool unordered = env_int("C8P_GATHER_UNORDERED", 1) != 0;
  // Without deferred consume a call waits for its own gathers before it returns: they then start while the call is still cutting —
  // every C8P_GATHER_CHUNK bytes of pieces go at once — and only the last one is still on its way when the call comes to wait
  size_t chunk_bytes = (size_t)env_int("C8P_GATHER_CHUNK", 1 << 20), pending_bytes = 0;
  bool want_flush(const engine_holder& E) const { return !E.d_cfg.defer_consume && pending_bytes >= chunk_bytes; }
  struct pend_t { uint64_t mark, scan; tick_t t_enq; uintptr_t lo, hi; };      // [lo, hi): the host addresses its gathers read
  // C8P_GATHER_STATS=1: what the waits of `settle` were, printed when the front goes (tools/probes/chain_quick.py)
  struct stats_t { bool on = env_int("C8P_GATHER_STATS", 0) != 0; long calls = 0, waits = 0, slow = 0; double us = 0, worst = 0, age_slow = 0, age_fast = 0, flush_bytes = 0; long flushes = 0; uint64_t items = 0; } stats;
  ~gather_lane()
  {
    if (stats.on && stats.calls)
      fprintf(stderr, "gather_lane: %ld settles, %ld with a wait (%ld over 5 us), %.0f us waited (worst %.0f), %.0f items a gather, depth %d; enqueue -> noticed done: "
              "%.0f us after a wait over 5 us, %.0f us after a shorter one; %ld gathers of %.0f KB\n", stats.calls, stats.waits, stats.slow, stats.us, stats.worst,
              (double)stats.items / std::max(1L, stats.calls), DEPTH, stats.age_slow / std::max(1L, stats.slow), stats.age_fast / std::max(1L, stats.waits - stats.slow),
              stats.flushes, stats.flush_bytes / 1024.0 / std::max(1L, stats.flushes));
  }
  std::deque<pend_t> pend;
  // How far this call looks.  A scheduler offers a block everything its input buffer holds; a front that cut all of it at once would then
  // find nothing new in the next call and could only wait for its own gather.  With deferred consume it takes a THIRD of the largest
  // window it has been offered per call: two gathers are on their way while the third of the buffer they have left is refilled —
  // the upstream block writes, this one cuts, the device copies, all at once.  (A window too small for that — under ~3 x 20 us of items —
  // just runs at the gathers' latency.)
  int hwm_window = 0;
  int cap_window(engine_holder& E, int used, int nproc)
  {
    hwm_window = std::max(hwm_window, nproc);
    if (!E.d_cfg.defer_consume) return nproc;
    const int third = std::max(hwm_window / (DEPTH + 1), 4096);
    return std::min(nproc, used + third);
  }
  void add(const void* src, uint64_t dst_off, uint64_t n, uint32_t fmt)
  {
    if (!n) return;
    const size_t isz = fmt == C8P_FMT_F32 ? 4 : 8;
    pending_bytes += isz * (size_t)n;
    // (a packet cut by a call boundary continues where the last piece ended, in the buffer and in the arena: one piece)
    if (!pieces.empty()) {
      c8p_piece& b = pieces.back();
      if (b.fmt == fmt && (const char*)b.src + b.n * isz == (const char*)src && b.dst_off + b.n == dst_off) {
        b.n += n;
        src_hi = std::max(src_hi, (uintptr_t)src + isz * (size_t)n);
        return;
      }
    }
    src_lo = std::min(src_lo, (uintptr_t)src); src_hi = std::max(src_hi, (uintptr_t)src + isz * (size_t)n);
    c8p_piece pc;
    memset(&pc, 0, sizeof(pc));
    pc.src = src; pc.dst_off = dst_off; pc.n = n; pc.trusted = 1; pc.fmt = fmt; pc.scale = 1.0f;
    pc.rsv = unordered ? C8P_PIECE_UNORDERED : 0u;     // (every piece has its own place in the batch's arena: the gathers of a lane may pass each other)
    pieces.push_back(pc);
  }
  void flush(engine_holder& E, void* dst)
  {
    if (pieces.empty()) return;
    // (at most C8P_GATHER_INLINE pieces a launch: a longer list goes through the engine's one table in pinned memory, and the call then
    // first waits for everything the copy stream holds — two antennas are two pieces per packet, 150 of them in a third of a window)
    for (size_t i = 0; i < pieces.size(); i += C8P_GATHER_INLINE)
      E.check(E.d_api.copy_gather(E.d_ctx, static_cast<c8p_c32*>(dst), pieces.data() + i, (int32_t)std::min<size_t>(C8P_GATHER_INLINE, pieces.size() - i)),
              "c8p_copy_gather");
    pieces.clear();
    if (stats.on) { stats.flushes++; stats.flush_bytes += (double)pending_bytes; }
    pending_bytes = 0;
    dirty = true;
  }
  // at the end of a call that looked at the items [rd, rd + used): how many of them may be consumed now
  int settle(engine_holder& E, uint64_t rd, int used, phase_clock& ph)
  {
    const bool fresh = rd + (uint64_t)used > scan_abs;                                // this call cut items no call had seen
    scan_abs = std::max(scan_abs, rd + (uint64_t)used);
    if (!dirty && pend.empty()) return used;                                         // nothing of this window is on its way
    const c8p_api& A = E.d_api;
    if (dirty) {
      const uint64_t m = A.copy_mark(E.d_ctx);
      if (!m) throw std::runtime_error("c8p_copy_mark failed");
      pend.push_back(pend_t{m, scan_abs, now_tick(), src_lo, src_hi});
      src_lo = UINTPTR_MAX; src_hi = 0;
      dirty = false;
    } else if (!pend.empty()) pend.back().scan = scan_abs;                           // (items without a gather of their own: skipped ones)
    sub_scope w(ph, phase_clock::GWAIT);
    // with deferral: all but the newest DEPTH gathers are waited for; a call that brought nothing new settles everything (the
    // scheduler has nothing more to offer until these items are let go of)
    const size_t keep = (E.d_cfg.defer_consume && fresh) ? (size_t)DEPTH : 0;
    uint64_t done_scan = rd;
    const tick_t t0 = stats.on ? now_tick() : tick_t();
    const bool waited = pend.size() > keep;
    const tick_t t_enq = waited ? pend.front().t_enq : tick_t();
    while (pend.size() > keep) {
      E.check(A.copy_wait_mark(E.d_ctx, pend.front().mark), "c8p_copy_wait_mark");
      done_scan = std::max(done_scan, pend.front().scan);
      pend.pop_front();
    }
    if (stats.on) {
      const double u = std::chrono::duration<double, std::micro>(now_tick() - t0).count();
      stats.calls++; stats.items += (uint64_t)used;
      if (waited) {
        const double age = std::chrono::duration<double, std::micro>(now_tick() - t_enq).count();
        stats.waits++; stats.us += u; stats.worst = std::max(stats.worst, u);
        if (u > 5.0) { stats.slow++; stats.age_slow += age; } else stats.age_fast += age;
      }
    }
    if (pend.empty()) return used;
    return (int)std::min<uint64_t>(done_scan - rd, (uint64_t)used);
  }
  // buffer_will_move: the newest gather that reads [base, base + bytes) has finished when this returns (a mark covers what came before it)
  void wait_range(engine_holder& E, const void* base, size_t bytes)
  {
    const uintptr_t lo = (uintptr_t)base, hi = lo + bytes;
    for (auto it = pend.rbegin(); it != pend.rend(); ++it)
      if (it->lo < hi && lo < it->hi) { E.check(E.d_api.copy_wait_mark(E.d_ctx, it->mark), "c8p_copy_wait_mark"); return; }
  }
  uintptr_t src_lo = UINTPTR_MAX, src_hi = 0;    // the pieces enqueued since the last mark
  bool dirty = false;                    // a gather has been enqueued since the last mark
};

// ================================================================================================ demod / demod2
class demod_front : public demod, public demod2
{
public:
  demod_front(int nports, int mupos, int mugid, const c8p_api& api, const batching_cfg& cfg)
      : d_eng(api, cfg), d_nports(nports), d_mupos(mupos), d_mugid(mugid),
        d_st{stage(&d_eng.d_api), stage(&d_eng.d_api), stage(&d_eng.d_api), stage(&d_eng.d_api)},
        d_timer(d_mu, [this](tick_t& w) { return next_deadline(w); }, [this] { on_timer(); })
  {
    d_st[0].state = stage::FILLING;
    // batches in flight: one less than the engine keeps slots (a submit into a slot whose batch is still running blocks inside the
    // engine), at least one, at most the ring allows
    d_max_fly = (int)std::max<int64_t>(1, std::min<int64_t>(NST - 1, api.counter(d_eng.d_ctx, C8P_CNT_SLOTS) - 1));
    if (cfg.flush_when_idle) d_timer.start();
  }
  ~demod_front() override
  {
    d_reg.leave();
    d_timer.stop();
    if (d_scat_stats.launches && env_int("C8P_GATHER_STATS", 0))
      fprintf(stderr, "llr scatter: %ld launches of %.0f items; waits: throttle %ld (%.0f us), space taken %ld (%.0f us), no progress %ld (%.0f us)\n",
              d_scat_stats.launches, d_scat_stats.items / (double)d_scat_stats.launches, d_scat_stats.n[0], d_scat_stats.us[0], d_scat_stats.n[1],
              d_scat_stats.us[1], d_scat_stats.n[2], d_scat_stats.us[2]);
    fused_purge(d_fused_ids);                      // the batches this front parked and no decode picked up
    llr_purge(d_llr_ids);
    d_outq.clear(); d_scat.clear(); d_pool.clear();    // (results own device memory of d_eng's ctx)
    for (stage& st : d_st) { d_eng.dev_release(st.d_iq); d_eng.dev_release(st.d_chan); }
  }

  // the io_signature's buffer type (c8p_blocks.h): pinned host memory for the item ports; in fused mode the output items are never written
  bool wants_pinned_input(unsigned) const override { return d_eng.d_cfg.pinned_io; }
  bool wants_pinned_output(unsigned) const override { return d_eng.d_cfg.pinned_io && !d_eng.d_cfg.fused; }
  void* buffer_alloc(size_t bytes) override { return d_eng.d_api.host_alloc(bytes); }
  void buffer_free(void* p) override { d_eng.d_api.host_free(p); }

  void forecast(int noutput_items, gr_vector_int& req) override   // demod_impl.cc:53-57
  {
    // results waiting for output space need no input to be handed out (the timer thread collected them while none arrived)
    std::lock_guard<std::mutex> g(d_mu);
    for (auto& r : req) r = (d_outq.empty() && !d_ready && !d_inflight) ? noutput_items : 0;
  }
  int output_ahead() override { std::lock_guard<std::mutex> g(d_mu); return (int)(d_ready + d_inflight); }
  int fused_state() override { std::lock_guard<std::mutex> g(d_mu); return d_mode_known ? (d_eng.d_cfg.fused ? 1 : 0) : -1; }
  void quiesce(const void* base, size_t bytes) override
  {
    std::lock_guard<std::mutex> g(d_mu);
    d_lane.wait_range(d_eng, base, bytes);
    const uintptr_t lo = (uintptr_t)base, hi = lo + bytes;
    for (auto it = d_scat.rbegin(); it != d_scat.rend(); ++it)
      if (it->lo < hi && lo < it->hi) { d_eng.check(d_eng.d_api.copy_wait_mark(d_eng.d_ctx, it->mark), "c8p_copy_wait_mark"); break; }
  }

  // a buffer moves (c8p_blocks.h, buffer_move_scope): the input side as quiesce(); items placed ahead in an output buffer that lies in
  // the range are waited for — every launch that writes them — and saved, and d_mu stays held until move_end has put them back
  int move_begin(const void* base, size_t bytes, int64_t timeout_us) override
  {
    if (!try_lock_for(d_mu, timeout_us)) return -1;
    bool held = false;
    try {
      d_lane.wait_range(d_eng, base, bytes);
      const uintptr_t lo = (uintptr_t)base, hi = lo + bytes, at = (uintptr_t)d_ahead_ptr;
      if (d_ready + d_inflight > 0 && d_ahead_ptr && at >= lo && at < hi) {
        while (!d_scat.empty()) scatter_collect(true, 1);            // all of [0, d_ready) is in the buffer now
        d_saved.assign(d_ahead_ptr, d_ahead_ptr + d_ready);
        held = true;
      }
    } catch (...) { d_mu.unlock(); throw; }
    if (!held) d_mu.unlock();
    return held ? 1 : 0;
  }
  void move_end(void* new_write_ptr) override      // (d_mu is held: move_begin returned true)
  {
    if (new_write_ptr && new_write_ptr != (void*)d_ahead_ptr) {
      memcpy(new_write_ptr, d_saved.data(), d_saved.size() * sizeof(float));
      d_ahead_ptr = static_cast<float*>(new_write_ptr);
      d_ahead_moves++;
    }
    d_saved.clear();
    d_mu.unlock();
  }

  std::string host_phase_us() override { std::lock_guard<std::mutex> g(d_mu); return d_ph.str(); }
  void host_phase_reset() override { std::lock_guard<std::mutex> g(d_mu); d_ph = phase_clock(); }

  int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& output_items) override
  {
    std::lock_guard<std::mutex> g(d_mu);
    if (!d_timer.error.empty()) throw std::runtime_error(d_timer.error);
    phase_clock::scope ph_all(d_ph, phase_clock::CUT);
    reap(false);                                   // batches whose completion word has arrived: their packets join the output queue
    if (!d_mode_known) {
      d_in_dev = d_eng.d_cfg.pinned_io;
      for (int p = 0; p < d_nports; p++) {
        const size_t cap = d_io->input_buffer_bytes((unsigned)p);
        d_in_dev = d_in_dev && d_io->input_is_pinned((unsigned)p) && (cap == 0 || cap >= gather_min_buffer());
      }
      // the cfg left fused / plain open: fused when nothing but decode mirrors reads the LLR port (c8p_blocks.h, batching_cfg::fused_auto)
      if (!d_eng.d_cfg.fused && d_eng.d_cfg.fused_auto && d_io->output_feeds_only_decode(0) == 1) {
        d_eng.d_cfg.fused = true;
        d_lean = d_eng.d_cfg.fused_lean_tags;      // nothing but decode mirrors reads this edge: the handle is the only tag they need
      }
      d_out_dev = d_eng.d_cfg.pinned_io && !d_eng.d_cfg.fused && d_io->output_is_pinned(0);
      d_mode_known = true;
    }
    int nproc = ninput_items[0];
    for (int p = 1; p < d_nports; p++) nproc = std::min(nproc, ninput_items[p]);
    const uint64_t rd = d_io->nitems_read(0);
    // A pinned output buffer: the LLR items that are ready start moving — one kernel that writes them where the scheduler wants them —
    // and are REPORTED at the end of the call in which they have arrived (scatter_place / scatter_report)
    if (d_out_dev) {
      // items placed ahead sit where this call's space begins: a scheduler that moved its write position without saying so
      // (buffer_move_scope, or calls that drain output_ahead first) would have this call report whatever lies here instead — say so
      if (d_ready + d_inflight > 0 && d_ahead_ptr != static_cast<float*>(output_items[0]))
        throw std::runtime_error("c8p_blocks: demod's output space moved while " + std::to_string(d_ready + d_inflight) +
                                 " LLR items placed ahead of produce() were not reported (the buffer must bracket its moves with "
                                 "c8p::buffer_move_scope, or call general_work(output_ahead()) until it is 0 first)");
      scatter_place(static_cast<float*>(output_items[0]), noutput_items);
    }
    int used = 0;
    if (d_in_dev) {                                // items cut by an earlier call whose gather had not been waited for: still here
      if (d_lane.scan_abs < rd) d_lane.scan_abs = rd;
      used = (int)std::min<uint64_t>(d_lane.scan_abs - rd, (uint64_t)nproc);
    }
    nproc = d_lane.cap_window(d_eng, used, nproc);
    const int used_before = used;                  // (items beyond this are new to the front: cutting them is progress, consumed or not)
    d_tc.load(d_io, rd + (uint64_t)used, rd + (uint64_t)nproc);
    while (used < nproc) {
      if (d_state == S_RDTAG) {
        tag_span tg;
        uint64_t next = 0;
        if (!d_tc.at(rd + (uint64_t)used, rd + (uint64_t)nproc, tg, next)) { used = (int)(next - rd); continue; }   // the reference would wait here forever; we resynchronise
        stage& st = d_st[d_fill];
        memset(&d_cur, 0, sizeof(d_cur));
        d_cur.seq = d_cur.lsig_mcs = d_cur.lsig_len = d_cur.lsig_nsamp = -1;
        const tag_value* ch = nullptr;
        for (const tag_t* t = tg.first; t != tg.last; ++t) {        // one pass over the item's tags (signal_impl.cc:135-143 writes eight)
          const std::string& k = t->key;
          const tag_value& v = t->value;
          const char* kc = k.data();
          switch (k.size()) {                                       // (by length, then one memcmp: no chain of string compares)
            case 3:
              if (!memcmp(kc, "cfo", 3)) d_cur.cfo = v.as_float();
              else if (!memcmp(kc, "snr", 3)) d_cur.snr = v.as_float();
              else if (!memcmp(kc, "seq", 3)) d_cur.seq = (int32_t)v.as_long();
              else if (!memcmp(kc, "mcs", 3)) d_cur.lsig_mcs = (int32_t)v.as_long();
              else if (!memcmp(kc, "len", 3)) d_cur.lsig_len = (int32_t)v.as_long();
              break;
            case 4:
              if (!memcmp(kc, "rssi", 4)) d_cur.rssi = v.as_float();
              else if (!memcmp(kc, "chan", 4)) ch = &v;
              break;
            case 5:
              if (!memcmp(kc, "nsamp", 5)) d_cur.lsig_nsamp = (int32_t)v.as_long();
              break;
            default: break;
          }
        }
        d_cur.n_ant = (uint8_t)d_nports; d_cur.mupos = (uint8_t)(d_mupos ? 1 : 0);
        d_cur.n_items = (uint32_t)std::max(d_cur.lsig_nsamp, 0) + 320u;     // demod_impl.cc:92
        d_cur.chan_off = (uint32_t)st.chan.size();
        gr_complex* c = st.chan.grow(64);
        if (ch && ch->v.size() >= 64) memcpy(static_cast<void*>(c), ch->v.data(), 64 * sizeof(gr_complex));
        else for (int i = 0; i < 64; i++) c[i] = ch && (int)ch->v.size() > i ? ch->v[i] : gr_complex(0.f, 0.f);
        for (int p = 0; p < d_nports; p++) d_cur.iq_off[p] = iq_alloc(st, d_cur.n_items);
        d_have = 0;
        d_state = S_COLLECT;
      }
      stage& st = d_st[d_fill];
      const int take = (int)std::min<int64_t>(nproc - used, (int64_t)d_cur.n_items - d_have);
      for (int p = 0; p < d_nports; p++) {
        const gr_complex* src = static_cast<const gr_complex*>(input_items[p]) + used;
        if (d_in_dev) d_lane.add(src, d_cur.iq_off[p] + (uint64_t)d_have, (uint64_t)take, C8P_FMT_C32);
        else memcpy(static_cast<void*>(st.iq.data() + d_cur.iq_off[p] + d_have), src, sizeof(gr_complex) * (size_t)take);
      }
      d_have += take; used += take;
      if (d_in_dev && d_lane.want_flush(d_eng)) { sub_scope pg(d_ph, phase_clock::GATHER); d_lane.flush(d_eng, st.d_iq.p); }
      if (d_have == (int)d_cur.n_items) {
        if (st.pk.empty()) st.t_first = now_tick();
        st.pk.push_back(d_cur);
        d_state = S_RDTAG;
        if ((int)st.pk.size() >= d_eng.d_cfg.max_batch) submit(false);
      }
    }
    // the call's last pieces start moving; what else the call has to do — replay the batches that have arrived, hand out items — is done
    // while they do, and only then are the gathers asked about (settle)
    if (d_in_dev) { sub_scope pg(d_ph, phase_clock::GATHER); d_lane.flush(d_eng, d_st[d_fill].d_iq.p); }
    if (d_eng.d_cfg.flush_when_idle) age_check();
    const int drained = d_out_dev ? 0 : drain(static_cast<float*>(output_items[0]), noutput_items);
    int cons = used;
    if (d_in_dev) {
      { sub_scope pg(d_ph, phase_clock::GATHER); d_lane.flush(d_eng, d_st[d_fill].d_iq.p); }     // (age_check may have moved to a new stage: nothing left then)
      cons = d_lane.settle(d_eng, rd, used, d_ph);
    }
    d_io->consume_each(cons);
    // (the timer sleeps without a deadline while this front holds nothing: tell it when that may have changed — not on every call)
    if (!d_st[d_fill].pk.empty() || !d_fly.empty()) d_timer.poke();
    if (d_out_dev) {
      const int produced = scatter_report(noutput_items, cons > 0 || used > used_before);
      d_ahead_ptr = static_cast<float*>(output_items[0]) + produced;     // where what is still ahead sits = where the next call's space begins
      return produced;
    }
    return drained;
  }

  void flush() override      // complete packets go now; a packet still being received stays behind
  {
    std::lock_guard<std::mutex> g(d_mu);
    if (!d_st[d_fill].pk.empty()) submit(true);
    while (!d_fly.empty()) reap(true);             // everything in flight has been replayed
  }

  // the timer thread (holds d_mu): when does the oldest thing this front is sitting on reach max_wait_us?
  bool next_deadline(tick_t& when)
  {
    const auto w = std::chrono::microseconds(d_eng.d_cfg.max_wait_us);
    bool have = false;
    if (!d_st[d_fill].pk.empty()) { when = d_st[d_fill].t_first + w; have = true; }
    if (!d_fly.empty()) {
      const stage& fl = d_st[d_fly.front()];
      if (!have || fl.t_submit + w < when) { when = fl.t_submit + w; have = true; }
    }
    return have;
  }
  void on_timer()
  {
    const size_t before = d_outq.size();
    age_check();
    if (d_outq.size() > before && d_io) d_io->wake();      // no input is coming that would get general_work called: ask for a call
  }

  // latency bound: a gathered batch is submitted once its first packet has waited max_wait_us (it stays in flight); a batch in flight
  // is replayed when its completion word is there, and waited for once it is that old
  void age_check()
  {
    const long w = d_eng.d_cfg.max_wait_us;
    if (!d_st[d_fill].pk.empty() && us_since(d_st[d_fill].t_first) >= w) submit(false);
    reap(false);
    if (!d_fly.empty() && us_since(d_st[d_fly.front()].t_submit) >= w) reap(true);
  }

private:
  struct result {             // downloaded arenas of one batch, shared by its packets until they have been drained
    result(engine_holder* e) : eng(e), cell(e->d_cell), llr(&e->d_api) {}
    ~result()
    {
      if (!dev_llr.p) return;
      std::lock_guard<std::mutex> g(cell->mu);
      cell->dev_free(cell->ctx, dev_llr.p);        // (ctx may be gone: c8p_dev_free(NULL, p) just frees)
    }
    engine_holder* eng;                // (only valid while the front lives: used by acquire() on its thread)
    std::shared_ptr<engine_holder::ctx_cell> cell;
    uint64_t n_llr = 0;               // floats of the arena copy
    pinned_vec<float> llr;            // pageable output buffer: the LLR arena on the host
    dev_buf dev_llr;                  // pinned output buffer: a device copy of the LLR arena (the engine's own is recycled with its slot)
  };
  struct stage {              // one batch being gathered / in flight: its arenas must stay untouched until the submit has completed
    explicit stage(const c8p_api* a) : iq(a), chan(a), tags(a), muchan(a), flag(a) {}
    enum { FREE, FILLING, INFLIGHT } state = FREE;
    pinned_vec<gr_complex> iq, chan;    // host staging (iq: pageable input buffers only)
    dev_buf d_iq, d_chan;               // pinned input buffers: the batch's arenas on the device, filled by the gathers
    size_t iq_items = 0;                // ... and the items laid out in d_iq so far
    std::vector<c8p_pkt_in> pk;
    tick_t t_first, t_submit;   // when the first packet was complete / when the batch was submitted
    pinned_vec<c8p_pkt_tags> tags;      // downloads enqueued at submit time, right behind the kernels (one c8p_fetch_gather): they run while
    pinned_vec<gr_complex> muchan;      // the next batches are gathered; the NDP captures ride along (1 KB per packet)
    pinned_vec<uint32_t> flag;          // ... and behind them the engine writes flag_want here: reap() looks at this word
    uint32_t flag_want = 0;
    std::shared_ptr<result> res;
    std::shared_ptr<fused_batch> fb;    // fused mode: the decoded results, handed to decode as they are
    long fb_id = 0;
  };
  struct out_pkt {             // one accepted packet waiting for output space: the downloaded tag record (its stream tags are made from it
    c8p_pkt_tags t;            // when its first item goes out: emit_tags), where its LLR items are, how many have gone out
    long handle = -1;          // fused mode: the "c8p_fused" value; plain mode with llr_handle: the "c8p_llr" value
    std::shared_ptr<result> res;
    const float* llr = nullptr;       // host copy (pageable output) or device copy (pinned output); null (fused mode): the items are
    bool on_device = false;           // produced, their values are not written
    size_t n = 0, sent = 0;
    std::vector<gr_complex> mu;       // an NDP's capture (tag "mu2x1chan"), empty otherwise
  };

  // the reference's tags on the packet's first item (demod_impl.cc:224-253, demod2_impl.cc:230-250), in its order, + this front's handle
  void emit_tags(const out_pkt& o, uint64_t off)
  {
    static const std::string K_cfo("cfo"), K_snr("snr"), K_s0("sssnr0"), K_s1("sssnr1"), K_rssi("rssi"), K_format("format"), K_mcs("mcs"), K_len("len"),
        K_cr("cr"), K_ampdu("ampdu"), K_trellis("trellis"), K_mu("mu2x1chan"), K_total("total"), K_fused("c8p_fused"), K_llr("c8p_llr");
    const c8p_pkt_tags& t = o.t;
    auto L = [&](const std::string& k, long v) { d_io->add_item_tag(0, off, k, tag_value::from_long(v)); };
    if (d_lean && o.handle >= 0) { L(K_fused, o.handle); return; }     // (the record is behind the handle: fused_batch::tags)
    auto F = [&](const std::string& k, float v) { d_io->add_item_tag(0, off, k, tag_value::from_float(v)); };
    F(K_cfo, t.cfo); F(K_snr, t.snr);
    if (t.format == C8P_F_VHT) { F(K_s0, t.sssnr0); if (d_nports == 2 && t.nss == 2) F(K_s1, t.sssnr1); }
    F(K_rssi, t.rssi);
    L(K_format, t.format); L(K_mcs, t.mcs); L(K_len, t.len); L(K_cr, t.cr); L(K_ampdu, t.ampdu); L(K_trellis, t.trellis);
    if (t.is_ndp) d_io->add_item_tag(0, off, K_mu, tag_value::from_c32vector(o.mu));
    L(K_total, t.total);
    if (o.handle >= 0) L(d_eng.d_cfg.fused ? K_fused : K_llr, o.handle);
  }

  // where the next packet's items go in the batch's arena
  uint64_t iq_alloc(stage& st, uint32_t n)
  {
    if (!d_in_dev) { const uint64_t off = st.iq.size(); st.iq.grow(n); return off; }
    const uint64_t off = st.iq_items;
    st.iq_items += n;
    if (st.iq_items * sizeof(gr_complex) > st.d_iq.cap) {
      // (pieces carry offsets; the base they are added to is about to change: what was cut so far goes to the old arena first)
      d_lane.flush(d_eng, st.d_iq.p);
      const size_t first = env_int("C8P_ARENA_MIN", 0) ? 0 : (size_t)std::max(d_eng.d_cfg.max_batch, 1) * 4096u * (size_t)d_nports * sizeof(gr_complex);
      d_eng.dev_reserve(st.d_iq, std::max(first, (size_t)st.iq_items * sizeof(gr_complex)), (size_t)off * sizeof(gr_complex));
    }
    return off;
  }

  // A result's pinned buffers are sized for the largest batch seen so far by ANY result (d_hwm_*): batches cut short by the
  // latency bound have all sizes, and a buffer that grows batch by batch is a hipHostMalloc + hipHostFree of tens of megabytes
  // each time (measured: 1.5-2.9 us per packet of the chain, profiles/r3c/blocks_phases.txt)
  std::shared_ptr<result> acquire(size_t llr_items, size_t n_pkt)
  {
    // (scaled to a full batch at once: batch sizes are random, and a mark that creeps up record by record re-allocates every
    // pooled buffer at each new record)
    const size_t full = (size_t)std::max(d_eng.d_cfg.max_batch, 1), np = std::max<size_t>(n_pkt, 1);
    d_hwm_llr = std::max(d_hwm_llr, (llr_items + np - 1) / np * full);
    std::shared_ptr<result> r;
    for (auto& q : d_pool) if (q.use_count() == 1) { r = q; break; }
    if (!r) { d_pool.push_back(std::make_shared<result>(&d_eng)); r = d_pool.back(); }
    r->llr.clear();
    if (llr_items && !d_out_dev) r->llr.reserve(d_hwm_llr);
    if (llr_items && d_out_dev) d_eng.dev_reserve(r->dev_llr, d_hwm_llr * sizeof(float), 0);
    return r;
  }
  std::shared_ptr<fused_batch> acquire_fused(size_t n_pkt, size_t pdu_bytes)
  {
    const size_t full = (size_t)std::max(d_eng.d_cfg.max_batch, 1), np = std::max<size_t>(n_pkt, 1);
    d_hwm_pkt = std::max(d_hwm_pkt, std::max(n_pkt, full));
    d_hwm_pdu = std::max(d_hwm_pdu, (pdu_bytes + np - 1) / np * full);
    std::shared_ptr<fused_batch> r;
    for (auto& q : d_fpool) if (q.use_count() == 1) { r = q; break; }     // (decode has let go of it and it has left the registry)
    if (!r) { d_fpool.push_back(std::make_shared<fused_batch>(&d_eng.d_api)); r = d_fpool.back(); }
    r->outs.reserve(d_hwm_pkt); r->pdu.reserve(d_hwm_pdu);
    r->outs.resize_uninit(n_pkt); r->pdu.resize_uninit(pdu_bytes);
    return r;
  }

  int free_stage() const
  {
    for (int i = 0; i < NST; i++) if (d_st[i].state == stage::FREE) return i;
    return -1;
  }

  // Submit the batch being gathered and go on with a free stage.  The batch's downloads and a completion word follow its kernels
  // (c8p_fetch_gather); reap() replays it once the word has arrived — the engine is not asked (no c8p_wait: a stream synchronisation
  // costs 5-10 us a batch even when the batch is long done).  At most d_max_fly batches are in flight: the oldest is waited for first.
  // last: return when nothing is in flight.
  void submit(bool last)
  {
    while ((int)d_fly.size() >= d_max_fly) reap(true);
    stage& st = d_st[d_fill];
    const c8p_api& A = d_eng.d_api;
    // a packet that is still being received lives at the end of this stage's arenas: it moves to the next stage
    std::vector<gr_complex> part_iq[2], part_chan;
    uint64_t part_off[2] = {0, 0};
    if (d_state == S_COLLECT) {
      for (int p = 0; p < d_nports; p++) {
        part_off[p] = d_cur.iq_off[p];
        if (!d_in_dev) part_iq[p].assign(st.iq.data() + d_cur.iq_off[p], st.iq.data() + d_cur.iq_off[p] + d_have);
      }
      part_chan.assign(st.chan.data() + d_cur.chan_off, st.chan.data() + d_cur.chan_off + 64);
    }
    c8p_batch_in b;
    memset(&b, 0, sizeof(b));
    b.n_pkt = (int32_t)st.pk.size(); b.pkts = st.pk.data();
    if (!d_in_dev) {
      b.mem = C8P_MEM_HOST;
      b.iq = reinterpret_cast<const c8p_c32*>(st.iq.data()); b.n_iq = st.iq.size();
      b.chan = reinterpret_cast<const c8p_c32*>(st.chan.data()); b.n_chan = st.chan.size();
    } else {
      // the items are on the device already or on their way (the calls' gathers); the channel estimates follow on the same stream, and
      // the submit's own stream waits for all of it ON THE DEVICE (c8p_copy_fence): no host thread waits for the link
      sub_scope pg(d_ph, phase_clock::GATHER);
      d_lane.flush(d_eng, st.d_iq.p);
      d_eng.dev_reserve(st.d_chan, std::max<size_t>(st.chan.size(), (size_t)std::max(d_eng.d_cfg.max_batch, 1) * 64u) * sizeof(gr_complex), 0);
      d_eng.check(A.copy_async(d_eng.d_ctx, st.d_chan.p, st.chan.data(), st.chan.size() * sizeof(gr_complex)), "chan upload");
      d_eng.check(A.copy_fence(d_eng.d_ctx), "c8p_copy_fence");
      b.mem = C8P_MEM_DEVICE;
      b.iq = static_cast<const c8p_c32*>(st.d_iq.p); b.n_iq = st.iq_items;
      b.chan = static_cast<const c8p_c32*>(st.d_chan.p); b.n_chan = st.chan.size();
    }
    sub_scope ph(d_ph, phase_clock::SUBMIT);
    d_eng.check(d_eng.d_cfg.fused ? A.rx(d_eng.d_ctx, &b) : A.demod(d_eng.d_ctx, &b));
    st.t_submit = now_tick();
    {
      sub_scope phq(d_ph, phase_clock::ENQ);
      // the batch's downloads — tags, NDP captures; fused mode: per-packet results + the PDU arena — and, behind them, the completion
      // word, with ONE launch behind its kernels (c8p_fetch_gather: the device writes the pinned buffers)
      const size_t n = st.pk.size();
      st.tags.resize_uninit(n);
      st.muchan.resize_uninit(128 * n);
      if (st.flag.empty()) st.flag.resize_uninit(1);
      st.flag_want = ++d_flag_seq ? d_flag_seq : ++d_flag_seq;      // never 0
      __atomic_store_n(st.flag.data(), 0u, __ATOMIC_RELAXED);
      c8p_fetch_piece fp[5];
      int nfp = 0;
      auto piece = [&](int which, void* dst, uint64_t bytes, uint64_t off) {
        c8p_fetch_piece& f = fp[nfp++];
        memset(&f, 0, sizeof(f));
        f.which = which; f.offset = off; f.bytes = bytes; f.dst = dst;
      };
      piece(C8P_BUF_TAGS, st.tags.data(), n * sizeof(c8p_pkt_tags), 0);
      piece(C8P_BUF_MUCHAN, st.muchan.data(), n * 128 * sizeof(gr_complex), 0);
      uint64_t bytes = 0, pbytes = 0;
      {
        sub_scope pha(d_ph, phase_clock::ALLOC);
        if (!d_eng.d_cfg.fused) {
          A.buffer(d_eng.d_ctx, C8P_BUF_LLR, &bytes);
          st.res = acquire(bytes / sizeof(float), n);
          st.res->n_llr = bytes / sizeof(float);
          if (!d_out_dev) st.res->llr.resize_uninit(bytes / sizeof(float));
        } else {
          A.buffer(d_eng.d_ctx, C8P_BUF_PDU, &pbytes);
          st.fb = acquire_fused(n, pbytes);
        }
      }
      if (!d_eng.d_cfg.fused) {       // the whole LLR arena (laid out for the worst case per n_items: ~1.8x what VHT packets use):
        // to the host (a pageable output buffer: drain() copies the packets' items out of it) or into a device buffer of this front
        // (a pinned one: scatter_place() moves the items from there, however long the scheduler takes to offer the space)
        if (bytes) d_eng.check(A.fetch_async(d_eng.d_ctx, C8P_BUF_LLR, d_out_dev ? st.res->dev_llr.p : (void*)st.res->llr.data(), 0, bytes), "llr");
      } else {
        piece(C8P_BUF_OUT, st.fb->outs.data(), n * sizeof(c8p_pkt_out), 0);
        if (pbytes) piece(C8P_BUF_PDU, st.fb->pdu.data(), pbytes & ~(uint64_t)3, 0);
      }
      piece(C8P_BUF_FLAG, st.flag.data(), 0, st.flag_want);
      d_eng.check(A.fetch_gather(d_eng.d_ctx, fp, nfp), "c8p_fetch_gather");
    }
    ph.end();
    // the batch is in flight; go on with a free stage (there is one: the ring has more stages than batches may be in flight)
    const int old = d_fill;
    st.state = stage::INFLIGHT;
    d_fly.push_back(old);
    if (last) while (!d_fly.empty()) reap(true);
    int nx = -1;
    // (device arenas: a packet that is still being received always moves to ANOTHER stage's arena, never inside one)
    for (int i = 0; i < NST; i++) if (i != old && d_st[i].state == stage::FREE) { nx = i; break; }
    if (nx < 0) throw std::runtime_error("c8p_blocks: no free stage");
    d_st[nx].state = stage::FILLING;
    d_fill = nx;
    if (d_state == S_COLLECT) {
      stage& from = d_st[old];
      stage& to = d_st[nx];
      d_cur.chan_off = (uint32_t)to.chan.size();
      memcpy(static_cast<void*>(to.chan.grow(64)), part_chan.data(), 64 * sizeof(gr_complex));
      for (int p = 0; p < d_nports; p++) {
        d_cur.iq_off[p] = iq_alloc(to, d_cur.n_items);
        if (!d_have) continue;
        if (!d_in_dev) memcpy(static_cast<void*>(to.iq.data() + d_cur.iq_off[p]), part_iq[p].data(), sizeof(gr_complex) * (size_t)d_have);
        else                                                        // device to device, behind the gathers that brought the items
          d_eng.check(A.copy_async(d_eng.d_ctx, static_cast<gr_complex*>(to.d_iq.p) + d_cur.iq_off[p],
                                   static_cast<const gr_complex*>(from.d_iq.p) + part_off[p], sizeof(gr_complex) * (size_t)d_have), "partial packet move");
      }
    }
  }

  // Replays every in-flight batch whose completion word has arrived, oldest first (block: waits for the oldest one's).  The word is
  // pinned host memory the engine writes behind the batch's kernels and downloads: a load, no call into the engine.  Caller holds d_mu.
  bool reap(bool block)
  {
    bool any = false;
    while (!d_fly.empty()) {
      stage& st = d_st[d_fly.front()];
      if (__atomic_load_n(st.flag.data(), __ATOMIC_ACQUIRE) != st.flag_want) {
        if (!block) break;
        sub_scope pw(d_ph, phase_clock::WAIT);
        int spins = 0;
        while (__atomic_load_n(st.flag.data(), __ATOMIC_ACQUIRE) != st.flag_want) {
          if (++spins < 4000) { __builtin_ia32_pause(); continue; }
          std::this_thread::sleep_for(std::chrono::microseconds(10));
          if (us_since(st.t_submit) > 20000000) throw std::runtime_error("c8p_blocks: a batch's completion word did not arrive within 20 s");
        }
      }
      block = false;
      sub_scope ph(d_ph, phase_clock::REPLAY);
      d_local.clear();
      replay(st, d_local);
      for (out_pkt& o : d_local) d_outq.push_back(std::move(o));
      if (d_eng.d_cfg.fused) {
        d_fused_ids.push_back(st.fb_id);
        if (d_fused_ids.size() > 2 * C8P_FUSED_MAX) d_fused_ids.erase(d_fused_ids.begin(), d_fused_ids.begin() + (std::ptrdiff_t)C8P_FUSED_MAX);
      }
      st.res.reset(); st.fb.reset();
      st.pk.clear(); st.iq.clear(); st.chan.clear(); st.iq_items = 0;
      st.state = stage::FREE;
      d_fly.pop_front();
      any = true;
    }
    return any;
  }

  // one completed batch -> out-packets (the reference's tags, and where its LLR items are)
  void replay(stage& st, std::vector<out_pkt>& out)
  {
    const size_t n = st.pk.size();
    const c8p_pkt_tags* tags = st.tags.data();
    const bool fused = d_eng.d_cfg.fused;
    std::shared_ptr<result> res = st.res;
    long fb_id = 0;
    if (fused) {                                   // the whole batch behind ONE handle; a packet's tag names its index
      fb_id = g_fused_next.fetch_add(1);
      st.fb_id = fb_id;
      if (d_lean) st.fb->tags.assign(tags, tags + n); else st.fb->tags.clear();
      fused_park(fb_id, st.fb);
    }
    long lb_id = 0;
    if (!fused && d_out_dev && d_eng.d_cfg.llr_handle && res && res->dev_llr.p) {   // the arena copy behind a handle (batching_cfg::llr_handle)
      std::shared_ptr<llr_batch> lb = std::make_shared<llr_batch>();
      lb->keep = res; lb->base = static_cast<const float*>(res->dev_llr.p); lb->n_floats = res->n_llr;
      lb->llr_off.resize(n);
      for (size_t p = 0; p < n; p++) lb->llr_off[p] = tags[p].llr_off;
      lb_id = g_fused_next.fetch_add(1);
      llr_park(lb_id, std::move(lb));
      d_llr_ids.push_back(lb_id);
      if (d_llr_ids.size() > 128) d_llr_ids.erase(d_llr_ids.begin(), d_llr_ids.begin() + 64);
    }
    out.reserve(n);
    for (size_t p = 0; p < n; p++) {
      const c8p_pkt_tags& t = tags[p];
      if (t.status != C8P_ST_OK) continue;          // DEMOD_S_CLEAN without tags: nothing reaches decode
      out.emplace_back();
      out_pkt& o = out.back();
      o.t = t;
      if (t.is_ndp) o.mu.assign(st.muchan.data() + 128 * p, st.muchan.data() + 128 * p + 128);
      o.n = (size_t)std::max(t.total, 0);
      if (!fused) {
        o.res = res; o.on_device = d_out_dev;
        o.llr = (d_out_dev ? static_cast<const float*>(res->dev_llr.p) : res->llr.data()) + t.llr_off;
        if (lb_id) o.handle = (lb_id << C8P_FUSED_IDX_BITS) | (long)p;
      } else o.handle = (fb_id << C8P_FUSED_IDX_BITS) | (long)p;
    }
  }

  int drain(float* out, int noutput_items)
  {
    sub_scope ph(d_ph, phase_clock::DRAIN);
    int produced = 0;
    while (!d_outq.empty() && produced < noutput_items) {
      out_pkt& o = d_outq.front();
      if (o.sent == 0) emit_tags(o, d_io->nitems_written(0) + (uint64_t)produced);
      const size_t k = std::min(o.n - o.sent, (size_t)(noutput_items - produced));
      // (fused mode: the items exist and are counted, their VALUES ARE NOT WRITTEN — c8p_blocks.h; 57 KB of zeros per packet was a
      // third of this thread's time, and nothing downstream of a fused demod may read them)
      if (o.llr) memcpy(out + produced, o.llr + o.sent, sizeof(float) * k);
      o.sent += k; produced += (int)k;
      if (o.sent == o.n) d_outq.pop_front();
    }
    return produced;
  }

  // drain() for a pinned output buffer: the items of every packet that fits go from the front's device copy of the LLR arena into the
  // scheduler's buffer with ONE c8p_copy_gather (device -> pinned host: the shader cores write over PCIe), only the bytes the packets
  // use (the arena's layout is ~1.8x that), no host copy in between.  The items are written AHEAD: a call places them behind what
  // earlier calls placed (scatter_place, at the start of general_work) and reports what has ARRIVED by its end (scatter_report) —
  // [0, d_ready) of the space it is given is complete, [d_ready, d_ready + d_inflight) is being written by up to SCAT_DEPTH launches
  // (d_scat).  A scheduler's output pointer continues where produce() stopped, so the items placed by this call are where a later
  // call's space begins (c8p_blocks.h, output_ahead).  Nothing is waited for while the call makes progress some other way: an 8 MB
  // scatter takes 180 us, and the version that waited for the previous call's scatter at the start of every call spent 0.9 us per
  // packet doing so (profiles/r5q/gather_waits.txt; the chain did not get faster for it — the link carries both directions out of
  // one budget, DESIGN.md §7 — but the items are reported a call earlier).
  enum { SCAT_DEPTH = 3 };
  struct scat_t { uint64_t mark; size_t n; std::vector<std::shared_ptr<result>> hold; uintptr_t lo = 0, hi = 0; };   // (its arenas are being read until the mark; [lo, hi): where it writes)
  struct scat_stats_t { long n[3] = {0, 0, 0}; double us[3] = {0, 0, 0}; long launches = 0; double items = 0; } d_scat_stats;   // C8P_GATHER_STATS
  void scatter_collect(bool block_oldest, int why = 0)
  {
    const c8p_api& A = d_eng.d_api;
    while (!d_scat.empty()) {
      scat_t& sc = d_scat.front();
      if (block_oldest) {
        sub_scope w(d_ph, phase_clock::GWAIT);
        const tick_t t0 = now_tick();
        d_eng.check(A.copy_wait_mark(d_eng.d_ctx, sc.mark), "c8p_copy_wait_mark");
        d_scat_stats.n[why]++; d_scat_stats.us[why] += std::chrono::duration<double, std::micro>(now_tick() - t0).count();
        block_oldest = false;
      } else {
        const int done = A.copy_mark_done(d_eng.d_ctx, sc.mark);
        if (done < 0) d_eng.check(done, "c8p_copy_mark_done");
        if (!done) return;
      }
      d_ready += sc.n; d_inflight -= sc.n;
      d_scat.pop_front();
    }
  }
  void scatter_place(float* out, int noutput_items)
  {
    sub_scope ph(d_ph, phase_clock::DRAIN);
    const c8p_api& A = d_eng.d_api;
    scatter_collect(false);
    if (d_outq.empty()) return;
    while (d_scat.size() >= (size_t)SCAT_DEPTH) scatter_collect(true, 0);
    size_t pos = d_ready + d_inflight;
    const size_t pos0 = pos;
    // one launch takes a THIRD of the space at most: the space then holds up to SCAT_DEPTH launches, the next is placed while the ones
    // before it are still being written, and the link always has one queued (a launch that fills the whole space is the only one under
    // way, and the link idles from its end until the call that notices)
    const size_t stop = std::min((size_t)noutput_items, pos0 + std::max<size_t>((size_t)noutput_items / SCAT_DEPTH, 1u << 16));
    d_opieces.clear();
    scat_t sc;
    while (!d_outq.empty() && pos < stop) {
      out_pkt& o = d_outq.front();
      if (o.sent == 0) emit_tags(o, d_io->nitems_written(0) + (uint64_t)pos);
      const size_t k = std::min(o.n - o.sent, stop - pos);
      if (k && o.llr) {
        c8p_piece pc;
        memset(&pc, 0, sizeof(pc));
        pc.src = o.llr + o.sent; pc.dst_off = (uint64_t)pos; pc.n = k; pc.trusted = 1; pc.fmt = C8P_FMT_F32; pc.scale = 1.0f;
        pc.rsv = C8P_PIECE_TO_HOST;                  // (the link's other direction: its own copy stream, beside the samples coming down)
        d_opieces.push_back(pc);
      }
      o.sent += k; pos += k;
      if (o.sent == o.n) { if (sc.hold.empty() || sc.hold.back() != o.res) sc.hold.push_back(std::move(o.res)); d_outq.pop_front(); }
    }
    if (pos == pos0) return;
    if (d_opieces.empty()) {                         // (packets without items: nothing to wait for, but they keep their place in the order)
      if (d_scat.empty()) d_ready += pos - pos0; else { d_scat.back().n += pos - pos0; d_inflight += pos - pos0; }
      return;
    }
    for (size_t i = 0; i < d_opieces.size(); i += C8P_GATHER_INLINE)       // (short lists: each launch stays on the device -> host stream and enqueues only)
      d_eng.check(A.copy_gather(d_eng.d_ctx, reinterpret_cast<c8p_c32*>(out), d_opieces.data() + i,
                                (int32_t)std::min<size_t>(C8P_GATHER_INLINE, d_opieces.size() - i)), "c8p_copy_gather (llr out)");
    sc.mark = A.copy_mark(d_eng.d_ctx);
    if (!sc.mark) throw std::runtime_error("c8p_copy_mark failed");
    sc.n = pos - pos0;
    sc.lo = (uintptr_t)(out + pos0); sc.hi = (uintptr_t)(out + pos);
    d_scat_stats.launches++; d_scat_stats.items += (double)sc.n;
    d_inflight += sc.n;
    d_scat.push_back(std::move(sc));
  }
  // the end of the call: what has arrived is reported.  A call that has cut or consumed input waits for nothing.  One that has not and
  // would report nothing waits for its oldest scatter (a call without any progress is repeated at once: it would spin) — also when its
  // space is taken up by what is under way (a scheduler at the end of its buffer keeps offering what is left of it until the block
  // has reported everything it placed there)
  int scatter_report(int noutput_items, bool progressed)
  {
    sub_scope ph(d_ph, phase_clock::DRAIN);
    scatter_collect(false);
    if (!d_scat.empty()) {
      const bool taken = d_ready + d_inflight >= (size_t)noutput_items;
      if (!progressed) {
        if (d_ready == 0) scatter_collect(true, taken ? 1 : 2);
      } else if (taken && d_ready == 0 && d_outq.size() > 2 * (size_t)d_eng.d_cfg.max_batch)
        scatter_collect(true, 0);                    // (input is being cut faster than the link takes the items away: no further ahead than this)
    }
    const size_t produced = std::min(d_ready, (size_t)noutput_items);
    d_ready -= produced;
    return (int)produced;
  }

  enum { S_RDTAG, S_COLLECT } d_state = S_RDTAG;
  std::mutex d_mu;                                 // general_work / flush (scheduler thread) against the latency timer's thread
  phase_clock d_ph;
  engine_holder d_eng;
  int d_nports, d_mupos, d_mugid;
  bool d_mode_known = false, d_in_dev = false, d_out_dev = false;     // pinned scheduler buffers on the input ports / the output port
  bool d_lean = false;                             // fused by fused_auto with lean tags: one tag per packet, the tag records travel with the batch
  c8p_pkt_in d_cur;
  int d_have = 0;
  enum { NST = 4 };
  stage d_st[NST];
  int d_fill = 0;                                  // the stage being gathered (FILLING); d_fly: the ones in flight, in submit order
  std::deque<int> d_fly;
  int d_max_fly = 2;                               // batches in flight at most
  std::vector<out_pkt> d_local;
  uint32_t d_flag_seq = 0;
  tag_cursor d_tc;
  gather_lane d_lane;                              // pieces always belong to d_st[d_fill]
  std::vector<c8p_piece> d_opieces;
  size_t d_ready = 0, d_inflight = 0;              // scatter_place: items placed in the output buffer and not reported yet (complete / being written)
  float* d_ahead_ptr = nullptr;                    // ... where they begin (the last call's output pointer + what it reported)
  std::vector<float> d_saved;                      // move_begin .. move_end: the items ahead, while the buffer moves
  long d_ahead_moves = 0;
  std::deque<scat_t> d_scat;                       // ... the launches that write them, oldest first
  std::vector<std::shared_ptr<result>> d_pool;
  std::vector<std::shared_ptr<fused_batch>> d_fpool;
  size_t d_hwm_llr = 0, d_hwm_pkt = 0, d_hwm_pdu = 0;
  std::deque<out_pkt> d_outq;
  std::vector<long> d_fused_ids;                   // batches this front parked in g_fused (the last 256-512 of them)
  std::vector<long> d_llr_ids;                     // ... and in g_llr
  latency_timer d_timer;                           // its thread uses everything above
  front_registration d_reg{this};                  // buffer_will_move finds the front from here to the top of its destructor
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
    // Process at the end of every call in which a pending flag can make progress (the scheduler never calls a block with zero
    // input items, so waiting for a dry call would park detected packets until max_batch flags have piled up, with every raw
    // sample since the oldest of them buffered): the first pending flag needs 224 items behind it, or — once its L-SIG is
    // known and the packet was incomplete (C8P_SIG_NEED_MORE) — the rest of its packet.
    if (!d_evts.empty() && avail_end() != d_last_end &&
        ((int)d_evts.size() >= d_eng.d_cfg.max_batch || (d_eng.d_cfg.flush_when_idle && avail_end() >= d_need_end))) {
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
      if (!submitted) { d_need_end = e.idx + 224; done = k; break; }     // not enough items yet: S_DEMOD waits here
      const c8p_pkt_in& p = pk[ri];
      if (p.n_items == 0) {
        if (p.rsv[0] == C8P_SIG_NEED_MORE) {                              // S_COPY would wait for the rest of the packet
          d_need_end = e.idx + 224 + (uint64_t)std::max(p.lsig_nsamp, 0);
          done = k; break;
        }
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
    if (d_evts.empty()) d_need_end = 0;
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
  uint64_t d_need_end = 0;                       // stream position the first pending flag waits for (0: unknown, try at once)
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
  decode_front(bool ifdebug, const c8p_api& api, const batching_cfg& cfg)
      : d_eng(api, cfg), d_debug(ifdebug), d_st{stage(&d_eng.d_api), stage(&d_eng.d_api), stage(&d_eng.d_api), stage(&d_eng.d_api)},
        d_timer(d_mu, [this](tick_t& w) { return next_deadline(w); }, [this] { age_check(); })
  {
    memset(d_vhtMcsCount, 0, sizeof(d_vhtMcsCount));
    memset(d_legacyMcsCount, 0, sizeof(d_legacyMcsCount));
    memset(d_htMcsCount, 0, sizeof(d_htMcsCount));
    d_st[0].state = stage::FILLING;
    d_max_fly = (int)std::max<int64_t>(1, std::min<int64_t>(NST - 1, api.counter(d_eng.d_ctx, C8P_CNT_SLOTS) - 1));
    if (cfg.flush_when_idle) d_timer.start();       // (its results are messages: published from the timer thread, no call needed)
  }
  ~decode_front() override
  {
    d_reg.leave();
    d_timer.stop();
    for (stage& st : d_st) { d_eng.dev_release(st.d_llr); d_eng.dev_release(st.d_mu); }
  }

  void quiesce(const void* base, size_t bytes) override { std::lock_guard<std::mutex> g(d_mu); d_lane.wait_range(d_eng, base, bytes); }
  int move_begin(const void* base, size_t bytes, int64_t timeout_us) override
  {
    if (!try_lock_for(d_mu, timeout_us)) return -1;
    std::lock_guard<std::mutex> g(d_mu, std::adopt_lock);
    d_lane.wait_range(d_eng, base, bytes);
    return 0;
  }

  bool is_decode_front() const override { return true; }
  // the io_signature's buffer type (c8p_blocks.h): the LLR stream arrives in pinned host memory
  bool wants_pinned_input(unsigned) const override { return d_eng.d_cfg.pinned_io; }
  void* buffer_alloc(size_t bytes) override { return d_eng.d_api.host_alloc(bytes); }
  void buffer_free(void* p) override { d_eng.d_api.host_free(p); }

  std::string host_phase_us() override { std::lock_guard<std::mutex> g(d_mu); return d_ph.str(); }
  void host_phase_reset() override { std::lock_guard<std::mutex> g(d_mu); d_ph = phase_clock(); }

  void forecast(int noutput_items, gr_vector_int& req) override   // decode_impl.cc:54-58
  {
    req[0] = noutput_items + 160;
  }

  int general_work(int /*noutput_items*/, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& /*output_items*/) override
  {
    std::lock_guard<std::mutex> g(d_mu);
    if (!d_timer.error.empty()) throw std::runtime_error(d_timer.error);
    phase_clock::scope ph_all(d_ph, phase_clock::CUT);
    reap(false);                                   // batches whose completion word has arrived are published
    if (!d_mode_known) {
      const size_t cap = d_io->input_buffer_bytes(0);
      d_in_dev = d_eng.d_cfg.pinned_io && d_io->input_is_pinned(0) && (cap == 0 || cap >= gather_min_buffer());
      d_mode_known = true;
    }
    const float* in = static_cast<const float*>(input_items[0]);
    int nproc = ninput_items[0];
    const uint64_t rd = d_io->nitems_read(0);
    int used = 0;
    if (d_in_dev) {                                // items cut by an earlier call whose gather had not been waited for: still here
      if (d_lane.scan_abs < rd) d_lane.scan_abs = rd;
      used = (int)std::min<uint64_t>(d_lane.scan_abs - rd, (uint64_t)nproc);
    }
    nproc = d_lane.cap_window(d_eng, used, nproc);
    d_tc.load(d_io, rd + (uint64_t)used, rd + (uint64_t)nproc);
    while (used < nproc) {
      if (d_state == S_IDLE) {
        tag_span tg;
        uint64_t next = 0;
        if (!d_tc.at(rd + (uint64_t)used, rd + (uint64_t)nproc, tg, next)) { used = (int)(next - rd); continue; }
        memset(&d_cur, 0, sizeof(d_cur));
        d_cur.status = C8P_ST_OK;
        d_cur.format = d_cur.len = d_cur.total = d_cur.cr = d_cur.mcs = d_cur.ampdu = d_cur.trellis = 99999;
        d_cur_fused = -1;
        long llr_h = -1;
        const tag_value* mu = nullptr;
        for (const tag_t* t = tg.first; t != tg.last; ++t) {        // one pass over the item's tags (demod_impl.cc:224-253 writes up to fourteen)
          const std::string& k = t->key;
          const tag_value& v = t->value;
          const char* kc = k.data();
          switch (k.size()) {                                       // (by length, then one memcmp)
            case 2: if (!memcmp(kc, "cr", 2)) d_cur.cr = (int32_t)v.as_long(); break;
            case 3:
              if (!memcmp(kc, "len", 3)) d_cur.len = (int32_t)v.as_long();
              else if (!memcmp(kc, "mcs", 3)) d_cur.mcs = (int32_t)v.as_long();
              else if (!memcmp(kc, "cfo", 3)) d_cur.cfo = v.as_float();
              else if (!memcmp(kc, "snr", 3)) d_cur.snr = v.as_float();
              break;
            case 4: if (!memcmp(kc, "rssi", 4)) d_cur.rssi = v.as_float(); break;
            case 5:
              if (!memcmp(kc, "total", 5)) d_cur.total = (int32_t)v.as_long();
              else if (!memcmp(kc, "ampdu", 5)) d_cur.ampdu = (int32_t)v.as_long();
              break;
            case 6:
              if (!memcmp(kc, "format", 6)) d_cur.format = (int32_t)v.as_long();
              else if (!memcmp(kc, "sssnr0", 6)) d_cur.sssnr0 = v.as_float();
              else if (!memcmp(kc, "sssnr1", 6)) d_cur.sssnr1 = v.as_float();
              break;
            case 7:
              if (!memcmp(kc, "trellis", 7)) d_cur.trellis = (int32_t)v.as_long();
              else if (!memcmp(kc, "c8p_llr", 7)) llr_h = v.as_long();
              break;
            case 9:
              if (!memcmp(kc, "c8p_fused", 9)) d_cur_fused = v.as_long();
              else if (!memcmp(kc, "mu2x1chan", 9)) mu = &v;
              break;
            default: break;
          }
        }
        if (d_cur_fused >= 0) {
          // fused mode: the batch behind the handle (the next batch: the one before it has been published and leaves the registry).  With
          // lean tags (batching_cfg::fused_lean_tags) the handle is the packet's only tag and its tag record is in the batch
          const long id = d_cur_fused >> C8P_FUSED_IDX_BITS;
          const size_t idx = (size_t)(d_cur_fused & ((1L << C8P_FUSED_IDX_BITS) - 1));
          if (id != d_fb_id) {
            if (d_fb) fused_purge(std::vector<long>(1, d_fb_id));
            d_fb = fused_find(id);
            d_fb_id = id;
          }
          if (d_cur.total == 99999) {                                 // no "total" on the item: a lean packet, or a foreign handle
            if (d_fb && idx < d_fb->tags.size()) d_cur = d_fb->tags[idx];
            else d_cur.total = 0;                                     // (nothing to count: step over the item)
          }
        }
        d_cur.is_ndp = d_cur.trellis == 0;
        d_cur_ext = false;
        if (d_cur_fused < 0) {
          // the packet's LLR items: behind a handle of a demod mirror of this process (on the device already: the items in the
          // stream are skipped), or in the stream (gathered / copied into the batch's arena)
          if (llr_h >= 0 && d_eng.d_cfg.llr_handle) {
            const long id = llr_h >> C8P_FUSED_IDX_BITS;
            const size_t idx = (size_t)(llr_h & ((1L << C8P_FUSED_IDX_BITS) - 1));
            if (id != d_lb_id) {                   // the next batch: the one before it leaves the registry (its packets hold it until decoded)
              if (d_lb) llr_purge(std::vector<long>(1, d_lb_id));
              d_lb = llr_find(id);
              d_lb_id = id;
            }
            if (d_lb && idx < d_lb->llr_off.size()) { d_cur_ext = true; d_cur.llr_off = d_lb->llr_off[idx]; }
          }
          if (!d_cur_ext) {
            // (a batch reads ONE arena: packets of the stream do not join a batch of handle packets)
            if (d_st[d_fill].src_id != 0 && !d_st[d_fill].tags.empty()) submit(false);
            d_st[d_fill].src_id = 0; d_st[d_fill].src.reset();
            d_cur.llr_off = llr_alloc(d_st[d_fill], (size_t)std::max(d_cur.total, 0));
          }
          d_cur_mu = mu != nullptr;
          for (int i = 0; i < 128; i++) d_cur_muv[i] = mu && (int)mu->v.size() > i ? mu->v[i] : gr_complex(0.f, 0.f);
        }
        d_have = 0;
        d_state = S_COLLECT;
      }
      const int take = (int)std::min<int64_t>(nproc - used, (int64_t)d_cur.total - d_have);
      if (take > 0 && d_cur_fused < 0 && !d_cur_ext) {
        stage& st = d_st[d_fill];
        if (d_in_dev) {
          d_lane.add(in + used, d_cur.llr_off + (uint64_t)d_have, (uint64_t)take, C8P_FMT_F32);
          if (d_lane.want_flush(d_eng)) { sub_scope pg(d_ph, phase_clock::GATHER); d_lane.flush(d_eng, st.d_llr.p); }
        } else memcpy(st.llr.data() + d_cur.llr_off + d_have, in + used, sizeof(float) * (size_t)take);
      }
      d_have += std::max(take, 0); used += std::max(take, 0);
      if (d_have >= d_cur.total) {
        d_state = S_IDLE;
        if (d_cur.total <= 0) used++;        // a tag set announcing no items: step over its item (the reference livelocks here)
        if (d_cur_fused >= 0) publish_fused();
        else {
          const long want_src = d_cur_ext ? d_lb_id : 0;
          if (!d_st[d_fill].tags.empty() && d_st[d_fill].src_id != want_src) submit(false);      // one arena per batch
          stage& st = d_st[d_fill];
          st.src_id = want_src;
          if (d_cur_ext) st.src = d_lb;
          if (st.tags.empty()) st.t_first = now_tick();
          st.tags.push_back(d_cur);
          if (d_cur_mu) st.any_mu = true;
          st.mu.insert(st.mu.end(), d_cur_muv, d_cur_muv + 128);
          if ((int)st.tags.size() >= d_eng.d_cfg.max_batch) submit(false);
        }
      }
    }
    used = std::min(used, nproc);
    if (d_in_dev) { sub_scope pg(d_ph, phase_clock::GATHER); d_lane.flush(d_eng, d_st[d_fill].d_llr.p); }      // (under way while the batches that arrived are published)
    if (d_eng.d_cfg.flush_when_idle) age_check();
    int cons = used;
    if (d_in_dev) {
      { sub_scope pg(d_ph, phase_clock::GATHER); d_lane.flush(d_eng, d_st[d_fill].d_llr.p); }
      cons = d_lane.settle(d_eng, rd, used, d_ph);
    }
    d_io->consume_each(cons);
    if (!d_st[d_fill].tags.empty() || !d_fly.empty()) d_timer.poke();
    return 0;
  }

  void flush() override      // complete packets go now; a packet still being received stays behind
  {
    std::lock_guard<std::mutex> g(d_mu);
    if (!d_st[d_fill].tags.empty()) submit(true);
    while (!d_fly.empty()) reap(true);             // everything in flight has been published
  }

  bool next_deadline(tick_t& when)     // see demod_front
  {
    const auto w = std::chrono::microseconds(d_eng.d_cfg.max_wait_us);
    bool have = false;
    if (!d_st[d_fill].tags.empty()) { when = d_st[d_fill].t_first + w; have = true; }
    if (!d_fly.empty()) {
      const stage& fl = d_st[d_fly.front()];
      if (!have || fl.t_submit + w < when) { when = fl.t_submit + w; have = true; }
    }
    return have;
  }

  void age_check()           // see demod_front::age_check
  {
    const long w = d_eng.d_cfg.max_wait_us;
    if (!d_st[d_fill].tags.empty() && us_since(d_st[d_fill].t_first) >= w) submit(false);
    reap(false);
    if (!d_fly.empty() && us_since(d_st[d_fly.front()].t_submit) >= w) reap(true);
  }

private:
  struct stage {
    explicit stage(const c8p_api* a) : flag(a), llr(a), outs(a), pdu(a) {}
    enum { FREE, FILLING, INFLIGHT } state = FREE;
    pinned_vec<uint32_t> flag;        // the engine writes flag_want here behind the batch's kernels and downloads
    uint32_t flag_want = 0;
    pinned_vec<float> llr;            // host staging (a pageable input buffer only)
    dev_buf d_llr, d_mu;              // a pinned input buffer: the batch's LLR arena on the device, filled by the calls' gathers
    size_t llr_items = 0;
    bool any_mu = false;              // a packet of the batch carries "mu2x1chan" (an NDP): its captures have to travel too
    long src_id = 0;                  // != 0: the batch's LLR items lie in a demod mirror's arena copy (llr_batch), not in this stage's
    std::shared_ptr<llr_batch> src;
    std::vector<c8p_pkt_tags> tags;
    std::vector<gr_complex> mu;
    tick_t t_first, t_submit;
    pinned_vec<c8p_pkt_out> outs;     // downloads enqueued at submit time, right behind the kernels (one c8p_fetch_gather)
    pinned_vec<uint8_t> pdu;          // (the whole fixed-stride PDU arena: ~1 MB per 256 packets)
  };

  // where the next packet's LLR items go in the batch's arena
  uint64_t llr_alloc(stage& st, size_t n)
  {
    if (!d_in_dev) { const uint64_t off = st.llr.size(); st.llr.grow(n); return off; }
    const uint64_t off = st.llr_items;
    st.llr_items += n;
    if (st.llr_items * sizeof(float) > st.d_llr.cap) {
      d_lane.flush(d_eng, st.d_llr.p);             // (what was cut so far goes to the old arena; the move below follows it)
      const size_t first = env_int("C8P_ARENA_MIN", 0) ? 0 : (size_t)std::max(d_eng.d_cfg.max_batch, 1) * 16384u * sizeof(float);
      d_eng.dev_reserve(st.d_llr, std::max(first, st.llr_items * sizeof(float)), (size_t)off * sizeof(float));
    }
    return off;
  }

  int free_stage() const
  {
    for (int i = 0; i < NST; i++) if (d_st[i].state == stage::FREE) return i;
    return -1;
  }

  // see demod_front::submit: nothing is collected here, reap() publishes the batch when its completion word has arrived
  void submit(bool last)
  {
    while ((int)d_fly.size() >= d_max_fly) reap(true);
    stage& st = d_st[d_fill];
    const c8p_api& A = d_eng.d_api;
    const bool part = d_state == S_COLLECT && d_cur_fused < 0 && !d_cur_ext;      // a packet still being received moves to the next stage
    std::vector<float> part_llr;
    const uint64_t part_off = d_cur.llr_off;
    if (part && !d_in_dev) part_llr.assign(st.llr.data() + d_cur.llr_off, st.llr.data() + d_cur.llr_off + d_have);
    c8p_decode_in b;
    memset(&b, 0, sizeof(b));
    b.n_pkt = (int32_t)st.tags.size(); b.tags = st.tags.data();
    if (st.src_id) {
      // every packet of the batch lies in a demod mirror's device copy of its LLR arena (complete: that batch was replayed before its
      // tags went out): decoded where it is
      sub_scope pg(d_ph, phase_clock::GATHER);
      if (st.any_mu) {
        d_eng.dev_reserve(st.d_mu, st.mu.size() * sizeof(gr_complex), 0);
        d_eng.check(A.copy_async(d_eng.d_ctx, st.d_mu.p, st.mu.data(), st.mu.size() * sizeof(gr_complex)), "mu2x1chan upload");
        d_eng.check(A.copy_fence(d_eng.d_ctx), "c8p_copy_fence");
      }
      b.mem = C8P_MEM_DEVICE;
      b.llr = st.src->base; b.n_llr = st.src->n_floats;
      b.mu2x1chan = st.any_mu ? static_cast<const c8p_c32*>(st.d_mu.p) : nullptr;
    } else if (!d_in_dev) {
      b.mem = C8P_MEM_HOST;
      if (st.llr.empty()) *st.llr.grow(1) = 0.f;
      b.llr = st.llr.data(); b.n_llr = st.llr.size();
      b.mu2x1chan = reinterpret_cast<const c8p_c32*>(st.mu.data());
    } else {
      // the LLR items are on the device or on their way; the submit's stream waits for them on the device (c8p_copy_fence)
      sub_scope pg(d_ph, phase_clock::GATHER);
      if (!st.d_llr.p) d_eng.dev_reserve(st.d_llr, 4096, 0);
      d_lane.flush(d_eng, st.d_llr.p);
      if (st.any_mu) {                             // NDP captures ride along (rare: one sounding per channel update)
        d_eng.dev_reserve(st.d_mu, st.mu.size() * sizeof(gr_complex), 0);
        d_eng.check(A.copy_async(d_eng.d_ctx, st.d_mu.p, st.mu.data(), st.mu.size() * sizeof(gr_complex)), "mu2x1chan upload");
      }
      d_eng.check(A.copy_fence(d_eng.d_ctx), "c8p_copy_fence");
      b.mem = C8P_MEM_DEVICE;
      b.llr = static_cast<const float*>(st.d_llr.p); b.n_llr = std::max<size_t>(st.llr_items, 1);
      b.mu2x1chan = st.any_mu ? static_cast<const c8p_c32*>(st.d_mu.p) : nullptr;
    }
    {
      sub_scope ph(d_ph, phase_clock::SUBMIT);
      d_eng.check(A.decode(d_eng.d_ctx, &b), "decode");
      st.t_submit = now_tick();
    }
    {
      sub_scope ph(d_ph, phase_clock::ENQ);
      uint64_t pbytes = 0;
      A.buffer(d_eng.d_ctx, C8P_BUF_PDU, &pbytes);
      {
        const size_t full = (size_t)std::max(d_eng.d_cfg.max_batch, 1), np = std::max<size_t>(st.tags.size(), 1);
        d_hwm_pdu = std::max<size_t>(d_hwm_pdu, (pbytes + np - 1) / np * full);
      }
      st.outs.reserve((size_t)std::max(d_eng.d_cfg.max_batch, (int)st.tags.size()));
      st.pdu.reserve(d_hwm_pdu);
      st.outs.resize_uninit(st.tags.size());
      st.pdu.resize_uninit(pbytes);
      if (st.flag.empty()) st.flag.resize_uninit(1);
      st.flag_want = ++d_flag_seq ? d_flag_seq : ++d_flag_seq;
      __atomic_store_n(st.flag.data(), 0u, __ATOMIC_RELAXED);
      c8p_fetch_piece fp[3];                       // results + PDU arena + the completion word with one launch behind the kernels
      memset(fp, 0, sizeof(fp));
      int nfp = 0;
      fp[nfp].which = C8P_BUF_OUT; fp[nfp].bytes = st.tags.size() * sizeof(c8p_pkt_out); fp[nfp].dst = st.outs.data(); nfp++;
      if (pbytes) { fp[nfp].which = C8P_BUF_PDU; fp[nfp].bytes = pbytes & ~(uint64_t)3; fp[nfp].dst = st.pdu.data(); nfp++; }
      fp[nfp].which = C8P_BUF_FLAG; fp[nfp].offset = st.flag_want; fp[nfp].dst = st.flag.data(); nfp++;
      d_eng.check(A.fetch_gather(d_eng.d_ctx, fp, nfp), "c8p_fetch_gather");
    }
    const int old = d_fill;
    st.state = stage::INFLIGHT;
    d_fly.push_back(old);
    if (last) while (!d_fly.empty()) reap(true);
    int nx = -1;
    for (int i = 0; i < NST; i++) if (i != old && d_st[i].state == stage::FREE) { nx = i; break; }      // (never a move inside one device arena)
    if (nx < 0) throw std::runtime_error("c8p_blocks: no free stage");
    d_st[nx].state = stage::FILLING;
    d_fill = nx;
    if (part) {
      stage& from = d_st[old];
      stage& to = d_st[nx];
      d_cur.llr_off = llr_alloc(to, (size_t)std::max(d_cur.total, 0));
      if (d_have) {
        if (!d_in_dev) memcpy(to.llr.data() + d_cur.llr_off, part_llr.data(), sizeof(float) * (size_t)d_have);
        else d_eng.check(A.copy_async(d_eng.d_ctx, static_cast<float*>(to.d_llr.p) + d_cur.llr_off, static_cast<const float*>(from.d_llr.p) + part_off,
                                      sizeof(float) * (size_t)d_have), "partial packet move");
      }
    }
  }

  // publishes every in-flight batch whose completion word has arrived, oldest first (see demod_front::reap)
  bool reap(bool block)
  {
    bool any = false;
    while (!d_fly.empty()) {
      stage& st = d_st[d_fly.front()];
      if (__atomic_load_n(st.flag.data(), __ATOMIC_ACQUIRE) != st.flag_want) {
        if (!block) break;
        sub_scope pw(d_ph, phase_clock::WAIT);
        int spins = 0;
        while (__atomic_load_n(st.flag.data(), __ATOMIC_ACQUIRE) != st.flag_want) {
          if (++spins < 4000) { __builtin_ia32_pause(); continue; }
          std::this_thread::sleep_for(std::chrono::microseconds(10));
          if (us_since(st.t_submit) > 20000000) throw std::runtime_error("c8p_blocks: a batch's completion word did not arrive within 20 s");
        }
      }
      block = false;
      sub_scope phr(d_ph, phase_clock::REPLAY);
      const size_t n = st.tags.size();
      for (size_t p = 0; p < n; p++) publish_packet(st.tags[p], st.outs.data()[p], st.pdu.data());
      st.tags.clear(); st.llr.clear(); st.mu.clear(); st.llr_items = 0; st.any_mu = false;
      st.src.reset(); st.src_id = 0;
      st.state = stage::FREE;
      d_fly.pop_front();
      any = true;
    }
    return any;
  }

  // one packet's records {u16 len, blob} out of a PDU arena -> messages, with the reference's debug lines
  void publish_packet(const c8p_pkt_tags& t, const c8p_pkt_out& o, const uint8_t* pdu)
  {
    size_t off = o.n_pdu > 0 ? (size_t)o.pdu_off : 0;
    for (int k = 0; k < o.n_pdu; k++) {
      const int len = pdu[off] | (pdu[off + 1] << 8);
      d_msg.len = len;
      d_msg.blob.assign(pdu + off + 2, pdu + off + 2 + len);
      off += 2 + (size_t)len;
      if (t.trellis != 0) debug_line(t, true);
      d_io->message_port_pub("out", d_msg);
    }
    for (int k = 0; k < o.n_crc_fail; k++) debug_line(t, false);
  }

  // fused mode: the packet was decoded on the device when `demod` ran c8p_rx; its batch's results wait behind the handle.
  // Everything gathered before it is published first, so the message order stays the arrival order.
  void publish_fused()
  {
    if (!d_st[d_fill].tags.empty()) submit(true);
    while (!d_fly.empty()) reap(true);
    const long id = d_cur_fused >> C8P_FUSED_IDX_BITS;
    const size_t idx = (size_t)(d_cur_fused & ((1L << C8P_FUSED_IDX_BITS) - 1));
    if (id != d_fb_id || !d_fb || idx >= d_fb->outs.size()) return;   // (resolved where the tags were read) a handle from another process / a replayed capture: nothing to publish
    sub_scope phr(d_ph, phase_clock::REPLAY);
    publish_packet(d_cur, d_fb->outs.data()[idx], d_fb->pdu.data());
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
  std::mutex d_mu;                                 // general_work / flush against the latency timer's thread
  phase_clock d_ph;
  engine_holder d_eng;
  bool d_debug;
  bool d_mode_known = false, d_in_dev = false;     // the input port's buffer is pinned host memory
  c8p_pkt_tags d_cur;
  int d_have = 0;
  long d_cur_fused = -1;
  bool d_cur_mu = false;                           // the packet being received carries "mu2x1chan"
  gr_complex d_cur_muv[128];                       // ... its values (zeros otherwise): joins the batch with the packet
  bool d_cur_ext = false;                          // ... its LLR items lie behind a handle ("c8p_llr"): the items in the stream are skipped
  long d_lb_id = 0;                                // the demod batch whose handle packets are arriving
  std::shared_ptr<llr_batch> d_lb;
  long d_fb_id = 0;                                // fused mode: the batch whose packets are arriving
  std::shared_ptr<fused_batch> d_fb;
  pdu_t d_msg;                                     // (reused: its blob keeps its capacity)
  enum { NST = 4 };
  stage d_st[NST];
  int d_fill = 0;                                  // the stage being gathered; d_fly: the ones in flight, in submit order
  std::deque<int> d_fly;
  int d_max_fly = 2;                               // batches in flight at most
  uint32_t d_flag_seq = 0;
  tag_cursor d_tc;
  gather_lane d_lane;                              // pieces always belong to d_st[d_fill]
  uint64_t d_nPktCorrect = 0, d_legacyMcsCount[8], d_vhtMcsCount[10], d_htMcsCount[8];
  size_t d_hwm_pdu = 0;
  latency_timer d_timer;                           // its thread uses everything above
  front_registration d_reg{this};
};

decode::sptr decode::make(bool ifdebug) { return make(ifdebug, default_api(), batching_cfg()); }
decode::sptr decode::make(bool ifdebug, const c8p_api& api, const batching_cfg& cfg)
{
  return std::shared_ptr<decode>(new decode_front(ifdebug, api, cfg));
}

C8P_MIRROR_NS_END
