// c8p_blocks.h — host-side mirror of the reference's block API for the hot path:
//
//   gr::ieee80211::demod::make(int mupos, int mugid)   include/gnuradio/ieee80211/demod.h:35
//   gr::ieee80211::demod2::make()                       include/gnuradio/ieee80211/demod2.h
//   gr::ieee80211::decode::make(bool ifdebug)           include/gnuradio/ieee80211/decode.h
//   gr::ieee80211::signal::make() / signal2::make()     include/gnuradio/ieee80211/signal.h, signal2.h   (§8f-1)
//
// Same factories, same general_work()/forecast() contract, same stream-tag keys and PDU layout as
// lib/{demod,demod2,decode}_impl.cc, but every block is a *batching front* for the C ABI of include/c8p_rx.h:
// it cuts the tagged item stream into packets, hands complete packets to c8p_demod()/c8p_decode() in batches,
// and replays the results (LLR items + tags, or PDUs) in arrival order.
//
// GNU Radio is not installed in the build image, so the runtime services a gr::block gets from the scheduler
// (item counters, tag access, consume, message ports) are reached through the small `block_io` interface below.
// INTEGRATION.md shows the 1:1 adapter onto gr::block / pmt for a machine that has GNU Radio 3.10.
#pragma once
#include <complex>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "c8p_rx.h"

// Stand-alone (this repository, tests, plain C++ users) the mirror classes carry the reference's own names,
// gr::ieee80211::demod ...; inside a real GNU Radio module those names belong to the public gr::block classes
// (gnuradio/lib/*_impl.h wrap the mirrors), so that build compiles this file with -DC8P_MIRROR_NS=c8p and gets c8p::demod ...
#ifdef C8P_MIRROR_NS
#define C8P_MIRROR_NS_BEGIN namespace C8P_MIRROR_NS {
#define C8P_MIRROR_NS_END }
#define C8P_MIRROR_QUAL C8P_MIRROR_NS
#include <gnuradio/types.h>          // gr_complex, gr_vector_*
#else
#define C8P_MIRROR_NS_BEGIN namespace gr { namespace ieee80211 {
#define C8P_MIRROR_NS_END } }
#define C8P_MIRROR_QUAL gr::ieee80211
typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;
#endif

C8P_MIRROR_NS_BEGIN

// value of a stream tag: the three PMT kinds the hot path uses (long, float, c32vector)
struct tag_value {
  enum kind_t { LONG, FLOAT, C32VECTOR } kind = LONG;
  long l = 0;
  float f = 0.f;
  std::vector<gr_complex> v;
  long as_long() const { return kind == FLOAT ? (long)f : l; }
  float as_float() const { return kind == LONG ? (float)l : f; }
  static tag_value from_long(long x) { tag_value t; t.kind = LONG; t.l = x; return t; }
  static tag_value from_float(float x) { tag_value t; t.kind = FLOAT; t.f = x; return t; }
  static tag_value from_c32vector(const std::vector<gr_complex>& x) { tag_value t; t.kind = C32VECTOR; t.v = x; return t; }
};
struct tag_t {
  uint64_t offset = 0;
  std::string key;
  tag_value value;
};
// a published message: pmt::cons(dict{"len": len}, blob)
struct pdu_t {
  long len = 0;
  std::vector<uint8_t> blob;
  int stream = -1;          // rx_chain only: which input port's radio (the reference's dict has no such entry: one flowgraph, one radio)
};

// the slice of gr::block a general_work() needs
class block_io
{
public:
  virtual ~block_io() {}
  virtual uint64_t nitems_read(unsigned port) = 0;
  virtual uint64_t nitems_written(unsigned port) = 0;
  virtual void get_tags_in_range(std::vector<tag_t>& v, unsigned port, uint64_t start, uint64_t end) = 0;
  virtual void add_item_tag(unsigned port, uint64_t offset, const std::string& key, const tag_value& value) = 0;
  virtual void consume_each(int n) = 0;
  virtual void message_port_pub(const std::string& port, const pdu_t& msg) = 0;
  virtual void debug_print(const std::string& line) = 0;     // the reference's `dout` (std::cout when ifdebug)
  // Called from the front's latency timer (another thread) when results became ready while no input was arriving: get the
  // scheduler to call general_work() again — forecast() then asks for no input, so the call is made even though there is none.
  // (A gr::block does it by posting a message to a port of its own, which wakes its scheduler thread.)
  virtual void wake() {}
  // GNU Radio 3.10 custom buffers.  A gr::block names a gr::buffer_type per port in its io_signature and the scheduler allocates the
  // buffer of that edge through it (flat_flowgraph::allocate_buffer; gnuradio-runtime's host_buffer is the template).  The mirrors' type
  // is PINNED HOST MEMORY (c8p_block::buffer_alloc): the upstream block writes its items where the device can read them, so a front
  // gathers the packets out of the scheduler's buffer with one kernel (c8p_copy_gather) instead of memcpy -> pinned staging -> H2D, and
  // demod's LLR items are written into the output buffer by the device.  These two say whether the scheduler did so for a port
  // (false: an ordinary pageable buffer — the fronts then stage through pinned copies of their own, as in rounds 1-4).
  virtual bool input_is_pinned(unsigned port) { return false; }
  virtual bool output_is_pinned(unsigned port) { return false; }
  // ... and how large the input buffer is (0: not known, taken to be large).  A gather takes ~18 us from enqueue to done whatever its
  // size: below C8P_GATHER_MIN_BUFFER a call's window is so small that copying it into pinned staging on the CPU (and letting a copy
  // engine upload the batch) is faster, and a front does that even when the buffer is pinned.
  virtual size_t input_buffer_bytes(unsigned port) { return 0; }
  // Who reads this output port?  1: ONE reader, and it is a decode mirror of this process (c8p_block::is_decode_front) — what rx.grc /
  // rx2.grc wire behind demod / demod2; 0: another reader, or more than one (a batch's hand-over is released by the reader that moves on
  // to the next batch: two decode blocks on one edge get the items themselves); -1: the scheduler side cannot tell.  A demod front whose cfg leaves the choice open
  // (batching_cfg::fused_auto) runs fused exactly when this says 1: nothing but the decode mirror ever looks at the item values then.
  virtual int output_feeds_only_decode(unsigned port) { return -1; }
};
#define C8P_GATHER_MIN_BUFFER (2u << 20)

// the C ABI the blocks call; defaults to the symbols of libc8p_rx.so, tests inject another table
struct c8p_api {
  c8p_ctx* (*create)(const c8p_cfg*);
  void (*destroy)(c8p_ctx*);
  const char* (*last_error)(const c8p_ctx*);
  int (*demod)(c8p_ctx*, const c8p_batch_in*);
  int (*decode)(c8p_ctx*, const c8p_decode_in*);
  int (*signal)(c8p_ctx*, const c8p_signal_in*);
  int (*detect)(c8p_ctx*, const c8p_detect_in*);
  int (*sync)(c8p_ctx*);
  const void* (*buffer)(c8p_ctx*, int, uint64_t*);
  int (*fetch)(c8p_ctx*, int, void*, uint64_t, uint64_t);
  // pipelining: submit batch i+1, then collect batch i (pinned staging, asynchronous copies, per-submit wait)
  int (*rx)(c8p_ctx*, const c8p_batch_in*);
  int (*ticket)(c8p_ctx*);
  int (*select)(c8p_ctx*, int);
  int (*wait)(c8p_ctx*, int);
  int (*done)(c8p_ctx*, int);
  int (*fetch_async)(c8p_ctx*, int, void*, uint64_t, uint64_t);
  void* (*host_alloc)(uint64_t);
  void (*host_free)(void*);
  // the raw-stream pipeline (c8p_rawchain): K streams per detection pass, arenas sized in advance, counters, caller-owned arenas
  int (*detect_multi)(c8p_ctx*, const c8p_detect_multi_in*);
  int (*reserve)(c8p_ctx*, const c8p_reserve_in*);
  int64_t (*counter)(c8p_ctx*, int);
  void* (*dev_alloc)(c8p_ctx*, uint64_t);
  void (*dev_free)(c8p_ctx*, void*);
  int (*copy_async)(c8p_ctx*, void*, const void*, uint64_t);
  int (*copy_spans)(c8p_ctx*, c8p_c32*, const c8p_c32*, const c8p_span*, int32_t);
  int (*copy_wait)(c8p_ctx*);
  int (*copy_gather)(c8p_ctx*, c8p_c32*, const c8p_piece*, int32_t);
  uint64_t (*copy_mark)(c8p_ctx*);
  int (*copy_wait_mark)(c8p_ctx*, uint64_t);
  int (*copy_mark_done)(c8p_ctx*, uint64_t);
  int (*copy_fence)(c8p_ctx*);
  int (*fetch_gather)(c8p_ctx*, const c8p_fetch_piece*, int32_t);
  int (*option)(c8p_ctx*, int, int64_t) = nullptr;      // c8p_option: optional (round 6) — a library without it is used as it is
};
// resolves the table from a shared library path (dlopen); throws std::runtime_error if a symbol is missing
c8p_api load_c8p_api(const std::string& so_path);
// the table the one-argument factories use: the library named by the environment's C8P_RX_LIB, or "libc8p_rx.so" on the loader path
c8p_api default_c8p_api();

struct batching_cfg {
  int max_batch = default_max_batch();   // packets per c8p_* call: 512, or the environment's C8P_MAX_BATCH.  (Measured on the demod -> decode chain,
                                // cfg-2 packets: 256 / 512 / 1024 = 565 / 693 / 690 k packets/s fused, profiles/r6c/chain_fused_sweep.txt; a batch that
                                // does not fill is cut short after max_wait_us, so the size bounds throughput, not latency.  Until round 6: 256.)
  int detect_items = 1 << 20;   // detect: raw items gathered per c8p_detect call
  // SAFE MODE (environment C8P_GR_SAFE=1): a front that uses nothing of its scheduler beyond what the reference's own blocks use — no
  // custom buffers (pinned_io off: items staged through the front's own pinned copies), no look at who reads the LLR edge (fused_auto
  // off: plain mode, every item value and every reference tag on the edge), consume in the call that cut the items, no handle tags,
  // no timer thread (flush_when_idle off: a partial batch goes out at the end of a general_work call that finds it old enough, and at
  // stop()).  Every default below returns its safe value then, whatever the other switches say.  The GNU Radio adapters
  // (gnuradio/lib/c8p_gr_adapter.h) name default buffers in their io_signatures in this mode; it is what a first run against an
  // untried GNU Radio installation should use, and what -DC8P_GR_STOCK_API_ONLY builds compile in for good.
  static bool safe_mode();
  bool flush_when_idle = default_flush_when_idle();  // submit a partial batch once its oldest packet has waited max_wait_us (bounds latency in a live
                                // flowgraph; a saturated one fills whole batches between two checks and is never cut short)
  int max_wait_us = default_max_wait_us();   // 1000, or the environment's C8P_MAX_WAIT_US; checked at the end of every general_work call AND by a timer thread of the front: the scheduler
                                // never calls a block that has no input, and demod's input is bursty (signal emits items only
                                // during packets), so without the timer the last packets of a burst would wait for the next one
  int device = 0;
  // Fused demod -> decode (forced on by the environment's C8P_FUSED=1; see fused_auto for the default).  When both mirrors sit in one flowgraph the float LLR stream
  // between them only exists to carry packets from one block to the next; in fused mode `demod` runs c8p_rx (demod + decode on
  // the device, the LLRs never leave HBM), emits the reference's tags plus "c8p_fused" (a handle) on the first of `total`
  // items whose VALUES ARE NOT WRITTEN, and `decode` publishes the PDUs behind the handle instead of decoding the items.
  // Tags, item counts, PDU bytes and debug lines are unchanged; anything else that reads the LLR items must not use it.
  bool fused = default_fused();
  // ... and when `fused` is false: decide at the first general_work call, from the flowgraph — fused if block_io::output_feeds_only_decode
  // says that every reader of the LLR port is a decode mirror (the reference's own flowgraphs), plain otherwise.  Default: on unless the
  // environment sets C8P_FUSED (1: always fused, 0: never, "auto": this).  A caller that sets `fused` itself should clear it.
  bool fused_auto = default_fused_auto();
  // When fused_auto chose fused — decode mirrors are the ONLY readers of the LLR port, so nothing else can see that edge's tags — demod
  // puts ONE tag on a packet's first item, the handle, instead of the reference's eleven to fourteen + the handle: decode finds the
  // packet's tag record (format, mcs, len, ..., total, cfo, snr, rssi, sssnr) behind the handle with the results.  Every tag is a string
  // key + a variant here and a PMT pair under the buffer's mutex in GNU Radio; item counts, PDU bytes and debug lines are unchanged.
  // Environment C8P_FUSED_TAGS=full keeps the reference's tags on the edge (a tag debugger on it would otherwise see one tag per packet).
  // A fused mode that was FORCED (cfg.fused / C8P_FUSED=1: the front does not know who reads) always writes the full set.
  bool fused_lean_tags = default_fused_lean_tags();
  // Custom buffers (block_io::input_is_pinned / output_is_pinned): use them when the scheduler provides them (default; environment
  // C8P_PINNED_IO=0 turns it off).  defer_consume: a front then consumes the items of a general_work call during the NEXT call, when
  // the gather that moves them has had a scheduler round trip to finish — nothing waits for the link; 0 (C8P_DEFER_CONSUME=0): the gather is
  // waited for at the end of the same call (what a single-mapped custom buffer that relocates unread items asks for).
  bool pinned_io = default_pinned_io();
  bool defer_consume = default_defer_consume();
  // Plain mode, demod's output buffer pinned: demod keeps the device copy of a batch's LLR arena it fills the output buffer from and
  // names it in a tag ("c8p_llr": batch id << 20 | packet index).  A decode mirror of the same process that meets the tag decodes the
  // packet out of that copy and skips its `total` items — which ARE in the stream, written, for whoever else reads them: the LLR
  // stream then crosses the link once (up) instead of twice.  Environment C8P_LLR_HANDLE=0 turns it off (decode gathers the items).
  bool llr_handle = default_llr_handle();
  static int default_max_batch();
  static bool default_llr_handle();
  static bool default_flush_when_idle();
  static bool default_fused();
  static bool default_fused_auto();
  static bool default_fused_lean_tags();
  static bool default_pinned_io();
  static bool default_defer_consume();
  static int default_max_wait_us();
};

class c8p_block
{
public:
  virtual ~c8p_block() {}
  virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required) = 0;
  virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                           gr_vector_void_star& output_items) = 0;
  // end of stream: submit whatever is buffered (a live flowgraph never calls this)
  virtual void flush() = 0;
  // where the host side of this block spent its time, in microseconds since it was made: "name=us;..." (tools/bench_blocks.py)
  virtual std::string host_phase_us() { return std::string(); }
  virtual void host_phase_reset() {}
  // The buffer type this block's io_signature names for its ports (see block_io::input_is_pinned): which ports want a buffer from
  // buffer_alloc, and the allocator itself (pinned host memory of the engine library; nullptr: the block has no engine to ask)
  virtual bool wants_pinned_input(unsigned port) const { return false; }
  virtual bool wants_pinned_output(unsigned port) const { return false; }
  virtual void* buffer_alloc(size_t bytes) { return nullptr; }
  virtual void buffer_free(void* p) {}
  // Items of output port 0 this block has placed in the output buffer BEYOND what it has reported as produced (demod with a pinned
  // output buffer: the device is still writing them; they are reported by the call at whose end they have arrived).  They sit at the
  // start of the space the next call is given — a scheduler's ring continues where the last call's produce() stopped — so whoever moves
  // the block's output pointer somewhere else (a buffer that is not a ring and has come to its end) first calls general_work with
  // noutput_items = this number, as often as it takes to bring it to zero: the block reports what has arrived (it waits for its oldest
  // launch when nothing has) and places nothing new.
  virtual int output_ahead() { return 0; }
  // true for the decode mirror (what a scheduler-side block_io::output_feeds_only_decode looks for in the readers of an edge)
  virtual bool is_decode_front() const { return false; }
  // demod / demod2: 1 / 0 = runs fused / plain (decided at the first general_work call when the cfg left it open), -1 = not decided
  // yet or not a demod front
  virtual int fused_state() { return -1; }
  // buffer_will_move (below): wait for every copy this block has under way that reads or writes [base, base + bytes)
  virtual void quiesce(const void* base, size_t bytes) {}
  // buffer_move_scope (below): quiesce(), and when this block has output items placed ahead (output_ahead() > 0) INSIDE the range: keep
  // its general_work out (the block's own mutex stays held), save those items, return true.  move_end (only after a true move_begin):
  // the saved items go to `new_write_ptr` — where the buffer's write position is now, i.e. where the next call's output space begins —
  // and general_work may run again.
  // Returns 1: held (move_end must follow); 0: nothing of this block is in the way; -1: the block could not be got hold of within
  // timeout_us (its general_work is running and does not return — e.g. it waits for a lock the mover holds): nothing is held, the
  // mover must not move now.
  virtual int move_begin(const void* base, size_t bytes, int64_t timeout_us) { quiesce(base, bytes); return 0; }
  virtual void move_end(void* new_write_ptr) {}
  void set_io(block_io* io) { d_io = io; }

protected:
  block_io* d_io = nullptr;
};

// A scheduler buffer that is about to MOVE items it has handed out says so first, from whatever thread: every front of this process waits
// for the copies it has under way that read or write [base, base + bytes) — items it has cut and not consumed yet (batching_cfg::
// defer_consume), LLR items it has placed and not reported (output_ahead).  GNU Radio's single-mapped custom buffers relocate unread
// items to the start of the buffer when a reader or the writer is blocked at its end (gr::buffer_single_mapped::input_blocked_callback /
// output_blocked_callback): gnuradio/lib/c8p_pinned_buffer.h calls this from both before it lets the base class move anything, which is
// what makes deferred consume safe on such a buffer.  The items themselves need no attention: what has been cut lives in the batch's
// device arena, and a front counts in absolute items, not addresses.
void buffer_will_move(const void* base, size_t bytes);

// A buffer that moves its WRITE position without asking the writer first — GNU Radio's single-mapped custom buffers do, from either
// blocked callback: input_blocked_callback_logic shifts [0, write) up by the reader's gap and sets d_write_index += gap,
// output_blocked_callback_logic shifts the unread items down — would leave the items a demod front has placed AHEAD of the write position
// (output_ahead) behind: the next call would report whatever lies at the new position.  The scope brackets the move:
//     { c8p::buffer_move_scope mv(base, bytes);  ... the buffer moves its items and its write index ...;  mv.moved_to(write_pointer()); }
// Its constructor does what buffer_will_move does (every front waits for its copies on the range) and, for the front that has items
// ahead in the range, holds that front and saves them; moved_to puts them back at the new write position; the destructor lets the front
// run again (without moved_to the items stay where they were: the buffer did not move).  A front is taken hold of with a TIMEOUT
// (its own mutex is tried, not waited for: a scheduler that calls the buffer's callback with a lock held that the front's running
// general_work needs — add_item_tag takes the buffer's — must not deadlock here): ok() == false means some front could not be held,
// nothing is held, and the buffer must NOT move now (its callback returns false: the scheduler comes back later).  gnuradio/lib/c8p_pinned_buffer.h uses it in both
// callbacks; the mirror's harness has an edge of that kind to test it with (c8ph_run, C8P_MIRROR_MOVES).
class buffer_move_scope
{
public:
  buffer_move_scope(const void* base, size_t bytes, int64_t timeout_us = 20000);
  bool ok() const { return d_ok; }
  void moved_to(void* new_write_ptr);
  ~buffer_move_scope();
  buffer_move_scope(const buffer_move_scope&) = delete;
  buffer_move_scope& operator=(const buffer_move_scope&) = delete;

private:
  std::vector<c8p_block*> d_held;
  bool d_done = false, d_ok = true;
};

class demod : virtual public c8p_block
{
public:
  typedef std::shared_ptr<demod> sptr;
  static sptr make(int mupos, int mugid);
  static sptr make(int mupos, int mugid, const c8p_api& api, const batching_cfg& cfg);
};
class demod2 : virtual public c8p_block
{
public:
  typedef std::shared_ptr<demod2> sptr;
  static sptr make();
  static sptr make(const c8p_api& api, const batching_cfg& cfg);
};
// gr::ieee80211::signal::make() / signal2::make()   include/gnuradio/ieee80211/signal.h, signal2.h
// in: port 0 = sync bytes (flag items carry the tags "rad", "snr", "rssi"), ports 1.. = raw samples per antenna;
// out: the CFO-corrected packet items per antenna, tags cfo/snr/rssi/seq/mcs/len/nsamp/chan on the first one
// (lib/signal_impl.cc:68-204).  SURVEY.md §8f-1.
class signal : virtual public c8p_block
{
public:
  typedef std::shared_ptr<signal> sptr;
  static sptr make();
  static sptr make(const c8p_api& api, const batching_cfg& cfg);
};
class signal2 : virtual public c8p_block
{
public:
  typedef std::shared_ptr<signal2> sptr;
  static sptr make();
  static sptr make(const c8p_api& api, const batching_cfg& cfg);
};
// The packet detector of rx.grc / rx2.grc as ONE block: the hier block presiso (examples/presiso.grc) + trigger::make()
// + sync::make().  in: the raw samples of antenna 0; out: sync's byte stream (1 item per input item), flagged items
// carry the tags "rad", "snr", "rssi" (lib/sync_impl.cc:111-141).  SURVEY.md §8f-2.
class detect : virtual public c8p_block
{
public:
  typedef std::shared_ptr<detect> sptr;
  static sptr make();
  static sptr make(const c8p_api& api, const batching_cfg& cfg);
};
class decode : virtual public c8p_block
{
public:
  typedef std::shared_ptr<decode> sptr;
  static sptr make(bool ifdebug);
  static sptr make(bool ifdebug, const c8p_api& api, const batching_cfg& cfg);
};

// The WHOLE receive chain of rx.grc (examples/rx.grc:753-767: presiso -> trigger -> sync -> signal -> demod -> decode) for K radios as
// ONE block: K complex input ports (port s = what radio s's source block delivers), no stream output, message port "out" = the PDUs
// decode publishes (lib/decode_impl.cc:105-120), pdu_t.stream = the port.  Behind it one host/c8p_rawchain handle: per stream its own
// trigger / sync state, commit point and scan position; every `items_per_round` items per stream (or when the input dries up) ONE
// detection pass, one signal call and one decode batch for all K streams.  The items are copied once, into pinned memory the device reads.
class rx_chain : virtual public c8p_block
{
public:
  typedef std::shared_ptr<rx_chain> sptr;
  static sptr make(int n_streams);
  static sptr make(int n_streams, const std::string& engine_lib, const batching_cfg& cfg);      // cfg.detect_items = items per stream and round
  // item_format = C8P_FMT_C32: gr_complex ports.  C8P_FMT_SC16 / C8P_FMT_SC8: the ports carry what the radio sends — std::complex<short> /
  // std::complex<char> items (a UHD source with cpu_format sc16 / sc8) — and the conversion UHD would do on the host
  // ((float)v * scale; 0: 1 / 32767, 1 / 127) happens in the engine's gather kernel: 4 / 2 bytes per item to copy and to send
  static sptr make(int n_streams, const std::string& engine_lib, const batching_cfg& cfg, int item_format, float scale);
};

C8P_MIRROR_NS_END
