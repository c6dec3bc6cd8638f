// c8p_gr_adapter.h — the scheduler services a batching front needs (c8p::block_io), implemented BY the gr::block itself.
//
// A front (c8p::demod / demod2 / decode of ../../gr-ieee80211_amd/host/c8p_blocks.h, compiled into namespace c8p in this
// module) cuts the tagged item stream into packets, batches them through the C ABI (libc8p_rx.so) and replays the results;
// everything it asks of GNU Radio goes through the eight block_io methods below.  c8p_gr_block<API> is the block class:
// API = the public class of include/gnuradio/ieee80211/ (demod, demod2, decode), which derives from gr::block, so the
// protected scheduler calls (get_tags_in_range, add_item_tag, ...) are made from inside the block, as in the reference's
// own *_impl classes.  Tag values cross as the three PMT kinds the reference's blocks use: long, float, c32vector.
//
// AUDIT LIST — every GNU Radio / PMT member this header and lib/{demod,demod2,decode}_impl.{h,cc} touch, with a place where the
// reference uses the same member the same way (paths relative to the reference tree).  None of this has met GNU Radio's own
// headers: tests/gr_api_decls/ declares these members as WE READ the 3.10 API documentation and the adapters are type-checked
// against that reading (tests/test_gnuradio_packaging.py); gnuradio/ci/run.sh is what checks them for real.
//
//   member (as used here)                                        the reference uses it at
//   ---------------------------------------------------------------------------------------------------------------------------
//   gr::block(name, io_signature::sptr in, out)  [ctor]          lib/demod_impl.cc:35-37, lib/decode_impl.cc:34-36
//   gr::io_signature::make(min, max, sizeof_item)                lib/demod_impl.cc:36-37, lib/decode_impl.cc:35-36
//   gnuradio::make_block_sptr<T>(args...)                        lib/demod_impl.cc:30, lib/decode_impl.cc:29
//   set_tag_propagation_policy(gr::block::TPP_DONT)              lib/demod_impl.cc:46, lib/decode_impl.cc:47
//   message_port_register_out(pmt::mp("out"))  [decode_impl.cc]  lib/decode_impl.cc:39
//   message_port_register_in(pmt::mp(..))                        lib/pktgen_impl.cc:46, lib/modulation2_impl.cc:45
//   set_msg_handler(pmt::mp(..), callable(pmt::pmt_t))           lib/pktgen_impl.cc:47 (boost::bind there, a lambda here)
//   forecast(int, gr_vector_int&)  [override]                    lib/demod_impl.cc:53-57, lib/decode_impl.cc:54-58
//   general_work(int, gr_vector_int&, gr_vector_const_void_star&, gr_vector_void_star&)  [override]
//                                                                lib/demod_impl.cc:59-63, lib/decode_impl.cc:60-64
//   stop()  [override, calls gr::block::stop()]                  NOT used by the reference (its blocks hold nothing back at the end
//                                                                of a finite flowgraph; a batching front does): gr::block::stop is
//                                                                virtual bool stop() in gnuradio/block.h of 3.10
//   nitems_read(unsigned), nitems_written(unsigned)              lib/demod_impl.cc:75, :259; lib/decode_impl.cc:70
//   get_tags_in_range(std::vector<gr::tag_t>&, port, start, end) lib/demod_impl.cc:75, lib/decode_impl.cc:70
//   gr::tag_t::offset / key / value                              lib/demod_impl.cc:79-80 (tag.key, tag.value)
//   add_item_tag(port, offset, key, value, alias_pmt())          lib/demod_impl.cc:258-262
//   consume_each(int)                                            lib/demod_impl.cc:102 ... :340
//   message_port_pub(pmt::mp("out"), pmt::cons(meta, blob))      lib/decode_impl.cc:120, :419, :516
//   _post(pmt::pmt_t port, pmt::pmt_t msg)  [wake()]             NOT used by the reference: gr::basic_block::_post is the public
//                                                                entry point message delivery uses (gnuradio/basic_block.h of 3.10)
//   pmt::mp(const char*) / pmt::mp(std::string)                  lib/demod_impl.cc:82 ...
//   pmt::make_dict(), pmt::dict_add(dict, key, value)            lib/demod_impl.cc:224-226, lib/decode_impl.cc:73-75
//   pmt::from_long(long), pmt::from_float(double)                lib/demod_impl.cc:225-237
//   pmt::to_long(pmt_t), pmt::to_double(pmt_t)                   lib/demod_impl.cc:85 (to_long); the reference reads floats with
//                                                                pmt::to_float (:82); to_double is the same accessor, 3.10 pmt.h
//   pmt::init_c32vector(size_t, const std::vector<std::complex<float>>&)     lib/demod_impl.cc:247
//   pmt::c32vector_elements(pmt_t) -> std::vector<std::complex<float>>       lib/demod_impl.cc:89, lib/decode_impl.cc:103
//   pmt::make_blob(const void*, size_t)                          lib/decode_impl.cc:119, :418, :515
//   pmt::cons(pmt_t, pmt_t)                                      lib/decode_impl.cc:120
//   pmt::is_integer / is_real / is_c32vector(pmt_t)              NOT used by the reference (it looks tags up by key and assumes the
//                                                                kind); predicates of 3.10 pmt.h
//   pmt::symbol_to_string(pmt_t), pmt::string_to_symbol(string)  lib/pad2_impl.cc:154 (string_to_symbol); symbol_to_string: 3.10 pmt.h
//   pmt::PMT_T                                                   3.10 pmt.h (the reference uses pmt::PMT_NIL, lib/demod_impl.cc:89)
//   input_signature() / output_signature() -> io_signature::sptr   gr::basic_block (3.10 basic_block.h); NOT used by the reference
//   io_signature::stream_buffer_type(size_t)                     3.10 io_signature.h (custom buffers: see c8p_pinned_buffer.h, which
//                                                                continues this list with the gr::buffer members the buffer type uses)
//   gr::block::detail() -> block_detail_sptr                     NOT used by the reference.  The four members behind
//   gr::block_detail::noutputs(), output(unsigned) -> buffer_sptr  output_feeds_only_decode(): who reads demod's LLR port?  3.10 block.h,
//   gr::buffer::nreaders(), reader(size_t) -> buffer_reader*     block_detail.h, buffer.h, buffer_reader.h as we read them.  Building with
//   gr::buffer_reader::link() -> block_sptr                      -DC8P_GR_NO_TOPOLOGY leaves them out (fused mode is then C8P_FUSED=1 only)
//
// Members without a call site in the reference (stop, _post, the three predicates + symbol_to_string / to_double / PMT_T, and the
// topology walk above): those are the first things to look at if the module does not build against a real GNU Radio.
#ifndef INCLUDED_IEEE80211_C8P_GR_ADAPTER_H
#define INCLUDED_IEEE80211_C8P_GR_ADAPTER_H

// SAFE MODE — the part of this header a first run against an untried GNU Radio should rely on.  Everything in the audit list that the
// reference does NOT use (custom buffers, the walk over the readers of an edge, _post) can be left out:
//   at run time   C8P_GR_SAFE=1 in the environment: the io_signatures name default buffers, the fronts stage items through their own pinned
//                 copies, the edge demod -> decode carries every item value and every reference tag (plain mode), no timer thread — the
//                 members marked (*) below are compiled in but never called;
//   at build time -DC8P_GR_STOCK_API_ONLY (cmake -DC8P_STOCK_API_ONLY=ON): the same, and those members are not even compiled — what is
//                 left is gr::block, io_signature::make(min, max, size), tags, messages and PMTs exactly as the reference's blocks use them.
// Without either, a port whose buffer turns out NOT to be one of the module's pinned buffers (the pointer general_work is handed is
// checked against the buffers c8p_pinned_buffer has allocated) is treated as an ordinary buffer, and a reader walk that throws or finds
// something unexpected answers "cannot tell", which means plain mode.
#include <gnuradio/block.h>
#if defined(C8P_GR_STOCK_API_ONLY) && !defined(C8P_GR_NO_TOPOLOGY)
#define C8P_GR_NO_TOPOLOGY
#endif
#ifndef C8P_GR_NO_TOPOLOGY
#include <gnuradio/block_detail.h>
#include <gnuradio/buffer.h>
#include <gnuradio/buffer_reader.h>
#endif
#include <gnuradio/ieee80211/decode.h>
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>

#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "c8p_blocks.h"
#ifndef C8P_GR_STOCK_API_ONLY
#include "c8p_pinned_buffer.h"
#endif

namespace gr {
namespace ieee80211 {

// true: nothing beyond the reference's own use of GNU Radio (see SAFE MODE above)
inline bool c8p_gr_safe()
{
#ifdef C8P_GR_STOCK_API_ONLY
  return true;
#else
  return c8p::batching_cfg::safe_mode();
#endif
}
// the io_signature of n stream ports of `item` bytes: the module's pinned buffer type, or GNU Radio's default in safe mode
inline gr::io_signature::sptr c8p_stream_sig(int n, size_t item)
{
#ifndef C8P_GR_STOCK_API_ONLY
  if (!c8p_gr_safe()) return gr::io_signature::make(n, n, (int)item, c8p_pinned_buffer::type);
#endif
  return gr::io_signature::make(n, n, (int)item);
}
// the cfg a block of this module hands its front
inline c8p::batching_cfg c8p_gr_cfg()
{
  c8p::batching_cfg cfg;                 // (its defaults are the safe ones under C8P_GR_SAFE=1)
#ifdef C8P_GR_STOCK_API_ONLY
  cfg.fused = cfg.fused_auto = cfg.pinned_io = cfg.defer_consume = cfg.llr_handle = cfg.flush_when_idle = false;
#endif
  return cfg;
}

template <class API>
class c8p_gr_block : public API, private c8p::block_io
{
public:
  c8p_gr_block(const std::string& name, gr::io_signature::sptr in, gr::io_signature::sptr out) : gr::block(name, in, out)
  {
    this->set_tag_propagation_policy(gr::block::TPP_DONT);          // demod_impl.cc:44, decode_impl.cc:41
    // wake(): a message to a port of its own is how a block gets its scheduler thread to run when no stream item arrives
    // (safe mode: no timer thread, nobody to wake, no extra port on the block)
    if (!c8p_gr_safe()) {
      this->message_port_register_in(pmt::mp("c8p_tick"));
      this->set_msg_handler(pmt::mp("c8p_tick"), [](pmt::pmt_t) {});
    }
  }

  // gr::block
  void forecast(int noutput_items, gr_vector_int& ninput_items_required) override { d_front->forecast(noutput_items, ninput_items_required); }
  int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& output_items) override
  {
    d_in_now = &input_items; d_out_now = &output_items;             // (input_is_pinned / output_is_pinned look at the pointers)
    const int produced = d_front->general_work(noutput_items, ninput_items, input_items, output_items);
    d_in_now = nullptr; d_out_now = nullptr;
    return produced;
  }
  // a finite flowgraph (file source) stops once the input is consumed: whatever the front still holds goes out now
  bool stop() override { d_front->flush(); return gr::block::stop(); }

protected:
  void attach(std::shared_ptr<c8p::c8p_block> front) { d_front = std::move(front); d_front->set_io(this); }

private:
  // c8p::block_io
  uint64_t nitems_read(unsigned port) override { return gr::block::nitems_read(port); }
  uint64_t nitems_written(unsigned port) override { return gr::block::nitems_written(port); }

  void get_tags_in_range(std::vector<c8p::tag_t>& v, unsigned port, uint64_t start, uint64_t end) override
  {
    std::vector<gr::tag_t> tags;
    gr::block::get_tags_in_range(tags, port, start, end);
    v.clear();
    for (const gr::tag_t& t : tags) {
      c8p::tag_t o;
      o.offset = t.offset;
      o.key = pmt::symbol_to_string(t.key);
      if (pmt::is_integer(t.value)) o.value = c8p::tag_value::from_long(pmt::to_long(t.value));
      else if (pmt::is_real(t.value)) o.value = c8p::tag_value::from_float((float)pmt::to_double(t.value));
      else if (pmt::is_c32vector(t.value)) o.value = c8p::tag_value::from_c32vector(pmt::c32vector_elements(t.value));
      else continue;                                   // a tag of another kind is none of the hot path's
      v.push_back(std::move(o));
    }
  }

  void add_item_tag(unsigned port, uint64_t offset, const std::string& key, const c8p::tag_value& val) override
  {
    pmt::pmt_t p;
    switch (val.kind) {
      case c8p::tag_value::LONG: p = pmt::from_long(val.l); break;
      case c8p::tag_value::FLOAT: p = pmt::from_float(val.f); break;
      default: p = pmt::init_c32vector(val.v.size(), val.v); break;
    }
    gr::block::add_item_tag(port, offset, key_symbol(key), p, this->alias_pmt());   // demod_impl.cc:224-253
  }

  // the interned symbol of a tag key: the fronts write 14 tags per packet with ~16 distinct keys, and pmt::string_to_symbol is a lookup
  // in the process-wide symbol table under its mutex — the block keeps the symbols it has used (scheduler thread only)
  pmt::pmt_t key_symbol(const std::string& key)
  {
    for (const auto& kv : d_key_syms)
      if (kv.first == key) return kv.second;
    d_key_syms.emplace_back(key, pmt::string_to_symbol(key));
    return d_key_syms.back().second;
  }
  std::vector<std::pair<std::string, pmt::pmt_t>> d_key_syms;

  void consume_each(int n) override { gr::block::consume_each(n); }

  // decode_impl.cc:359-361: pmt::cons(dict{"len": n}, blob)
  void message_port_pub(const std::string& port, const c8p::pdu_t& m) override
  {
    pmt::pmt_t meta = pmt::dict_add(pmt::make_dict(), pmt::mp("len"), pmt::from_long(m.len));
    gr::block::message_port_pub(pmt::mp(port), pmt::cons(meta, pmt::make_blob(m.blob.data(), m.blob.size())));
  }

  void debug_print(const std::string& line) override { std::cout << line << std::endl; }   // the reference's dout

  // the front's latency timer collected results while no input was arriving: after the (empty) handler the scheduler thread
  // evaluates forecast() — which asks for no input while results wait for output space — and calls general_work()
#ifdef C8P_GR_STOCK_API_ONLY
  void wake() override {}
  bool input_is_pinned(unsigned) override { return false; }
  bool output_is_pinned(unsigned) override { return false; }
  size_t input_buffer_bytes(unsigned) override { return 0; }
#else
  void wake() override { if (!c8p_gr_safe()) this->_post(pmt::mp("c8p_tick"), pmt::PMT_T); }      // (*)

  // custom buffers (c8p_pinned_buffer.h): the scheduler allocated this port's buffer through the module's buffer type when the
  // io_signature names it — which the impl classes' constructors do — AND the pointer this call was handed lies in a buffer that type
  // has allocated (the fronts ask during general_work): a flowgraph that overrides the type, or a scheduler that did not honour it,
  // gets pageable staging back instead of a device read of memory the device cannot see
  bool port_is_pinned(gr::io_signature::sptr sig, unsigned port, const void* now)                  // (*)
  {
    if (c8p_gr_safe() || !(sig->stream_buffer_type(port) == c8p_pinned_buffer::type)) return false;
    if (now && !c8p_pinned_buffer::owns(now)) {
      if (!d_warned) std::cerr << "ieee80211 (c8p): a port's io_signature names the pinned buffer type but its items are not in one: staged copies" << std::endl;
      d_warned = true;
      return false;
    }
    return true;
  }
  bool input_is_pinned(unsigned port) override
  {
    return port_is_pinned(this->input_signature(), port, d_in_now && port < d_in_now->size() ? (*d_in_now)[port] : nullptr);
  }
  bool output_is_pinned(unsigned port) override
  {
    return port_is_pinned(this->output_signature(), port, d_out_now && port < d_out_now->size() ? (*d_out_now)[port] : nullptr);
  }
  // what c8p_pinned_buffer::make_buffer allocates at least: megabytes, so the fronts gather (c8p_blocks.h: C8P_GATHER_MIN_BUFFER)
  size_t input_buffer_bytes(unsigned port) override { return input_is_pinned(port) ? (size_t)C8P_PINNED_BUFFER_MIN_BYTES : 0; }
#endif

  // Who reads this output port (c8p::block_io::output_feeds_only_decode)?  The edge's buffer must have ONE reader, this module's decode
  // block, for a demod front to run fused by default (c8p::batching_cfg::fused_auto): nothing else ever looks at the LLR item values
  // then.  The flat flowgraph has attached detail, buffers and readers before any scheduler thread calls general_work.
  int output_feeds_only_decode(unsigned port) override
  {
#ifdef C8P_GR_NO_TOPOLOGY
    (void)port;
    return -1;
#else
    if (c8p_gr_safe()) return -1;
    try {
      gr::block_detail_sptr d = this->detail();
      if (!d || (int)port >= d->noutputs()) return -1;
      gr::buffer_sptr b = d->output(port);
      if (!b || b->nreaders() == 0) return -1;
      // exactly ONE reader, and it is the decode block: a batch's hand-over leaves the registry when its reader moves on to the next
      // batch, so two decode blocks on one edge must get the items themselves (plain mode)
      if (b->nreaders() != 1) return 0;
      return std::dynamic_pointer_cast<gr::ieee80211::decode>(b->reader(0)->link()) ? 1 : 0;
    } catch (...) {                                    // (a reader whose block is gone: the flowgraph is being torn down)
      return -1;
    }
#endif
  }

  std::shared_ptr<c8p::c8p_block> d_front;
  const gr_vector_const_void_star* d_in_now = nullptr;     // the pointers of the general_work call in progress
  const gr_vector_void_star* d_out_now = nullptr;
  bool d_warned = false;
};

} // namespace ieee80211
} // namespace gr

#endif
