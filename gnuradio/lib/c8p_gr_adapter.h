// c8p_gr_adapter.h — the scheduler services a batching front needs (c8p::block_io), implemented BY the gr::block itself.
//
// A front (c8p::demod / demod2 / decode of ../../gr-ieee80211_amd/host/c8p_blocks.h, compiled into namespace c8p in this
// module) cuts the tagged item stream into packets, batches them through the C ABI (libc8p_rx.so) and replays the results;
// everything it asks of GNU Radio goes through the eight block_io methods below.  c8p_gr_block<API> is the block class:
// API = the public class of include/gnuradio/ieee80211/ (demod, demod2, decode), which derives from gr::block, so the
// protected scheduler calls (get_tags_in_range, add_item_tag, ...) are made from inside the block, as in the reference's
// own *_impl classes.  Tag values cross as the three PMT kinds the reference's blocks use: long, float, c32vector.
#ifndef INCLUDED_IEEE80211_C8P_GR_ADAPTER_H
#define INCLUDED_IEEE80211_C8P_GR_ADAPTER_H

#include <gnuradio/block.h>
#include <gnuradio/io_signature.h>
#include <pmt/pmt.h>

#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "c8p_blocks.h"

namespace gr {
namespace ieee80211 {

template <class API>
class c8p_gr_block : public API, private c8p::block_io
{
public:
  c8p_gr_block(const std::string& name, gr::io_signature::sptr in, gr::io_signature::sptr out) : gr::block(name, in, out)
  {
    this->set_tag_propagation_policy(gr::block::TPP_DONT);          // demod_impl.cc:44, decode_impl.cc:41
    // wake(): a message to a port of its own is how a block gets its scheduler thread to run when no stream item arrives
    this->message_port_register_in(pmt::mp("c8p_tick"));
    this->set_msg_handler(pmt::mp("c8p_tick"), [](pmt::pmt_t) {});
  }

  // gr::block
  void forecast(int noutput_items, gr_vector_int& ninput_items_required) override { d_front->forecast(noutput_items, ninput_items_required); }
  int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                   gr_vector_void_star& output_items) override
  {
    return d_front->general_work(noutput_items, ninput_items, input_items, output_items);
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
    gr::block::add_item_tag(port, offset, pmt::string_to_symbol(key), p, this->alias_pmt());   // demod_impl.cc:224-253
  }

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
  void wake() override { this->_post(pmt::mp("c8p_tick"), pmt::PMT_T); }

  std::shared_ptr<c8p::c8p_block> d_front;
};

} // namespace ieee80211
} // namespace gr

#endif
