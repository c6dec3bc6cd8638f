#include "decode_impl.h"

namespace gr {
namespace ieee80211 {

decode::sptr decode::make(bool ifdebug) { return gnuradio::make_block_sptr<decode_impl>(ifdebug); }

decode_impl::decode_impl(bool ifdebug)
    : gr::block("decode", c8p_stream_sig(1, sizeof(float)), gr::io_signature::make(0, 0, 0)),            // decode_impl.cc:35-38
      c8p_gr_block<decode>("decode", c8p_stream_sig(1, sizeof(float)), gr::io_signature::make(0, 0, 0))
{
  message_port_register_out(pmt::mp("out"));     // decode_impl.cc:39
  c8p::batching_cfg cfg = c8p_gr_cfg();
  attach(c8p::decode::make(ifdebug, c8p::default_c8p_api(), cfg));
}

decode_impl::~decode_impl() {}

} // namespace ieee80211
} // namespace gr
