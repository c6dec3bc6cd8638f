#include "demod_impl.h"

namespace gr {
namespace ieee80211 {

demod::sptr demod::make(int mupos, int mugid) { return gnuradio::make_block_sptr<demod_impl>(mupos, mugid); }

#ifndef C8P_GR_STOCK_API_ONLY
// the module's one definition of its buffer type (c8p_pinned_buffer.h)
gr::buffer_type c8p_pinned_buffer::type(buftype_C8P_PINNED{});
#endif

// ports as the reference's (1 x gr_complex in, 1 x float out, demod_impl.cc:36-37); c8p_stream_sig: their buffers of the module's
// pinned type — the edge from signal and the edge to decode are then allocated through it (flat_flowgraph::allocate_buffer) — or, in
// safe mode (C8P_GR_SAFE=1 / -DC8P_GR_STOCK_API_ONLY), exactly the reference's io_signature::make(1, 1, size)
demod_impl::demod_impl(int mupos, int mugid)
    : gr::block("demod", c8p_stream_sig(1, sizeof(gr_complex)), c8p_stream_sig(1, sizeof(float))),
      c8p_gr_block<demod>("demod", c8p_stream_sig(1, sizeof(gr_complex)), c8p_stream_sig(1, sizeof(float)))
{
  c8p::batching_cfg cfg = c8p_gr_cfg();
  attach(c8p::demod::make(mupos, mugid, c8p::default_c8p_api(), cfg));   // loads libc8p_rx.so (C8P_RX_LIB or the loader path); throws without a HIP device
}

demod_impl::~demod_impl() {}

} // namespace ieee80211
} // namespace gr
