#include "demod_impl.h"

namespace gr {
namespace ieee80211 {

demod::sptr demod::make(int mupos, int mugid) { return gnuradio::make_block_sptr<demod_impl>(mupos, mugid); }

// the module's one definition of its buffer type (c8p_pinned_buffer.h)
gr::buffer_type c8p_pinned_buffer::type(buftype_C8P_PINNED{});

// ports as the reference's (1 x gr_complex in, 1 x float out, demod_impl.cc:36-37), their buffers of the module's pinned type: the
// edge from signal and the edge to decode are then allocated through it (flat_flowgraph::allocate_buffer)
demod_impl::demod_impl(int mupos, int mugid)
    : gr::block("demod", gr::io_signature::make(1, 1, sizeof(gr_complex), c8p_pinned_buffer::type),
                gr::io_signature::make(1, 1, sizeof(float), c8p_pinned_buffer::type)),
      c8p_gr_block<demod>("demod", gr::io_signature::make(1, 1, sizeof(gr_complex), c8p_pinned_buffer::type),
                          gr::io_signature::make(1, 1, sizeof(float), c8p_pinned_buffer::type))
{
  c8p::batching_cfg cfg;
  attach(c8p::demod::make(mupos, mugid, c8p::default_c8p_api(), cfg));   // loads libc8p_rx.so (C8P_RX_LIB or the loader path); throws without a HIP device
}

demod_impl::~demod_impl() {}

} // namespace ieee80211
} // namespace gr
