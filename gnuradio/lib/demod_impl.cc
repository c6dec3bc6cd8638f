#include "demod_impl.h"

namespace gr {
namespace ieee80211 {

demod::sptr demod::make(int mupos, int mugid) { return gnuradio::make_block_sptr<demod_impl>(mupos, mugid); }

demod_impl::demod_impl(int mupos, int mugid)
    : gr::block("demod", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(float))),
      c8p_gr_block<demod>("demod", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(float)))
{
  attach(c8p::demod::make(mupos, mugid));        // loads libc8p_rx.so (C8P_RX_LIB or the loader path); throws without a HIP device
}

demod_impl::~demod_impl() {}

} // namespace ieee80211
} // namespace gr
