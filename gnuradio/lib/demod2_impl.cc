#include "demod2_impl.h"

namespace gr {
namespace ieee80211 {

demod2::sptr demod2::make() { return gnuradio::make_block_sptr<demod2_impl>(); }

demod2_impl::demod2_impl()
    : gr::block("demod2", c8p_stream_sig(2, sizeof(gr_complex)), c8p_stream_sig(1, sizeof(float))),     // demod2_impl.cc:36-37
      c8p_gr_block<demod2>("demod2", c8p_stream_sig(2, sizeof(gr_complex)), c8p_stream_sig(1, sizeof(float)))
{
  c8p::batching_cfg cfg = c8p_gr_cfg();
  attach(c8p::demod2::make(c8p::default_c8p_api(), cfg));
}

demod2_impl::~demod2_impl() {}

} // namespace ieee80211
} // namespace gr
