// demod2_impl.h — gr::ieee80211::demod2 (2 antennas, rx2.grc) on the MI355X engine.  Replaces lib/demod2_impl.{h,cc}:
// same make() (include/gnuradio/ieee80211/demod2.h), 2 x gr_complex in, 1 x float out (demod2_impl.cc:36-37).
#ifndef INCLUDED_IEEE80211_DEMOD2_IMPL_H
#define INCLUDED_IEEE80211_DEMOD2_IMPL_H

#include <gnuradio/ieee80211/demod2.h>

#include "c8p_gr_adapter.h"

namespace gr {
namespace ieee80211 {

class demod2_impl : public c8p_gr_block<demod2>
{
public:
  demod2_impl();
  ~demod2_impl() override;
};

} // namespace ieee80211
} // namespace gr

#endif /* INCLUDED_IEEE80211_DEMOD2_IMPL_H */
