// demod_impl.h — gr::ieee80211::demod on the MI355X engine.  Replaces lib/demod_impl.{h,cc} of the reference: same class,
// same make(mupos, mugid) (include/gnuradio/ieee80211/demod.h:35), same ports (1 x gr_complex in, 1 x float out,
// demod_impl.cc:36-37), same tags; general_work() is the batching front c8p::demod.
#ifndef INCLUDED_IEEE80211_DEMOD_IMPL_H
#define INCLUDED_IEEE80211_DEMOD_IMPL_H

#include <gnuradio/ieee80211/demod.h>

#include "c8p_gr_adapter.h"

namespace gr {
namespace ieee80211 {

class demod_impl : public c8p_gr_block<demod>
{
public:
  demod_impl(int mupos, int mugid);
  ~demod_impl() override;
};

} // namespace ieee80211
} // namespace gr

#endif /* INCLUDED_IEEE80211_DEMOD_IMPL_H */
