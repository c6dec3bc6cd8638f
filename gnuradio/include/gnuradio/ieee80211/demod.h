/* gr::ieee80211::demod — public API, unchanged from the reference (include/gnuradio/ieee80211/demod.h:35):
 * 1 x gr_complex in (packets tagged by signal), 1 x float out (LLRs tagged for decode). */
#ifndef INCLUDED_IEEE80211_DEMOD_H
#define INCLUDED_IEEE80211_DEMOD_H

#include <gnuradio/block.h>
#include <gnuradio/ieee80211/api.h>

namespace gr {
namespace ieee80211 {

class IEEE80211_API demod : virtual public gr::block
{
public:
  typedef std::shared_ptr<demod> sptr;
  /* mupos: MU-MIMO user position (0 / 1), mugid: MU-MIMO group id */
  static sptr make(int mupos, int mugid);
};

} // namespace ieee80211
} // namespace gr

#endif /* INCLUDED_IEEE80211_DEMOD_H */
