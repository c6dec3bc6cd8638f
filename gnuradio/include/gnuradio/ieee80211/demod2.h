/* gr::ieee80211::demod2 — public API, unchanged from the reference (include/gnuradio/ieee80211/demod2.h):
 * 2 x gr_complex in (two antennas, packets tagged by signal2), 1 x float out. */
#ifndef INCLUDED_IEEE80211_DEMOD2_H
#define INCLUDED_IEEE80211_DEMOD2_H

#include <gnuradio/block.h>
#include <gnuradio/ieee80211/api.h>

namespace gr {
namespace ieee80211 {

class IEEE80211_API demod2 : virtual public gr::block
{
public:
  typedef std::shared_ptr<demod2> sptr;
  static sptr make();
};

} // namespace ieee80211
} // namespace gr

#endif /* INCLUDED_IEEE80211_DEMOD2_H */
