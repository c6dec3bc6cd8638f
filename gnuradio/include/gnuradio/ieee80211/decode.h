/* gr::ieee80211::decode — public API, unchanged from the reference (include/gnuradio/ieee80211/decode.h):
 * 1 x float in (LLRs tagged by demod / demod2), message port "out" (one PDU per FCS-correct (sub)frame, NDP reports). */
#ifndef INCLUDED_IEEE80211_DECODE_H
#define INCLUDED_IEEE80211_DECODE_H

#include <gnuradio/block.h>
#include <gnuradio/ieee80211/api.h>

namespace gr {
namespace ieee80211 {

class IEEE80211_API decode : virtual public gr::block
{
public:
  typedef std::shared_ptr<decode> sptr;
  /* ifdebug: print the "ieee80211 decode, ... crc32 correct|wrong, total:..." lines tools/performance scrapes */
  static sptr make(bool ifdebug);
};

} // namespace ieee80211
} // namespace gr

#endif /* INCLUDED_IEEE80211_DECODE_H */
