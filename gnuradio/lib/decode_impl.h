// decode_impl.h — gr::ieee80211::decode on the MI355X engine.  Replaces lib/decode_impl.{h,cc}: same make(ifdebug)
// (include/gnuradio/ieee80211/decode.h), 1 x float in, no stream output, message port "out" (decode_impl.cc:35-39),
// forecast noutput_items + 160 (decode_impl.cc:54-58), the same PDU blobs and debug lines.
#ifndef INCLUDED_IEEE80211_DECODE_IMPL_H
#define INCLUDED_IEEE80211_DECODE_IMPL_H

#include <gnuradio/ieee80211/decode.h>

#include "c8p_gr_adapter.h"

namespace gr {
namespace ieee80211 {

class decode_impl : public c8p_gr_block<decode>
{
public:
  explicit decode_impl(bool ifdebug);
  ~decode_impl() override;
};

} // namespace ieee80211
} // namespace gr

#endif /* INCLUDED_IEEE80211_DECODE_IMPL_H */
